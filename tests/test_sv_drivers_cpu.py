"""Structural-variant drivers against fixtures generated from the reference's own
`orca_predict.process_*` and `orca_utils.StructuralChange2` (tools/make_golden.py G11 / G12).

The drivers' numerics are `genomepredict`'s (pinned by G7/G8); what is pinned here is everything around it: the
coordinate algebra of the mutated chromosome, window clipping, piece-wise sequence assembly incl. reverse
complements, inserted strings and 'N' padding, anchors, labels and scaled annotations.  Both sides run the same cheap
stand-in models (`orca_amd.standins.FakeModel32`) on the CPU.
"""
import json
import os

import numpy as np
import pytest

from orca_amd import orca_predict, orca_utils, synth
from tests import standins

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_structural_change_scripts_match_reference():
    scripts = json.load(open(os.path.join(GOLD, "G12_structural_change.json")))
    nq = 0
    for sc_ref in scripts:
        sc = orca_utils.StructuralChange2("chrA", sc_ref["L"])
        for op in sc_ref["ops"]:
            if op[0] == "concat":
                other = orca_utils.StructuralChange2("chrB", op[1])
                other.invert(op[2], op[3])
                sc = sc + other
            elif op[0] == "insert":
                sc.insert(op[1], op[2], strand=op[3])
            else:
                getattr(sc, op[0])(op[1], op[2])
        assert list(sc.coord_points) == sc_ref["points"]
        assert [[g.len] + list(g.ref) for g in sc.segments] == sc_ref["segments"]
        for q in sc_ref["queries"]:
            s0, e0 = q["q"]
            try:
                got = [list(x) for x in sc[s0:e0]]
            except ValueError:
                got = "ValueError"
            assert got == q["pieces"], (sc_ref["ops"], q["q"])
            rc, cc = sc.query_ref("chrA", s0, e0)
            assert [[int(v) for v in r] for r in rc] == q["ref"]
            assert [[int(r[0]), int(r[1]), r[2]] for r in cc] == q["cur"]
            nq += 1
    assert nq > 500


def test_process_anno_and_errors():
    assert orca_utils.process_anno([[10, 20, "black"], [15, "double"]], base=10, window_radius=5) == [[0.0, 1.0, "black"], [0.5, "double"]]
    with pytest.raises(ValueError):
        orca_utils.process_anno([[1]])
    g = synth.sv_driver_genome()
    with pytest.raises(ValueError):
        orca_predict.process_del("chrS", 1, 2, g, custom_models=[object()], window_radius=1234, use_cuda=False)
    with pytest.raises(NotImplementedError):   # plotting is the reference's job
        orca_predict.process_del("chrS", 1, 2, g, custom_models=[object()], file="x", use_cuda=False)


def test_packed_genome_selene_semantics():
    g = synth.sv_driver_genome()
    assert dict(g.get_chr_lens()) == {"chrS": 40_000_000, "chrT": 36_000_000}
    e = g.get_encoding_from_coords("chrS", 1000, 1100)
    assert e.shape == (100, 4) and e.dtype == np.float32 and np.all(e.sum(axis=1) == 1.0)
    r = g.get_encoding_from_coords("chrS", 1000, 1100, strand="-")
    np.testing.assert_array_equal(r, e[::-1, ::-1])                      # selene_utils2.py:259-260
    p = g.get_encoding_from_coords("chrS", -5, 10, pad=True)
    assert p.shape == (15, 4) and np.all(p[:5] == 0.25) and np.array_equal(p[5:], g.get_encoding_from_coords("chrS", 0, 10))
    n = dict(g.get_chr_lens())["chrT"]
    q = g.get_encoding_from_coords("chrT", n - 3, n + 4, pad=True)
    assert q.shape == (7, 4) and np.all(q[3:] == 0.25)
    with pytest.raises(AssertionError):
        g.get_encoding_from_coords("chrT", n - 3, n + 4)
    np.testing.assert_array_equal(g.sequence_to_encoding("AcgTNx"),
                                  np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [.25] * 4, [.25] * 4], np.float32))


@pytest.fixture(scope="module")
def sv_setup():
    saved = dict(orca_predict.model_dict_global)
    orca_predict.model_dict_global["h1esc"] = standins.FakeModel32(0)
    orca_predict.model_dict_global["hff"] = standins.FakeModel32(1)
    yield synth.sv_driver_genome(), np.load(os.path.join(GOLD, "G11_sv_drivers.npz"))
    orca_predict.model_dict_global.clear()
    orca_predict.model_dict_global.update(saved)


@pytest.mark.parametrize("case", [c[0] for c in synth.sv_driver_cases()])
def test_sv_driver_matches_reference(sv_setup, case):
    genome, gold = sv_setup
    name, fn, a, kw = next(c for c in synth.sv_driver_cases() if c[0] == case)
    cm = None if fn == "process_ins" else [orca_predict.model_dict_global["h1esc"]]
    outs = getattr(orca_predict, fn)(*a, genome, custom_models=cm, target=False, use_cuda=False, **kw)
    got = synth.summarize_outputs(outs)
    keys = [k for k in gold.files if k.startswith(name + ".")]
    assert len(keys) == len(got) and len(keys) > 0
    for k in keys:
        want, have = gold[k], got[k[len(name) + 1:]]
        if want.dtype.kind in "US":
            assert str(want[0]) == str(have[0]), k
        elif want.dtype.kind == "i":
            np.testing.assert_array_equal(want, have, err_msg=k)
        else:
            np.testing.assert_allclose(have, want, rtol=2e-5, atol=2e-5, err_msg=k)


@pytest.fixture(scope="module")
def sv_setup_256():
    saved, saved_fn = dict(orca_predict.model_dict_global), orca_predict.genomepredict_256Mb
    orca_predict.model_dict_global["h1esc_256m"] = standins.Background256(0)
    orca_predict.model_dict_global["hff_256m"] = standins.Background256(1)
    rec = standins.Recorder256()
    orca_predict.genomepredict_256Mb = rec
    # .to("cpu"): the drivers take their packed-codes route (what they do with the genome in HBM), on the host
    yield synth.sv_driver_genome_256().to("cpu"), np.load(os.path.join(GOLD, "G13_sv_drivers_256.npz")), rec
    orca_predict.genomepredict_256Mb = saved_fn
    orca_predict.model_dict_global.clear()
    orca_predict.model_dict_global.update(saved)


@pytest.mark.parametrize("case", [c[0] for c in synth.sv_driver_cases_256()])
def test_sv_driver_256mb_views_match_reference(sv_setup_256, case):
    """window_radius=128000000: every view the reference's driver hands to genomepredict_256Mb - the 256 Mb sequence
    (exact position-weighted digest), the per-model distance backgrounds and targets from `_retrieve_multi`, chromosome
    label, rounded length, anchor, window centre, annotation - recorded on both sides instead of running a model."""
    genome, gold, rec = sv_setup_256
    name, fn, a, kw = next(c for c in synth.sv_driver_cases_256() if c[0] == case)
    first = len(rec.calls)
    tgt = [standins.FakeTarget256()] if fn == "process_del" else False    # the reference's process_del needs targets at 256 Mb
    outs = getattr(orca_predict, fn)(*a, genome, custom_models=[object(), object()], target=tgt,
                                     use_cuda=True, window_radius=128000000, padding_chr="chr1", **kw)
    got = rec.summary(first)
    got["order"] = np.array([o["call"] - first for o in outs])
    keys = [k for k in gold.files if k.startswith(name + ".")]
    assert len(keys) == len(got) and len(keys) > 10
    for k in keys:
        want, have = gold[k], got[k[len(name) + 1:]]
        if want.dtype.kind in "US":
            assert str(want[0]) == str(have[0]), k
        elif want.dtype.kind in "ib":
            np.testing.assert_array_equal(want, have, err_msg=k)
        elif k.endswith("_seq"):
            np.testing.assert_array_equal(want, have, err_msg=k)        # exact: sums of quarter-integers in float64
        else:
            np.testing.assert_allclose(have, want, rtol=1e-12, atol=0, err_msg=k)
    rec.calls[first:] = [{} for _ in rec.calls[first:]]                 # free the digests' memory


def test_sv_drivers_256mb_unsupported_forms():
    g = synth.sv_driver_genome()
    for fn, a in (("process_ins", ("chrS", 5, "ACGT")), ("process_custom", ([], [], 0))):
        with pytest.raises(NotImplementedError):
            getattr(orca_predict, fn)(*a, g, custom_models=[object()], window_radius=128000000, use_cuda=False)


def test_process_seqstr_matches_reference(monkeypatch):
    """`process_seqstr` (orca_predict.py:3060-3161) against G19 = the reference's function run with the same stand-in for
    the absent `seqstr` package and the same stand-in model: exactly 32 Mb with the default zoom, and an odd-length longer
    string chopped to its middle 32 Mb with a zoom position.  Error behaviour as the reference's."""
    import sys
    import types
    gold = np.load(os.path.join(GOLD, "G19_seqstr.npz"))
    monkeypatch.setitem(sys.modules, "seqstr", None)                     # not installed
    with pytest.raises(ImportError, match="Seqstr is not installed"):
        orca_predict.process_seqstr("[32000000,41]")
    monkeypatch.setitem(sys.modules, "seqstr", types.SimpleNamespace(seqstr=standins.FakeSeqstr()))
    with pytest.raises(ValueError, match="at least 32Mb"):
        orca_predict.process_seqstr("[31999999,1]", custom_models=[standins.FakeModel32(0)], use_cuda=False)
    with pytest.raises(NotImplementedError):
        orca_predict.process_seqstr("[32000000,41]", file="x", custom_models=[standins.FakeModel32(0)], use_cuda=False)
    h1 = standins.FakeModel32(0)
    for name, spec, mpos in synth.seqstr_cases():
        got = synth.summarize_outputs(orca_predict.process_seqstr(spec, mpos=mpos, custom_models=[h1], use_cuda=False))
        keys = [k for k in gold.files if k.startswith(name + ".")]
        assert len(keys) == len(got) and len(keys) > 0
        for k in keys:
            want, have = gold[k], got[k[len(name) + 1:]]
            if want.dtype.kind in "US":
                assert str(want[0]) == str(have[0]), k
            elif want.dtype.kind == "i":
                np.testing.assert_array_equal(want, have, err_msg=k)
            else:
                np.testing.assert_allclose(have, want, rtol=2e-5, atol=2e-5, err_msg=k)
