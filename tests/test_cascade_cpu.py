"""Host-logic parity of orca_amd.orca_predict (genomepredict / genomepredict_256Mb)
against fixtures produced by the REAL reference functions (tools/make_golden.py G7/G9).
Runs on CPU: the cascade is driven with oracle-backed foreign nn.Modules through the
duck-typed model protocol (use_cuda=False), so what is tested here is the product's
zoom arithmetic, slicing, strand handling, background coarse-graining, target /
annotation bookkeeping and output dict - not the kernels."""
import numpy as np
import pytest
import torch

from oracle import orca_oracle as O
from orca_amd import orca_predict as P
from tests import standins
from orca_amd import synth
from tests.util import golden, maxabs, stats, synth_sd


class OracleModel32(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        self.net0 = standins.FakeNet0(nbins=8000, seed=seed)
        self.net = O.OracleModule(O.encoder2_forward, synth_sd("Encoder2", seed))
        self.denets = {lv: O.OracleModule(O.decoder_forward, synth_sd("Decoder", seed + lv, upsample_mode="bilinear"),
                                          upsample_mode="bilinear") for lv in (1, 2, 4, 8, 16, 32)}
        self.denet_1_pt = O.OracleModule(O.decoder_1m_forward, synth_sd("Decoder_1m", seed))
        self.normmats, self.epss = synth.synth_normmats_32m()


class OracleModel256(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        self.net0 = standins.FakeNet0(nbins=64000, seed=seed)
        self.net1 = O.OracleModule(O.encoder2_forward, synth_sd("Encoder2", seed))
        self.net = O.OracleModule(O.encoder3_forward, synth_sd("Encoder3", seed))
        self.denets = {lv: O.OracleModule(O.decoder_forward, synth_sd("Decoder", seed + lv, upsample_mode="bilinear"),
                                          upsample_mode="bilinear") for lv in (32, 64, 128, 256)}


@pytest.fixture(scope="module")
def model32():
    return OracleModel32(0)


def test_genomepredict_matches_reference(model32):
    g = golden("G7_cascade32.npz")
    seq = synth.synth_sequence(320000, seed=41)
    for ci in (1, 2):   # the clip-at-0 / clip-at-125 cases; case 0 runs in test_genomepredict_targets_and_annotation, all four on the GPU (test_gpu_e2e)
        mpos, wpos = (int(v) for v in g[f"c{ci}_args"])
        out = P.genomepredict(seq, "chrS", mpos, wpos, models=[model32], use_cuda=False)
        assert out["start_coords"] == list(g[f"c{ci}_start"])
        assert out["end_coords"] == list(g[f"c{ci}_end"])
        assert out["chr"] == "chrS" and out["experiments"] is None and out["annos"] is None
        assert len(out["predictions"]) == 1 and len(out["predictions"][0]) == 6
        assert len(out["normmats"][0]) == 6
        for j, p in enumerate(out["predictions"][0]):
            assert p.shape == (250, 250) and p.dtype == np.float32
            ref = g[f"c{ci}_sub_{j}"]
            assert maxabs(p if ci == 0 else p[::5, ::5], ref) < 5e-5, (ci, j)
            np.testing.assert_allclose(stats(p), g[f"c{ci}_stats_{j}"], rtol=2e-4, atol=1e-3)


def test_genomepredict_targets_and_annotation(model32):
    g = golden("G7_cascade32.npz")
    seq = synth.synth_sequence(320000, seed=41)
    mpos, wpos = (int(v) for v in g["c0_args"])
    tgt = np.abs(np.random.RandomState(42).randn(1, 8000, 8000).astype(np.float32)) * 1e-3
    tgt[0, 100:140, :] = np.nan
    anno = [(0.1, 0.3, "a"), (0.52, "b"), (0.9, 0.95, "c")]
    out = P.genomepredict(seq, "chrS", mpos, wpos, models=[model32], targets=[torch.from_numpy(tgt)], annotation=anno,
                          use_cuda=False)
    assert out["start_coords"] == list(g["c0_start"]) and out["end_coords"] == list(g["c0_end"])
    for j, p in enumerate(out["predictions"][0]):
        assert maxabs(p, g[f"c0_sub_{j}"]) < 5e-5
        np.testing.assert_allclose(stats(p), g[f"c0_stats_{j}"], rtol=2e-4, atol=1e-3)
    for j, e in enumerate(out["experiments"][0]):
        ref = g[f"tgt_sub_{j}"]
        got = np.asarray(e)[::5, ::5]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.nanmax(np.abs(got - ref)) < 1e-5
    mine = repr([[tuple(float(v) if not isinstance(v, str) else v for v in r) for r in lv] for lv in out["annos"]])
    assert mine == str(g["annos_repr"][0])


def test_genomepredict_256mb_matches_reference():
    g = golden("G9_cascade256.npz")
    model = OracleModel256(0)
    seq = synth.synth_sequence(512000, seed=51)
    for ci in (0, 2):   # the chromosome-end case 1 runs on the GPU with the same fixture (test_gpu_e2e)
        mpos, wpos, chrlen = (int(v) for v in g[f"c{ci}_args"])
        nm = synth.synth_normmat_256m(chrlen, seed=0)
        out = P.genomepredict_256Mb(seq, "chrS", [nm], chrlen, mpos, wpos, models=[model], padding_chr="chrP", use_cuda=False)
        assert out["start_coords"] == list(g[f"c{ci}_start"])
        assert [int(v) for v in out["end_coords"]] == list(g[f"c{ci}_end"])
        assert out["padding_chr"] == "chrP" and len(out["normmats"]) == 2
        for j, p in enumerate(out["predictions"][0]):
            ref = g[f"c{ci}_sub_{j}"]
            assert maxabs(p if ci == 0 else p[::5, ::5], ref) < 5e-5, (ci, j)


def test_product_models_refuse_cpu():
    from orca_amd import orca_models as M
    from orca_amd._lib import OrcaHipError
    m = M.H1esc(synthetic_seed=0)
    assert set(m.denets) == {1, 2, 4, 8, 16, 32} and set(m.normmats) == set(m.epss) == set(m.denets)
    assert m.normmats[32].shape == (250, 250)
    if not torch.cuda.is_available():
        with pytest.raises(OrcaHipError):
            P.genomepredict(synth.synth_sequence(32000, seed=1), "chrS", 16000000, 16000000, models=[m], use_cuda=False)
    with pytest.raises(FileNotFoundError):
        M.H1esc(model_dir="/nonexistent")
