#!/usr/bin/env python
"""Generate golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  Imports
/root/reference/orca_modules.py and orca_predict.py read-only (third-party
modules that are not installed - selene_sdk, cooler, cooltools, pyfaidx,
pyranges, tabix, pygenometracks - are stubbed in sys.modules; none of them is
touched by the functions exercised here), loads deterministic synthetic weights
from orca_amd/synth.py into the reference nn.Modules, runs seeded inputs and
stores the outputs as small .npz fixtures.  The reference source itself never
travels: fixtures are data only (inputs are regenerated from seeds).

Usage:  python tools/make_golden.py [--full32m | --config3 | --coarsegrain | --full256m | --genome | --svreal]   (--full32m adds the ~8 min
full 32 Mb two-strand forward, G8)
"""
import argparse
import os
import sys
import time
import types
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from orca_amd import synth  # noqa: E402
from tests import standins

GOLD = os.path.join(REPO, "tests", "golden")


def _stub_third_party():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    class _Genome(_Dummy):
        # selene's one-hot encoder is not vendored (encoding parity unpinned): process_ins gets the
        # build's definition (A,C,G,T channel order, anything else 0.25 x 4)
        from orca_amd.genome import sequence_to_encoding as _s2e
        sequence_to_encoding = staticmethod(_s2e)

    mod("selene_sdk")
    mod("selene_sdk.sequences", Genome=_Genome)
    mod("selene_sdk.samplers", OnlineSampler=_Dummy)
    mod("selene_sdk.utils", get_indices_and_probabilities=lambda *a, **k: None)
    mod("selene_sdk.targets", Target=_Dummy)
    mod("cooltools")
    mod("cooltools.lib")
    mod("cooltools.lib.numutils", adaptive_coarsegrain=lambda *a, **k: None)
    for n in ("cooler", "pyranges", "pyfaidx", "tabix", "pkg_resources"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                mod(n)
    mod("pygenometracks")
    mod("pygenometracks.plotTracks")
    import matplotlib
    matplotlib.use("Agg")


def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=seed)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    module.eval()
    return module


def stats(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), (a * a).sum(), np.abs(a).max()])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full32m", action="store_true")
    ap.add_argument("--config3", action="store_true", help="G17: rows 0 and 5 of BASELINE config 3 (HFF-shaped model, 8 x 32 Mb), ~10 min of CPU")
    ap.add_argument("--svreal", action="store_true", help="G22-G25: the reference's process_* drivers with the real orca_modules networks (G22-24: 32 Mb, 13-20 min of CPU each; G25: process_del at 256 Mb, ~75 min)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    _stub_third_party()
    import orca_modules as om

    torch.set_num_threads(os.cpu_count())
    only = set(args.only.split(",")) if args.only else None

    def want(name):
        return only is None or name in only

    with torch.no_grad():
        # ---- G0: state-dict key/shape manifest of every hot-path module -----------
        if want("G0"):
            man = {}
            for cls in ("Encoder", "Encoder2", "Encoder2b", "Encoder3", "Decoder", "Decoder_1m", "Net"):
                m = om.Net(num_1d=32) if cls == "Net" else getattr(om, cls)()
                man[cls] = np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()])
            np.savez_compressed(os.path.join(GOLD, "G0_manifest.npz"), **man)
            print("G0 done")

        # ---- G1: Encoder, 3 blocks (first / interior / last), with N runs ---------
        if want("G1"):
            t = time.time()
            enc = load_synth(om.Encoder(), seed=0)
            L = 1712000
            x = torch.from_numpy(synth.synth_sequence(L, seed=11, n_frac=0.01)).transpose(1, 2)
            y = enc(x)[0].numpy()
            # G2: block-size invariance pair
            om.Blocksize = 4000 * 50
            y200 = enc(x)[0].numpy()
            om.Blocksize = 4000 * 200
            # single block, length not a multiple of anything big
            L2 = 4000 * 37
            x2 = torch.from_numpy(synth.synth_sequence(L2, seed=12)).transpose(1, 2)
            y2 = enc(x2)[0].numpy()
            # arbitrary float (non one-hot) input
            rs = np.random.RandomState(13)
            x3 = torch.from_numpy(rs.rand(1, 4, 4000 * 12).astype(np.float32))
            y3 = enc(x3)[0].numpy()
            np.savez_compressed(os.path.join(GOLD, "G1_encoder.npz"), y=y, y_block200k=y200, y_single=y2, y_float=y3)
            print("G1 done %.1fs  max|y-y200|=%.3g  |y|max=%.3g" % (time.time() - t, np.abs(y - y200).max(), np.abs(y).max()))

        # ---- G3/G4: Encoder2 / Encoder3 ------------------------------------------
        if want("G3"):
            e2 = load_synth(om.Encoder2(), seed=0)
            rs = np.random.RandomState(21)
            x = torch.from_numpy((rs.rand(1, 128, 800) * 0.5).astype(np.float32))
            ys = e2(x)
            xl = torch.from_numpy((np.random.RandomState(22).rand(1, 128, 8000) * 0.5).astype(np.float32))
            yl = e2(xl)
            e3 = load_synth(om.Encoder3(), seed=0)
            x3 = torch.from_numpy((np.random.RandomState(23).rand(1, 128, 2000) * 0.5).astype(np.float32))
            y3 = e3(x3)
            d = {f"e2_{i}": y[0].numpy() for i, y in enumerate(ys)}
            d.update({f"e2L_stats_{i}": stats(y.numpy()) for i, y in enumerate(yl)})
            d.update({f"e2L_head_{i}": y[0, :, :16].numpy() for i, y in enumerate(yl)})
            d.update({f"e3_{i}": y[0, :, ::5].numpy() for i, y in enumerate(y3)})
            d.update({f"e3_stats_{i}": stats(y.numpy()) for i, y in enumerate(y3)})
            np.savez_compressed(os.path.join(GOLD, "G3_encoder23.npz"), **d)
            print("G3/G4 done")

        # ---- G5/G6: Decoder (no y / y bilinear / y nearest), Decoder_1m ----------
        if want("G5"):
            nm, _ = synth.synth_normmats_32m()
            x = torch.from_numpy((np.random.RandomState(31).rand(1, 128, 250) * 0.5).astype(np.float32))
            de = torch.log(torch.from_numpy(nm[8][None, None].astype(np.float32)))
            dec = load_synth(om.Decoder(upsample_mode="bilinear"), seed=0)
            p0 = dec(x, de)
            yc = p0[:, :, 37:162, 37:162]
            p1 = dec(x, de, yc)
            decn = load_synth(om.Decoder(upsample_mode="nearest"), seed=0)
            p2 = decn(x, de, yc)
            d1m = load_synth(om.Decoder_1m(), seed=0)
            p3 = d1m(x)
            np.savez_compressed(os.path.join(GOLD, "G5_decoder.npz"), noy=p0[0, 0].numpy(), y_bilinear=p1[0, 0].numpy(),
                                y_nearest=p2[0, 0].numpy(), dec1m=p3[0, 0].numpy())
            print("G5/G6 done; sym err", float((p1 - p1.transpose(2, 3)).abs().max()))

        # ---- G7: genomepredict cascade through the REAL orca_predict -------------
        if want("G7"):
            import orca_predict as op

            class Container(torch.nn.Module):
                def __init__(self, seed):
                    super().__init__()
                    self.net0 = standins.FakeNet0(nbins=8000, seed=seed)
                    self.net = load_synth(om.Encoder2(), seed=seed)
                    self.denets = {lv: load_synth(om.Decoder(upsample_mode="bilinear"), seed=seed + lv)
                                   for lv in (1, 2, 4, 8, 16, 32)}
                    self.denet_1_pt = load_synth(om.Decoder_1m(), seed=seed)
                    self.normmats, self.epss = synth.synth_normmats_32m()

            model = Container(0)
            seq = synth.synth_sequence(320000, seed=41)
            cases = [(16000000 + 1234567, 16000000), (3000000, 16000000), (31500000, 16000000),
                     (50000000 + 9876543, 50000000 + 1)]
            d = {}
            for ci, (mpos, wpos) in enumerate(cases):
                t = time.time()
                out = op.genomepredict(seq, "chrS", mpos, wpos, models=[model], use_cuda=False)
                d[f"c{ci}_args"] = np.array([mpos, wpos], dtype=np.int64)
                d[f"c{ci}_start"] = np.array(out["start_coords"], dtype=np.int64)
                d[f"c{ci}_end"] = np.array(out["end_coords"], dtype=np.int64)
                for j, p in enumerate(out["predictions"][0]):
                    d[f"c{ci}_stats_{j}"] = stats(p)
                    d[f"c{ci}_sub_{j}"] = (p if ci == 0 else p[::5, ::5]).astype(np.float32)
                print("G7 case", ci, "%.1fs" % (time.time() - t), out["start_coords"])
            # targets + annotation bookkeeping on case 0
            tgt = np.abs(np.random.RandomState(42).randn(1, 8000, 8000).astype(np.float32)) * 1e-3
            tgt[0, 100:140, :] = np.nan
            anno = [(0.1, 0.3, "a"), (0.52, "b"), (0.9, 0.95, "c")]
            out = op.genomepredict(seq, "chrS", cases[0][0], cases[0][1], models=[model],
                                   targets=[torch.from_numpy(tgt)], annotation=anno, use_cuda=False)
            for j, e in enumerate(out["experiments"][0]):
                d[f"tgt_sub_{j}"] = np.asarray(e)[::5, ::5].astype(np.float32)
            d["annos_repr"] = np.array([repr([[tuple(float(v) if not isinstance(v, str) else v for v in r) for r in lv]
                                              for lv in out["annos"]])])
            np.savez_compressed(os.path.join(GOLD, "G7_cascade32.npz"), **d)
            print("G7 done")

        # ---- G9: genomepredict_256Mb cascade -------------------------------------
        if want("G9"):
            import orca_predict as op

            class Container256(torch.nn.Module):
                def __init__(self, seed):
                    super().__init__()
                    self.net0 = standins.FakeNet0(nbins=64000, seed=seed)
                    self.net1 = load_synth(om.Encoder2(), seed=seed)
                    self.net = load_synth(om.Encoder3(), seed=seed)
                    self.denets = {lv: load_synth(om.Decoder(upsample_mode="bilinear"), seed=seed + lv)
                                   for lv in (32, 64, 128, 256)}

            model = Container256(0)
            seq = synth.synth_sequence(512000, seed=51)
            chrlen = 138368000
            d = {}
            for ci, (mpos, wpos) in enumerate([(70000000, 128000000), (130000000, 128000000), (5000000, 128000000)]):
                nm = synth.synth_normmat_256m(chrlen, seed=0)
                out = op.genomepredict_256Mb(seq, "chrS", [nm], chrlen, mpos, wpos, models=[model],
                                             padding_chr="chrP", use_cuda=False)
                d[f"c{ci}_args"] = np.array([mpos, wpos, chrlen], dtype=np.int64)
                d[f"c{ci}_start"] = np.array(out["start_coords"], dtype=np.int64)
                d[f"c{ci}_end"] = np.array([int(v) for v in out["end_coords"]], dtype=np.int64)
                for j, p in enumerate(out["predictions"][0]):
                    d[f"c{ci}_stats_{j}"] = stats(p)
                    d[f"c{ci}_sub_{j}"] = (p if ci == 0 else p[::5, ::5]).astype(np.float32)
                print("G9 case", ci, out["start_coords"])
            np.savez_compressed(os.path.join(GOLD, "G9_cascade256.npz"), **d)
            print("G9 done")

        # ---- G10: coordinate helpers of the SV drivers (orca_utils.py:1009-1060) --
        if want("G10"):
            import orca_utils as ou
            rs = np.random.RandomState(7)
            rows = []
            for _ in range(400):
                chrlen = int(rs.randint(33_000_000, 250_000_000))
                pos = int(rs.randint(0, chrlen))
                if rs.rand() < 0.2:
                    pos = int(rs.choice([rs.randint(0, 200000), chrlen - rs.randint(0, 200000)]))
                rows.append((pos, chrlen, int(ou.coord_clip(pos, chrlen)), int(ou.coord_round(pos)),
                             int(ou.coord_clip(pos, max(chrlen, 260_000_000), binsize=1024000, window_radius=128000000))))
            np.savez_compressed(os.path.join(GOLD, "G10_coords.npz"), rows=np.array(rows, dtype=np.int64))
            print("G10 done")

        # ---- G11: structural-variant drivers through the REAL orca_predict.process_* ---------
        if want("G11"):
            import orca_predict as op
            genome = synth.sv_driver_genome()
            op.model_dict_global["h1esc"] = standins.FakeModel32(0)
            op.model_dict_global["hff"] = standins.FakeModel32(1)
            d = {}
            for name, fn, a, kw in synth.sv_driver_cases():
                t = time.time()
                # process_ins forgets `models=` in its alt.r call and falls back to the default pair: run that case
                # with the default pair throughout so that every view uses the same models
                cm = None if fn == "process_ins" else [op.model_dict_global["h1esc"]]
                outs = getattr(op, fn)(*a, genome, custom_models=cm, target=False, use_cuda=False, **kw)
                for k, v in synth.summarize_outputs(outs).items():
                    d[f"{name}.{k}"] = v
                print("G11", name, "%.1fs" % (time.time() - t))
            np.savez_compressed(os.path.join(GOLD, "G11_sv_drivers.npz"), **d)
            print("G11 done")

        # ---- G19: process_seqstr through the REAL orca_predict.process_seqstr (the `seqstr` package replaced by a stand-in) ----
        if want("G19"):
            import orca_predict as op
            op.seqstr = standins.FakeSeqstr()
            h1 = standins.FakeModel32(0)
            d = {}
            for name, spec, mpos in synth.seqstr_cases():
                t = time.time()
                out = op.process_seqstr(spec, mpos=mpos, custom_models=[h1], use_cuda=False)
                for k, v in synth.summarize_outputs(out).items():
                    d[f"{name}.{k}"] = v
                print("G19", name, "%.1fs" % (time.time() - t))
            np.savez_compressed(os.path.join(GOLD, "G19_seqstr.npz"), **d)
            print("G19 done")

        # ---- G15: Encoder2b (orca_modules.py:1173-1276), the HCTnoc variant of Encoder2 ----------------------
        if want("G15"):
            e2b = load_synth(om.Encoder2b(), seed=0)
            x = torch.from_numpy((np.random.RandomState(33).rand(1, 128, 2048) * 0.5).astype(np.float32))
            outs = e2b(x)
            np.savez_compressed(os.path.join(GOLD, "G15_encoder2b.npz"), **{f"o{i}": o[0].numpy() for i, o in enumerate(outs) if i > 0})
            print("G15 done", [tuple(o.shape) for o in outs])

        # ---- G16: multi-target decoders of orca_leukemia.py (Decoder(num_2d) :512-990, Decoder_1m(num_2d) :996-1316,
        # Encoder2 :1499-1601 = the up-path-only U-net).  The module cannot be imported as a whole - it needs ORCA_PATH
        # set before line 10 and instantiates OrcaLeukemiaA/B (checkpoint + resource files) at import, :1872-1873 - so
        # the class definitions above `class OrcaLeukemiaA` are executed in a scratch module.
        if want("G16"):
            import types
            src = open(os.path.join(REF, "orca_leukemia.py")).read()
            ol = types.ModuleType("orca_leukemia_classes")
            ol.__dict__["ORCA_PATH"] = REF
            exec(compile(src[:src.index("class OrcaLeukemiaA")], os.path.join(REF, "orca_leukemia.py"), "exec"), ol.__dict__)
            man = {}
            for name, m in (("Decoder2", ol.Decoder(2)), ("Decoder6", ol.Decoder(6)), ("Decoder_1m2", ol.Decoder_1m(2)),
                            ("Encoder2", ol.Encoder2()), ("Net2_3", ol.Net(num_2d=2, num_1d=3))):
                man[name] = np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()])
            d = {f"manifest_{k}": v for k, v in man.items()}
            nm, _ = synth.synth_normmats_32m()
            x = torch.from_numpy((np.random.RandomState(71).rand(1, 128, 250) * 0.5).astype(np.float32))
            for T, lv in ((2, 8), (6, 2)):
                bg = np.stack([nm[lv] * (1.0 + 0.15 * t) for t in range(T)])
                de = torch.log(torch.from_numpy(bg[None].astype(np.float32)))
                dec = load_synth(ol.Decoder(T), seed=5)
                p0 = dec(x, de)
                yc = p0[:, :, 29:154, 29:154]
                p1 = dec(x, de, yc)
                if T == 2:
                    d["T2_noy"], d["T2_y"] = p0[0].numpy(), p1[0].numpy()
                else:   # every 3rd row/column + whole-map statistics keep the fixture small
                    d[f"T{T}_noy_sub"], d[f"T{T}_y_sub"] = p0[0, :, ::3, ::3].numpy(), p1[0, :, ::3, ::3].numpy()
                    d[f"T{T}_noy_stats"], d[f"T{T}_y_stats"] = stats(p0), stats(p1)
            d1m = load_synth(ol.Decoder_1m(2), seed=5)
            d["T2_dec1m"] = d1m(x)[0].numpy()
            np.savez_compressed(os.path.join(GOLD, "G16_multitarget.npz"), **d)
            print("G16 done", {k: v.shape for k, v in d.items() if not k.startswith("manifest")})

        # ---- G14: Net, the 1 Mb model (orca_modules.py:1409-1900), one 1 Mb sequence with an N run ----------
        if want("G14"):
            t = time.time()
            net = load_synth(om.Net(num_1d=32), seed=0)
            seq = synth.synth_sequence(1_000_000, seed=61, n_frac=0.002)
            x = torch.from_numpy(seq).transpose(1, 2)
            pred, out1d = net(x)
            net0 = load_synth(om.Net(), seed=3)
            pred0 = net0(x)
            np.savez_compressed(os.path.join(GOLD, "G14_net1m.npz"), pred=pred[0, 0].numpy(), out1d=out1d[0].numpy(),
                                pred_no1d_stats=stats(pred0), pred_no1d_sub=pred0[0, 0, ::5, ::5].numpy())
            print("G14 done %.1fs" % (time.time() - t), float(pred.abs().max()), float(out1d.min()), float(out1d.max()))

        # ---- G13: the 256 Mb structural-variant views through the REAL process_* (model forward replaced by a recorder)
        if want("G13"):
            import orca_predict as op
            genome = synth.sv_driver_genome_256()
            op.h1esc_256m, op.hff_256m = standins.Background256(0), standins.Background256(1)   # what _retrieve_multi reads
            rec = standins.Recorder256()
            op.genomepredict_256Mb = rec
            op.genomeplot_256Mb = lambda *a, **k: None       # process_dup needs a file name at 256 Mb (:1365-1367)
            op.target_dict_global["fake"] = type("T", (standins.FakeTarget256, op.Genomic2DFeatures), {"__init__": lambda self: None})()
            d = {}
            for name, fn, a, kw in synth.sv_driver_cases_256():
                t = time.time()
                first = len(rec.calls)
                extra = {"file": "/tmp/g13"} if fn == "process_dup" else {}
                tgt = ["fake"] if fn == "process_del" else False   # process_del cannot run without targets at 256 Mb
                outs = getattr(op, fn)(*a, genome, custom_models=[object(), object()], target=tgt, use_cuda=False,
                                       window_radius=128000000, padding_chr="chr1", **extra, **kw)
                d[f"{name}.order"] = np.array([o["call"] - first for o in outs])
                for k, v in rec.summary(first).items():
                    d[f"{name}.{k}"] = v
                print("G13", name, len(rec.calls) - first, "views", "%.1fs" % (time.time() - t))
            np.savez_compressed(os.path.join(GOLD, "G13_sv_drivers_256.npz"), **d)
            print("G13 done")

        # ---- G12: StructuralChange2 edit scripts (orca_utils.py:737-965) -------------------------
        if want("G12"):
            import orca_utils as ou
            rs = np.random.RandomState(12)
            scripts = []
            for trial in range(120):
                L = int(rs.randint(1000, 5000))
                sc = ou.StructuralChange2("chrA", L)
                ops = []
                if trial % 5 == 0:
                    sc2 = ou.StructuralChange2("chrB", L // 2)
                    sc2.invert(3, L // 4)
                    sc = sc + sc2
                    ops.append(["concat", L // 2, 3, L // 4])
                for _ in range(rs.randint(1, 7)):
                    n = sc.coord_points[-1]
                    if n < 10:
                        break
                    s0 = int(rs.randint(0, n - 2)); e0 = int(rs.randint(s0 + 1, n + 1))
                    kind = ["delete", "duplicate", "invert", "insert"][rs.randint(4)]
                    if kind == "insert":
                        st = "+-"[rs.randint(2)]
                        sc.insert(s0, e0 - s0, strand=st)
                        ops.append([kind, s0, e0 - s0, st])
                    else:
                        getattr(sc, kind)(s0, e0)
                        ops.append([kind, s0, e0])
                n = sc.coord_points[-1]
                queries = []
                for _ in range(6):
                    if n < 3:
                        break
                    s0 = int(rs.randint(0, n - 1)); e0 = int(rs.randint(s0 + 1, n + 2))
                    try:
                        res = [list(x) for x in sc[s0:e0]]
                    except ValueError:
                        res = "ValueError"
                    rc, cc = sc.query_ref("chrA", s0, e0)
                    queries.append({"q": [s0, e0], "pieces": res, "ref": [[int(v) for v in r] for r in rc],
                                    "cur": [[int(r[0]), int(r[1]), r[2]] for r in cc]})
                scripts.append({"L": L, "ops": ops, "points": list(sc.coord_points),
                                "segments": [[g.len] + list(g.ref) for g in sc.segments], "queries": queries})
            import json
            with open(os.path.join(GOLD, "G12_structural_change.json"), "w") as f:
                json.dump(scripts, f)
            print("G12 done", len(scripts))

        class Full(torch.nn.Module):
            """The reference's own sub-networks (orca_modules) in the container shape genomepredict expects, synthetic weights."""

            def __init__(self, seed):
                super().__init__()
                self.net0 = load_synth(om.Encoder(), seed=seed)
                self.net = load_synth(om.Encoder2(), seed=seed)
                self.denets = {lv: load_synth(om.Decoder(upsample_mode="bilinear"), seed=seed + lv)
                               for lv in (1, 2, 4, 8, 16, 32)}
                self.denet_1_pt = load_synth(om.Decoder_1m(), seed=seed)
                self.normmats, self.epss = synth.synth_normmats_32m()

        # ---- G22: ONE structural-variant driver of the reference - process_del - with the REAL networks at full size (three
        #      genomepredict calls: both reference-allele views and the alternative allele; ~13 min of CPU) ------------------
        if args.svreal and want("G22"):
            import orca_predict as op
            genome = synth.sv_driver_genome()
            t = time.time()
            outs = op.process_del(*synth.SV_REAL_CASE, genome, custom_models=[Full(0)], target=False, use_cuda=False)
            d = {f"del.{k}": v for k, v in synth.summarize_outputs(outs, stride=5).items()}
            d["t_cpu_s"] = np.array([time.time() - t])
            np.savez_compressed(os.path.join(GOLD, "G22_sv_del_real_nets.npz"), **d)
            print("G22 done %.1fs" % (time.time() - t), len(outs), "views")

        # ---- G23: process_dup and process_inv of the reference with the REAL networks (3 + 4 genomepredict calls, ~17 min of CPU):
        #      the alternative alleles whose windows repeat a segment / take it from the other strand -------------------------------
        if args.svreal and want("G23"):
            import orca_predict as op
            genome = synth.sv_driver_genome()
            t = time.time()
            d = {}
            model = Full(0)
            for name, fn, a in synth.SV_REAL_CASES_G23:
                t1 = time.time()
                outs = getattr(op, fn)(*a, genome, custom_models=[model], target=False, use_cuda=False)
                d.update({f"{name}.{k}": v for k, v in synth.summarize_outputs(outs, stride=5).items()})
                d[f"{name}.t_cpu_s"] = np.array([time.time() - t1])
                print("G23", name, len(outs), "views %.1fs" % (time.time() - t1), flush=True)
            np.savez_compressed(os.path.join(GOLD, "G23_sv_dup_inv_real_nets.npz"), **d)
            print("G23 done %.1fs" % (time.time() - t))

        # ---- G24: process_ins / process_single_breakpoint / process_custom of the reference with the REAL networks (3 + 3 + 2 genomepredict
        #      calls, ~20 min of CPU): alternative alleles with an inserted string, pieces of two chromosomes, a '-' piece ---------------------
        if args.svreal and want("G24"):
            import orca_predict as op
            genome = synth.sv_driver_genome()
            t = time.time()
            d = {}
            model = Full(0)
            # (the reference's process_ins forgets `models=` in its alt.r call, orca_predict.py:2474, and falls back to the registered default
            # pair: both names get the same real model, the fixture's alt.r view then carries it twice - the test compares model 0)
            op.model_dict_global["h1esc"] = op.model_dict_global["hff"] = model
            path24 = os.path.join(GOLD, "G24_sv_ins_bp_custom_real_nets.npz")
            if os.path.exists(path24):                      # resumable: the file is rewritten after every case, finished cases are kept
                d.update({k: v for k, v in np.load(path24).items()})
            for name, fn, a, kw in synth.sv_real_cases_g24():
                if f"{name}.t_cpu_s" in d:
                    continue
                t1 = time.time()
                outs = getattr(op, fn)(*a, genome, custom_models=[model], target=False, use_cuda=False, **kw)
                d.update({f"{name}.{k}": v for k, v in synth.summarize_outputs(outs, stride=5).items()})
                d[f"{name}.t_cpu_s"] = np.array([time.time() - t1])
                print("G24", name, len(outs), "views %.1fs" % (time.time() - t1), flush=True)
                np.savez_compressed(path24, **d)
            print("G24 done %.1fs" % (time.time() - t))

        # ---- G25: a 256 Mb structural-variant driver of the reference - process_del with window_radius=128000000 - with the REAL networks
        #      (orca_modules Encoder / Encoder2 / Encoder3 / four Decoders): three genomepredict_256Mb calls, ~25 min of CPU each.  Resumable
        #      view by view (every finished genomepredict_256Mb call is kept under /tmp/g25_view<i>.pkl) ------------------------------------------
        if args.svreal and want("G25"):
            import pickle
            import orca_predict as op
            genome = synth.sv_driver_genome_256()
            op.h1esc_256m, op.hff_256m = standins.Background256(0), standins.Background256(1)   # what _retrieve_multi reads
            op.target_dict_global["fake"] = type("T", (standins.FakeTarget256, op.Genomic2DFeatures), {"__init__": lambda self: None})()

            class Ref256(torch.nn.Module):
                def __init__(self, seed):
                    super().__init__()
                    self.net0 = load_synth(om.Encoder(), seed=seed)
                    self.net1 = load_synth(om.Encoder2(), seed=seed)
                    self.net = load_synth(om.Encoder3(), seed=seed)
                    self.denets = {lv: load_synth(om.Decoder(upsample_mode="bilinear"), seed=seed + lv) for lv in (32, 64, 128, 256)}

            real, ncall, case_ = op.genomepredict_256Mb, [0], [""]

            def kept(*a, **k):
                path = "/tmp/g25_%s_view%d.pkl" % (case_[0], ncall[0])
                ncall[0] += 1
                if os.path.exists(path):
                    return pickle.load(open(path, "rb"))
                t1 = time.time()
                out = real(*a, **k)
                pickle.dump(out, open(path, "wb"))
                print("G25", case_[0], "view", ncall[0] - 1, "%.1fs" % (time.time() - t1), flush=True)
                return out

            op.genomepredict_256Mb = kept
            path25 = os.path.join(GOLD, "G25_sv_del256_real_nets.npz")
            d = {k: v for k, v in np.load(path25).items()} if os.path.exists(path25) else {}
            model = Ref256(0)
            # $G25_CASES: names of synth.sv_driver_cases_256() (default del256; inv256 = four views, the inverted piece from the other strand)
            for name, fn, a, kw in synth.sv_driver_cases_256():
                if name not in os.environ.get("G25_CASES", "del256").split(",") or f"{name}.t_cpu_s" in d:
                    continue
                t = time.time()
                case_[0], ncall[0] = name, 0
                tgt = ["fake"] if fn == "process_del" else False     # the reference's process_del cannot run without targets at 256 Mb
                outs = getattr(op, fn)(*a, genome, custom_models=[model], target=tgt, use_cuda=False, window_radius=128000000, padding_chr="chr1", **kw)
                d.update({f"{name}.{k}": v for k, v in synth.summarize_outputs(outs, stride=5).items()})
                d[f"{name}.t_cpu_s"] = np.array([time.time() - t])
                np.savez_compressed(path25 + ".tmp.npz", **d)
                os.replace(path25 + ".tmp.npz", path25)          # (atomic: an interrupted run never leaves half a fixture)
                print("G25", name, "done %.1fs" % (time.time() - t), len(outs), "views", flush=True)
            op.genomepredict_256Mb = real

        # ---- G8: one full 32 Mb H1-ESC-shaped forward, both strands ---------------
        if args.full32m and want("G8"):
            import orca_predict as op

            model = Full(0)
            seq = synth.synth_sequence(32000000, seed=1)
            t = time.time()
            enc_f = model.net0(torch.from_numpy(seq).transpose(1, 2))
            t_enc = time.time() - t
            out = op.genomepredict(seq, "chrS", 16000000 + 1234567, 16000000, models=[model], use_cuda=False)
            d = {"enc_fwd": enc_f[0].numpy(), "t_encoder_cpu_s": np.array([t_enc]),
                 "t_total_cpu_s": np.array([time.time() - t]), "ncores": np.array([os.cpu_count()]),
                 "start": np.array(out["start_coords"], dtype=np.int64)}
            for j, p in enumerate(out["predictions"][0]):
                d[f"pred_{j}"] = p.astype(np.float32)
            np.savez_compressed(os.path.join(GOLD, "G8_full32m.npz"), **d)
            print("G8 done: encoder %.1fs, total %.1fs" % (t_enc, time.time() - t))


CONFIG3 = {"seed": 7, "rows": (0, 5), "row_seed0": 10, "L": 32_000_000, "mpos": 17_234_567, "wpos": 16_000_000}


def config3_golden(om):
    """G17 - BASELINE.json configs[2]: HFF-shaped 32 Mb model, batch of 8 random 32 Mb sequences, module-level forward
    (net0 -> net -> six Decoder levels + denet_1_pt, forward strand; `genomepredict` itself keeps only row 0,
    orca_predict.py:514-523).  Rows 0 and 5 through the REFERENCE's nn.Modules on CPU; the coarse-to-fine bookkeeping
    is the build's run_cascade (pinned to the reference's genomepredict by G7)."""
    from orca_amd import orca_predict as P
    c = CONFIG3
    s = c["seed"]

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net0 = load_synth(om.Encoder(), seed=s)
            self.net = load_synth(om.Encoder2(), seed=s)
            self.denets = {lv: load_synth(om.Decoder(upsample_mode="bilinear"), seed=s + lv) for lv in (1, 2, 4, 8, 16, 32)}
            self.denet_1_pt = load_synth(om.Decoder_1m(), seed=s)
            e = synth.synth_expected_log(8000, s)
            idx = np.abs(np.arange(8000)[None, :] - np.arange(8000)[:, None])
            nm = np.exp(e[idx])
            self.normmats = {lv: np.reshape(nm[: 250 * lv, : 250 * lv], (250, lv, 250, lv)).mean(axis=1).mean(axis=2) for lv in (1, 2, 4, 8, 16, 32)}

    model = Ref()
    d = {}
    for b in c["rows"]:
        t = time.time()
        codes = synth.synth_base_codes(c["L"], seed=c["row_seed0"] + b)
        x = torch.zeros(1, 4, c["L"])
        x[0, torch.from_numpy(codes.astype(np.int64)), torch.arange(c["L"])] = 1.0
        enc0 = model.net0(x)
        encs = dict(zip([1, 2, 4, 8, 16, 32], model.net(enc0)))
        de = {lv: torch.log(torch.from_numpy(model.normmats[lv][None, None].astype(np.float32))) for lv in model.normmats}
        preds, starts = P.run_cascade(model, encs, [32, 16, 8, 4, 2, 1], lambda lv: lv, 1, [False], lambda lv, k, st: de[lv],
                                      lambda lv, st, rev: P.zoom_index_32m(lv, st, c["mpos"], c["wpos"], rev), add_1m_level=1)
        e0 = enc0[0].numpy()     # 4 MB per row: keep the two ends and the moments
        d[f"enc0_first64_row{b}"], d[f"enc0_last64_row{b}"], d[f"enc0_stats_row{b}"] = e0[:, :64].copy(), e0[:, -64:].copy(), stats(e0)
        d[f"maps_row{b}"] = np.stack([p[0, 0].numpy() for p in preds]).astype(np.float32)
        d[f"starts_row{b}"] = np.array(starts[0], dtype=np.int64)
        print("G17 row %d done in %.1fs" % (b, time.time() - t), flush=True)
    np.savez_compressed(os.path.join(GOLD, "G17_config3.npz"), **d)


def coarsegrain_golden():
    """G18: the reference's own `_adaptive_coarsegrain(cuda=True)` -> `adaptive_coarsegrain_gpu` (selene_utils2.py:274-504) on
    synthetic observed Hi-C blocks.  The function hard-codes the CUDA default tensor type (:346) and uses `np.int`
    (:397, removed from numpy >= 1.24): both are patched IN THIS PROCESS ONLY so that it runs on the CPU here."""
    _stub_third_party()
    np.int = int
    torch.set_default_tensor_type = lambda *a, **k: None
    import selene_utils2 as su
    d = {}
    for name, n, m, seed in synth.COARSEGRAIN_CASES:
        a, c = synth.synth_hic(n, seed, m=m)
        out = su._adaptive_coarsegrain(a.copy(), c.copy(), cuda=True).astype(np.float32)
        if n * m > 100000:      # large case: moments + a corner + the NaN pattern's size
            fin = np.isfinite(out)
            d[name + "_stats"] = np.array([out[fin].astype(np.float64).sum(), (out[fin].astype(np.float64) ** 2).sum(), float((~fin).sum())])
            d[name + "_corner"] = out[:64, :64].copy()
        else:
            d[name] = out
        print("G18", name, out.shape, int(np.isnan(out).sum()), flush=True)
    np.savez_compressed(os.path.join(GOLD, "G18_coarsegrain.npz"), **d)


GENOME26 = {"seed": 26, "chroms": (("chr1", 100_003), ("chr10", 1_237), ("chr2", 64), ("chrX", 7), ("chrM", 20_000))}


def genome26_strings():
    """The synthetic multi-chromosome FASTA text behind G26: upper and lower case bases, N runs, IUPAC symbols."""
    rs = np.random.RandomState(GENOME26["seed"])
    recs = {}
    for name, n in GENOME26["chroms"]:
        s = rs.choice(list("ACGTacgt"), size=n, p=[0.2] * 4 + [0.05] * 4)
        for _ in range(max(1, n // 9000)):
            a = int(rs.randint(0, max(1, n - 3)))
            s[a:a + int(rs.randint(1, max(2, min(700, n // 3))))] = "N"
        for a in rs.randint(0, n, max(1, n // 400)):
            s[a] = rs.choice(list("RYnKMswb"))
        recs[name] = "".join(s)
    recs["chr1"] = "N" * 11 + recs["chr1"][11:-5] + "nnnnn"      # N at both chromosome ends
    return recs


def genome_golden():
    """G26 - the reference's genome store itself: `selene_utils2.MemmapGenome.get_encoding_from_coords` /
    `get_encoding_from_coords_check_unk` (selene_utils2.py:186-272) on a synthetic multi-chromosome genome.  The object is
    made with `object.__new__` and given the state `_unpicklable_init` (:97-157) leaves behind - `sequence_data` [4, total]
    float32 in sorted-chromosome order, `inds`, `len_chrs`, `initialized` - because pyfaidx and selene are not installed;
    the one-hot columns come from the build's encoder (selene's `sequence_to_encoding` is not vendored: SURVEY 8c).
    Everything the QUERY does - slicing, 0.25 padding, the '-' strand as `[::-1, ::-1]`, the asserts, `pad=strand` in
    `_check_unk`, the first-row unknown test - is the reference's own code."""
    _stub_third_party()
    import selene_utils2 as su
    from orca_amd.genome import sequence_to_encoding
    recs = genome26_strings()
    g = object.__new__(su.MemmapGenome)
    g.chrs = sorted(recs)
    g.len_chrs = {c: len(recs[c]) for c in g.chrs}
    g.lens = np.array([g.len_chrs[c] for c in g.chrs])
    g.inds = {c: ind for c, ind in zip(g.chrs, np.concatenate([[0], np.cumsum(g.lens)]))}
    g.sequence_data = np.zeros((4, int(g.lens.sum())), dtype=np.float32)
    for c in g.chrs:
        g.sequence_data[:, g.inds[c]: g.inds[c] + g.len_chrs[c]] = sequence_to_encoding(recs[c]).T
    g.initialized = True

    rs = np.random.RandomState(GENOME26["seed"] + 1)
    queries = []
    for c in g.chrs:
        n = g.len_chrs[c]
        span = min(n, 300)
        fixed = [(0, n), (0, 0), (n, n), (0, min(n, 5)), (max(0, n - 5), n), (-7, min(n, 4)), (max(0, n - 3), n + 9), (-4, n + 4),
                 (-10, 0), (n, n + 10), (-10, -2), (n + 2, n + 12), (n + 1, n + 1), (-3, -3), (5, 2), (min(n, 3), min(n, 3)),
                 (-1, 1), (n - 1, n + 1)]
        rnd = []
        for _ in range(24):
            a = int(rs.randint(-span // 4 - 2, n + span // 4 + 2))
            rnd.append((a, a + int(rs.randint(0, span + 1))))
        for (a, b) in fixed + rnd:
            for strand in ("+", "-", "."):
                for pad in (False, True):
                    queries.append((c, int(a), int(b), strand, pad))
    # Encoder-sized windows (whole 4 kb bins, N runs inside) for the -m gpu comparison of the HBM-resident stores and of
    # `Encoder.forward_2bit` with the Encoder on the reference's float rows
    for (a, b) in ((0, 96_000), (4_003, 100_003), (-2_000, 22_000), (80_003, 104_003)):
        for strand in ("+", "-"):
            queries.append(("chr1", a, b, strand, True))
    queries.append(("chrM", 0, 20_000, "-", False))

    d = {"chrs": np.array(g.chrs), "q_chrom": np.array([q[0] for q in queries]), "q_start": np.array([q[1] for q in queries], dtype=np.int64),
         "q_end": np.array([q[2] for q in queries], dtype=np.int64), "q_strand": np.array([q[3] for q in queries]),
         "q_pad": np.array([q[4] for q in queries], dtype=bool)}
    for c in g.chrs:
        d["seq_" + c] = np.array(recs[c])
    status, dtypes, rows, offs, unk_status, unk_flag, unk_same = [], [], [], [0], [], [], []
    for (c, a, b, strand, pad) in queries:
        try:
            enc = g.get_encoding_from_coords(c, a, b, strand=strand, pad=pad)
            assert enc.shape == (b - a, 4)
            status.append("ok"); dtypes.append(str(enc.dtype)); rows.append(np.asarray(enc, dtype=np.float32))
            assert np.array_equal(rows[-1].astype(enc.dtype), enc)
        except AssertionError:
            status.append("AssertionError"); dtypes.append(""); rows.append(np.zeros((0, 4), np.float32))
        except Exception as e:       # anything else the reference raises is recorded by type
            status.append(type(e).__name__); dtypes.append(""); rows.append(np.zeros((0, 4), np.float32))
        offs.append(offs[-1] + rows[-1].shape[0])
        try:      # the check_unk form ignores the caller's pad (pad=strand is always true: selene_utils2.py:271)
            enc2, unk = g.get_encoding_from_coords_check_unk(c, a, b, strand=strand, pad=pad)
            unk_status.append("ok"); unk_flag.append(bool(unk))
            ref_pad = g.get_encoding_from_coords(c, a, b, strand=strand, pad=True)
            unk_same.append(bool(np.array_equal(enc2, ref_pad)))
        except AssertionError:
            unk_status.append("AssertionError"); unk_flag.append(False); unk_same.append(True)
        except Exception as e:
            unk_status.append(type(e).__name__); unk_flag.append(False); unk_same.append(True)
    assert all(unk_same)
    d.update(status=np.array(status), dtype=np.array(dtypes), rows=np.concatenate(rows, axis=0), row_offsets=np.array(offs, dtype=np.int64),
             unk_status=np.array(unk_status), unk_flag=np.array(unk_flag, dtype=bool))
    np.savez_compressed(os.path.join(GOLD, "G26_genome.npz"), **d)
    from collections import Counter
    print("G26 done:", len(queries), "queries;", Counter(status), Counter(unk_status), Counter(dtypes), "rows", d["rows"].shape)


FULL256 = {"seed": 0, "seq_seed": 2, "L": 256_000_000, "chrlen": 138_368_000, "mpos": 70_000_000, "wpos": 128_000_000}


def full256_sample_bins(nb=64000):
    """Bins of the [128, 64000] Encoder output kept in G20: both ends, every 125th, and +-8 around each multiple of
    8000 (the product's 32 Mb chunk seams)."""
    idx = set(range(64)) | set(range(nb - 64, nb)) | set(range(0, nb, 125))
    for k in range(1, nb // 8000):
        idx |= set(range(k * 8000 - 8, k * 8000 + 8))
    return np.array(sorted(idx), dtype=np.int64)


def full256_golden(om, threads):
    """G20 - BASELINE.json configs[3] at FULL size through the reference's own code: `genomepredict_256Mb`
    (orca_predict.py:543-878) with the REAL `net0` = `orca_modules.Encoder` (:803-980) on a seeded 256 Mb sequence, both
    strands; Encoder2(64 000 bins) -> [-1] -> Encoder3 -> four Decoders.  Stored: the four maps, start/end coords, and of
    each strand's [128, 64000] Encoder output a column sample + moments.  ~20 min of CPU on 8 cores."""
    import orca_predict as op
    c = FULL256
    torch.set_num_threads(threads)

    class Rec(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner, self.outs = inner, []

        def forward(self, x):
            y = self.inner(x)
            self.outs.append(y[0].numpy().copy())
            return y

    class Ref256(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            self.net0 = Rec(load_synth(om.Encoder(), seed=seed))
            self.net1 = load_synth(om.Encoder2(), seed=seed)
            self.net = load_synth(om.Encoder3(), seed=seed)
            self.denets = {lv: load_synth(om.Decoder(upsample_mode="bilinear"), seed=seed + lv) for lv in (32, 64, 128, 256)}

    model = Ref256(c["seed"])
    t = time.time()
    seq = synth.synth_sequence(c["L"], seed=c["seq_seed"])
    nm = synth.synth_normmat_256m(c["chrlen"], seed=0)
    print("G20 inputs ready %.1fs" % (time.time() - t), flush=True)
    out = op.genomepredict_256Mb(seq, "chrS", [nm], c["chrlen"], c["mpos"], c["wpos"], models=[model], padding_chr="chrP", use_cuda=False)
    d = {"args": np.array([c["mpos"], c["wpos"], c["chrlen"], c["L"], c["seq_seed"], c["seed"]], dtype=np.int64),
         "start": np.array(out["start_coords"], dtype=np.int64), "end": np.array([int(v) for v in out["end_coords"]], dtype=np.int64),
         "t_total_cpu_s": np.array([time.time() - t]), "ncores": np.array([threads]), "bins": full256_sample_bins()}
    for j, p in enumerate(out["predictions"][0]):
        d[f"pred_{j}"] = p.astype(np.float32)
    assert len(model.net0.outs) == 2
    for k, e in enumerate(model.net0.outs):      # 0 = forward strand, 1 = reverse complement
        d[f"enc_{k}_cols"] = e[:, d["bins"]].copy()
        d[f"enc_{k}_stats"] = stats(e)
        e64 = e.astype(np.float64)
        d[f"enc_{k}_chan_sum"] = e64.sum(axis=1)
        d[f"enc_{k}_chan_sq"] = (e64 * e64).sum(axis=1)
    np.savez_compressed(os.path.join(GOLD, "G20_full256m.npz"), **d)
    print("G20 done in %.1fs" % (time.time() - t), out["start_coords"], flush=True)


if __name__ == "__main__":
    if "--coarsegrain" in sys.argv:
        coarsegrain_golden()
    elif "--genome" in sys.argv:
        genome_golden()
    elif "--full256m" in sys.argv:
        _stub_third_party()
        import orca_modules as _om
        with torch.no_grad():
            full256_golden(_om, int(os.environ.get("GOLDEN_THREADS", os.cpu_count())))
    elif "--config3" in sys.argv:
        _stub_third_party()
        import orca_modules as _om
        torch.set_num_threads(os.cpu_count())
        with torch.no_grad():
            config3_golden(_om)
    else:
        main()
