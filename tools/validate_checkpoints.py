"""Validation of the default arithmetic on REAL checkpoints - for users who have the Zenodo files of the reference
(/root/reference/README.md:63-72; layout /root/reference/orca_models.py:53-123).  Offline builders have only synthetic weights: the fp16
split ("f16x2", 22 significant bits, range 65 504) has never seen trained activations.  This tool closes that on the user's machine:

    python tools/validate_checkpoints.py --orca-path /path/to/orca            # expects <path>/models/*.statedict, <path>/resources/*.npy
    python tools/validate_checkpoints.py --write-synthetic /tmp/ck --orca-path /tmp/ck     # self-test: writes the reference's layout first

What it reports (one JSON document on stdout, human-readable lines on stderr):
 1. load: every sub-network of H1esc / Hff (and the 256 Mb containers when their files exist) through `orca_amd.orca_models`' own loader;
 2. weights: per folded conv (BatchNorm folded in float64, as the engine uploads it) max |w| and max |b| against the fp16 range;
 3. activations: a LAYER-BY-LAYER fp32 walk of the Encoder (on --sample-bp bases), Encoder2 and every Decoder level + Decoder_1m on the
    exact-fp32 kernels (`orca_conv1d_forward` / `orca_conv2d_forward`), max |activation| per layer against 65 504 - the number the f16x2
    guard compares with on the device;
 4. cascade: `genomepredict` on a 32 Mb sequence (random, or --sequence file.npy of base codes 0..4) in the default f16x2 arithmetic and in
    precision="f32", both strands, all six levels: max-abs and Pearson per level, and whether the device range guard fired (= whether the
    default silently fell back to bf16x3 / f32 for a module).
Exit code 0 when every level agrees to --tol (default 1e-4, the north star) and nothing exceeded the fp16 range, 1 otherwise.
Steps 3-4 need the MI355X; --no-gpu stops after step 2."""
import argparse
import collections
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from orca_amd import engine, orca_models as M, orca_modules as pm, synth

F16_MAX = 65504.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def write_synthetic(root):
    """Checkpoint + resource files in the reference's layout (DataParallel prefixes, the double-prefixed stage-a `net0` dict) from the
    synthetic weights - what tests/test_checkpoints_cpu.py writes, for both cell types."""
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    os.makedirs(os.path.join(root, "resources"), exist_ok=True)

    def sd(module, seed):
        shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
        return {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state_dict(shapes, seed=seed).items()}

    def save(name, d, prefix):
        torch.save(collections.OrderedDict((prefix + k, v) for k, v in d.items()), os.path.join(root, "models", name))

    for cell, seed, res in (("h1esc", 0, "4DNFI9GMP2J8"), ("hff", 1, "4DNFI643OYP9")):
        save(f"orca_{cell}.net0.statedict", sd(pm.Net(num_1d=32), seed), "module.module.")
        save(f"orca_{cell}.net.statedict", sd(pm.Encoder2(), seed), "module.")
        for lv in (1, 2, 4, 8, 16, 32):
            save(f"orca_{cell}.d{lv}.statedict", sd(pm.Decoder(upsample_mode="bilinear"), seed + lv), "module.")
        for r, n in (("4000", 8000), ("1000", 1200)):
            np.save(os.path.join(root, "resources", f"{res}.rebinned.mcool.expected.res{r}.npy"), synth.synth_expected_log(n, seed))
    log(f"wrote synthetic checkpoints in the reference's layout under {root}")


def weight_report(name, module):
    rows = []
    for i, c in enumerate(module._fold_sequentials(module._conv_items())):
        rows.append({"net": name, "conv": i, "cout": c["cout"], "cin": c["cin"], "max_w": float(np.abs(c["w"]).max()), "max_b": float(np.abs(c["b"]).max())})
    return rows


def _mx(t):
    return float(t.abs().max())


def encoder_walk(enc, x):
    """x [1,4,n] one-hot floats; the reference's stage loop (orca_modules.py:935-950) one conv at a time on the exact-fp32 kernel."""
    convs = enc._fold_sequentials(enc._conv_items())
    rows, cur = [], x
    for st in range(7):
        la, lb, ca, cb = convs[4 * st: 4 * st + 4]
        if pm.ENCODER_POOLS[st] > 1:
            cur = engine.maxpool1d(cur, pm.ENCODER_POOLS[st])
        t = engine.conv1d(cur, la["w"], la["b"]); rows.append((f"lconv{st + 1}.a", _mx(t)))
        lout = engine.conv1d(t, lb["w"], lb["b"]); rows.append((f"lconv{st + 1}.b", _mx(lout)))
        t = engine.conv1d(lout, ca["w"], ca["b"], relu=True); rows.append((f"conv{st + 1}.a", _mx(t)))
        out = engine.conv1d(t, cb["w"], cb["b"], relu=True); rows.append((f"conv{st + 1}.b", _mx(out)))
        cur = out + lout if st < 6 else out
        if st < 6:
            rows.append((f"stage{st + 1} out+lout", _mx(cur)))
    return rows, cur


def unet_walk(net, x, nlev, tag):
    """Encoder2 / Encoder3 (orca_modules.py:1151-1169, :1388-1406)."""
    import torch.nn.functional as F
    convs = net._fold_sequentials(net._conv_items())
    rows, encs, out = [], [x], x
    for i in range(nlev):
        a, b, c, d = convs[4 * i: 4 * i + 4]
        lout = engine.conv1d(engine.conv1d(engine.maxpool1d(out, 2), a["w"], a["b"]), b["w"], b["b"])
        t = engine.conv1d(lout, c["w"], c["b"], relu=True)
        out = engine.conv1d(t, d["w"], d["b"], relu=True) + lout
        rows.append((f"{tag} down{i}", max(_mx(lout), _mx(t), _mx(out))))
        encs.append(out)
    outs = [None] * (nlev + 1)
    outs[nlev] = cur = encs[nlev]
    for i in range(nlev):
        lev = nlev - 1 - i
        a, b, c, d = convs[4 * nlev + 4 * i: 4 * nlev + 4 * i + 4]
        up = F.interpolate(cur, scale_factor=2, mode="nearest")
        lout = engine.conv1d(engine.conv1d(up, a["w"], a["b"]), b["w"], b["b"])
        t = engine.conv1d(lout, c["w"], c["b"], relu=True)
        cur = engine.conv1d(t, d["w"], d["b"], relu=True) + lout + encs[lev]
        rows.append((f"{tag} up{i}", max(_mx(lout), _mx(t), _mx(cur))))
        outs[lev] = cur
    return rows, outs


def decoder_walk(dec, x, distenc, y, tag):
    """Decoder.forward (orca_modules.py:461-488) / Decoder_1m.forward (:782-800) conv by conv on the exact-fp32 kernel."""
    import torch.nn.functional as F
    convs = dec._fold_sequentials(dec._conv_items())
    rows = []
    c2 = lambda t, c, relu=False, r=None: engine.conv2d(t, c["w"], c["b"], dilation=c["dil"], relu=relu, r=r)
    mat = x[:, :, :, None] + x[:, :, None, :]
    if distenc is not None:
        mat = torch.cat([mat, distenc], dim=1)
        m0 = c2(c2(mat, convs[0]), convs[1]); rows.append((f"{tag} lcombinerD", _mx(m0)))
        mat = c2(c2(m0, convs[2], True), convs[3], True, m0); rows.append((f"{tag} combinerD", _mx(mat)))
        pairs = convs[8:-2]
        if y is not None:
            cur = torch.cat([mat, F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=False)], dim=1)
            l0 = c2(c2(cur, convs[4]), convs[5])
            cur = c2(c2(l0, convs[6], True), convs[7], True, l0); rows.append((f"{tag} lcombiner/combiner", max(_mx(l0), _mx(cur))))
        else:
            l0 = c2(c2(mat, pairs[0]), pairs[1])
            cur = c2(c2(l0, pairs[2], True), pairs[3], True, l0)
    else:
        pairs = convs[:-2]
        l0 = c2(c2(mat, pairs[0]), pairs[1])
        cur = c2(c2(l0, pairs[2], True), pairs[3], True, l0)
    for i in range(1, len(pairs) // 4):
        p = pairs[4 * i: 4 * i + 4]
        t = c2(cur, p[0]); oth = c2(t, p[1], False, cur)
        t2 = c2(oth, p[2], True); cur = c2(t2, p[3], True, oth)
        rows.append((f"{tag} block{i} (d={p[0]['dil']})", max(_mx(t), _mx(oth), _mx(t2), _mx(cur))))
    return rows, cur


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--orca-path", required=True)
    ap.add_argument("--write-synthetic", metavar="DIR", help="first write synthetic checkpoints in the reference's layout there (self-test)")
    ap.add_argument("--models", default="h1esc,hff")
    ap.add_argument("--sequence", help=".npy of base codes (uint8 0..3 = ACGT, 4 = N), >= 32 000 000 long; default: random, --seed")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sample-bp", type=int, default=3_200_000, help="bases of the layer-by-layer Encoder walk")
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args()
    if args.write_synthetic:
        write_synthetic(args.write_synthetic)
    report = {"orca_path": args.orca_path, "models": {}, "ok": True}
    classes = {"h1esc": M.H1esc, "hff": M.Hff}
    models = {}
    for name in args.models.split(","):
        m = classes[name](model_dir=args.orca_path)
        models[name] = m
        w = []
        for part, mod in [("net0", m.net0), ("net", m.net), ("denet_1_pt", m.denet_1_pt)] + [(f"denet_{lv}", m.denets[lv]) for lv in m.levels]:
            w += weight_report(part, mod)
        worst = max(w, key=lambda r: max(r["max_w"], r["max_b"]))
        over = [r for r in w if max(r["max_w"], r["max_b"]) >= F16_MAX]
        report["models"][name] = {"convs": len(w), "max_folded_weight": worst, "weights_over_fp16_range": over}
        report["ok"] &= not over
        log(f"[{name}] loaded {len(w)} convs; largest folded |w| or |b|: {max(worst['max_w'], worst['max_b']):.4g} ({worst['net']} conv {worst['conv']}); fp16 range {F16_MAX:.0f}")
    if args.no_gpu:
        print(json.dumps(report))
        return 0 if report["ok"] else 1

    dev = torch.device("cuda:0")
    if args.sequence:
        codes = np.load(args.sequence).astype(np.uint8)[:32_000_000]
    else:
        codes = synth.synth_base_codes(32_000_000, seed=args.seed)
    assert codes.shape[0] == 32_000_000, "a 32 Mb sequence is needed"
    codes_dev = torch.from_numpy(codes)[None].to(dev)
    from orca_amd import orca_predict as P
    for name, m in models.items():
        m.cuda()
        rep = report["models"][name]
        # ---- 3. layer-by-layer fp32 walk -----------------------------------------------------------------------------
        n = args.sample_bp - args.sample_bp % 4000
        lut = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [.25, .25, .25, .25]], dtype=torch.float32, device=dev)
        x = lut[codes_dev[0, :n].long()].t()[None].contiguous()
        rows, _ = encoder_walk(m.net0, x)
        del x
        with torch.no_grad():
            enc0 = m.net0.forward_codes(codes_dev)                   # the whole window (default arithmetic) feeds the walks below
            r2, encs = unet_walk(m.net, enc0, 5, "Encoder2")
            rows += r2
            ypred = None
            for lv in (32, 16, 8, 4, 2, 1):
                xs = encs[{1: 0, 2: 1, 4: 2, 8: 3, 16: 4, 32: 5}[lv]][:, :, :250].contiguous()
                de = P._cached_log_background(m, lv, True)
                r, cur = decoder_walk(m.denets[lv], xs, de.expand(1, -1, -1, -1).contiguous(), ypred, f"Decoder {lv}Mb")
                rows += r
                ypred = m.denets[lv](xs, de.expand(1, -1, -1, -1), ypred)[:, :, :125, :125].contiguous()
            r, _ = decoder_walk(m.denet_1_pt, encs[0][:, :, :250].contiguous(), None, None, "Decoder_1m")
            rows += r
        worst = max(rows, key=lambda t: t[1])
        rep["activations"] = {"per_layer_max_abs": [[k, round(v, 4)] for k, v in rows], "worst": list(worst), "fp16_range": F16_MAX,
                              "headroom": F16_MAX / max(worst[1], 1e-30)}
        report["ok"] &= worst[1] < F16_MAX
        log(f"[{name}] largest |activation| in the fp32 walk: {worst[1]:.4g} at {worst[0]} (fp16 range {F16_MAX:.0f}: headroom {F16_MAX / max(worst[1], 1e-30):.3g}x)")
        # ---- 4. full cascade: default arithmetic vs exact fp32 ------------------------------------------------------------
        outs = {}
        for prec in ("f16x2", "f32"):
            for mod in [m.net0, m.net, m.denet_1_pt] + [m.denets[lv] for lv in m.levels]:
                mod.precision = prec
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                outs[prec] = P.genomepredict(codes_dev, "chrV", 16_000_000 + 1_234_567, 16_000_000, models=[m], targets=False, use_cuda=True)
            if prec == "f16x2":
                fired = [str(w.message) for w in wlist if "fp16 range" in str(w.message)]
        levels = []
        for j, (a, b) in enumerate(zip(outs["f16x2"]["predictions"][0], outs["f32"]["predictions"][0])):
            d = float(np.abs(a - b).max())
            r = float(np.corrcoef(a.ravel(), b.ravel())[0, 1])
            levels.append({"level_mb": [32, 16, 8, 4, 2, 1][j], "max_abs": d, "pearson": r})
            report["ok"] &= d < args.tol
        rep["cascade_f16x2_vs_f32"] = {"levels": levels, "range_guard_fired": fired, "tol": args.tol}
        log(f"[{name}] f16x2 vs f32 on the full 32 Mb cascade: worst max-abs {max(l['max_abs'] for l in levels):.3g}, worst Pearson "
            f"{min(l['pearson'] for l in levels):.8f}; range guard fired: {bool(fired)}")
    print(json.dumps(report))
    return 0 if report["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
