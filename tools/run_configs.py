"""One-off measurements of BASELINE.json configs 3-5 on one MI355X (numbers quoted in DESIGN.md section 6)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import engine, orca_models as M, orca_predict as P, sv, synth

dev = torch.device("cuda:0")
def sync(): torch.cuda.synchronize(dev)
if len(sys.argv) > 1 and sys.argv[1] == "config5_1024":
    # BASELINE.json configs[4] at its stated size on ONE GPU: all 1024 synthetic SVs x (reference + alternative allele), as SURVEY 8(d) draws
    # them (unaligned, the default) and on rounds 3-5's 4 kb grid (argv[2] = "aligned" | "unaligned" | both when absent)
    h1 = M.H1esc(synthetic_seed=0)
    g = torch.Generator(device=dev).manual_seed(5)
    genome = torch.randint(0, 4, (40_000_000,), device=dev, generator=g, dtype=torch.uint8)
    out = {}
    for label, align in (("unaligned", 1), ("aligned_4kb", 4000)):
        if len(sys.argv) > 2 and not label.startswith(sys.argv[2]):
            continue
        svs = sv.synth_svs(1024, 40_000_000, align=align)
        sv.sv_screen([h1], genome, svs[:2], 40_000_000, min_uses=1); sync()
        acc = {"chk": 0.0, "kinds": {}}

        def keep(i, v):        # only checksums of the 12 288 maps are kept
            acc["kinds"][v["sv"].kind] = acc["kinds"].get(v["sv"].kind, 0) + 1
            acc["chk"] += float(sum(np.sum(m, dtype=np.float64) for a in ("ref", "alt") for m in v[a]["predictions"][0]))
        stats = {}
        t = time.perf_counter()
        sv.sv_screen([h1], genome, svs, 40_000_000, stats=stats, on_result=keep)      # chromosome encodings (where phases are shared) included
        sync(); dt = time.perf_counter() - t
        out[label] = {"svs": 1024, "coordinates": f"synth_svs(align={align})", "s_total": round(dt, 2), "s_per_sv_ref_plus_alt": round(dt / 1024, 4), "svs_per_s": round(1024 / dt, 2),
                      "strand_Mb_per_s": round(1024 * 4 * 32 / dt, 1), "kinds": acc["kinds"], "maps": 1024 * 12, "maps_checksum": round(acc["chk"], 3),
                      "encoder_bins_encoded_frac": round(stats["bins_encoded"] / stats["bins_total"], 4), "chromosome_encodings": stats["chromosome_encodings"],
                      "whole_window_runs": stats.get("whole_window_runs"), "stage3_cache": stats.get("stage3_cache")}
    out["mode"] = ("orca_amd/sv.py sv_screen: chromosome encoded once per strand and 4 kb phase that >= 3 window runs share, windows re-encode ends + junctions; windows whose "
                   "phase is not held (every window of the unaligned set) take stages 1-3 of the Encoder from the chromosome's stage-3 cache (sv.Stage3Cache: 16 phases x 2 "
                   "strands, built inside the timed region) and run their ends, junctions and stages 4-7; ref + alt of two variants as one decoder batch")
    print(json.dumps({"config5_sv_screen_1024_1gpu": out}))
    sys.exit(0)
def rand_codes(B, L, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randint(0, 4, (B, L), device=dev, generator=g, dtype=torch.uint8)
res = {}

# ---- config 3: HFF-shaped 32 Mb model, batch of 8 random 32 Mb sequences, bf16 throughput mode vs the fp32-class default ----
hff = M.Hff(synthetic_seed=7)
codes = torch.from_numpy(np.stack([synth.synth_base_codes(32_000_000, seed=10 + b) for b in range(8)])).to(dev)
de = {lv: torch.log(torch.from_numpy(hff.normmats[lv][None, None].astype(np.float32))).to(dev) for lv in hff.levels}
def fwd8(prec, enc_only=False):
    dprec = {"bf16+f16": "f16"}.get(prec, prec)
    hff.net0.precision = prec.split("+")[0]
    for lv in hff.levels: hff.denets[lv].precision = dprec
    hff.denet_1_pt.precision = dprec
    enc0 = hff.net0.forward_codes(codes)
    if enc_only: return enc0
    encs = dict(zip([1, 2, 4, 8, 16, 32], hff.net(enc0)))
    return P.run_cascade(hff, encs, [32, 16, 8, 4, 2, 1], lambda lv: lv, 8, [False], lambda lv, k, st: de[lv],
                         lambda lv, st, rev: P.zoom_index_32m(lv, st, 17_234_567, 16_000_000, rev), add_1m_level=1)[0]
out = {}
for prec in ("bf16", "bf16+f16", "f16x2"):
    fwd8(prec); sync(); t = time.perf_counter(); p = fwd8(prec); sync(); dt = time.perf_counter() - t
    t = time.perf_counter(); fwd8(prec, True); sync(); dte = time.perf_counter() - t
    out[prec] = [x.cpu().numpy() for x in p]
    res[f"config3_{prec}_B8_single_strand"] = {"s": round(dt, 4), "Mb_per_s": round(8 * 32 / dt, 1), "maps_per_s": round(48 / dt, 1),
                                                "encoder_s": round(dte, 4), "encoder_Mb_per_s": round(8 * 32 / dte, 1)}
for m in ("bf16", "bf16+f16"):
    err = max(float(np.abs(a - b).max()) for a, b in zip(out[m], out["f16x2"]))
    r = min(float(np.corrcoef(a.ravel(), b.ravel())[0, 1]) for a, b in zip(out[m], out["f16x2"]))
    res[f"config3_{m}_vs_f16x2"] = {"max_abs": round(err, 4), "min_pearson_r": round(r, 6)}
print(json.dumps(res), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "config3": sys.exit(0)
del codes, hff, out; engine.get_context(dev).release_workspace(); torch.cuda.empty_cache()

# ---- config 4 (one GPU): H1esc_256M-shaped model on a random 256 Mb sequence ------------------------------------
m256 = M.H1esc_256M(synthetic_seed=0)
c256 = rand_codes(1, 256_000_000, 2)
chrlen = 138_368_000
nm = synth.synth_normmat_256m(chrlen, seed=0)
m256.net0.forward_codes(c256); sync()
t = time.perf_counter(); e = m256.net0.forward_codes(c256); sync(); t_enc = time.perf_counter() - t
P.genomepredict_256Mb(c256, "chrS", [nm], chrlen, 70_000_000, 128_000_000, models=[m256]); sync()      # first call: NaN fill of the background, allocations
t = time.perf_counter(); o = P.genomepredict_256Mb(c256, "chrS", [nm], chrlen, 70_000_000, 128_000_000, models=[m256]); sync()
t_all = time.perf_counter() - t
res["config4_256Mb_1gpu"] = {"encoder_one_strand_s": round(t_enc, 4), "encoder_Mb_per_s": round(256 / t_enc, 1),
                             "genomepredict_256Mb_s_incl_host_background_coarsegraining": round(t_all, 3), "start_coords": o["start_coords"]}
del c256, m256; engine.get_context(dev).release_workspace(); torch.cuda.empty_cache()

# ---- config 5 (one GPU): SV screen from a packed 40 Mb chromosome -------------------------------------------------
h1 = M.H1esc(synthetic_seed=0)
genome = rand_codes(1, 40_000_000, 5)[0]
svs = sv.synth_svs(6, 40_000_000)
sv.sv_screen([h1], genome, svs[:2], 40_000_000); sync()      # warm-up: workspace growth, both window shapes
t = time.perf_counter(); r5 = sv.sv_screen([h1], genome, svs, 40_000_000); sync(); dt = time.perf_counter() - t
res["config5_sv_screen_1gpu"] = {"svs": len(svs), "s_per_sv_ref_plus_alt": round(dt / len(svs), 4), "svs_per_s": round(len(svs) / dt, 2),
                                 "kinds": [v.kind for v in svs]}

# ---- config 5 through the reference-signature drivers (orca_amd.sv_drivers): genome resident in HBM vs host route --
from orca_amd import synth
g_dev = synth.sv_driver_genome().to(dev)
g_host = synth.sv_driver_genome()
calls = [("process_del", ("chrS", 15_200_000, 15_850_000)), ("process_dup", ("chrS", 20_000_000, 21_500_000)),
         ("process_inv", ("chrS", 30_100_000, 33_000_000))]
P.process_del("chrS", 15_200_000, 15_850_000, g_dev, custom_models=[h1], target=False); sync()
for label, g in (("device_genome", g_dev), ("host_route", g_host)):
    t = time.perf_counter(); nviews = 0
    for fn, a in calls:
        nviews += len(getattr(P, fn)(*a, g, custom_models=[h1], target=False))
    sync(); dt = time.perf_counter() - t
    res["config5_drivers_" + label] = {"variants": len(calls), "views": nviews, "s_per_view": round(dt / nviews, 4),
                                       "s_per_variant": round(dt / len(calls), 4)}

# ---- 1 Mb model (Net / H1esc_1M): batches of random 1 Mb sequences ----------------------------------
del h1; engine.get_context(dev).release_workspace(); torch.cuda.empty_cache()
m1 = M.H1esc_1M(synthetic_seed=0).to(dev)
for B in (1, 8, 32):
    xb = torch.from_numpy(synth.synth_sequence(1_000_000, seed=70, batch=B)).to(dev).transpose(1, 2)
    m1(xb); sync()
    t = time.perf_counter()
    for _ in range(3):
        m1(xb)
    sync(); dt = (time.perf_counter() - t) / 3
    res[f"model_1M_B{B}"] = {"ms_per_forward": round(dt * 1e3, 2), "sequences_per_s": round(B / dt, 1), "Mb_per_s": round(B / dt, 1),
                             "tflops_algorithmic": round(B * (0.4656 + 0.1774) / dt, 1)}
print(json.dumps(res, indent=1))
