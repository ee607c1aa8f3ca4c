"""Turn gpurun_out/<tag>/ (tools/profile_run.sh) into the summaries committed under profiles/:
   <tag>_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-configs`
   <tag>_bench_dominant_launches.txt   every n = 32 M launch of the dominant kernel's instantiations (us) + averages
   <tag>_bench.json / _bench_under_rocprof.json
   <tag>_pmc_<run>.txt                 per-kernel sums of the counter passes (tools/pmc_summary.py format)
   pmc_traffic.json                    HBM bytes per launch of the dominant kernel (FETCH_SIZE x 2 + WRITE_SIZE, KiB)
usage: python tools/profile_collect.py r02"""
import collections, csv, json, os, re, shutil, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for f in ("bench.json", "bench_under_rocprof.json"):
    lines = [l for l in open(os.path.join(src, f)).read().splitlines() if l.startswith("{")]
    open(os.path.join(dst, f"{tag}_{f}"), "w").write(lines[-1] + "\n")

for f in ("configs.json", "config5_1024.json", "bench_gaps.txt"):
    if os.path.exists(os.path.join(src, f)) and os.path.getsize(os.path.join(src, f)) > 2:
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))

# ---- dominant kernel: per-launch durations from the trace
rows = list(csv.DictReader(open(os.path.join(src, "bench", "bench_kernel_trace.csv"))))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if n.startswith("void conv1d_k9_p16_kernel<") or n.startswith("void conv1d_k9_p16w1_kernel<") or n.startswith("void conv1d_k9_p16x_kernel<") or n.startswith("void conv1d_k9_p16p5_kernel<") or n.startswith("void conv1d_k9_ws_kernel") or n.startswith("void conv1d_first_mfma_p16_kernel"):
        by[n.replace("void ", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(os.path.join(dst, f"{tag}_bench_dominant_launches.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-configs`: launches of the planar conv kernels (bench.py's `roofline` names the\n"
            "# instantiation with the largest time share), split by problem size; template arguments <CT, MW, NW, WM, out_mode, residual, ABL,\n"
            "# fused-first-layer, format, residual-from-bases> (conv_p16w1.h: <out_mode, residual, format>).  'big' = launches >= 2.5 ms (stage 1 at n = 32 M, stage 2 at n = 8 M of a 32 Mb strand\n"
            "# or chunk), the rest are stages 3-4.\n")
    allbig, allsmall = [], []
    for k, v in sorted(by.items()):
        big, small = [d for d in v if d >= 2500], [d for d in v if d < 2500]
        allbig += big; allsmall += small
        if big:
            f.write(f"{k}  big: {len(big)} launches, avg {sum(big) / len(big):.1f} us, min {min(big):.1f}, max {max(big):.1f}\n")
            f.write("   " + " ".join(f"{d:.0f}" for d in big) + "\n")
        if small:
            f.write(f"{k}  small: {len(small)} launches, avg {sum(small) / len(small):.1f} us\n")
    if allbig:
        f.write(f"ALL big launches: {len(allbig)}, avg {sum(allbig) / len(allbig):.1f} us\n")

# ---- counter passes
for run in sorted(os.listdir(src)):
    p = os.path.join(src, run, "p_counter_collection.csv")
    if os.path.exists(p):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), p], capture_output=True, text=True).stdout
        open(os.path.join(dst, f"{tag}_{run}.txt"), "w").write(out)

def pmc_sum(run, counter, prefix, min_us=1500.0):
    """sum of `counter` over the launches of kernels named prefix* that ran >= min_us (the n = 32 M launches)"""
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(os.path.join(src, run, "p_counter_collection.csv"))):
        if (prefix.search(r["Kernel_Name"]) if hasattr(prefix, "search") else r["Kernel_Name"].startswith(prefix)) and r["Counter_Name"] == counter and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 >= min_us:
            tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    return tot, len(disp)

traffic = {"_source": f"profiles/{tag}_pmc_enc_f16x2_fetch.txt + profiles/{tag}_pmc_enc_f16x2_write.txt (and the bf16 pair)"}
for mode, key, prefix, min_us in (("f16x2", "conv1d_k9_p16_kernel<cout=64,f16x2>", "void conv1d_k9_p16_kernel<64,", 4000.0),
                                  ("f16x2", "conv1d_k9_p16x_kernel<cout=96,f16x2>", re.compile(r"^void conv1d_k9_p16x_kernel<\d, (true|false), 3, 96>"), 2500.0),   # 96 -> 96, plain and pooled + residual (and the 17-tap 64 -> 96)
                                  ("f16x2", "conv1d_first_mfma_p16_kernel<0,0,25>", "void conv1d_first_mfma_p16_kernel<0, 0, 25>", 800.0),
                                  ("bf16", "conv1d_k9_p16w1_kernel<cout=96,bf16>", re.compile(r"^void conv1d_k9_p16w1_kernel<\d, (true|false), 1>"), 800.0),
                                  ("bf16", "conv1d_k9_ws_kernel<cout=64,bf16>", "void conv1d_k9_ws_kernel<1, 64, 64,", 1200.0)):
    try:
        fe, n1 = pmc_sum(f"pmc_enc_{mode}_fetch", "FETCH_SIZE", prefix, min_us)
        wr, n2 = pmc_sum(f"pmc_enc_{mode}_write", "WRITE_SIZE", prefix, min_us)
        if n1 and n1 == n2:
            traffic[key] = (2 * fe + wr) * 1024 / n1
            traffic[key + "_launches"] = n1
    except FileNotFoundError:
        pass
traffic["_note"] = ("HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches over every launch of the kernel in ONE run of "
                    "tools/prof_encoder.py 32 <mode> 1 codes (2 forwards of a 32 Mb strand, packed input): separate rocprofv3 --pmc passes, FETCH_SIZE doubled "
                    "per the gfx950 correction of MI355X_MICROARCH.md (HBM section), WRITE_SIZE as reported, both in KiB.  Only the n = 32 M launches (stage 1: "
                    "the 64 -> 64 convs, the population of bench.py's roofline.achieved): f16x2 = 3 per strand, the first with its input produced in LDS from "
                    "1 byte/base (reads 72 MB instead of 8.2 GB), the last pooled (writes 2 GB) with a residual (reads 8.2 GB more): algorithmic "
                    "(0 + 8.2) + (8.2 + 8.2) + (8.2 + 8.2 + 2.05) = 43.1 GB = 14.4 GB/launch... of which the P16 planes are 4 B/element; bf16 (B16 planes, "
                    "2 B/element) = 2 ws launches per strand + the pooled one: (4.1 + 4.1) + (4.1 + 4.1) + (4.1 + 4.1 + 1.0) = 25.6 GB = 8.5 GB/launch.  conv1d_k9_p16x_kernel / conv1d_k9_p16w1_kernel<cout=96,...>: the average over stage 2's three "
                    "launches per strand at n = 8 M (17-tap 64 -> 96, 96 -> 96, 96 -> 96 pooled + residual: algorithmic 5.12 + 6.14 + 6.91 GB = 6.06 GB/launch in f16x2, "
                    "half of that on B16 planes); bench.py's roofline.achieved for that kernel is over the two 96 -> 96 launches.")
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_")}, indent=1))
print(open(os.path.join(dst, f"{tag}_bench_dominant_launches.txt")).read()[-700:])
