"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name."""
import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
            agg[k]["_LDS"] = float(r.get("LDS_Block_Size", 0) or 0); agg[k]["_VGPR"] = float(r.get("VGPR_Count", 0) or 0)
            agg[k]["_AGPR"] = float(r.get("Accum_VGPR_Count", 0) or 0); agg[k]["_WG"] = float(r.get("Workgroup_Size", 0) or 0)
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0))):
    print(f"{k}  calls={cnt[k]}")
    print("   ", {a: (round(b) if abs(b) > 10 else b) for a, b in sorted(d.items())})
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and d.get("GRBM_GUI_ACTIVE"):
        # counters are sums over the 8 XCDs: SIMD cycles of the launches = GRBM_GUI_ACTIVE / 8 x 1024 SIMDs (256 CUs x 4)
        print(f"    matrix pipe busy: {100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0):.1f} % of the SIMD cycles")
