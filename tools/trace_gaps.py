"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: where the GPU waits for the host (or for a launch boundary).
usage: python tools/trace_gaps.py <..._kernel_trace.csv> [min_gap_us = 10]
Prints the total busy / idle time, the idle time by size class, and the largest gaps with the kernels on either side."""
import csv, sys, collections

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
busy = 0; end = rows[0][0]; gaps = []
for i, (s, e, n) in enumerate(rows):
    if s > end:
        gaps.append(((s - end) / 1e3, rows[i - 1][2] if i else "", n))
    busy += max(0, e - max(s, end)); end = max(end, e)
span = (rows[-1][1] - rows[0][0]) / 1e3
idle = sum(g[0] for g in gaps)
print(f"{len(rows)} launches, span {span/1e3:.2f} ms, busy {busy/1e6:.2f} ms, idle {idle/1e3:.2f} ms ({100*idle/span:.1f} %)")
cls = collections.Counter(); cnt = collections.Counter()
for g, a, b in gaps:
    k = "<2us" if g < 2 else "<5us" if g < 5 else "<20us" if g < 20 else "<100us" if g < 100 else "<1ms" if g < 1000 else ">=1ms"
    cls[k] += g; cnt[k] += 1
for k in ("<2us", "<5us", "<20us", "<100us", "<1ms", ">=1ms"):
    print(f"  gaps {k:7s}: {cnt[k]:7d} x, {cls[k]/1e3:9.2f} ms")
by_pair = collections.defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    if g >= min_gap:
        by_pair[(a, b)][0] += 1; by_pair[(a, b)][1] += g
print(f"gaps >= {min_gap} us by (kernel before -> kernel after), largest total first:")
for (a, b), (c, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t/1e3:8.2f} ms in {c:5d} x   {a}  ->  {b}")
