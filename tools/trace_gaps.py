"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: where the GPU waits for the host (or for a launch boundary).
usage: python tools/trace_gaps.py <..._kernel_trace.csv> [min_gap_us = 10] [split_ms = 5]
The trace is cut into SEGMENTS at gaps >= split_ms (model set-up, fixture loading, the sections of bench.py); for every segment of >= 2 000
launches (a timed loop) it prints busy / idle time, the idle time by gap size, and the largest gaps with the kernels on either side."""
import csv, sys, collections

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
split_us = (float(sys.argv[3]) if len(sys.argv) > 3 else 5.0) * 1e3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()


def report(seg, idx):
    busy = 0; end = seg[0][0]; gaps = []
    for i, (s, e, n) in enumerate(seg):
        if s > end:
            gaps.append(((s - end) / 1e3, seg[i - 1][2] if i else "", n))
        busy += max(0, e - max(s, end)); end = max(end, e)
    span = (seg[-1][1] - seg[0][0]) / 1e3
    idle = sum(g[0] for g in gaps)
    top = collections.Counter(n for _, _, n in seg).most_common(1)[0]
    print(f"segment {idx}: {len(seg)} launches (most frequent: {top[0]} x {top[1]}), span {span/1e3:.2f} ms, busy {busy/1e6:.2f} ms, idle {idle/1e3:.2f} ms ({100*idle/span:.2f} %)")
    cls = collections.Counter(); cnt = collections.Counter()
    for g, a, b in gaps:
        k = "<2us" if g < 2 else "<5us" if g < 5 else "<20us" if g < 20 else "<100us" if g < 100 else "<1ms" if g < 1000 else ">=1ms"
        cls[k] += g; cnt[k] += 1
    print("   " + "  ".join(f"{k}: {cnt[k]} x {cls[k]/1e3:.2f} ms" for k in ("<2us", "<5us", "<20us", "<100us", "<1ms", ">=1ms")))
    by_pair = collections.defaultdict(lambda: [0, 0.0])
    for g, a, b in gaps:
        if g >= min_gap:
            by_pair[(a, b)][0] += 1; by_pair[(a, b)][1] += g
    for (a, b), (c, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"   {t/1e3:8.2f} ms in {c:5d} gaps >= {min_gap:.0f} us   {a}  ->  {b}")


segs = [[rows[0]]]; end = rows[0][1]
for r in rows[1:]:
    if r[0] - end >= split_us * 1e3:
        segs.append([])
    segs[-1].append(r); end = max(end, r[1])
print(f"{len(rows)} launches in {len(segs)} segments (cut at gaps >= {split_us/1e3:.0f} ms); segments of >= 2000 launches:")
for i, seg in enumerate(segs):
    if len(seg) >= 2000:
        report(seg, i)
