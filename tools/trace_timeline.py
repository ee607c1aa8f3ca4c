"""Launch-ordered timeline of the LAST `n` kernel launches of a rocprofv3 kernel_trace.csv: start offset, duration, gap to the previous
kernel's end (us) and the kernel name (template arguments kept, parameter list dropped).  usage: trace_timeline.py <csv> [n] [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = rows[-n:]
t0, prev_end, tot = int(rows[0]["Start_Timestamp"]), None, 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = max(e, prev_end or e)
    tot += (e - s) / 1e3
    if (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:8.1f}  {r['Kernel_Name'].replace('void ', '').split('(')[0][:110]}")
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot:.1f} us, launches {len(rows)}")
