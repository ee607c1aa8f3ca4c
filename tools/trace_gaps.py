"""Idle gaps of the GPU in a rocprofv3 kernel_trace.csv: over the LAST `span_ms` of the trace, every interval >= min_us in which NO kernel was
running, with the kernels on either side.  usage: trace_gaps.py <csv> [span_ms] [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 70.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60]) for r in rows))
t_end = max(e for _, e, _ in ev)
ev = [x for x in ev if x[0] >= t_end - span_ms * 1e6]
cur_end, last, idle, n = ev[0][1], ev[0][2], 0.0, 0
for s, e, name in ev[1:]:
    if s > cur_end:
        g = (s - cur_end) / 1e3
        idle += g
        if g >= min_us:
            n += 1
            print(f"{(cur_end - ev[0][0]) / 1e3:10.1f} us  idle {g:8.1f} us   {last}  ->  {name}")
    if e > cur_end:
        cur_end, last = e, name
import collections
by = collections.defaultdict(lambda: [0, 0.0])
for s_, e_, name in ev:
    by[name][0] += 1; by[name][1] += (e_ - s_) / 1e3
for name, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"   {us:10.1f} us  {c:5d} launches  {name}")
print(f"window {(t_end - ev[0][0]) / 1e3:.1f} us, idle total {idle:.1f} us in gaps of any size, {n} gaps >= {min_us} us, {len(ev)} launches")
