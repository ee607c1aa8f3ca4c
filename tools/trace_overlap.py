"""Concurrency of the kernels in a rocprofv3 kernel_trace.csv after the LAST n launches' first start: sum of kernel durations, the time at
least one kernel runs (union), per-queue kernel time, and the time-weighted mean number of kernels in flight.
usage: trace_overlap.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
tot = sum(e - s for s, e in iv)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = max(e for _, e in iv) - iv[0][0]
queues = {}
for r in rows:
    q = r.get("Queue_Id", "?")
    queues[q] = queues.get(q, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"launches {len(rows)}: span {span / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us, >= 1 kernel running {union / 1e3:.1f} us, "
      f"mean kernels in flight while busy {tot / union:.2f}")
print("kernel time per queue (us):", {q: round(v / 1e3, 1) for q, v in sorted(queues.items())})
