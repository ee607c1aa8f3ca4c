"""Print the per-launch durations (us) of kernels matching a substring from a rocprofv3 kernel_trace.csv, in launch order."""
import csv, sys
path, pat = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(d)
print(pat, "calls", len(d), "last", n)
print(" ".join(f"{v:.1f}" for v in d[-n:]))
