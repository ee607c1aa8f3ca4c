"""Summarise a rocprofv3 rocpd database (`--kernel-trace`, default sqlite output) per kernel, and list the launches of
one kernel in dispatch order.  usage: rocpd_stats.py results.db [--list SUBSTR] [--last N]   (-> CSV on stdout)"""
import re, sqlite3, sys

def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)

def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start").fetchall()
    if "--last" in sys.argv:   # only the last N dispatches (e.g. the timed repetitions)
        rows = rows[-int(sys.argv[sys.argv.index("--last") + 1]):]
    if "--list" in sys.argv:
        sub = sys.argv[sys.argv.index("--list") + 1]
        print("kernel,start_us,duration_us,grid,wg")
        t0 = rows[0][1]
        for n, s, e, g, w, *_ in rows:
            if sub in n:
                print(f"\"{short(n)}\",{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.2f},{g},{w}")
        return
    agg = {}
    for n, s, e, g, w, v, av, lds in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0, v, av, lds])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{n}\",{a[0]},{a[1] / 1e3:.1f},{a[1] / a[0] / 1e3:.2f},{a[2] / 1e3:.2f},{a[3] / 1e3:.2f},{100 * a[1] / tot:.2f},{a[4]},{a[5]},{a[6]}")

main()
