"""What a structural-variant driver call costs on a resident genome: python tools/time_sv_drivers.py [ncalls]

`process_del / process_dup / process_inv` on the synthetic genome (chrS 40 Mb) in HBM, one H1-ESC-shaped model:
  * whole-window route (rounds 1-4; ORCA_SV_INCREMENTAL=0): every view through `genomepredict`;
  * incremental route, FIRST call on a cold cache; the SECOND and later calls on the chromosome, (a) coordinates on the 4 kb grid (a (strand,
    phase) that keeps coming back is encoded once for the whole chromosome), (b) coordinates off the grid (every call its own phases: only
    the views of the call share work - until, after 16 window strands nobody could serve, the store builds the chromosome's STAGE-3 cache
    (round 6, sv.Stage3Cache: 41 GB for chrS, ~2 s once), from which a window strand at any phase costs ~3 instead of 24.6 ms: `tail_calls_ms` =
    the median over the last third of the calls).
Prints one JSON line (ms per call, wall clock around the call incl. the device -> host copy of the maps)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import orca_models as M, orca_predict as P, sv_drivers, synth

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
model = M.H1esc(synthetic_seed=0)
genome = synth.sv_driver_genome().to(dev)
rs = np.random.RandomState(11)


def variants(n, aligned):
    out = []
    for k in range(n):
        size = int(rs.randint(50_000, 2_000_000))
        start = int(rs.randint(8_000_000, 30_000_000))
        if aligned:
            size, start = size - size % 4000 + 4000, start - start % 4000
        out.append((start, start + size))
    return out


def timed(fn, a):
    torch.cuda.synchronize()
    t = time.perf_counter()
    outs = getattr(P, fn)("chrS", a[0], a[1], genome, custom_models=[model], target=False, use_cuda=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, outs


res = {"model": "H1esc-shaped, synthetic weights", "genome": "chrS 40 Mb + chrT 36 Mb, 1 byte/base in HBM", "calls_per_row": ncalls}
for fn in ("process_del", "process_dup", "process_inv"):
    row = {}
    for aligned in (True, False):
        vs = variants(ncalls + 1, aligned)
        timed(fn, vs[0])                                        # warm-up: weights, workspaces
        os.environ["ORCA_SV_INCREMENTAL"] = "0"
        whole = [timed(fn, v)[0] for v in vs[1:]]
        del os.environ["ORCA_SV_INCREMENTAL"]
        sv_drivers.clear_encoding_cache()
        stats = []
        inc = []
        for v in vs[1:]:
            ms, outs = timed(fn, v)
            inc.append(ms)
        key = "grid4kb" if aligned else "offgrid"
        row[key] = {"whole_window_ms": round(float(np.median(whole)), 2), "incremental_first_call_ms": round(inc[0], 2),
                    "incremental_later_calls_ms": round(float(np.median(inc[2:])), 2), "all_incremental_ms": [round(x, 1) for x in inc],
                    "speedup_later_calls": round(float(np.median(whole) / np.median(inc[2:])), 2),
                    "tail_calls_ms": round(float(np.median(inc[-max(1, len(inc) // 3):])), 2),
                    "stage3_caches": {c: len(ce.stage3.entries) for c, ce in sv_drivers._store(genome, model.net0).chroms.items() if ce.stage3 is not None}}
    res[fn] = row
print(json.dumps(res))
