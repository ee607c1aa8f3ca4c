"""SV screen timing (BASELINE configs[4]) on one GPU: python tools/time_sv_screen.py [n_svs] [incremental 0|1] [streams: auxiliary contexts of the local encodes, 0 = none, -1 = default] [align: 1 = unaligned (default), 4000 = the 4 kb grid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orca_amd import orca_models, sv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inc = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
streams = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) >= 0 else None
align = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
h1 = orca_models.H1esc(synthetic_seed=0)
g = torch.Generator(device=dev).manual_seed(5)
genome = torch.randint(0, 4, (40_000_000,), device=dev, generator=g, dtype=torch.uint8)
svs = sv.synth_svs(n + 2, 40_000_000, align=align)
sv.sv_screen([h1], genome, svs[:2], 40_000_000, incremental=inc, min_uses=1, streams=streams)
torch.cuda.synchronize()
t0 = time.perf_counter()
st = {}
sv.sv_screen([h1], genome, svs[2:], 40_000_000, incremental=inc, stats=st, streams=streams)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"incremental={inc} streams={streams} {n} SVs: {dt / n * 1e3:.1f} ms per SV = {n / dt:.2f} SV/s  {st}")

if inc:   # where a variant's time goes (on the grid: bins from the chromosome encodings; off it: the stage-3 cache)
    import time as T
    cache = sv.ChromEncodings(h1.net0, genome)
    if align == 4000:
        for k in (("+", 0), ("-", 0), ("+", 2000), ("-", 2000)):
            cache.get(*k)
    else:
        cache.stage3 = sv.Stage4Cache(h1.net0, genome)
        torch.cuda.synchronize(); t = T.perf_counter()
        cache.stage3.build_all()
        torch.cuda.synchronize(); print(f"stage-4 cache: 160 entries in {T.perf_counter() - t:.2f} s")
    enc0 = torch.empty((4, 128, 8000), device=dev)
    from orca_amd import engine
    ns = 4 if streams is None else streams
    pool = engine.context_pool(dev, ns) if ns > 0 else None
    acc = {"assemble": 0.0, "encode_window": 0.0, "cascade": 0.0, "to_host": 0.0}
    for v in svs[2:18]:
        rp, rw, rm, ap, aw, am = sv.sv_windows(v, 40_000_000)
        torch.cuda.synchronize(); t = T.perf_counter()
        codes = torch.stack([sv.assemble_codes(genome, rp), sv.assemble_codes(genome, ap)])
        torch.cuda.synchronize(); acc["assemble"] += T.perf_counter() - t; t = T.perf_counter()
        with engine.defer_overflow_guard():      # as in sv_screen: no range check (= stream sync) per call
            sv.encode_windows(cache, [rp, ap], codes, enc0, build=False, pool=pool)
        torch.cuda.synchronize(); acc["encode_window"] += T.perf_counter() - t; t = T.perf_counter()
        with engine.defer_overflow_guard():
            merged, starts = sv._cascade_windows(h1, enc0, [(rm, rw), (am, aw)])
        torch.cuda.synchronize(); acc["cascade"] += T.perf_counter() - t; t = T.perf_counter()
        sv._window_outputs(h1, merged, starts, [(rm, rw), (am, aw)], "c")
        acc["to_host"] += T.perf_counter() - t
    print({k: round(v / 16 * 1e3, 2) for k, v in acc.items()}, "ms per SV")
