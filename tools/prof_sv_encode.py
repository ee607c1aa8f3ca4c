"""rocprofv3 driver: the local (ends + junctions) Encoder calls of ONE variant of the incremental SV screen.
usage: prof_sv_encode.py [streams: contexts of an engine.ContextPool the calls are dealt to, 0 = the caller's stream]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orca_amd import orca_models, sv
dev = torch.device("cuda:0")
h1 = orca_models.H1esc(synthetic_seed=0)
g = torch.Generator(device=dev).manual_seed(5)
genome = torch.randint(0, 4, (40_000_000,), device=dev, generator=g, dtype=torch.uint8)
v = sv.synth_svs(4, 40_000_000, align=4000)[2]      # an on-grid variant: the local encodes are what is measured
cache = sv.ChromEncodings(h1.net0, genome)
rp, rw, rm, ap, aw, am = sv.sv_windows(v, 40_000_000)
codes = torch.stack([sv.assemble_codes(genome, rp), sv.assemble_codes(genome, ap)])
enc0 = torch.empty((4, 128, 8000), device=dev)
from orca_amd import engine
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pool = engine.context_pool(dev, ns) if ns > 0 else None
for _ in range(2):
    sv.encode_windows(cache, [rp, ap], codes, enc0, pool=pool)
torch.cuda.synchronize()
print("MARK")
import time
t = time.perf_counter()
with engine.defer_overflow_guard():      # as in sv_screen: no range check (= stream sync) per call
    n = sv.encode_windows(cache, [rp, ap], codes, enc0, build=False, pool=pool)
t_host = time.perf_counter() - t
torch.cuda.synchronize()
print(v, n, "bins", "streams", ns, "host enqueue", round(t_host * 1e3, 2), "ms, done after", round((time.perf_counter() - t) * 1e3, 2), "ms")
