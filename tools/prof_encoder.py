"""Small driver for rocprofv3 counter passes: one Encoder forward on an L-Mb random sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import orca_modules as pm, synth
from tests.util import product_module
L = int(sys.argv[1]) * 1_000_000 if len(sys.argv) > 1 else 8_000_000
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
enc = product_module("Encoder", 0); enc.precision = prec
x = torch.from_numpy(synth.synth_sequence(L, seed=1)).to(dev).transpose(1, 2)
packed = len(sys.argv) > 4 and sys.argv[4] == "codes"   # 1 byte/base input (fused first layer) instead of the float view
if packed:
    from orca_amd import engine
    codes, ok = engine.pack_sequence(x)
    assert ok
    run = lambda: enc.forward_codes(codes)
else:
    run = lambda: enc(x)
y = run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps): y = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"encoder {prec}{' packed input' if packed else ''} L={L/1e6:.0f}Mb: {dt*1e3:.2f} ms  -> {L/1e6/dt:.1f} Mb/s, {465555.5*L/dt/1e12:.1f} TFLOP/s algorithmic")
