"""bf16 (B16) Encoder vs the fp32-class (f16x2) Encoder on the same sequence: error statistics + timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import engine, synth
from tests.util import product_module
L = int(sys.argv[1]) * 1_000_000 if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda:0")
enc = product_module("Encoder", 0)
x = torch.from_numpy(synth.synth_sequence(L, seed=1, n_frac=0.001)).to(dev).transpose(1, 2)
codes, ok = engine.pack_sequence(x)
out = {}
for prec in ("f16x2", "bf16"):
    enc.precision = prec
    y = enc.forward_codes(codes); torch.cuda.synchronize()
    t = time.perf_counter(); y = enc.forward_codes(codes); torch.cuda.synchronize(); dt = time.perf_counter() - t
    out[prec] = y.cpu().numpy()
    print(f"{prec}: {dt*1e3:.2f} ms, {L/1e6/dt:.1f} Mb/s")
enc.precision = "bf16"
yf = enc(x).cpu().numpy()
a, b = out["f16x2"], out["bf16"]
print("bf16 vs f16x2: max-abs %.4g, rms %.4g, ref rms %.4g, pearson %.6f" % (np.abs(a - b).max(), np.sqrt(((a - b) ** 2).mean()), np.sqrt((a ** 2).mean()), np.corrcoef(a.ravel(), b.ravel())[0, 1]))
print("bf16 float-input vs codes: max-abs %.4g" % np.abs(yf - b).max())
