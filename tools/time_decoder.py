"""Decoder forward timing (both strands batched, 250 x 250 maps) in a chosen arithmetic mode: python tools/time_decoder.py [f16x2|f16|bf16] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import product_module
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
dec = product_module("Decoder", 0, device=dev)
dec.precision = prec
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(B, 128, 250).astype(np.float32)).to(dev)
de = torch.from_numpy(rs.randn(B, 1, 250, 250).astype(np.float32)).to(dev)
y = torch.from_numpy(rs.randn(B, 1, 125, 125).astype(np.float32)).to(dev)
for _ in range(3):
    out = dec(x, de, y)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    out = dec(x, de, y)
ev1.record(); torch.cuda.synchronize()
print(f"decoder {prec} B={B}: {ev0.elapsed_time(ev1) / 10:.3f} ms per forward")
