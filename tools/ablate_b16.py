import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import engine
dev = torch.device("cuda:0")
cin = int(sys.argv[1]); cout = int(sys.argv[2]); n = int(sys.argv[3]); prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.randn(1, n, cin).astype(np.float32)).to(dev)
w = (rs.randn(cout, cin, 9) / np.sqrt(cin * 9)).astype(np.float32); b = np.zeros(cout, np.float32)
ctx = engine.get_context(dev)
engine.conv1d_nlc(x, w, b, prec)
ctx.set_timing(True)
for _ in range(3): engine.conv1d_nlc(x, w, b, prec)
ts = [r[5] for r in ctx.get_timing()]
fl = 2.0 * 9 * cin * cout * n
print(f"ablate={os.environ.get('ORCA_B16_ABLATE','0')} {prec} {cin}->{cout} n={n}: {min(ts):.3f} ms  {fl/min(ts)/1e9:.1f} TFLOP/s-equivalent")
