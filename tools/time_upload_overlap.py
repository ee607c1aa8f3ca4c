"""Does a host-to-device copy run under the Encoder on this platform?  (round 4, VERDICT item 5: no - see profiles/HISTORY.md)"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from orca_amd import orca_models as M, orca_predict as P, synth, engine
model = M.H1esc(synthetic_seed=0)
seq = synth.synth_sequence(32000000, seed=1)
dev = torch.device("cuda:0")
cs = torch.cuda.Stream(dev)
def upl(busy):
    torch.cuda.synchronize()
    codes = torch.from_numpy(synth.synth_base_codes(32000000, seed=1)[None]).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if busy:
        e = model.net0.forward_codes(codes)     # ~24 ms of GPU work queued
    ts = []
    for i in range(4):
        with torch.cuda.stream(cs):
            p = torch.from_numpy(seq[:, i * 8000000:(i + 1) * 8000000]).to(dev)
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("busy" if busy else "idle", [round(t * 1e3, 2) for t in ts], "total", round((time.perf_counter() - t0) * 1e3, 2))
for b in (0, 0, 1, 1):
    upl(b)
# pinned
pin = torch.from_numpy(seq).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
e = model.net0.forward_codes(torch.from_numpy(synth.synth_base_codes(32000000, seed=1)[None]).cuda())
with torch.cuda.stream(cs):
    p = pin.to(dev, non_blocking=True)
torch.cuda.synchronize(); print("pinned 512 MB under the encoder:", round((time.perf_counter() - t0) * 1e3, 2))
torch.cuda.synchronize(); t0 = time.perf_counter()
p = pin.to(dev, non_blocking=True)
torch.cuda.synchronize(); print("pinned 512 MB idle:", round((time.perf_counter() - t0) * 1e3, 2))
