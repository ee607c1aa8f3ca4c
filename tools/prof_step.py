"""Small driver for rocprofv3 --kernel-trace over whole genomepredict steps (device-resident packed sequence): 2 warm-up + 2 timed calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import orca_models as M, orca_predict as P, synth
model = M.H1esc(synthetic_seed=0)
codes = torch.from_numpy(synth.synth_base_codes(32000000, seed=1)[None]).cuda()
for i in range(4):
    t = time.perf_counter()
    out = P.genomepredict(codes, "chrS", 16000000 + 1234567, 16000000, models=[model], use_cuda=True)
    torch.cuda.synchronize()
    print("genomepredict(device codes) wall: %.2f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
