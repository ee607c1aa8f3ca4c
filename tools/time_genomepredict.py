import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from orca_amd import orca_models as M, orca_predict as P, synth
model = M.H1esc(synthetic_seed=0)
seq = synth.synth_sequence(32000000, seed=1)
for i in range(3):
    t = time.perf_counter()
    out = P.genomepredict(seq, "chrS", 16000000 + 1234567, 16000000, models=[model], use_cuda=True)
    torch.cuda.synchronize()
    print("genomepredict(host float [1,32e6,4]) wall: %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
codes = torch.from_numpy(synth.synth_base_codes(32000000, seed=1)[None]).cuda()
for i in range(3):
    t = time.perf_counter()
    out = P.genomepredict(codes, "chrS", 16000000 + 1234567, 16000000, models=[model], use_cuda=True)
    torch.cuda.synchronize()
    print("genomepredict(device codes) wall: %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
out = P.genomepredict(seq, "chrS", 16000000 + 1234567, 16000000, models=[model], use_cuda=True); torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
