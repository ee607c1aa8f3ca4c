"""cProfile of one genomepredict_256Mb call on the MI355X (host time vs GPU time; quoted in DESIGN.md section 6)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import engine, orca_models as M, orca_predict as P, synth
dev = torch.device("cuda:0")
m256 = M.H1esc_256M(synthetic_seed=0)
g = torch.Generator(device=dev).manual_seed(2)
c256 = torch.randint(0, 4, (1, 256_000_000), device=dev, generator=g, dtype=torch.uint8)
chrlen = 138_368_000
nm = synth.synth_normmat_256m(chrlen, seed=0)
P.genomepredict_256Mb(c256, "chrS", [nm], chrlen, 70_000_000, 128_000_000, models=[m256]); torch.cuda.synchronize()
t = time.perf_counter(); o = P.genomepredict_256Mb(c256, "chrS", [nm], chrlen, 70_000_000, 128_000_000, models=[m256]); torch.cuda.synchronize()
print("wall", time.perf_counter() - t)
pr = cProfile.Profile(); pr.enable()
o = P.genomepredict_256Mb(c256, "chrS", [nm], chrlen, 70_000_000, 128_000_000, models=[m256]); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
