import os, time, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
from oracle import orca_oracle as O
from orca_amd import synth
from tests.util import synth_sd
sd0 = synth_sd("Encoder", 0)
x = torch.from_numpy(synth.synth_sequence(912000, seed=1)).transpose(1, 2)
sdd = synth_sd("Decoder", 0, upsample_mode="bilinear")
nm,_ = synth.synth_normmats_32m()
e = torch.from_numpy((np.random.RandomState(1).rand(1,128,250)*0.5).astype(np.float32))
de = torch.log(torch.from_numpy(nm[4][None,None].astype(np.float32)))
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    O.encoder_forward(sd0, x[:, :, :200000])
    t=time.perf_counter(); O.encoder_forward(sd0, x); te=time.perf_counter()-t
    t=time.perf_counter(); O.decoder_forward(sdd, e, de); td=time.perf_counter()-t
    print("threads", nt, "encoder 912kb %.2fs" % te, "decoder %.2fs" % td, flush=True)
