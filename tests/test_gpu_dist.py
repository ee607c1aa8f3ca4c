"""-m gpu: the multi-GPU exchange.  On a 1-GPU box only the single-rank communicator runs (RCCL is
loaded, a communicator of one rank is created through the C ABI, the all-gather is a copy); with >= 2 GPUs a
torch.distributed.run world of 2 checks the sharded Encoder (bin ranges + ONE all-gather, orca_predict.py:675-683 /
orca_modules.py:955-977) against the unsharded one, through both the C ABI's RCCL communicator and torch.distributed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_communicator_single_rank(cuda):
    from orca_amd import dist as D
    comm = D.AbiComm(cuda, world=1, rank=0)
    x = torch.from_numpy(np.random.RandomState(0).randn(2, 128, 77).astype(np.float32)).to(cuda)
    y = comm.all_gather(x)
    torch.cuda.synchronize()
    assert y.shape == (1, 2, 128, 77) and torch.equal(y[0], x)
    comm.close()


def _tail_inputs(cuda):
    from orca_amd import orca_models, orca_predict, synth
    model = orca_models.H1esc_256M(synthetic_seed=0)
    enc0 = torch.from_numpy(np.random.RandomState(3).randn(2, 128, 64000).astype(np.float32)).to(cuda)
    chrlen = 138_368_000
    nm = synth.synth_normmat_256m(chrlen, seed=0)
    de = {}
    for lv in (256, 128, 64, 32):
        w = 250 * (lv // 8)
        de[lv] = torch.log(torch.from_numpy(orca_predict._coarse_grain(nm[None, :w, :w], lv // 8, 1).astype(np.float32))[None]).to(cuda)
    return model, enc0, chrlen, de


def test_strand_parallel_tail_single_process(cuda):
    """The per-rank share of the 256 Mb tail (one strand, dist.strand_tail_256m) against the batched two-strand cascade:
    what ranks 0 and 1 of a world would all-gather, merged, must BE the single-rank result (same kernels per map)."""
    from orca_amd import dist as D
    from orca_amd import engine
    model, enc0, chrlen, de = _tail_inputs(cuda)
    mpos, wpos = 70_000_000, 128_000_000
    whole = D.strand_parallel_cascade_256m(model, enc0, mpos, wpos, chrlen, de)          # no process group: both strands here
    fwd = D.strand_tail_256m(model, enc0, 0, mpos, wpos, chrlen, de)
    rev = D.strand_tail_256m(model, enc0, 1, mpos, wpos, chrlen, de)
    assert fwd.shape == (4, 1, 250, 250) and len(whole) == 4
    assert float((fwd - rev).abs().max()) > 1e-3                                        # the strands do differ
    for j in range(4):
        assert torch.equal(engine.strand_merge(fwd[j, 0], rev[j, 0]), whole[j][0]), j


def test_sharded_encoder_world2_rccl(cuda):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 8-GPU node); the partition / reassembly logic is covered on CPU by tests/test_dist_cpu.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    rows = [json.loads(l.split("DIST_RESULT ", 1)[1]) for l in p.stdout.splitlines() if "DIST_RESULT " in l]
    assert len(rows) == 2
    for r in rows:
        for k, v in r.items():
            if k not in ("rank", "world"):
                assert v == 0.0, (k, v, r)     # same kernels on the same bins: bit-identical to the unsharded encoding


def test_bench_two_ranks_on_one_gpu_over_gloo(cuda):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with both ranks on cuda:0 and gloo as
    the process group (ORCA_BENCH_ONE_DEVICE / ORCA_BENCH_BACKEND test hooks): the whole N > 1 control flow - replica mode with the
    max-over-ranks clock, the 256 Mb section with bin-sharded Encoders, the all-gathers, one strand's tail per rank parity and the map
    gather - runs on a 1-GPU box.  The sharded maps must be the single-rank ones (checksum of the four merged maps)."""
    env = dict(os.environ, ORCA_BENCH_ONE_DEVICE="1", ORCA_BENCH_BACKEND="gloo")
    base = [os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--sharded-steps", "1", "--no-cpu-baseline"]
    p1 = subprocess.run([sys.executable] + base + ["--gpus", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert p1.returncode == 0, p1.stderr[-2000:]
    one = json.loads([l for l in p1.stdout.splitlines() if l.startswith("{")][-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 300)] + base + ["--gpus", "2"]
    p2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p2.returncode == 0, p2.stdout[-1500:] + p2.stderr[-1500:]
    lines = [l for l in p2.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # ONE JSON line, from rank 0
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["metric"] == one["metric"]
    s1, s2 = one["sharded_256mb"], two["sharded_256mb"]
    assert "error" not in s2, s2
    assert s2["n_gpus"] == 2 and s2["bins_this_rank"] == [0, 32000] and s1["bins_this_rank"] == [0, 64000]
    assert abs(s2["maps_checksum"] - s1["maps_checksum"]) <= 1e-6 * abs(s1["maps_checksum"]), (s1["maps_checksum"], s2["maps_checksum"])
