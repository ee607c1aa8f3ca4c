"""Worker of tests/test_gpu_dist.py (launched by torch.distributed.run, one rank per GPU): the sharded Encoder over RCCL -
through the C ABI's communicator and through torch.distributed - against the same rank's unsharded encoding."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from orca_amd import dist as D
    from orca_amd import engine, synth
    from tests.util import product_module
    rank, world, dev = D.init_from_env("nccl")
    L = 4000 * 1500 + 4000 * 3          # uneven split over the ranks
    codes = torch.from_numpy(synth.synth_base_codes(L, seed=21)[None]).to(dev)
    enc = product_module("Encoder", 0)
    full = enc.forward_codes(codes)
    res = {"rank": rank, "world": world}
    comm = D.AbiComm(dev)
    for name, c in (("abi", comm), ("torch", None)):
        for rev in (False, True):
            ref = enc.forward_codes(codes, reverse=rev)
            out = D.ShardedEncoder(enc, comm=c).forward_codes(codes, reverse=rev)
            res[f"{name}_rev{int(rev)}"] = float((out - ref).abs().max())
    x = torch.from_numpy(synth.synth_sequence(4000 * 300, seed=22)).to(dev).transpose(1, 2)
    res["float_input"] = float((D.ShardedEncoder(enc, comm=comm)(x) - enc(x)).abs().max())
    # the 256 Mb tail with one strand per rank parity + one all-gather of the maps, against both strands on this rank
    from tests.test_gpu_dist import _tail_inputs
    model, enc0, chrlen, de = _tail_inputs(dev)
    f = D.strand_tail_256m(model, enc0, 0, 70_000_000, 128_000_000, chrlen, de)
    r = D.strand_tail_256m(model, enc0, 1, 70_000_000, 128_000_000, chrlen, de)
    ref = [torch.stack([engine.strand_merge(f[j, 0], r[j, 0])]) for j in range(4)]
    for name, c in (("abi", comm), ("torch", None)):
        out = D.strand_parallel_cascade_256m(model, enc0, 70_000_000, 128_000_000, chrlen, de, comm=c)
        res[f"tail_{name}"] = max(float((a - b).abs().max()) for a, b in zip(out, ref))
    comm.close()
    torch.distributed.barrier()
    print("DIST_RESULT " + json.dumps(res), flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
