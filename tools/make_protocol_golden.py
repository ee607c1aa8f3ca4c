"""G21: the CALL PROTOCOL the reference's own `genomepredict` / `genomepredict_256Mb` (orca_predict.py:231-540, :543-878) use on a
model object - recorded by running those two functions, imported from /root/reference (third-party imports stubbed as in
tools/make_golden.py), on recording stand-in sub-networks that return zeros of the reference modules' output shapes.

For every call the fixture holds the attribute that was called, each tensor argument's shape / strides / dtype / storage offset /
contiguity and the returned shapes.  tests/test_gpu_protocol.py replays the log against orca_amd's containers on the MI355X (same
shapes, same strides - transposed sequence view, sliced encodings, expanded and flipped distance matrices, cropped coarse predictions)
and tests/test_protocol_cpu.py checks the attribute surface and the forward signatures without a GPU: together they are the check
that the reference's driver can drive this package's models unchanged.  Data only - no reference source.

usage: python tools/make_protocol_golden.py            (needs /root/reference; ~3 min, ~15 GB of host memory for the 256 Mb call)
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from tools.make_golden import _stub_third_party  # noqa: E402

LOG = []


def _desc(t):
    return {"shape": list(t.shape), "stride": list(t.stride()), "dtype": str(t.dtype).replace("torch.", ""), "offset": int(t.storage_offset()),
            "contiguous": bool(t.is_contiguous())}


class Rec(nn.Module):
    """Stand-in sub-network: logs the call, returns zeros of the reference module's output shape(s)."""

    def __init__(self, name, out_shapes):
        super().__init__()
        self.name, self.out_shapes = name, out_shapes

    def forward(self, *args, **kwargs):
        B = args[0].shape[0]
        outs = [torch.zeros((B,) + tuple(s)) for s in self.out_shapes(args)]
        LOG.append({"call": self.name, "args": [None if a is None else _desc(a) for a in args], "kwargs": sorted(kwargs),
                    "returns": [list(o.shape) for o in outs], "returns_list": len(outs) > 1 or self.name in ("net", "net1")})
        return outs if len(outs) > 1 or self.name in ("net", "net1") else outs[0]


class Fake32M(nn.Module):
    def __init__(self):
        super().__init__()
        self.net0 = Rec("net0", lambda a: [(128, a[0].shape[2] // 4000)])
        self.net = Rec("net", lambda a: [(128, a[0].shape[2] >> i) for i in range(6)])
        self.denets = {lv: Rec(f"denets[{lv}]", lambda a: [(1, 250, 250)]) for lv in (1, 2, 4, 8, 16, 32)}
        self.denet_1_pt = Rec("denet_1_pt", lambda a: [(1, 250, 250)])
        self.normmats = {lv: np.ones((250, 250), dtype=np.float64) * (1.0 + lv) for lv in (1, 2, 4, 8, 16, 32)}
        self.epss = {lv: 1e-3 for lv in (1, 2, 4, 8, 16, 32)}


class Fake256M(nn.Module):
    def __init__(self):
        super().__init__()
        self.net0 = Rec("net0", lambda a: [(128, a[0].shape[2] // 4000)])
        self.net1 = Rec("net1", lambda a: [(128, a[0].shape[2] >> i) for i in range(6)])
        self.net = Rec("net", lambda a: [(128, a[0].shape[2] >> i) for i in range(4)])
        self.denets = {lv: Rec(f"denets[{lv}]", lambda a: [(1, 250, 250)]) for lv in (32, 64, 128, 256)}


def main():
    _stub_third_party()
    import orca_predict as ref  # the reference, read-only

    out = {}
    with torch.no_grad():
        LOG.clear()
        seq = np.zeros((1, 32_000_000, 4), dtype=np.float32)
        res = ref.genomepredict(seq, "chrS", 16_000_000 + 1_234_567, 16_000_000, models=[Fake32M()], use_cuda=False)
        out["genomepredict"] = {"calls": list(LOG), "output_keys": sorted(res.keys()),
                                "predictions": [[list(np.shape(p)) for p in lvls] for lvls in res["predictions"]]}
        LOG.clear()
        del seq
        seq = np.zeros((1, 256_000_000, 4), dtype=np.float32)
        normmat = np.ones((8000, 8000), dtype=np.float64)
        res = ref.genomepredict_256Mb(seq, "chrS", [normmat], 138_368_000, 128_000_000 + 1_234_567, 128_000_000, models=[Fake256M()],
                                      use_cuda=False)
        out["genomepredict_256Mb"] = {"calls": list(LOG), "output_keys": sorted(res.keys()),
                                      "predictions": [[list(np.shape(p)) for p in lvls] for lvls in res["predictions"]]}
    path = os.path.join(REPO, "tests", "golden", "G21_protocol.json")
    json.dump(out, open(path, "w"), indent=0)
    print(path, {k: len(v["calls"]) for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
