import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orca_amd import synth, engine
from tests.util import product_module
dev = torch.device("cuda:0")
enc = product_module("Encoder", 0); enc.precision = "bf16"
for L in (1_200_000, 12_000, 256_000):
    codes = torch.from_numpy(synth.synth_base_codes(L, seed=3))[None].to(dev)
    codes[0, L // 3: L // 3 + 300] = 4
    enc.precision = "bf16"
    os.environ.pop("ORCA_NO_STAGE1_FUSE", None)
    a = enc.forward_codes(codes).clone(); ar = enc.forward_codes(codes, reverse=True).clone()
    os.environ["ORCA_NO_STAGE1_FUSE"] = "1"
    b = enc.forward_codes(codes).clone(); br = enc.forward_codes(codes, reverse=True).clone()
    os.environ.pop("ORCA_NO_STAGE1_FUSE", None)
    enc.precision = "f32"
    c = enc.forward_codes(codes).clone(); cr = enc.forward_codes(codes, reverse=True).clone()
    print(L, "fused vs two-launch max", float((a - b).abs().max()), "rev", float((ar - br).abs().max()),
          "| vs f32: fused mean", float((a - c).abs().mean()), "max", float((a - c).abs().max()), "two-launch mean", float((b - c).abs().mean()), "max", float((b - c).abs().max()),
          "| rev fused", float((ar - cr).abs().mean()), "two", float((br - cr).abs().mean()))
