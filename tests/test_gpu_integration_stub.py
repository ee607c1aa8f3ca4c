"""-m gpu: the raw binding INTEGRATION.md section 3 shows a reference maintainer - ctypes straight onto liborca_hip.so, no orca_amd host
code on the path: BatchNorm folded in numpy here, the 28 folded convs handed to orca_net_create in forward order, orca_encoder_forward
on the reference's own input view (`seq.transpose(1, 2)` of a [B, L, 4] array, orca_predict.py:334) and orca_encoder_forward_codes on
packed bases - checked against the CPU oracle.  Keeps the documented stub honest (argument order, struct layout, strides)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import orca_oracle as O
from orca_amd import synth
from tests.util import maxabs, synth_sd

pytestmark = pytest.mark.gpu
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orca_amd", "csrc", "liborca_hip.so")


class ConvDesc(ctypes.Structure):          # orca_conv_desc (include/orca_hip.h)
    _fields_ = [("weight_host", ctypes.c_void_p), ("bias_host", ctypes.c_void_p),
                ("cout", ctypes.c_int32), ("cin", ctypes.c_int32), ("ksize", ctypes.c_int32), ("dilation", ctypes.c_int32)]


def _fold(sd, conv, bn):
    """eval-mode BatchNorm into the conv in front of it: w' = w * g / sqrt(var + 1e-5), b' = (b - mean) * g / sqrt(var + 1e-5) + beta"""
    w, b = sd[conv + ".weight"].astype(np.float64), sd[conv + ".bias"].astype(np.float64)
    s = sd[bn + ".weight"].astype(np.float64) / np.sqrt(sd[bn + ".running_var"].astype(np.float64) + 1e-5)
    return (np.ascontiguousarray((w * s[:, None, None]).astype(np.float32)),
            np.ascontiguousarray(((b - sd[bn + ".running_mean"]) * s + sd[bn + ".bias"]).astype(np.float32)))


def test_raw_ctypes_stub_encoder(cuda):
    lib = ctypes.CDLL(LIB)                 # after `import torch`: shares torch's HIP runtime
    lib.orca_last_error.restype = ctypes.c_char_p
    sd = synth_sd("Encoder", 0)
    keep, descs = [], []
    for i in range(1, 8):
        for pre, idx in ((f"lconv{i}", (0, 2) if i == 1 else (1, 3)), (f"conv{i}", (0, 3))):     # lconv_i>1 starts with its MaxPool1d
            for ci in idx:
                w, b = _fold(sd, f"{pre}.{ci}", f"{pre}.{ci + 1}")
                keep += [w, b]
                descs.append(ConvDesc(w.ctypes.data, b.ctypes.data, w.shape[0], w.shape[1], 9, 1))
    assert len(descs) == 28
    arr = (ConvDesc * 28)(*descs)
    ctx, net = ctypes.c_void_p(), ctypes.c_void_p()
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.orca_ctx_create(0, ctypes.c_void_p(stream), ctypes.byref(ctx)) == 0, lib.orca_last_error()
    assert lib.orca_net_create(ctx, 1, arr, 28, 1, ctypes.byref(net)) == 0, lib.orca_last_error()      # 1 = ORCA_NET_ENCODER
    i64 = ctypes.c_int64
    L = 4000 * 30
    seq = synth.synth_sequence(L, seed=8, n_frac=0.02)
    x = torch.from_numpy(seq).cuda().transpose(1, 2)            # [B,4,L] view of [B,L,4] storage
    out = torch.empty((1, 128, L // 4000), device="cuda")
    rc = lib.orca_encoder_forward(ctx, net, ctypes.c_void_p(x.data_ptr()), i64(x.stride(0)), i64(x.stride(1)), i64(x.stride(2)),
                                  1, i64(L), i64(0), i64(0), ctypes.c_void_p(out.data_ptr()), i64(out.stride(0)), i64(out.stride(1)), i64(0))
    assert rc == 0, lib.orca_last_error()
    torch.cuda.synchronize()
    ref = O.encoder_forward(sd, torch.from_numpy(seq).transpose(1, 2)).numpy()
    assert maxabs(out.cpu().numpy(), ref) < 1e-4          # default precision of a fresh net: exact fp32 MFMA
    # packed bases (orca_pack_sequence + orca_encoder_forward_codes), reverse complement from the same buffer, f16x2 arithmetic
    assert lib.orca_net_set_precision(net, 4) == 0, lib.orca_last_error()                                  # 4 = ORCA_PRECISION_F16X2
    codes = torch.empty((1, L), dtype=torch.uint8, device="cuda")
    packable = ctypes.c_int(0)
    assert lib.orca_pack_sequence(ctx, ctypes.c_void_p(x.data_ptr()), i64(x.stride(1)), i64(x.stride(2)), i64(L),
                                  ctypes.c_void_p(codes.data_ptr()), ctypes.byref(packable)) == 0, lib.orca_last_error()
    assert packable.value == 1
    for rev in (0, 1):
        rc = lib.orca_encoder_forward_codes(ctx, net, ctypes.c_void_p(codes.data_ptr()), i64(codes.stride(0)), rev, 1, i64(L), i64(0), i64(0),
                                            ctypes.c_void_p(out.data_ptr()), i64(out.stride(0)), i64(out.stride(1)), i64(0))
        assert rc == 0, lib.orca_last_error()
        flag = ctypes.c_int(0)
        assert lib.orca_ctx_take_overflow(ctx, ctypes.byref(flag)) == 0 and flag.value == 0
        r = ref if rev == 0 else O.encoder_forward(sd, torch.from_numpy(np.ascontiguousarray(seq[:, ::-1, ::-1])).transpose(1, 2)).numpy()
        assert maxabs(out.cpu().numpy(), r) < 1e-4, rev
    assert lib.orca_net_free(net) == 0 and lib.orca_ctx_destroy(ctx) == 0
