"""-m gpu: single HIP kernels (through the C ABI) vs the torch fp32 CPU reference of
the same op.  fp32 MFMA is an exact fmaf chain, so the only difference is summation
order: tolerance 2e-5 abs on O(1) outputs (observed ~1e-6)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from orca_amd import engine

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _ref_conv1d(x, w, b, relu, r1=None, r2=None):
    y = F.conv1d(x, torch.from_numpy(w), torch.from_numpy(b), padding=4)
    if relu:
        y = F.relu(y)
    if r1 is not None:
        y = y + r1
    if r2 is not None:
        y = y + r2
    return y


@pytest.mark.parametrize("cin,cout,n,tile", [
    (4, 64, 1000, 0), (64, 64, 777, 0), (64, 96, 1024, 0), (96, 96, 515, 0), (96, 128, 300, 32),
    (128, 128, 250, 32), (128, 128, 250, 64), (128, 128, 1000, 128), (128, 128, 1001, 256), (128, 128, 4096, 0),
])
def test_conv1d_k9(cuda, cin, cout, n, tile):
    rs = np.random.RandomState(cin * 1000 + cout + n)
    B = 2
    x = torch.from_numpy(rs.randn(B, cin, n).astype(np.float32))
    w = (rs.randn(cout, cin, 9) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(B, cout, n).astype(np.float32))
    r2 = torch.from_numpy(rs.randn(B, cout, n).astype(np.float32))
    for relu, ra, rb in [(False, None, None), (True, r1, None), (True, r1, r2)]:
        y = engine.conv1d(x.to(cuda), w, b, relu, None if ra is None else ra.to(cuda), None if rb is None else rb.to(cuda), tile)
        ref = _ref_conv1d(x, w, b, relu, ra, rb)
        err = float((y.cpu() - ref).abs().max())
        assert err < TOL, (cin, cout, n, tile, relu, err)


def test_conv1d_is_transpose_sensitive(cuda):
    """asymmetric weights / identity-like input catch swapped operands or cout/pos maps."""
    n, cin, cout = 256, 64, 64
    x = torch.zeros(1, cin, n)
    x[0, 3, 100] = 1.0
    w = np.zeros((cout, cin, 9), dtype=np.float32)
    w[7, 3, 2] = 5.0  # tap 2 -> output position 100 + (4 - 2) = 102
    y = engine.conv1d(x.to(cuda), w, np.zeros(cout, dtype=np.float32)).cpu()
    assert float(y[0, 7, 102]) == 5.0
    assert float(y.abs().sum()) == 5.0


@pytest.mark.parametrize("cin,cout,dil,n", [
    (64, 32, 1, 250), (32, 64, 2, 250), (64, 64, 1, 250), (64, 32, 64, 250), (32, 64, 32, 250), (129, 64, 1, 250),
    (65, 64, 1, 250), (128, 32, 1, 250), (64, 32, 16, 126), (32, 64, 8, 64),
])
def test_conv2d_3x3_dilated(cuda, cin, cout, dil, n):
    rs = np.random.RandomState(cin * 100 + cout + dil)
    B = 2 if n < 250 else 1
    x = torch.from_numpy(rs.randn(B, cin, n, n).astype(np.float32))
    w = (rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r = torch.from_numpy(rs.randn(B, cout, n, n).astype(np.float32))
    for relu, rr in [(False, None), (True, r)]:
        y = engine.conv2d(x.to(cuda), w, b, dil, relu, None if rr is None else rr.to(cuda)).cpu()
        ref = F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=dil, dilation=dil)
        if relu:
            ref = F.relu(ref)
        if rr is not None:
            ref = ref + rr
        err = float((y - ref).abs().max())
        assert err < TOL, (cin, cout, dil, n, relu, err)


@pytest.mark.parametrize("k", [2, 4, 5])
def test_maxpool(cuda, k):
    x = torch.from_numpy(np.random.RandomState(k).randn(2, 7, 1003).astype(np.float32))
    y = engine.maxpool1d(x.to(cuda), k).cpu()
    assert torch.equal(y, F.max_pool1d(x, k, k))


@pytest.mark.parametrize("cin,cout,n", [(64, 64, 1000), (64, 96, 515), (96, 96, 777), (96, 128, 300), (128, 128, 2049), (128, 128, 250)])
@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-5), ("f16x2", 2e-5), ("bf16x2", 3e-4), ("bf16", 5e-2)])
def test_conv1d_split_bf16_channel_last(cuda, cin, cout, n, precision, tol):
    """conv_bf16s.h: bf16 MFMA with split fp32 operands; bf16x3 (6 products) must be fp32-class."""
    rs = np.random.RandomState(cin + cout + n)
    B = 2
    x = torch.from_numpy(rs.randn(B, cin, n).astype(np.float32))
    w = (rs.randn(cout, cin, 9) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(B, cout, n).astype(np.float32))
    for relu, ra in [(False, None), (True, r1)]:
        y = engine.conv1d_nlc(x.transpose(1, 2).contiguous().to(cuda), w, b, precision, relu,
                              None if ra is None else ra.transpose(1, 2).contiguous().to(cuda))
        ref = _ref_conv1d(x, w, b, relu, ra, None)
        err = float((y.cpu().transpose(1, 2) - ref).abs().max())
        assert err < tol, (cin, cout, n, precision, relu, err)


@pytest.mark.parametrize("cin,cout,n", [(128, 128, 1), (128, 128, 7), (128, 128, 33), (128, 128, 55), (128, 128, 275), (128, 128, 2048), (64, 96, 100), (96, 64, 513)])
@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-5), ("f16x2", 2e-5), ("bf16x2", 3e-4), ("bf16", 5e-2)])
def test_conv1d_short_rows_channel_last(cuda, cin, cout, n, precision, tol):
    """conv_small.h: rows of <= 2048 positions (the Encoder's stages 5-7 of a local re-encode) run the K-chunks of a tile side by side - against
    torch fp32 (the chunk-after-chunk kernel of conv_bf16s.h, which longer rows run, has the same operand splits and products and another fp32
    summation order).  A batch row must not depend on the batch."""
    rs = np.random.RandomState(cin + cout + n)
    B = 3
    x = torch.from_numpy(rs.randn(B, cin, n).astype(np.float32))
    w = (rs.randn(cout, cin, 9) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(B, cout, n).astype(np.float32))
    xd, rd = x.transpose(1, 2).contiguous().to(cuda), r1.transpose(1, 2).contiguous().to(cuda)
    for relu, ra, rad in [(False, None, None), (True, r1, rd)]:
        y = engine.conv1d_nlc(xd, w, b, precision, relu, rad)
        ref = _ref_conv1d(x, w, b, relu, ra, None)
        err = float((y.cpu().transpose(1, 2) - ref).abs().max())
        assert err < tol, (cin, cout, n, precision, relu, err)
        one = engine.conv1d_nlc(xd[1:2].contiguous(), w, b, precision, relu, None if rad is None else rad[1:2].contiguous())
        assert torch.equal(one, y[1:2])


def test_conv1d_split_bf16_wide_dynamic_range(cuda):
    """3-way split keeps fp32 exponent range: inputs spanning 1e-6..1e4 stay fp32-accurate (relative)."""
    rs = np.random.RandomState(0)
    x = torch.from_numpy((rs.randn(1, 64, 512) * np.exp(rs.uniform(-14, 9, (1, 64, 512)))).astype(np.float32))
    w = (rs.randn(64, 64, 9) / 24).astype(np.float32)
    y = engine.conv1d_nlc(x.transpose(1, 2).contiguous().to(cuda), w, np.zeros(64, np.float32), "bf16x3").cpu().transpose(1, 2)
    ref = F.conv1d(x.double(), torch.from_numpy(w).double(), None, padding=4)
    scale = F.conv1d(x.double().abs(), torch.from_numpy(w).double().abs(), None, padding=4)
    assert float(((y.double() - ref).abs() / scale).max()) < 1e-6


@pytest.mark.parametrize("cin,cout,n", [(64, 64, 2000), (64, 64, 513), (64, 96, 1500), (96, 96, 777), (96, 128, 1030), (128, 128, 3000),
                                        (64, 64, 2051), (64, 64, 4097), (64, 64, 300000), (96, 96, 300001), (64, 96, 70000), (96, 128, 66001)])   # the last three: 512-position tiles (conv_p16x.h / conv_p16w1.h / conv_p16f.h)
@pytest.mark.parametrize("out_mode", [0, 1, 2])
def test_conv1d_p16_dma(cuda, cin, cout, n, out_mode):
    """conv_p16.h: planar split-fp16 activations + LDS-DMA staging; optional fused MaxPool1d(4); vs torch fp32."""
    rs = np.random.RandomState(cin + cout + n + out_mode)
    x = torch.from_numpy(rs.randn(1, cin, n).astype(np.float32))
    w = (rs.randn(cout, cin, 9) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(1, cout, n).astype(np.float32))
    for relu, ra in [(False, None), (True, r1)]:
        y = engine.conv1d_p16(x[0].t().contiguous().to(cuda), w, b, relu, None if ra is None else ra[0].t().contiguous().to(cuda), out_mode)
        ref = _ref_conv1d(x, w, b, relu, ra, None)
        if out_mode == 1:
            ref = F.max_pool1d(ref, 4, 4)
        err = float((y.cpu().t()[None] - ref).abs().max())
        assert y.shape[0] == ref.shape[2]
        assert err < 2e-5, (cin, cout, n, out_mode, relu, err)


@pytest.mark.parametrize("cin,k,n", [(96, 9, 300001), (64, 17, 70000), (96, 9, 30001), (64, 17, 9000)])
@pytest.mark.parametrize("out_mode,res", [(0, False), (1, True)])
def test_conv1d_p16_96_couts(cuda, cin, k, n, out_mode, res):
    """The 96-cout layers of P16 planes run on conv_p16x.h (16 x 16 x 32 MFMAs, 512-position tiles) from 65 536 positions on and on the
    256-position tile of conv_p16.h below: both against torch fp32 (k9 and the composed 17-tap form; plain and ReLU + residual + MaxPool1d(4))."""
    rs = np.random.RandomState(cin + k + n + out_mode)
    x = torch.from_numpy(rs.randn(1, cin, n).astype(np.float32))
    w = (rs.randn(96, cin, k) / np.sqrt(cin * k)).astype(np.float32)
    b = rs.randn(96).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(1, 96, n).astype(np.float32)) if res else None
    ref = F.conv1d(x.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=k // 2)
    ref = F.relu(ref) + r1.double() if res else ref
    ref = F.max_pool1d(ref, 4, 4) if out_mode == 1 else ref
    y = engine.conv1d_p16(x[0].t().contiguous().to(cuda), w, b, res, None if r1 is None else r1[0].t().contiguous().to(cuda), out_mode)
    err = float((y.cpu().t()[None].double() - ref).abs().max())
    assert err < 2e-5, (cin, k, n, out_mode, err)


@pytest.mark.parametrize("cin,n", [(128, 2000), (128, 323), (96, 1603), (128, 200004), (128, 4)])
@pytest.mark.parametrize("fmt", ["p16", "b16"])
def test_conv1d_p16_pool5_fused(cuda, cin, n, fmt):
    """conv_p16p5.h (out_mode 3): a 128-cout k9 conv with ReLU, residual and MaxPool1d(5) in one launch - positions dealt to the lanes with
    stride 5, the pool a register-local max over five accumulator tiles - against torch fp32 (conv, relu, + residual, max_pool1d(5, 5):
    the ragged last window is dropped), ragged tiles and sizes below one window included; P16 and B16 (bf16-rounded operands, one final
    rounding of the pooled output to bf16) formats."""
    if fmt == "b16" and cin % 32:
        pytest.skip("B16: 32 input channels per step")
    rs = np.random.RandomState(cin + n)
    x = torch.from_numpy(rs.randn(1, cin, n).astype(np.float32))
    w = (rs.randn(128, cin, 9) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(128).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(1, 128, n).astype(np.float32))
    if fmt == "b16":
        x, r1 = _bf16(x), _bf16(r1)
        w = _bf16(torch.from_numpy(w)).numpy()
    for relu, ra in [(True, r1), (False, None)]:
        y = engine.conv1d_p16(x[0].t().contiguous().to(cuda), w, b, relu, None if ra is None else ra[0].t().contiguous().to(cuda), 3, fmt=fmt)
        ref = _ref_conv1d(x, w, b, relu, ra, None)
        ref = F.max_pool1d(ref, 5, 5) if n >= 5 else ref[:, :, :0]
        assert y.shape[0] == ref.shape[2] == n // 5
        if n >= 5:
            d = (y.cpu().t()[None] - ref).abs()
            bound = 2e-5 + (ref.abs() * 2.0 ** -8 if fmt == "b16" else 0.0)
            assert bool((d <= bound).all()), (cin, n, fmt, relu, float(d.max()))


@pytest.mark.parametrize("cin,cout,n", [(64, 96, 1500), (96, 128, 1031), (64, 64, 4097), (64, 96, 300000), (96, 128, 70003)])
@pytest.mark.parametrize("fmt", ["p16", "b16"])
def test_conv1d_17_taps_planar(cuda, cin, cout, n, fmt):
    """The 17-tap form of the planar conv (a composed linear pair: ConvP16Args.k17 = twice the K-chunks, the second tap half on the input
    shifted by 9, its ninth tap skipped) against torch's own 17-tap conv1d in fp32 - kernel level, zero padding of 8 at both ends,
    ragged last tile, with and without residual."""
    rs = np.random.RandomState(cin + cout + n)
    x = torch.from_numpy(rs.randn(1, cin, n).astype(np.float32))
    w = (rs.randn(cout, cin, 17) / np.sqrt(cin * 17)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r1 = torch.from_numpy(rs.randn(1, cout, n).astype(np.float32))
    if fmt == "b16":
        x, r1 = _bf16(x), _bf16(r1)
        w = _bf16(torch.from_numpy(w)).numpy()
    for relu, ra in [(False, None), (True, r1)]:
        y = engine.conv1d_p16(x[0].t().contiguous().to(cuda), w, b, relu, None if ra is None else ra[0].t().contiguous().to(cuda), 0, fmt=fmt)
        ref = F.conv1d(x.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=8)
        if relu:
            ref = F.relu(ref)
        if ra is not None:
            ref = ref + ra.double()
        err = float((y.cpu().t()[None].double() - ref).abs().max())
        assert err < (2e-5 if fmt == "p16" else 3e-2), (cin, cout, n, fmt, relu, err)


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("cin,cout,n", [(64, 64, 2000), (64, 64, 513), (64, 96, 1500), (96, 96, 777), (96, 128, 1030), (128, 128, 3000),
                                        (64, 64, 4097), (64, 64, 300000), (96, 96, 300001), (64, 96, 70000)])
@pytest.mark.parametrize("out_mode", [0, 1, 2])
def test_conv1d_b16_dma(cuda, cin, cout, n, out_mode):
    """conv_p16.h, FMT = 1 (B16): single-plane bf16 activations, bf16 weights, ONE MFMA product, fp32 accumulate.
    Products of bf16 operands are exact in fp32, so against torch fp32 on the SAME bf16-rounded operands only the
    summation order differs (2e-5) - plus, where the output goes back to the planar bf16 storage (out_mode 0 / 1),
    one final round-to-nearest-even to bf16 (half an ulp = 2^-9 relative)."""
    rs = np.random.RandomState(cin + cout + n + out_mode)
    x = _bf16(torch.from_numpy(rs.randn(1, cin, n).astype(np.float32)))
    w = _bf16(torch.from_numpy((rs.randn(cout, cin, 9) / np.sqrt(cin * 9)).astype(np.float32))).numpy()
    b = rs.randn(cout).astype(np.float32) * 0.1
    r1 = _bf16(torch.from_numpy(rs.randn(1, cout, n).astype(np.float32)))
    for relu, ra in [(False, None), (True, r1)]:
        y = engine.conv1d_p16(x[0].t().contiguous().to(cuda), w, b, relu, None if ra is None else ra[0].t().contiguous().to(cuda), out_mode, fmt="b16")
        ref = _ref_conv1d(x, w, b, relu, ra, None)
        if out_mode == 1:
            ref = F.max_pool1d(ref, 4, 4)
        assert y.shape[0] == ref.shape[2]
        d = (y.cpu().t()[None] - ref).abs()
        bound = 2e-5 + (ref.abs() * 2.0 ** -8 if out_mode != 2 else 0.0)
        assert bool((d <= bound).all()), (cin, cout, n, out_mode, relu, float(d.max()))


@pytest.mark.parametrize("cin,cout,dil,n", [(64, 32, 1, 250), (32, 64, 2, 250), (64, 64, 1, 250), (64, 32, 8, 250), (32, 64, 4, 250), (129, 64, 1, 250),
                                            (65, 64, 1, 250), (128, 32, 1, 250), (64, 32, 8, 126), (32, 64, 8, 64), (64, 64, 2, 256), (64, 32, 4, 30)])
@pytest.mark.parametrize("kernel", ["four_row", "one_row"])
def test_conv2d_m16_dilated(cuda, cin, cout, dil, n, kernel):
    """conv2d_m16q.h / conv2d_m16.h: the Decoders' conv on M16 maps (two fp16 planes, LDS-DMA'd operands, 3 products) vs torch fp32, on
    the four-row kernel (what a batch runs on: B = 2 here) and on the one-row kernel (a single map).  The maps make a round trip
    through the 22-bit storage (inputs, residual and output each rounded once: 2^-22 relative)."""
    rs = np.random.RandomState(cin * 100 + cout + dil + n)
    B = 2 if kernel == "four_row" else 1
    x = torch.from_numpy(rs.randn(B, cin, n, n).astype(np.float32))
    w = (rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) * 0.1
    r = torch.from_numpy(rs.randn(B, cout, n, n).astype(np.float32))
    for relu, rr in [(False, None), (True, r)]:
        y = engine.conv2d_m16(x.to(cuda), w, b, dil, relu, None if rr is None else rr.to(cuda)).cpu()
        ref = F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=dil, dilation=dil)
        if relu:
            ref = F.relu(ref)
        if rr is not None:
            ref = ref + rr
        err = float((y - ref).abs().max())
        assert err < TOL, (cin, cout, dil, n, relu, err)


@pytest.mark.parametrize("precision,ulp", [("bf16", 2.0 ** -8), ("f16", 2.0 ** -11)])
@pytest.mark.parametrize("cin,cout,dil,n", [(64, 32, 1, 250), (32, 64, 8, 250), (144, 64, 1, 126), (64, 64, 4, 64)])
@pytest.mark.parametrize("kernel", ["four_row", "one_row"])
def test_conv2d_m16_single_plane(cuda, cin, cout, dil, n, precision, ulp, kernel):
    """The single-plane modes: on operands that are exactly representable the only differences from torch fp32 are the
    summation order and ONE final rounding of the output to the plane's 16-bit type (four-row kernel: a batch of 2; one-row: a single map)."""
    rs = np.random.RandomState(cin + cout + dil + n)
    dt = torch.bfloat16 if precision == "bf16" else torch.float16
    q = lambda t: t.to(dt).to(torch.float32)
    B = 2 if kernel == "four_row" else 1
    x = q(torch.from_numpy(rs.randn(B, cin, n, n).astype(np.float32)))
    w = q(torch.from_numpy((rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9)).astype(np.float32))).numpy()
    b = rs.randn(cout).astype(np.float32) * 0.1
    r = q(torch.from_numpy(rs.randn(B, cout, n, n).astype(np.float32)))
    y = engine.conv2d_m16(x.to(cuda), w, b, dil, True, r.to(cuda), precision=precision).cpu()
    ref = F.relu(F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), padding=dil, dilation=dil)) + r
    d = (y - ref).abs()
    assert bool((d <= 2e-5 + ref.abs() * ulp).all()), (cin, cout, dil, n, precision, float(d.max()))
