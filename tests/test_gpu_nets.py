"""-m gpu: whole modules on the MI355X (HIP path through the C ABI) vs (a) the golden
fixtures generated from the real reference and (b) the CPU oracle on the same inputs.
Tolerance from BASELINE.json north_star: 1e-4 max-abs on O(1) fp32 outputs, plus
Pearson r per output."""
import numpy as np
import pytest
import torch

from oracle import orca_oracle as O
from orca_amd import synth
from tests.util import golden, maxabs, pearson, product_module, stats, synth_sd

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("bf16x3", 1e-4), ("f16x2", 1e-4), ("bf16x2", 1e-3)])
def test_encoder_vs_golden_and_chunking(cuda, precision, tol):
    TOL = tol
    g = golden("G1_encoder.npz")
    enc = product_module("Encoder", 0)
    enc.precision = precision
    x = torch.from_numpy(synth.synth_sequence(1712000, seed=11, n_frac=0.01)).to(cuda).transpose(1, 2)
    y = enc(x)[0].cpu().numpy()
    assert y.shape == (128, 428)
    assert maxabs(y, g["y"]) < TOL and pearson(y, g["y"]) > 0.99999
    # internal chunking (halo 112 kb) must not change the result: 400 kb chunks vs one chunk
    y2 = enc(x, chunk_bp=400000)[0].cpu().numpy()
    assert maxabs(y2, y) < max(2e-5, 0.05 * tol)       # (stages 5-7 of a 400 kb chunk are short rows: conv_small.h, another fp32 summation order; bf16x2 is a 3e-5-class mode)
    # bin sub-range (multi-GPU shard) equals the slice of the full result
    y3 = enc(x, bin_lo=100, bin_hi=300)[0].cpu().numpy()
    assert maxabs(y3, y[:, 100:300]) < 2e-5
    x2 = torch.from_numpy(synth.synth_sequence(4000 * 37, seed=12)).to(cuda).transpose(1, 2)
    assert maxabs(enc(x2)[0].cpu().numpy(), g["y_single"]) < TOL
    x3 = torch.from_numpy(np.random.RandomState(13).rand(1, 4, 4000 * 12).astype(np.float32)).to(cuda)
    assert maxabs(enc(x3)[0].cpu().numpy(), g["y_float"]) < TOL


def test_encoder_ragged_length_and_batch_vs_oracle(cuda):
    enc = product_module("Encoder", 3)
    sd = synth_sd("Encoder", 3)
    L = 4000 * 9 + 1234  # not a multiple of the bin size
    x = torch.from_numpy(synth.synth_sequence(L, seed=5, batch=2, n_frac=0.02)).transpose(1, 2)
    ref = O.encoder_forward(sd, x).numpy()
    y = enc(x.to(cuda)).cpu().numpy()
    assert y.shape == ref.shape == (2, 128, 9)
    assert maxabs(y, ref) < TOL


@pytest.mark.parametrize("precision", ["f16x2", "bf16", "f32", "bf16x3"])
def test_encoder_composed_linear_pairs_vs_two_conv_form_and_oracle(cuda, precision):
    """lconv1..3 run as single 17-tap convs and conv1.a o lconv1 as a 25-tap conv from the bases (weights composed on the host, ends
    redone by the edge-fix chain).  Against
    (a) the CPU oracle = the reference's two-conv form, on sequences so short that the 4 + 4 end positions of every stage
    carry weight (1 and 2 bins: 250 / 500 positions at stage 3), one-hot with N runs, reverse strand from codes, and raw
    floats; (b) the library's own less composed forms (`Encoder.form`: what extreme weights and float rows fall back to) on the same inputs."""
    from orca_amd import engine
    enc = product_module("Encoder", 5)
    enc.precision = precision
    sd = synth_sd("Encoder", 5)
    tol = 0.15 if precision == "bf16" else 1e-4      # ("f32": stage 1 composed as fp32 FMA tap sums, first_taps_f32_kernel)
    for L, seed in ((4000, 3), (8000, 4), (4000 * 3 + 777, 6)):
        xs = synth.synth_sequence(L, seed=seed, n_frac=0.03)
        x = torch.from_numpy(xs).transpose(1, 2)
        ref = O.encoder_forward(sd, x).numpy()
        xc = x.to(cuda)
        y = enc(xc).cpu().numpy()
        assert maxabs(y, ref) < tol, (L, "float rows")
        codes, ok = engine.pack_sequence(xc)
        assert ok
        yc = enc.forward_codes(codes).cpu().numpy()
        assert maxabs(yc, ref) < tol, (L, "codes")
        xr = torch.from_numpy(np.ascontiguousarray(xs[:, ::-1, ::-1])).transpose(1, 2)
        refr = O.encoder_forward(sd, xr).numpy()
        yr = enc.forward_codes(codes, reverse=True).cpu().numpy()
        assert maxabs(yr, refr) < tol, (L, "reverse codes")
        # two-conv form everywhere / conv1.a as its own launch / lout1 stored (on bf16 planes that is also stage 1 as two launches instead of
        # the one kernel that produces conv1.b's input tiles from the bases, conv_stage1.h)
        for form in ("two_conv", "lconv1_only", "stored_residual"):
            enc.form = form
            try:
                y2 = enc(xc).cpu().numpy()
                yc2 = enc.forward_codes(codes).cpu().numpy()
            finally:
                enc.form = "default"
            assert maxabs(y2, ref) < tol and maxabs(yc2, ref) < tol, form
            if precision != "bf16":
                assert maxabs(y, y2) < 2e-5 and maxabs(yc, yc2) < 2e-5, form
            elif form == "stored_residual":    # same products but for the 25-tap weights' lo part; a1 rounded to bf16 either way
                assert maxabs(yc, yc2) < 0.03, (form, maxabs(yc, yc2))
    xf = torch.from_numpy(np.random.RandomState(14).rand(1, 4, 4000 * 2).astype(np.float32))     # arbitrary float rows
    reff = O.encoder_forward(sd, xf).numpy()
    assert maxabs(enc(xf.to(cuda)).cpu().numpy(), reff) < tol
    big = torch.from_numpy(np.random.RandomState(15).rand(1, 8, 4000 * 2).astype(np.float32))     # strided rows: gathered first
    assert maxabs(enc(big.to(cuda)[:, ::2, :]).cpu().numpy(), O.encoder_forward(sd, big[:, ::2, :]).numpy()) < tol


@pytest.mark.parametrize("seed,gain", [(1, 1.0), (2, 1.0), (3, 1.6), (4, 0.6)])
def test_encoder_weight_seed_and_gain_sweep_vs_oracle(cuda, seed, gain):
    """The goldens pin ONE set of synthetic weights; the composed forms multiply weights together, so their error depends on the weight
    statistics: other seeds and other conv gains (activations 0.2x .. 4x as large) against the CPU oracle on the same input, both
    strands from packed bases, tolerance 1e-4 relative to the output's range."""
    from orca_amd import engine
    from orca_amd import orca_modules as pm
    from tests.util import shapes_of
    sd = synth.synth_state_dict(shapes_of("Encoder"), seed=seed, relu_gain=gain)
    enc = pm.Encoder()
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    enc.eval()
    xs = synth.synth_sequence(4000 * 114, seed=20 + seed, n_frac=0.01)
    x = torch.from_numpy(xs).transpose(1, 2)
    codes, ok = engine.pack_sequence(x.to(cuda))
    assert ok
    for rev in (False, True):
        xin = torch.from_numpy(np.ascontiguousarray(xs[:, ::-1, ::-1])).transpose(1, 2) if rev else x
        ref = O.encoder_forward(sd, xin).numpy()
        y = enc.forward_codes(codes, reverse=rev).cpu().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        assert maxabs(y, ref) < 1e-4 * scale and pearson(y, ref) > 0.999999, (seed, gain, rev, maxabs(y, ref), scale)


@pytest.mark.parametrize("kind", ["Encoder", "Encoder2", "Decoder"])
def test_gain_sweep_up_to_the_trip_point_of_the_fp16_range_guard(cuda, kind):
    """Range evidence without the published checkpoints (tools/range_headroom.py): the synthetic weights' conv gain is raised
    until the device range guard of the f16x2 arithmetic FIRES.  Below the trip gain the f16x2 forward must equal the exact-fp32 mode to
    1e-4 of the output's range with no warning; AT the trip gain the module must warn, redo the forward in its range-safe arithmetic
    (bf16x3 / f32) and still equal the fp32 mode - a tripped guard costs time (2.4 x per step), never a wrong result.  The trip has to
    lie well above gain 1 (the weights every other test uses) and the fp32 walk's largest activation there has to be of the fp16 range's
    order (the guard compares what the kernels split, which is not every tensor of the layer-by-layer walk: composed groups)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import range_headroom
    rep = range_headroom.trip_sweep(kinds=(kind,), dev=cuda)[kind]
    rows = rep["rows"]
    assert rep["trip_gain"] is not None and rep["trip_gain"] > 1.0, rep
    assert rows[-1]["guard_fired"] and not any(r["guard_fired"] for r in rows[:-1]), rows
    for r in rows:
        assert r["finite"] and r["rel_err_vs_f32"] < 1e-4, (kind, r)
    assert rows[-1]["max_abs_activation"] > 65504.0 / 4 and all(r["max_abs_activation"] < 65504.0 * 4 for r in rows[:-1]), rows
    assert rep["headroom_at_gain_1"] > 100, rep


def test_encoder_default_dispatch_reaches_the_current_kernels(cuda):
    """Dispatch by shape makes silent fallbacks easy: one Encoder forward from packed bases must put stage 2 on
    conv_p16x.h (timing tag -14), stage 3's pooled conv on conv_p16p5.h (-12) and stage 1's `conv1.b` / stage 3's other convs on the 64-cout
    tiles of conv_p16.h (-5), as recorded by the per-launch HIP-event timing (launches over >= 65 536 positions)."""
    from orca_amd import engine
    enc = product_module("Encoder", 0)
    codes, ok = engine.pack_sequence(torch.from_numpy(synth.synth_sequence(4000 * 300, seed=5)).transpose(1, 2).to(cuda))
    assert ok
    ctx = engine.get_context(cuda)
    enc.forward_codes(codes)
    ctx.set_timing(True)
    enc.forward_codes(codes)
    recs = ctx.get_timing()
    ctx.set_timing(False)
    tags = {(cout, tile) for cout, cin, tile, batch, n, ms, ksize in recs}
    assert (96, -14) in tags and (128, -12) in tags and (64, -5) in tags and (128, -5) in tags, sorted(tags)
    assert not any(tile in (-9, -10, -11) for _, tile in tags), sorted(tags)     # (conv_p16w1.h serves B16 planes only)


def test_encoder_composed_weights_outside_fp16_keep_the_two_conv_form(cuda):
    """A composed weight is a sum of products of folded weights and may leave the fp16 range although every single layer fits (extreme
    checkpoints): such a group must keep the reference's two-conv form instead of packing infinities.  lconv1's two convs are scaled by
    3000 each (singles ~5e2, composed ~1e6), the input by 1e-6 so that the activations stay in range; the result must be finite and
    equal the forced two-conv form and the CPU oracle."""
    from orca_amd import orca_modules as pm
    sd = {k: np.array(v, copy=True) for k, v in synth_sd("Encoder", 9).items()}
    for k in ("lconv1.0.weight", "lconv1.2.weight"):
        sd[k] = sd[k] * 3000.0
    for k in ("lconv1.0.bias", "lconv1.1.bias", "lconv1.1.running_mean"):
        sd[k] = sd[k] * 0.0
    enc = pm.Encoder()
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    enc.eval()
    x = torch.from_numpy((np.random.RandomState(3).rand(1, 4, 4000 * 3) * 1e-6).astype(np.float32))
    ref = O.encoder_forward(sd, x).numpy()
    y = enc(x.to(cuda)).cpu().numpy()
    assert np.isfinite(y).all()
    scale = float(np.abs(ref).max())
    assert maxabs(y, ref) < 1e-4 * max(1.0, scale)
    enc.form = "two_conv"
    y2 = enc(x.to(cuda)).cpu().numpy()
    assert maxabs(y, y2) <= 1e-5 * max(1.0, scale)      # lconv1 ran uncomposed in both (lconv2 / lconv3 still differ in form)


@pytest.mark.parametrize("precision", ["f16x2", "f32", "bf16x3"])
def test_encoder2_encoder3_vs_golden(cuda, precision):
    g = golden("G3_encoder23.npz")
    e2 = product_module("Encoder2", 0, precision=precision)
    x = torch.from_numpy((np.random.RandomState(21).rand(1, 128, 800) * 0.5).astype(np.float32)).to(cuda)
    ys = e2(x)
    for i, y in enumerate(ys):
        assert maxabs(y[0].cpu().numpy(), g[f"e2_{i}"]) < TOL
    xl = torch.from_numpy((np.random.RandomState(22).rand(1, 128, 8000) * 0.5).astype(np.float32)).to(cuda)
    for i, y in enumerate(e2(xl)):
        assert maxabs(y[0, :, :16].cpu().numpy(), g[f"e2L_head_{i}"]) < TOL
        np.testing.assert_allclose(stats(y.cpu().numpy()), g[f"e2L_stats_{i}"], rtol=1e-4)
    e3 = product_module("Encoder3", 0, precision=precision)
    x3 = torch.from_numpy((np.random.RandomState(23).rand(1, 128, 2000) * 0.5).astype(np.float32)).to(cuda)
    for i, y in enumerate(e3(x3)):
        assert maxabs(y[0, :, ::5].cpu().numpy(), g[f"e3_{i}"]) < TOL


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_encoder2_batch_and_strided_input_vs_oracle(cuda, precision):
    e2 = product_module("Encoder2", 1, precision=precision)
    sd = synth_sd("Encoder2", 1)
    big = torch.from_numpy((np.random.RandomState(7).rand(2, 128, 700) * 0.5).astype(np.float32))
    x = big[:, :, 30:670]  # non-contiguous, 640 bins
    ref = O.encoder2_forward(sd, x)
    ys = e2(big.to(cuda)[:, :, 30:670])
    for a, b in zip(ys, ref):
        assert maxabs(a.cpu().numpy(), b.numpy()) < TOL
    # position-strided input (every second bin of a longer tensor): the other staging route of the channel-last path
    xs = torch.from_numpy((np.random.RandomState(8).rand(2, 128, 1280) * 0.5).astype(np.float32))
    ref2 = O.encoder2_forward(sd, xs[:, :, ::2])
    for a, b in zip(e2(xs.to(cuda)[:, :, ::2]), ref2):
        assert maxabs(a.cpu().numpy(), b.numpy()) < TOL


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_decoders_vs_golden(cuda, precision, monkeypatch):
    monkeypatch.setenv("ORCA_DECODER_PRECISION", precision)
    g = golden("G5_decoder.npz")
    nm, _ = synth.synth_normmats_32m()
    x = torch.from_numpy((np.random.RandomState(31).rand(1, 128, 250) * 0.5).astype(np.float32)).to(cuda)
    de = torch.log(torch.from_numpy(nm[8][None, None].astype(np.float32))).to(cuda)
    dec = product_module("Decoder", 0, upsample_mode="bilinear")
    p0 = dec(x, de)
    assert maxabs(p0[0, 0].cpu().numpy(), g["noy"]) < TOL
    assert float((p0 - p0.transpose(2, 3)).abs().max()) == 0.0  # exactly symmetric
    yc = torch.from_numpy(g["noy"][None, None]).to(cuda)[:, :, 37:162, 37:162]  # strided crop, as in the cascade
    p1 = dec(x, de, yc)
    assert maxabs(p1[0, 0].cpu().numpy(), g["y_bilinear"]) < TOL
    assert pearson(p1[0, 0].cpu().numpy(), g["y_bilinear"]) > 0.99999
    decn = product_module("Decoder", 0, upsample_mode="nearest")
    assert maxabs(decn(x, de, yc)[0, 0].cpu().numpy(), g["y_nearest"]) < TOL
    d1m = product_module("Decoder_1m", 0)
    p3 = d1m(x)
    assert maxabs(p3[0, 0].cpu().numpy(), g["dec1m"]) < TOL
    # accumulate form used for `+ denet_1_pt(...)` (orca_predict.py:362-366)
    acc = p1.clone()
    d1m.forward_into(acc, x, accumulate=True)
    assert maxabs(acc.cpu().numpy(), (p1 + p3).cpu().numpy()) < 1e-6


@pytest.mark.parametrize("precision", ["f16x2", "f16", "bf16"])
def test_decoder_four_row_kernel_vs_one_row_kernel(cuda, precision):
    """conv2d_3x3_m16q_kernel (what a BATCH runs on: tiles of four output rows x 128 pixels, whole batch per launch) against the reference
    fixture and against the one-row kernel of rounds 2-3 (what a SINGLE map runs on) on the same inputs: a batch of 3 and its rows one by one.
    Both kernels add the same products in the same order (kernel-column-major taps, accumulators started from the bias) into fp32
    accumulators: the maps are bit-identical."""
    g = golden("G5_decoder.npz")
    nm, _ = synth.synth_normmats_32m()
    x = torch.from_numpy((np.random.RandomState(31).rand(1, 128, 250) * 0.5).astype(np.float32)).to(cuda)
    de = torch.log(torch.from_numpy(nm[8][None, None].astype(np.float32))).to(cuda)
    yc = torch.from_numpy(g["noy"][None, None]).to(cuda)[:, :, 37:162, 37:162]
    dec = product_module("Decoder", 0, upsample_mode="bilinear", precision=precision)
    xb = torch.cat([x, x.flip(2), 0.5 * x], dim=0)
    p3 = dec(xb, de.expand(3, -1, -1, -1), yc.expand(3, -1, -1, -1))            # four-row kernel
    rows = [dec(xb[b: b + 1], de, yc) for b in range(3)]                          # one-row kernel
    if precision == "f16x2":
        assert maxabs(rows[0][0, 0].cpu().numpy(), g["y_bilinear"]) < TOL
    for b in range(3):
        assert torch.equal(p3[b: b + 1], rows[b]), b      # bit for bit: a batch size must not change a map (multi-GPU strand tails run B = 1)


@pytest.mark.parametrize("precision", ["f16x2", "f16", "bf16"])
def test_decoder_batches_of_several_rounds_walk_the_maps(cuda, precision):
    """Batches of more than one round of workgroups (the SV screen's 4 strands, config 3's 8): the four-row kernel's grid holds one round and a
    workgroup walks the maps b, b + grid.y, ... of its tile, requesting the next map's first piece under the last piece of the current one
    (conv2d_m16q.h).  Same arithmetic per map: the rows of batches of 5 (odd: the walkers have 3 and 2 maps) and 8 equal the same maps decoded
    two by two (one workgroup per map and tile: no walk) and one by one (the one-row kernel) bit for bit, with and without the coarse prediction."""
    nm, _ = synth.synth_normmats_32m()
    rs = np.random.RandomState(77)
    x = torch.from_numpy((rs.rand(8, 128, 250) * 0.5).astype(np.float32)).to(cuda)
    de = torch.log(torch.from_numpy(nm[8][None, None].astype(np.float32))).to(cuda)
    yc = torch.from_numpy(rs.randn(8, 1, 125, 125).astype(np.float32)).to(cuda)
    dec = product_module("Decoder", 0, upsample_mode="bilinear", precision=precision)
    for with_y in (True, False):
        pairs = torch.cat([dec(x[b: b + 2], de.expand(2, -1, -1, -1), yc[b: b + 2] if with_y else None) for b in range(0, 8, 2)])
        for B in (5, 8):
            walk = dec(x[:B], de.expand(B, -1, -1, -1), yc[:B] if with_y else None)
            assert torch.equal(walk, pairs[:B]), (B, with_y)
    one = dec(x[4:5], de, yc[4:5])
    assert torch.equal(dec(x[:5], de.expand(5, -1, -1, -1), yc[:5])[4:5], one)


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_decoder_batch_sliced_input_vs_oracle(cuda, precision, monkeypatch):
    monkeypatch.setenv("ORCA_DECODER_PRECISION", precision)
    nm, _ = synth.synth_normmats_32m()
    sd = synth_sd("Decoder", 2, upsample_mode="bilinear")
    dec = product_module("Decoder", 2, upsample_mode="bilinear")
    enc = torch.from_numpy((np.random.RandomState(9).rand(2, 128, 400) * 0.5).astype(np.float32))
    de = torch.log(torch.from_numpy(nm[2][None, None].astype(np.float32))).expand(2, -1, -1, -1)
    yc = torch.from_numpy(np.random.RandomState(10).randn(2, 1, 125, 125).astype(np.float32))
    ref = O.decoder_forward(sd, enc[:, :, 77:327], de, yc, "bilinear").numpy()
    out = dec(enc.to(cuda)[:, :, 77:327], de.to(cuda), yc.to(cuda)).cpu().numpy()
    assert maxabs(out, ref) < TOL


def test_f16x2_overflow_guard_falls_back(cuda):
    """|activation| beyond the fp16 range must not produce a silent wrong result: the device flag
    fires and the forward is redone with the range-safe bf16x3 split."""
    import warnings
    enc = product_module("Encoder", 0)
    sd = synth_sd("Encoder", 0)
    L = 4000 * 8
    x = torch.from_numpy(synth.synth_sequence(L, seed=3)).transpose(1, 2) * 3.0e6   # first-layer output ~1e6
    ref = O.encoder_forward(sd, x).numpy()
    enc.precision = "f16x2"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        y = enc(x.to(cuda)).cpu().numpy()
    assert any("fp16 range" in str(m.message) for m in w)
    assert np.isfinite(y).all()
    assert maxabs(y / np.abs(ref).max(), ref / np.abs(ref).max()) < 1e-4


def test_decoder_f16x2_overflow_guard_falls_back(cuda):
    import warnings
    nm, _ = synth.synth_normmats_32m()
    sd = synth_sd("Decoder", 0, upsample_mode="bilinear")
    dec = product_module("Decoder", 0, upsample_mode="bilinear")
    dec.precision = "f16x2"
    x = torch.from_numpy((np.random.RandomState(5).rand(1, 128, 250) * 1.0e5).astype(np.float32))   # x_i + x_j > 65504
    de = torch.log(torch.from_numpy(nm[8][None, None].astype(np.float32)))
    ref = O.decoder_forward(sd, x, de, None, "bilinear").numpy()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = dec(x.to(cuda), de.to(cuda)).cpu().numpy()
    assert any("fp16 range" in str(m.message) for m in w)
    s = np.abs(ref).max()
    assert maxabs(out / s, ref / s) < 1e-4


def test_encoder_from_packed_codes_and_reverse_complement(cuda):
    """orca_pack_sequence + orca_encoder_forward_codes: 1 byte/base input, reverse complement derived on the device,
    vs the oracle on the explicit float strands; non-packable rows are detected."""
    from orca_amd import engine
    enc = product_module("Encoder", 0)
    sd = synth_sd("Encoder", 0)
    L = 4000 * 30
    seq = synth.synth_sequence(L, seed=21, n_frac=0.02, batch=2)          # [2, L, 4] with N runs
    x = torch.from_numpy(seq).to(cuda).transpose(1, 2)
    codes, ok = engine.pack_sequence(x)
    assert ok and codes.shape == (2, L) and int(codes.max()) == 4
    for precision in ("f16x2", "bf16x3"):
        enc.precision = precision
        yf = enc.forward_codes(codes).cpu().numpy()
        yr = enc.forward_codes(codes, reverse=True).cpu().numpy()
        ref_f = O.encoder_forward(sd, torch.from_numpy(seq).transpose(1, 2)).numpy()
        ref_r = O.encoder_forward(sd, torch.from_numpy(np.ascontiguousarray(seq[:, ::-1, ::-1])).transpose(1, 2)).numpy()
        assert maxabs(yf, ref_f) < TOL and maxabs(yr, ref_r) < TOL, precision
        # a bin sub-range on the reverse strand (multi-GPU shard) and small internal chunks
        part = enc.forward_codes(codes, reverse=True, bin_lo=7, bin_hi=19, chunk_bp=40000).cpu().numpy()
        assert maxabs(part, yr[:, :, 7:19]) < 2e-5
    bad = x.clone()
    bad[0, :, 1234] = torch.tensor([0.3, 0.2, 0.4, 0.1], device=cuda)
    _, ok2 = engine.pack_sequence(bad)
    assert not ok2


def test_net_1mb_model_vs_reference_and_oracle(cuda):
    """Net / H1esc_1M on the HIP kernels: the reference's output for one 1 Mb sequence (G14), a batch of
    two against the oracle, the container's forward, and DataParallel-prefixed checkpoint keys."""
    from orca_amd import orca_models as M
    g = golden("G14_net1m.npz")
    net = product_module("Net", 0, device=cuda, num_1d=32)
    x = torch.from_numpy(synth.synth_sequence(1_000_000, seed=61, n_frac=0.002)).to(cuda).transpose(1, 2)
    pred, out1d = net(x)
    assert pred.shape == (1, 1, 250, 250) and out1d.shape == (1, 32, 250)
    assert maxabs(pred[0, 0].cpu().numpy(), g["pred"]) < 1e-4
    assert maxabs(out1d[0].cpu().numpy(), g["out1d"]) < 1e-5
    net0 = product_module("Net", 3, device=cuda)
    assert maxabs(net0(x)[0, 0, ::5, ::5].cpu().numpy(), g["pred_no1d_sub"]) < 1e-4
    # batch of two shorter sequences (n = 126 bins) against the oracle
    xb = torch.from_numpy(synth.synth_sequence(504_000, seed=62, batch=2)).transpose(1, 2)
    pb, ob = net(xb.to(cuda))
    rb, rob = O.net_forward(synth_sd("Net", 0, num_1d=32), xb, num_1d=32)
    assert maxabs(pb.cpu().numpy(), rb.numpy()) < 1e-4 and maxabs(ob.cpu().numpy(), rob.numpy()) < 1e-5
    # container: same weights as Net seed 0, map only; reference checkpoint keys carry 'module.' prefixes
    m = M.H1esc_1M(synthetic_seed=0).to(cuda)
    assert maxabs(m(x)[0, 0].cpu().numpy(), g["pred"]) < 1e-4 and sorted(m.normmats) == [1]
    sd = {"module." + k: v for k, v in net.state_dict().items()}
    net2 = type(net)(num_1d=32)
    net2.load_state_dict(sd)
    assert maxabs(net2.to(cuda)(x)[0].cpu().numpy(), pred.cpu().numpy()) == 0.0


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_encoder2b_vs_reference_and_hctnoc_container(cuda, precision):
    """Encoder2b (the HCTnoc variant: contracting path only) vs the reference fixture G15 and, batched and strided,
    vs the oracle; the HCTnoc container exposes the reference's attributes (no denet_1_pt, nearest-upsampling decoders)."""
    from orca_amd import orca_models as M
    g = golden("G15_encoder2b.npz")
    e2b = product_module("Encoder2b", 0, device=cuda, precision=precision)     # split-operand channel-last path / exact fp32 kernels
    x = torch.from_numpy((np.random.RandomState(33).rand(1, 128, 2048) * 0.5).astype(np.float32))
    outs = e2b(x.to(cuda))
    assert len(outs) == 6 and maxabs(outs[0].cpu().numpy(), x.numpy()) == 0.0
    for i in range(1, 6):
        assert maxabs(outs[i][0].cpu().numpy(), g[f"o{i}"]) < 1e-4
    xb = torch.from_numpy(np.random.RandomState(34).randn(2, 128, 640).astype(np.float32) * 0.4)
    ref = O.encoder2b_forward(synth_sd("Encoder2b", 0), xb[:, :, ::2])
    for a, b in zip(e2b(xb.to(cuda)[:, :, ::2]), ref):
        assert maxabs(a.cpu().numpy(), b.numpy()) < 1e-4
    m = M.HCTnoc(synthetic_seed=0)
    assert sorted(m.denets) == [1, 2, 4, 8, 16, 32] and not hasattr(m, "denet_1_pt") and type(m.net).__name__ == "Encoder2b"
    assert m.denets[8]._upsample == 0 and sorted(m.normmats) == [1, 2, 4, 8, 16, 32]


@pytest.mark.parametrize("nbins", [1, 2, 3, 5, 7, 28, 29, 57, 113, 251, 253, 512, 1025, 2051])
def test_encoder_length_sweep_split_fp16_vs_exact_fp32(cuda, nbins):
    """The default arithmetic (P16 / split-fp16 MFMA kernels, fused pooling, ragged last tiles) against the independent
    exact-fp32-MFMA code path of the same library over awkward lengths, float and packed input, both strands."""
    L = 4000 * nbins
    seq = synth.synth_sequence(L, seed=100 + nbins, n_frac=0.01)
    x = torch.from_numpy(seq).to(cuda).transpose(1, 2)
    enc = product_module("Encoder", 0, device=cuda)
    enc.precision = "f32"
    ref = enc(x)
    enc.precision = "f16x2"
    got = enc(x)
    assert got.shape == (1, 128, nbins) and maxabs(got.cpu().numpy(), ref.cpu().numpy()) < 1e-4
    from orca_amd import engine
    codes, ok = engine.pack_sequence(x)
    assert ok
    assert maxabs(enc.forward_codes(codes).cpu().numpy(), ref.cpu().numpy()) < 1e-4
    xr = torch.from_numpy(np.ascontiguousarray(seq[:, ::-1, ::-1])).to(cuda).transpose(1, 2)
    enc.precision = "f32"
    ref_r = enc(xr)
    enc.precision = "f16x2"
    assert maxabs(enc.forward_codes(codes, reverse=True).cpu().numpy(), ref_r.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("n", [2, 4, 18, 62, 126, 250, 256])
def test_decoder_size_sweep_split_fp16_vs_exact_fp32(cuda, n):
    """Decoder / Decoder_1m on map sizes from 2 to the 256-pixel pitch: split-fp16 kernels (XCD-banded rows, LDS-transposed
    epilogue, border tap skipping with dilations up to 64 > n) against the exact-fp32 kernels of the same library."""
    rs = np.random.RandomState(n)
    x = torch.from_numpy(rs.randn(2, 128, n).astype(np.float32) * 0.5).to(cuda)
    de = torch.from_numpy(rs.randn(2, 1, n, n).astype(np.float32)).to(cuda)
    y = torch.from_numpy(rs.randn(2, 1, n // 2, n // 2).astype(np.float32)).to(cuda)
    for cls, args in (("Decoder", (x, de)), ("Decoder", (x, de, y)), ("Decoder_1m", (x,))):
        kw = {"upsample_mode": "bilinear"} if cls == "Decoder" else {}
        m = product_module(cls, 1, device=cuda, **kw)
        m.precision = "f32"
        ref = m(*args)
        m.precision = "f16x2"
        got = m(*args)
        assert got.shape == (2, 1, n, n) and bool(torch.isfinite(got).all())
        assert maxabs(got.cpu().numpy(), ref.cpu().numpy()) < 1e-4, (cls, len(args))
        assert maxabs(got.cpu().numpy(), got.transpose(2, 3).cpu().numpy()) < 1e-6     # symmetrised output
