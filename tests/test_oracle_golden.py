"""Pin the CPU oracle (oracle/orca_oracle.py) against fixtures generated from the
REAL reference modules by tools/make_golden.py (the reference has no tests of its
own).  CPU only."""
import numpy as np
import torch

from oracle import orca_oracle as O
from orca_amd import synth
from tests.util import golden, maxabs, stats, synth_sd

TOL = 2e-5  # same ATen CPU kernels as the reference; only op grouping may differ


def test_manifest_matches_product_modules():
    from orca_amd import orca_modules as pm
    man = golden("G0_manifest.npz")
    for cls in ("Encoder", "Encoder2", "Encoder2b", "Encoder3", "Decoder", "Decoder_1m", "Net"):
        m = pm.Net(num_1d=32) if cls == "Net" else getattr(pm, cls)()
        mine = [f"{k}|{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
        assert mine == list(man[cls]), cls


def test_encoder_blocks_and_edges():
    g = golden("G1_encoder.npz")
    sd = synth_sd("Encoder", 0)
    x = torch.from_numpy(synth.synth_sequence(1712000, seed=11, n_frac=0.01)).transpose(1, 2)
    y = O.encoder_forward(sd, x)[0].numpy()
    assert y.shape == (128, 428)
    assert maxabs(y, g["y"]) < TOL
    # block-size invariance (G2): 200 kb blocks give the same bins
    y200 = O.encoder_forward(sd, x, blocksize=4000 * 50)[0].numpy()
    assert maxabs(y200, g["y_block200k"]) < TOL
    assert maxabs(y200, y) < 1e-4
    x2 = torch.from_numpy(synth.synth_sequence(4000 * 37, seed=12)).transpose(1, 2)
    assert maxabs(O.encoder_forward(sd, x2)[0].numpy(), g["y_single"]) < TOL
    x3 = torch.from_numpy(np.random.RandomState(13).rand(1, 4, 4000 * 12).astype(np.float32))
    assert maxabs(O.encoder_forward(sd, x3)[0].numpy(), g["y_float"]) < TOL


def test_encoder_full_256mb_fixture_window():
    """G20 (the reference's Encoder on the full 256 Mb seed-2 sequence): the oracle on the 304 kb window around the 32 Mb seam
    reproduces the stored columns there (bins 7992..8007: the window's halo is the reference's own 112 kb), both strands."""
    g = golden("G20_full256m.npz")
    L, seq_seed = int(g["args"][3]), int(g["args"][4])
    codes = synth.synth_base_codes(L, seed=seq_seed)
    bins = list(g["bins"])
    sd = synth_sd("Encoder", int(g["args"][5]))
    lo_bin, hi_bin = 7992, 8008
    cols = [bins.index(b) for b in range(lo_bin, hi_bin)]
    for k in range(2):
        if k == 0:
            w = codes[lo_bin * 4000 - 112000: hi_bin * 4000 + 112000]
        else:      # reverse complement: strand position p is base L-1-p, complemented
            w = 3 - codes[L - (hi_bin * 4000 + 112000): L - (lo_bin * 4000 - 112000)][::-1]
        x = torch.zeros(1, 4, w.shape[0])
        x[0, torch.from_numpy(w.astype(np.int64)), torch.arange(w.shape[0])] = 1.0
        y = O.encoder_forward(sd, x)[0].numpy()[:, 28:28 + (hi_bin - lo_bin)]
        assert maxabs(y, g[f"enc_{k}_cols"][:, cols]) < TOL, k


def test_encoder2_encoder3():
    g = golden("G3_encoder23.npz")
    sd2 = synth_sd("Encoder2", 0)
    x = torch.from_numpy((np.random.RandomState(21).rand(1, 128, 800) * 0.5).astype(np.float32))
    ys = O.encoder2_forward(sd2, x)
    assert [y.shape[2] for y in ys] == [800, 400, 200, 100, 50, 25]
    for i, y in enumerate(ys):
        assert maxabs(y[0].numpy(), g[f"e2_{i}"]) < TOL
    xl = torch.from_numpy((np.random.RandomState(22).rand(1, 128, 8000) * 0.5).astype(np.float32))
    for i, y in enumerate(O.encoder2_forward(sd2, xl)):
        assert maxabs(y[0, :, :16].numpy(), g[f"e2L_head_{i}"]) < TOL
        np.testing.assert_allclose(stats(y.numpy()), g[f"e2L_stats_{i}"], rtol=1e-4)
    sd3 = synth_sd("Encoder3", 0)
    x3 = torch.from_numpy((np.random.RandomState(23).rand(1, 128, 2000) * 0.5).astype(np.float32))
    for i, y in enumerate(O.encoder3_forward(sd3, x3)):
        assert maxabs(y[0, :, ::5].numpy(), g[f"e3_{i}"]) < TOL


def test_decoders():
    g = golden("G5_decoder.npz")
    nm, _ = synth.synth_normmats_32m()
    x = torch.from_numpy((np.random.RandomState(31).rand(1, 128, 250) * 0.5).astype(np.float32))
    de = torch.log(torch.from_numpy(nm[8][None, None].astype(np.float32)))
    sd = synth_sd("Decoder", 0)
    p0 = O.decoder_forward(sd, x, de)
    assert maxabs(p0[0, 0].numpy(), g["noy"]) < TOL
    assert float((p0 - p0.transpose(2, 3)).abs().max()) == 0.0  # exact symmetry, orca_modules.py:488
    yc = torch.from_numpy(g["noy"][None, None, 37:162, 37:162].copy())
    assert maxabs(O.decoder_forward(sd, x, de, yc, "bilinear")[0, 0].numpy(), g["y_bilinear"]) < TOL
    assert maxabs(O.decoder_forward(sd, x, de, yc, "nearest")[0, 0].numpy(), g["y_nearest"]) < TOL
    assert maxabs(O.decoder_1m_forward(synth_sd("Decoder_1m", 0), x)[0, 0].numpy(), g["dec1m"]) < TOL


def test_zoom_index_arithmetic_matches_reference_cascade():
    g = golden("G7_cascade32.npz")
    for ci in range(4):
        mpos, wpos = (int(v) for v in g[f"c{ci}_args"])
        for reverse in (False,):
            starts = [0]
            for j, level in enumerate([32, 16, 8, 4, 2, 1]):
                starts.append(starts[j] + O.zoom_index_32m(level, starts[j], mpos, wpos, reverse) * level)
            coords = [wpos - 16000000 + s * 4000 for s in starts[:-1]]
            assert coords == list(g[f"c{ci}_start"])


def test_net_1mb_model():
    """Net (the 1 Mb model, orca_modules.py:1409-1900): oracle vs the reference on one 1 Mb sequence, with and
    without the auxiliary 1-D head."""
    g = golden("G14_net1m.npz")
    x = torch.from_numpy(synth.synth_sequence(1_000_000, seed=61, n_frac=0.002)).transpose(1, 2)
    pred, out1d = O.net_forward(synth_sd("Net", 0, num_1d=32), x, num_1d=32)
    assert pred.shape == (1, 1, 250, 250) and out1d.shape == (1, 32, 250)
    assert maxabs(pred[0, 0].numpy(), g["pred"]) < TOL
    assert maxabs(out1d[0].numpy(), g["out1d"]) < 1e-5
    pred0 = O.net_forward(synth_sd("Net", 3), x)
    assert maxabs(pred0[0, 0, ::5, ::5].numpy(), g["pred_no1d_sub"]) < TOL


def test_encoder2b():
    g = golden("G15_encoder2b.npz")
    x = torch.from_numpy((np.random.RandomState(33).rand(1, 128, 2048) * 0.5).astype(np.float32))
    outs = O.encoder2b_forward(synth_sd("Encoder2b", 0), x)
    assert [o.shape[2] for o in outs] == [2048, 1024, 512, 256, 128, 64] and outs[0] is not None
    for i in range(1, 6):
        assert maxabs(outs[i][0].numpy(), g[f"o{i}"]) < TOL
