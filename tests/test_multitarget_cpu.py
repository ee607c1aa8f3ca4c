"""Multi-target decoders (reference orca_leukemia.py): the oracle against the reference's outputs (G16), the
state-dict manifests of the product's classes against the reference's, the container protocol.  No GPU."""
import numpy as np
import torch

from oracle import orca_oracle as O
from orca_amd import orca_leukemia as L
from orca_amd import synth

from .util import golden, maxabs, stats

TOL = 2e-5


def _sd(module, seed):
    return synth.synth_state_dict({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed=seed)


def _inputs(T, lv):
    nm, _ = synth.synth_normmats_32m()
    x = torch.from_numpy((np.random.RandomState(71).rand(1, 128, 250) * 0.5).astype(np.float32))
    bg = np.stack([nm[lv] * (1.0 + 0.15 * t) for t in range(T)])
    return x, torch.log(torch.from_numpy(bg[None].astype(np.float32)))


def test_manifests_match_reference_classes():
    g = golden("G16_multitarget.npz")
    for name, m in (("Decoder2", L.Decoder(2)), ("Decoder6", L.Decoder(6)), ("Decoder_1m2", L.Decoder_1m(2)),
                    ("Encoder2", L.Encoder2()), ("Net2_3", L.Net(num_2d=2, num_1d=3))):
        mine = [f"{k}|{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
        assert mine == list(g[f"manifest_{name}"]), name


def test_oracle_multitarget_decoders_vs_reference():
    g = golden("G16_multitarget.npz")
    x, de = _inputs(2, 8)
    sd = _sd(L.Decoder(2), 5)
    p0 = O.decoder_forward(sd, x, de, None, "nearest")
    assert tuple(p0.shape) == (1, 2, 250, 250)
    assert maxabs(p0[0].numpy(), g["T2_noy"]) < TOL
    yc = torch.from_numpy(g["T2_noy"][None, :, 29:154, 29:154].copy())
    assert maxabs(O.decoder_forward(sd, x, de, yc, "nearest")[0].numpy(), g["T2_y"]) < TOL
    assert maxabs(O.decoder_1m_forward(_sd(L.Decoder_1m(2), 5), x)[0].numpy(), g["T2_dec1m"]) < TOL
    # six targets: hidden width of the final head = 6
    x, de = _inputs(6, 2)
    sd = _sd(L.Decoder(6), 5)
    p0 = O.decoder_forward(sd, x, de, None, "nearest")
    assert maxabs(p0[0, :, ::3, ::3].numpy(), g["T6_noy_sub"]) < TOL
    assert np.allclose(stats(p0.numpy()), g["T6_noy_stats"], rtol=1e-5)
    p1 = O.decoder_forward(sd, x, de, p0[:, :, 29:154, 29:154], "nearest")
    assert maxabs(p1[0, :, ::3, ::3].numpy(), g["T6_y_sub"]) < TOL
    assert np.allclose(stats(p1.numpy()), g["T6_y_stats"], rtol=1e-5)


def test_container_protocol():
    m = L.OrcaLeukemiaA(synthetic_seed=3)
    assert sorted(m.denets) == [1, 2, 4, 8, 16, 32]
    for lv in m.levels:
        assert m.normmats[lv].shape == (2, 250, 250) and m.epss[lv] == np.min(m.normmats[lv])
        assert m.denets[lv].num_2d == 2
    assert m.denet_1_pt.num_2d == 2 and type(m.net).__mro__[1].__name__ == "Encoder2b"
    # block means as the reference computes them (orca_leukemia.py:1703-1717)
    e = np.exp(synth.synth_expected_log(8000, 3))
    full = e[np.abs(np.arange(8000)[:, None] - np.arange(8000)[None, :])][None]
    ref4 = np.reshape(full[:, :1000, :1000], (1, 250, 4, 250, 4)).mean(axis=4).mean(axis=2)
    assert np.array_equal(m.normmats[4][0], ref4[0])
    try:
        L.Decoder(9)
    except ValueError:
        pass
    else:
        raise AssertionError("num_2d beyond the library's limit must be rejected")
