"""Multi-target decoders (reference orca_leukemia.py) on the HIP kernels: against the reference's outputs (G16) and,
through genomepredict with 3-D backgrounds, against the oracle cascade composed here."""
import numpy as np
import pytest
import torch

from oracle import orca_oracle as O
from orca_amd import orca_leukemia as L
from tests import standins
from orca_amd import orca_predict as P
from orca_amd import synth

from .util import golden, maxabs, pearson, stats

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _sd(module, seed):
    return synth.synth_state_dict({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed=seed)


def _load(module, seed):
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in _sd(module, seed).items()}, strict=True)
    return module.eval()


def _inputs(T, lv, cuda):
    nm, _ = synth.synth_normmats_32m()
    x = torch.from_numpy((np.random.RandomState(71).rand(1, 128, 250) * 0.5).astype(np.float32)).to(cuda)
    bg = np.stack([nm[lv] * (1.0 + 0.15 * t) for t in range(T)])
    return x, torch.log(torch.from_numpy(bg[None].astype(np.float32))).to(cuda)


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_multitarget_decoders_vs_reference(cuda, precision, monkeypatch):
    monkeypatch.setenv("ORCA_DECODER_PRECISION", precision)
    g = golden("G16_multitarget.npz")
    x, de = _inputs(2, 8, cuda)
    dec = _load(L.Decoder(2), 5)
    p0 = dec(x, de)
    assert tuple(p0.shape) == (1, 2, 250, 250)
    assert maxabs(p0[0].cpu().numpy(), g["T2_noy"]) < TOL
    assert float((p0 - p0.transpose(2, 3)).abs().max()) == 0.0
    yc = torch.from_numpy(g["T2_noy"][None]).to(cuda)[:, :, 29:154, 29:154]   # strided crop, as in the cascade
    p1 = dec(x, de, yc)
    assert maxabs(p1[0].cpu().numpy(), g["T2_y"]) < TOL
    assert pearson(p1[0].cpu().numpy(), g["T2_y"]) > 0.99999
    d1m = _load(L.Decoder_1m(2), 5)
    p3 = d1m(x)
    assert maxabs(p3[0].cpu().numpy(), g["T2_dec1m"]) < TOL
    acc = p1.clone()
    d1m.forward_into(acc, x, accumulate=True)
    assert maxabs(acc.cpu().numpy(), (p1 + p3).cpu().numpy()) < 1e-6
    # six targets (hidden width 6), batch of 2 with a broadcast background
    x6, de6 = _inputs(6, 2, cuda)
    dec6 = _load(L.Decoder(6), 5)
    q0 = dec6(torch.cat([x6, x6 * 0.5]), de6.expand(2, -1, -1, -1))
    assert maxabs(q0[0, :, ::3, ::3].cpu().numpy(), g["T6_noy_sub"]) < TOL
    assert np.allclose(stats(q0[0].cpu().numpy()), g["T6_noy_stats"], rtol=1e-4)
    q1 = dec6(x6, de6, q0[:1, :, 29:154, 29:154])
    assert maxabs(q1[0, :, ::3, ::3].cpu().numpy(), g["T6_y_sub"]) < TOL
    ref_b1 = O.decoder_forward(_sd(L.Decoder(6), 5), (x6 * 0.5).cpu(), de6.cpu(), None, "nearest")
    assert maxabs(q0[1].cpu().numpy(), ref_b1[0].numpy()) < TOL


def test_wrong_channel_counts_are_rejected(cuda):
    x, de = _inputs(2, 8, cuda)
    dec = _load(L.Decoder(2), 5)
    with pytest.raises(ValueError):
        dec(x, de[:, :1])
    with pytest.raises(ValueError):
        dec(x, de, torch.zeros(1, 1, 125, 125, device=cuda))


class _FakeEncoderLeukemia(torch.nn.Module):
    def __init__(self, full):
        super().__init__()
        self.net0 = standins.FakeNet0(nbins=8000, seed=0).cuda()
        self.net, self.denets, self.denet_1_pt = full.net, full.denets, full.denet_1_pt
        self.normmats, self.epss = full.normmats, full.epss


def test_genomepredict_with_multitarget_model_vs_oracle(cuda):
    """genomepredict over a 2-dataset model (3-D normmats, 2-channel coarse predictions, `+ denet_1_pt`), the Encoder
    replaced by the cheap stand-in so that the oracle side finishes in seconds."""
    full = L.OrcaLeukemiaA(synthetic_seed=4)
    model = _FakeEncoderLeukemia(full)
    seq = synth.synth_sequence(320000, seed=43)
    mpos, wpos = 16000000 + 2345678, 16000000
    out = P.genomepredict(seq, "chrS", mpos, wpos, models=[model], use_cuda=True)
    preds = out["predictions"][0]
    assert len(preds) == 6 and all(p.shape == (2, 250, 250) for p in preds)
    # oracle cascade (orca_predict.py:316-523), composed from the pinned pieces
    sds = {"net": _sd(full.net, 4), "pt": _sd(full.denet_1_pt, 4), "d": {lv: _sd(full.denets[lv], 4 + lv) for lv in full.levels}}
    fake = standins.FakeNet0(nbins=8000, seed=0)
    allp, starts0 = [], None
    for k, s in enumerate([seq, seq[:, ::-1, ::-1].copy()]):
        enc0 = fake(torch.from_numpy(np.ascontiguousarray(s)).transpose(1, 2))
        encs = dict(zip([1, 2, 4, 8, 16, 32], O.encoder2b_forward(sds["net"], enc0)))
        ps, starts, si = [], [0], 0
        for j, level in enumerate([32, 16, 8, 4, 2, 1]):
            b = int(starts[j] / level)
            e = encs[level][:, :, b: b + 250]
            de = torch.log(torch.from_numpy(full.normmats[level].astype(np.float32)))[None]
            coarse = ps[j - 1][:, :, si: si + 125, si: si + 125] if j > 0 else None
            p = O.decoder_forward(sds["d"][level], e, de, coarse, "nearest")
            if level == 1:
                p = p + O.decoder_1m_forward(sds["pt"], e)
            si = O.zoom_index_32m(level, starts[j], mpos, wpos, k == 1)
            starts.append(starts[j] + si * level)
            ps.append(p)
        allp.append(ps)
        starts0 = starts0 or starts[:-1]
    assert out["start_coords"] == [wpos - 16000000 + s * 4000 for s in starts0]
    for j in range(6):
        ref = allp[0][j].numpy()[0] * 0.5 + allp[1][j].numpy()[0, :, ::-1, ::-1] * 0.5
        assert maxabs(preds[j], ref) < TOL, j
        assert pearson(preds[j], ref) > 0.99999
