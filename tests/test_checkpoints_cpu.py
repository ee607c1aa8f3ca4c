"""The reference's checkpoint files load unchanged (SURVEY.md 8b "checkpoint format to preserve",
/root/reference/orca_models.py:53-123, :449-542, :545-649): `torch.save`d OrderedDicts with DataParallel `module.`
prefixes; `orca_<cell>.net0.statedict` is the stage-a `Net` dict with a DOUBLE prefix, out of which Encoder / Decoder_1m /
Net keys are picked.  The files here are written in exactly that layout from synthetic tensors (the real 1.3 GB download
is unavailable offline) and must come back bit-identical through the containers' `model_dir=` route."""
import collections
import os

import numpy as np
import torch

from orca_amd import orca_models as M
from orca_amd import orca_modules as pm
from orca_amd import synth


def _sd(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    return {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_state_dict(shapes, seed=seed).items()}


def _save(path, sd, prefix):
    torch.save(collections.OrderedDict((prefix + k, v) for k, v in sd.items()), path)


def _same(module, sd):
    own = module.state_dict()
    assert set(own) == set(sd)
    return all(torch.equal(own[k].cpu(), sd[k].to(own[k].dtype)) for k in sd)


def test_reference_checkpoint_layout_loads(tmp_path):
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "models"))
    os.makedirs(os.path.join(root, "resources"))
    mdl = lambda name: os.path.join(root, "models", name)
    net_sd = _sd(pm.Net(num_1d=32), 5)                                   # stage-a dict: Encoder + Decoder_1m + final_1d keys
    _save(mdl("orca_h1esc.net0.statedict"), net_sd, "module.module.")
    enc2_sd = _sd(pm.Encoder2(), 6)
    _save(mdl("orca_h1esc.net.statedict"), enc2_sd, "module.")
    three = [_sd(pm.Decoder(upsample_mode="bilinear"), 10 + k) for k in range(3)]     # three distinct decoders, cycled over the levels
    dec_sd = {lv: three[i % 3] for i, lv in enumerate((1, 2, 4, 8, 16, 32, 64, 128, 256))}
    for lv in (1, 2, 4, 8, 16, 32):
        _save(mdl(f"orca_h1esc.d{lv}.statedict"), dec_sd[lv], "module.")
    np.save(os.path.join(root, "resources", "4DNFI9GMP2J8.rebinned.mcool.expected.res4000.npy"), synth.synth_expected_log(8000, 1))
    np.save(os.path.join(root, "resources", "4DNFI9GMP2J8.rebinned.mcool.expected.res1000.npy"), synth.synth_expected_log(1200, 2))

    m = M.H1esc(model_dir=root)
    assert _same(m.net, enc2_sd)
    assert all(_same(m.denets[lv], dec_sd[lv]) for lv in m.levels)
    assert _same(m.net0, {k: v for k, v in net_sd.items() if k in m.net0.state_dict()})
    assert _same(m.denet_1_pt, {k: v for k, v in net_sd.items() if k in m.denet_1_pt.state_dict()})
    ref_nm, ref_eps = synth.synth_normmats_32m(1)
    assert all(np.array_equal(m.normmats[lv], ref_nm[lv]) and m.epss[lv] == ref_eps[lv] for lv in m.levels)

    m1 = M.H1esc_1M(model_dir=root)                                      # orca_models.py:449-493: the whole Net from the same file
    assert _same(m1.net, net_sd) and m1.normmats[1].shape == (250, 250)
    e = np.exp(synth.synth_expected_log(1200, 2)[:1000])
    nm = e[np.abs(np.arange(1000)[None, :] - np.arange(1000)[:, None])].reshape(250, 4, 250, 4).mean(axis=1).mean(axis=2)
    assert np.array_equal(m1.normmats[1], nm) and m1.epss[1] == nm.min()

    enc3_sd = _sd(pm.Encoder3(), 7)                                      # 256 Mb model: own Encoder3 + decoders, shared net0 / Encoder2
    _save(mdl("orca_h1esc_256m.net.statedict"), enc3_sd, "module.")
    for lv in (32, 64, 128, 256):
        _save(mdl(f"orca_h1esc_256m.d{lv}.statedict"), dec_sd[lv], "module.")
    np.save(os.path.join(root, "resources", "4DNFI9GMP2J8.rebinned.mcool.expected.res32000.mono.npy"), -np.log1p(np.arange(8000.0)))
    np.save(os.path.join(root, "resources", "4DNFI9GMP2J8.rebinned.mcool.expected.res32000.trans.npy"), np.float64(-11.0))
    m2 = M.H1esc_256M(model_dir=root)
    assert _same(m2.net, enc3_sd) and _same(m2.net1, enc2_sd) and all(_same(m2.denets[lv], dec_sd[lv]) for lv in m2.levels)
    assert _same(m2.net0, {k: v for k, v in net_sd.items() if k in m2.net0.state_dict()})
    assert m2.background_cis.shape == (10000,) and np.isnan(m2.background_cis[8000:]).all() and m2.background_trans == np.exp(-11.0)
