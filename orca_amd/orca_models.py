"""Model containers with the reference's attribute protocol
(/root/reference/orca_models.py: H1esc :17-175, Hff :178-333, H1esc_1M :449-493, Hff_1M :496-542,
H1esc_256M :545-649, Hff_256M :652-760):

    .net0 .net (.net1) .denet_<level> .denet_1_pt .denets{} .normmats{} .epss{}
    (256M: .background_cis .background_trans)

Checkpoints: the reference's ``models/orca_<cell>.<part>.statedict`` files
(torch.save'd OrderedDicts, DataParallel ``module.`` prefixes; ``net0`` and
``denet_1_pt`` are filtered out of the stage-a ``Net`` dict whose keys carry a
double prefix, orca_models.py:104-123) load unchanged.  Because the 1.3 GB
weight download is not available offline, every container can alternatively be
built with deterministic synthetic weights (``synthetic_seed=...``) - that is
what the tests and bench.py use.

Differences from the reference, by design: sub-networks are NOT wrapped in
``nn.DataParallel`` (one process per GPU; multi-GPU goes through
orca_amd/dist.py), and modules hold their weights on the MI355X in the HIP
library's own layout.
"""
import os
import pathlib

import numpy as np
import torch
from torch import nn

from . import synth
from .orca_modules import Decoder, Decoder_1m, Encoder, Encoder2, Encoder2b, Encoder3, Net

ORCA_PATH = os.environ.get("ORCA_PATH", str(pathlib.Path(__file__).parent.absolute()))


def _synth_into(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=seed)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return module


def _load_file(module, path, filtered=False):
    """torch.load a reference checkpoint into ``module``.  ``filtered`` mirrors
    orca_models.py:104-123: pick this module's keys out of a larger dict."""
    sd = torch.load(path, map_location=torch.device("cpu"))
    if filtered:
        own = module.state_dict().keys()
        picked = {}
        for k in own:
            for cand in (k, "module." + k, "module.module." + k):
                if cand in sd:
                    picked[k] = sd[cand]
                    break
            else:
                raise KeyError(f"{path}: no entry for '{k}'")
        sd = picked
    module.load_state_dict(sd, strict=True)
    return module


def _pyramid(normmat, levels):
    """Block means of the 8000x8000 background (orca_models.py:139-156)."""
    normmats, epss = {}, {}
    for lv in levels:
        m = np.reshape(normmat[: 250 * lv, : 250 * lv], (250, lv, 250, lv)).mean(axis=1).mean(axis=2)
        normmats[lv] = m
        epss[lv] = np.min(m)
    return normmats, epss


class _Orca32M(nn.Module):
    """Orca 1-32 Mb model: Encoder + Encoder2 + six Decoders + Decoder_1m."""

    modelstr = None
    expected_file = None
    levels = (1, 2, 4, 8, 16, 32)

    def __init__(self, model_dir=None, synthetic_seed=None):
        super().__init__()
        self.net0 = Encoder()
        self.net = Encoder2()
        for lv in self.levels:
            setattr(self, f"denet_{lv}", Decoder(upsample_mode="bilinear"))
        self.denet_1_pt = Decoder_1m()
        if synthetic_seed is not None:
            s = int(synthetic_seed)
            _synth_into(self.net0, s)
            _synth_into(self.net, s)
            for lv in self.levels:
                _synth_into(getattr(self, f"denet_{lv}"), s + lv)
            _synth_into(self.denet_1_pt, s)
            expected_log = synth.synth_expected_log(8000, s)
        else:
            root = model_dir or ORCA_PATH
            base = os.path.join(root, "models", "orca_" + self.modelstr)
            if not os.path.exists(base + ".net.statedict"):
                raise FileNotFoundError(
                    f"{base}.net.statedict not found. Download the Orca resources (reference README.md:63-72) into "
                    f"{root}/models and {root}/resources, set ORCA_PATH, or pass synthetic_seed=... for synthetic weights.")
            _load_file(self.net, base + ".net.statedict")
            for lv in self.levels:
                _load_file(getattr(self, f"denet_{lv}"), f"{base}.d{lv}.statedict")
            _load_file(self.net0, base + ".net0.statedict", filtered=True)
            _load_file(self.denet_1_pt, base + ".net0.statedict", filtered=True)
            expected_log = np.load(os.path.join(root, "resources", self.expected_file))
        self.eval()
        idx = np.abs(np.arange(8000)[None, :] - np.arange(8000)[:, None])
        self.normmats, self.epss = _pyramid(np.exp(expected_log[idx]), self.levels)
        self.denets = {lv: getattr(self, f"denet_{lv}") for lv in self.levels}


class H1esc(_Orca32M):
    """Orca H1-ESC model (1-32Mb), orca_models.py:17-175."""
    modelstr = "h1esc"
    expected_file = "4DNFI9GMP2J8.rebinned.mcool.expected.res4000.npy"


class Hff(_Orca32M):
    """Orca HFF model (1-32Mb), orca_models.py:178-333."""
    modelstr = "hff"
    expected_file = "4DNFI643OYP9.rebinned.mcool.expected.res4000.npy"


class _Orca256M(nn.Module):
    """Orca 32-256 Mb model: Encoder + Encoder2 (net1) + Encoder3 (net) + four Decoders."""

    modelstr = None
    base32 = None
    bg_prefix = None
    levels = (32, 64, 128, 256)

    def __init__(self, model_dir=None, synthetic_seed=None):
        super().__init__()
        self.net0 = Encoder()
        self.net1 = Encoder2()
        self.net = Encoder3()
        for lv in self.levels:
            setattr(self, f"denet_{lv}", Decoder(upsample_mode="bilinear"))
        if synthetic_seed is not None:
            s = int(synthetic_seed)
            _synth_into(self.net0, s)   # shared with the 32 Mb model (orca_models.py:611-626)
            _synth_into(self.net1, s)
            _synth_into(self.net, s)
            for lv in self.levels:
                _synth_into(getattr(self, f"denet_{lv}"), s + lv)
            d = np.arange(8000, dtype=np.float64)
            cis_log = -1.1 * np.log1p(d) - 2.0 + 0.02 * np.cos(d / 53.0 + s)
            trans_log = np.float64(-12.5)
        else:
            root = model_dir or ORCA_PATH
            base = os.path.join(root, "models", "orca_" + self.modelstr)
            b32 = os.path.join(root, "models", "orca_" + self.base32)
            if not os.path.exists(base + ".net.statedict"):
                raise FileNotFoundError(f"{base}.net.statedict not found (see H1esc for how to obtain the weights)")
            _load_file(self.net, base + ".net.statedict")
            for lv in self.levels:
                _load_file(getattr(self, f"denet_{lv}"), f"{base}.d{lv}.statedict")
            _load_file(self.net0, b32 + ".net0.statedict", filtered=True)
            _load_file(self.net1, b32 + ".net.statedict", filtered=True)
            cis_log = np.load(os.path.join(root, "resources", self.bg_prefix + ".rebinned.mcool.expected.res32000.mono.npy"))
            trans_log = np.load(os.path.join(root, "resources", self.bg_prefix + ".rebinned.mcool.expected.res32000.trans.npy"))
        self.eval()
        self.background_cis = np.hstack([np.exp(cis_log), np.repeat(np.nan, 2000)])
        self.background_trans = np.exp(trans_log)
        self.denets = {lv: getattr(self, f"denet_{lv}") for lv in self.levels}


class H1esc_256M(_Orca256M):
    """Orca H1-ESC model (32-256Mb), orca_models.py:545-649."""
    modelstr, base32, bg_prefix = "h1esc_256m", "h1esc", "4DNFI9GMP2J8"


class Hff_256M(_Orca256M):
    """Orca HFF model (32-256Mb), orca_models.py:652-760."""
    modelstr, base32, bg_prefix = "hff_256m", "hff", "4DNFI643OYP9"


class _Orca1M(nn.Module):
    """Orca 1 Mb model (orca_models.py:449-542): ``Net`` with the auxiliary 1-D head, loaded from the stage-a
    checkpoint ``orca_<cell>.net0.statedict``; ``forward(x)`` returns the [B,1,250,250] map only.
    ``normmats[1]`` / ``epss[1]``: 4 kb block means of the first 1000 entries of the 1 kb expected curve."""

    modelstr = None
    expected_file = None
    num_1d = None
    expected_is_log = True

    def __init__(self, model_dir=None, synthetic_seed=None):
        super().__init__()
        self.net = Net(num_1d=self.num_1d)
        if synthetic_seed is not None:
            _synth_into(self.net, int(synthetic_seed))
            expected = np.exp(synth.synth_expected_log(1000, int(synthetic_seed)))
        else:
            root = model_dir or ORCA_PATH
            path = os.path.join(root, "models", "orca_" + self.modelstr + ".net0.statedict")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found (reference README.md:63-72), or pass synthetic_seed=...")
            _load_file(self.net, path, filtered=True)
            expected = np.exp(np.load(os.path.join(root, "resources", self.expected_file))[:1000])
        self.eval()
        normmat = expected[np.abs(np.arange(1000)[None, :] - np.arange(1000)[:, None])]
        normmat_r = np.reshape(normmat, (250, 4, 250, 4)).mean(axis=1).mean(axis=2)
        self.normmats = {1: normmat_r}
        self.epss = {1: np.min(normmat_r)}

    def forward(self, x):
        pred, _ = self.net.forward(x)
        return pred


class H1esc_1M(_Orca1M):
    """Orca H1-ESC 1 Mb model, orca_models.py:449-493."""
    modelstr = "h1esc"
    expected_file = "4DNFI9GMP2J8.rebinned.mcool.expected.res1000.npy"
    num_1d = 32


class Hff_1M(_Orca1M):
    """Orca HFF 1 Mb model, orca_models.py:496-542."""
    modelstr = "hff"
    expected_file = "4DNFI643OYP9.rebinned.mcool.expected.res1000.npy"
    num_1d = 22


class HCTnoc(nn.Module):
    """Orca HCT116 cohesin-depleted model (orca_models.py:335-446): Encoder + Encoder2b (no expanding path) + six
    Decoders with the default nearest-neighbour upsampling, and NO ``denet_1_pt`` - as in the reference, it therefore
    cannot go through ``genomepredict``'s 1 Mb level; its sub-networks are used directly."""

    modelstr = "hctnoc"
    expected_file = "4DNFILP99QJS.HCT_auxin6h.rebinned.mcool.expected.res4000.npy"
    levels = (1, 2, 4, 8, 16, 32)

    def __init__(self, model_dir=None, synthetic_seed=None):
        super().__init__()
        self.net0 = Encoder()
        self.net = Encoder2b()
        for lv in self.levels:
            setattr(self, f"denet_{lv}", Decoder())
        if synthetic_seed is not None:
            s = int(synthetic_seed)
            _synth_into(self.net0, s)
            _synth_into(self.net, s)
            for lv in self.levels:
                _synth_into(getattr(self, f"denet_{lv}"), s + lv)
            expected_log = synth.synth_expected_log(8000, s)
        else:
            root = model_dir or ORCA_PATH
            base = os.path.join(root, "models", "orca_" + self.modelstr)
            if not os.path.exists(base + ".net.statedict"):
                raise FileNotFoundError(f"{base}.net.statedict not found (reference README.md:63-72), or pass synthetic_seed=...")
            _load_file(self.net, base + ".net.statedict")
            for lv in self.levels:
                _load_file(getattr(self, f"denet_{lv}"), f"{base}.d{lv}.statedict")
            _load_file(self.net0, base + ".net0.statedict")
            expected_log = np.load(os.path.join(root, "resources", self.expected_file))
        self.eval()
        idx = np.abs(np.arange(8000)[None, :] - np.arange(8000)[:, None])
        self.normmats, self.epss = _pyramid(np.exp(expected_log[idx]), self.levels)
        self.denets = {lv: getattr(self, f"denet_{lv}") for lv in self.levels}
