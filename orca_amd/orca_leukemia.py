"""Multi-target Orca variants with the reference's names (/root/reference/orca_leukemia.py): the leukemia models
predict several Hi-C datasets at once - `Decoder(num_2d)` (:512-990) and `Decoder_1m(num_2d)` (:996-1316) take a
`num_2d`-channel background (`distenc`) and coarse prediction and end in a `64 -> max(5, num_2d) -> num_2d` head;
`Encoder2` (:1499-1601) is the contracting-path-only U-net (the same network as orca_modules.Encoder2b); `Encoder`
(:1318-1496) is the Orca Encoder; `Net(num_2d, num_1d)` (:16-509) is the 1 Mb model with a `num_2d`-map head.

They run on the same HIP kernels as the Orca models (orca_modules.py of this package: the decoders through
`orca_decoder_forward_mt`, include/orca_hip.h); state-dict keys and shapes are the reference's
(tests/golden/G16_multitarget.npz holds the manifest), so `orca_leukemia{A,B}.*.statedict` load unchanged.

Containers `OrcaLeukemiaA` (2 datasets) / `OrcaLeukemiaB` (6 datasets) keep the attribute protocol genomepredict
consumes - `.net0 .net .denets{} .denet_1_pt .normmats{} .epss{}` with 3-D `[num_2d, 250, 250]` backgrounds
(:1636-1733).  Unlike the reference module (which instantiates both models at import, :1872-1873) nothing is built
at import time; `synthetic_seed=...` gives deterministic synthetic weights for tests.
"""
import os

import numpy as np
import torch
from torch import nn

from . import orca_modules as _om
from . import synth
from .orca_models import ORCA_PATH, _load_file, _synth_into

Encoder = _om.Encoder


class Encoder2(_om.Encoder2b):
    """orca_leukemia.py:1499-1601: `lblocks` / `blocks` only; returns the six contracting-path encodings."""


class Decoder(_om.Decoder):
    """orca_leukemia.py:512-990.  `Decoder(num_2d)`; the upsampling of the coarse prediction is nearest-neighbour
    (`nn.Upsample(scale_factor=(2, 2))`, :930)."""

    def __init__(self, num_2d, precision=None):
        super().__init__(upsample_mode="nearest", precision=precision, num_2d=num_2d)


class Decoder_1m(_om.Decoder_1m):
    """orca_leukemia.py:996-1316."""

    def __init__(self, num_2d, precision=None):
        super().__init__(precision=precision, num_2d=num_2d)


class Net(_om.Net):
    """orca_leukemia.py:16-509: the 1 Mb model with `num_2d` maps (and optionally `num_1d` 1-D targets)."""

    def __init__(self, num_2d=1, num_1d=None, precision=None):
        super().__init__(num_1d=num_1d, precision=precision, num_2d=num_2d)


class _OrcaLeukemia(nn.Module):
    modelstr = None
    normmat_files = ()
    levels = (1, 2, 4, 8, 16, 32)

    def __init__(self, model_dir=None, synthetic_seed=None):
        super().__init__()
        T = len(self.normmat_files)
        self.net0 = Encoder()
        self.net = Encoder2()
        for lv in self.levels:
            setattr(self, f"denet_{lv}", Decoder(T))
        self.denet_1_pt = Decoder_1m(T)
        if synthetic_seed is not None:
            s = int(synthetic_seed)
            _synth_into(self.net0, s)
            _synth_into(self.net, s)
            for lv in self.levels:
                _synth_into(getattr(self, f"denet_{lv}"), s + lv)
            _synth_into(self.denet_1_pt, s)
            expected = [np.exp(synth.synth_expected_log(8000, s + 101 * t)) for t in range(T)]
        else:
            root = model_dir or ORCA_PATH
            base = os.path.join(root, "models", "orca_" + self.modelstr)
            if not os.path.exists(base + ".net.statedict"):
                raise FileNotFoundError(f"{base}.net.statedict not found (reference README.md:63-72), or pass synthetic_seed=...")
            _load_file(self.net, base + ".net.statedict")
            for lv in self.levels:
                _load_file(getattr(self, f"denet_{lv}"), f"{base}.d{lv}.statedict")
            _load_file(self.net0, base + ".net0.statedict", filtered=True)
            _load_file(self.denet_1_pt, base + ".net0.statedict", filtered=True)
            expected = [np.exp(np.load(os.path.join(root, "resources", f)))[:8000] for f in self.normmat_files]
        self.eval()
        idx = np.abs(np.arange(8000)[:, None] - np.arange(8000)[None, :])
        self.normmats, self.epss = {}, {}
        for lv in self.levels:   # block means per dataset (orca_leukemia.py:1703-1717)
            m = np.stack([np.reshape(e[idx[: 250 * lv, : 250 * lv]], (250, lv, 250, lv)).mean(axis=3).mean(axis=1) for e in expected])
            self.normmats[lv] = m
            self.epss[lv] = np.min(m)
        self.denets = {lv: getattr(self, f"denet_{lv}") for lv in self.levels}


class OrcaLeukemiaA(_OrcaLeukemia):
    """Orca Leukemia model A, 1-32 Mb, 2 datasets (orca_leukemia.py:1604-1733)."""
    modelstr = "leukemiaA"
    normmat_files = ("GSE134761_TALL_all.hg38.no_filter.1000.mcool.expected.res4000.npy",
                     "THP1.hg38.no_filter.1000.mcool.expected.res4000.npy")


class OrcaLeukemiaB(_OrcaLeukemia):
    """Orca Leukemia model B, 1-32 Mb, 6 datasets (orca_leukemia.py:1736-1869)."""
    modelstr = "leukemiaB"
    normmat_files = ("4DNFIXP4QG5B.mcool.rebinned.mcool.expected.res4000.npy",
                     "NALM6.hg38.no_filter.1000.mcool.expected.res4000.npy",
                     "GSE146901_T_ALL_NonETP.hg38.no_filter.1000.mcool.expected.res4000.npy",
                     "GSE146901_T_ALL_ETP.hg38.no_filter.1000.mcool.expected.res4000.npy",
                     "GSE63525_K562.hg38.no_filter.1000.mcool.expected.res4000.npy",
                     "GSE63525_KBM7.hg38.no_filter.1000.mcool.expected.res4000.npy")
