"""CPU-only checks of the C-ABI library: it loads, exports every symbol declared in
include/orca_hip.h, and fails loudly (no fallback) without a GPU."""
import os
import re

import pytest
import torch

from orca_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "orca_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(orca_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in orca_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype in orca_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.orca_abi_version() == 1


def test_num_bins_floor_chain():
    lib = _lib.load()
    assert lib.orca_encoder_num_bins(32000000) == 8000
    assert lib.orca_encoder_num_bins(1712000) == 428
    assert lib.orca_encoder_num_bins(4000 * 37 + 3999) == 37
    assert lib.orca_encoder_num_bins(3999) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_no_silent_cpu_fallback():
    from orca_amd import orca_modules as pm
    from orca_amd._lib import OrcaHipError
    lib = _lib.load()
    assert lib.orca_device_count() == 0
    enc = pm.Encoder().eval()
    with pytest.raises(OrcaHipError):
        enc(torch.zeros(1, 4, 8000))
    dec = pm.Decoder(upsample_mode="bilinear").eval()
    with pytest.raises(OrcaHipError):
        dec(torch.zeros(1, 128, 250), torch.zeros(1, 1, 250, 250))
