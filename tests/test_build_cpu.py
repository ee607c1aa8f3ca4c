"""Build-time properties of the hot kernels, checked by cross-compiling for gfx950 (no GPU needed): none of the MFMA kernels may spill
to scratch - an accumulator array that ends up in scratch memory (a lambda that was not inlined, a dynamically indexed register array)
still passes every parity test and silently costs 40-70 % (seen once with conv_p16p5.h)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ("conv1d_k9_p16", "conv1d_k9_ws", "conv1d_stage1", "conv1d_first_mfma", "conv2d_3x3_m16", "conv2d_dblock", "conv1d_k9_bf16s", "conv1d_k9_small", "conv2d_3x3_f16s")
# deliberate, bounded spills: the single-plane Decoder block is held to 128 VGPRs for two workgroups per CU (conv2d_dblock.h: 100 bytes per
# lane), the three-way-split fallback tile of conv_bf16s.h spills 20
TOLERATED = {"_Z20conv2d_dblock_kernelILi1ELi1ELi0EEv10DBlockArgs": 128, "_Z22conv1d_k9_bf16s_kernelILi64ELi2ELi2ELi4ELi1ELi3ELi0ELi0EEv11ConvB16Args": 32}


def test_hot_kernels_do_not_spill(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "orca_amd", "csrc")
    from concurrent.futures import ThreadPoolExecutor

    def compile_unit(unit):      # the two translation units that hold MFMA kernels
        return subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-c",
                               os.path.join(src, unit), "-o", str(tmp_path / (unit + ".o")), "-Rpass-analysis=kernel-resource-usage"],
                              capture_output=True, text=True, cwd=src, timeout=900)
    with ThreadPoolExecutor(2) as ex:
        runs = list(ex.map(compile_unit, ("orca_encoder.hip", "orca_decoder.hip")))
    for r in runs:
        assert r.returncode == 0, r.stderr[-2000:]
    stderr = "\n".join(r.stderr for r in runs)
    name, seen, spilled = None, 0, []
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and any(h in name for h in HOT):
            seen += 1
            if int(m.group(1)) > TOLERATED.get(name, 0):
                spilled.append((name, int(m.group(1))))
    assert seen > 40, seen          # the remark format changed?
    assert not spilled, spilled
