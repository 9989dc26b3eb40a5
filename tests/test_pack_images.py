"""Host logic: the LDS-image weight packers (layout_dm_amd/csrc/ldm_pack.h) against the fused kernels' LDS read
formulas, compiled and run on the CPU (no GPU, no HIP)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_images_match_kernel_read_formulas(tmp_path):
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = tmp_path / "cpu_pack_check"
    subprocess.run([cxx, "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpu_pack_check.cpp"), "-o", str(exe)],
                   check=True, cwd=ROOT)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
