"""Host logic: the LDS-image weight packers (layout_dm_amd/csrc/ldm_pack.h) against the fused kernels' LDS read
formulas, compiled and run on the CPU (no GPU, no HIP)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitize", [False, True], ids=["plain", "asan+ubsan"])
def test_weight_images_match_kernel_read_formulas(tmp_path, sanitize):
    """sanitize=True is the ASan/UBSan build of the pure-host part of the C-ABI layer: the packers
    index four differently padded arrays with three index maps each; an out-of-bounds write aborts the test."""
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = tmp_path / "cpu_pack_check"
    flags = ["-O1", "-g", "-std=c++17"]
    if sanitize:
        flags += ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    r = subprocess.run([cxx, *flags, os.path.join(ROOT, "tests", "cpu_pack_check.cpp"), "-o", str(exe)],
                       capture_output=True, text=True, cwd=ROOT)
    if sanitize and r.returncode != 0 and ("asan" in r.stderr or "ubsan" in r.stderr or "sanitize" in r.stderr):
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
