"""Host logic: the lgkmcnt bookkeeping of the fused kernels' continuous LDS-read / MFMA pipelines
(layout_dm_amd/csrc/ldm_stream_sched.h) against an independent replay of the issue order with an in-order LDS queue,
compiled and run on the CPU (no GPU, no HIP)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_head_stream_counted_waits(tmp_path):
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = tmp_path / "cpu_sched_check"
    r = subprocess.run([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(ROOT, "tests", "cpu_sched_check.cpp"), "-o", str(exe)],
                       capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0 and "sanitize" in r.stderr:
        r = subprocess.run([cxx, "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpu_sched_check.cpp"), "-o", str(exe)],
                           capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "OK:" in out.stdout, out.stdout + out.stderr
