"""Builds libldm_hip.so (gfx950) in-tree with hipcc.  No torch headers, no cmake."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libldm_hip.so")
SOURCES = ["ldm_api.cpp", "ldm_weights.cpp", "ldm_denoise.cpp", "ldm_loop.cpp", "ldm_dev.cpp", "ldm_fid_api.cpp", "kernels_fid.hip", "kernels_prdc.hip", "kernels_metrics.hip", "kernels_norm.hip", "kernels_gemm.hip", "kernels_gemm16.hip", "kernels_lngemm.hip", "kernels_ffn16.hip", "kernels_attn.hip",
           "kernels_attn16.hip", "kernels_attnout.hip", "kernels_stack.hip", "kernels_post.hip", "kernels_decode.hip", "kernels_relation.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


DIGEST_MARKER = b"LDM_SRC_DIGEST="


def source_digest() -> str:
    """sha256 over everything the library is built from: csrc/*, include/ldm_hip.h, the source list and the flags."""
    import hashlib

    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "ldm_hip.h"))
    for f in files:
        h.update(os.path.relpath(f, os.path.dirname(HERE)).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(SOURCES + FLAGS).encode())
    return h.hexdigest()


def built_digest(lib: str = LIB):
    """The source digest a built library carries (ldm_build_source_digest), or None."""
    if not os.path.exists(lib):
        return None
    with open(lib, "rb") as fh:
        data = fh.read()
    i = data.find(DIGEST_MARKER)
    if i < 0:
        return None
    d = data[i + len(DIGEST_MARKER):i + len(DIGEST_MARKER) + 64]
    return d.decode("ascii", "replace")


def needs_build() -> bool:
    """Content-based: a pushed tree whose prebuilt .so lags its sources rebuilds, whatever the mtimes say."""
    return built_digest() != source_digest()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    digest = source_digest()
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [cc, "-x", "hip", *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if src == "ldm_api.cpp":
            cmd.insert(3, f'-DLDM_SRC_DIGEST="{digest}"')
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    tmp = LIB + f".tmp{os.getpid()}"     # link under a temporary name, then rename: nobody dlopens a half-written file
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
