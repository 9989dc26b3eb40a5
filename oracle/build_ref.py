"""TEST INFRASTRUCTURE ONLY — recipe that builds `oracle/_ref/`: the reference's own sampling path as an importable,
SOURCELESS Python package, so that the REAL reference can be timed (and spot-checked) on the GPU box, where
/root/reference does not exist.

    python -m oracle.build_ref            # /root/reference/src/trainer/trainer/**/*.py  ->  oracle/_ref/trainer/**/*.pyc

What it does: every module of the reference's `trainer` package is byte-compiled WHERE IT LIES (py_compile with an
explicit output path: nothing is written under /root/reference, no source text is copied) into `oracle/_ref/`, the
CPython "sourceless distribution" layout (`pkg/mod.pyc` next to where `pkg/mod.py` would be), plus a manifest
(`oracle/_ref/MANIFEST.json`: interpreter magic, sha256 of every source the bytecode came from).  `oracle/_ref/` is
git-ignored (build output, like libldm_hip.so) and NOT gpurun-ignored, so it travels with the snapshot to the GPU box —
same image, same interpreter.  `__graft_entry__.build()` runs this when /root/reference is present.

Only `bench.py`'s `cpu_baseline` leg (kind "reference"), `__graft_entry__.smoke()` and tests may import the result, through
`oracle/ref_harness.py` (which supplies the in-memory stubs for hydra / omegaconf / torch_geometric / ...).  Nothing in the
product path does.

Reference path timed: `ConstrainedMaskAndReplaceDiffusion.sample` as `trainer/test.py:194-203` calls and times it."""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REFERENCE_ROOT = os.environ.get("LAYOUTDM_REFERENCE", "/root/reference")
SRC_PKG = os.path.join(REFERENCE_ROOT, "src", "trainer", "trainer")


def source_files():
    out = []
    for root, _dirs, files in os.walk(SRC_PKG):
        for f in sorted(files):
            if f.endswith(".py"):
                out.append(os.path.join(root, f))
    return sorted(out)


def _sha(path: str) -> str:
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def up_to_date() -> bool:
    man = os.path.join(OUT, "MANIFEST.json")
    if not os.path.exists(man):
        return False
    try:
        m = json.load(open(man))
    except Exception:
        return False
    if m.get("magic") != importlib.util.MAGIC_NUMBER.hex():
        return False
    if not os.path.isdir(SRC_PKG):
        return True  # nothing to compare with (GPU box): keep what travelled
    cur = {os.path.relpath(p, SRC_PKG): _sha(p) for p in source_files()}
    return cur == m.get("sources")


def build(force: bool = False, verbose: bool = False) -> str:
    """Returns the directory to put on sys.path (it contains the sourceless `trainer` package)."""
    if not os.path.isdir(SRC_PKG):
        if os.path.isdir(os.path.join(OUT, "trainer")):
            return OUT
        raise RuntimeError(f"{SRC_PKG} not found and no prebuilt oracle/_ref/: run this where the reference tree is")
    if not force and up_to_date():
        return OUT
    sys.dont_write_bytecode = True   # /root/reference stays pristine
    tmp = OUT + ".tmp"
    shutil.rmtree(tmp, ignore_errors=True)
    sources = {}
    for src in source_files():
        rel = os.path.relpath(src, SRC_PKG)
        dst = os.path.join(tmp, "trainer", rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the name tracebacks show (the reference's own path, so that a failure points at its file:line)
        py_compile.compile(src, cfile=dst, dfile=os.path.join("reference/src/trainer/trainer", rel), doraise=True, optimize=0)
        sources[rel] = _sha(src)
        if verbose:
            print(f"compiled {rel}", file=sys.stderr)
    json.dump({"magic": importlib.util.MAGIC_NUMBER.hex(), "python": sys.version.split()[0], "from": SRC_PKG,
               "what": "byte-compiled modules of the reference's `trainer` package (no source text); built by oracle/build_ref.py",
               "sources": sources}, open(os.path.join(tmp, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(OUT, ignore_errors=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
