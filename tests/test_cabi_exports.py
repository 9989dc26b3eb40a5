"""CPU: the C-ABI library builds/loads and exports every symbol include/ldm_hip.h declares;
without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from layout_dm_amd import build

    return build.build(verbose=False)


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ldm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ldm_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ldm_hip.h but not exported"
    from layout_dm_amd import binding

    assert set(binding.EXPORTS) == set(syms)
    assert lib.ldm_abi_version() == binding.ABI_VERSION


def test_built_library_carries_the_digest_of_this_tree(lib_path, tmp_path):
    """Build provenance: the library embeds the sha256 of the sources it was built from; a prebuilt .so
    that lags the tree is detected by content (not by mtime) and rebuilt, and `ldm_describe` reports the digest."""
    from layout_dm_amd import build

    want = build.source_digest()
    assert len(want) == 64 and build.built_digest(lib_path) == want and not build.needs_build()
    lib = ctypes.CDLL(lib_path)
    marker = (ctypes.c_char * 80).in_dll(lib, "ldm_build_source_digest")
    assert marker.value == b"LDM_SRC_DIGEST=" + want.encode()
    # a library built from other sources is recognised whatever its mtime is
    stale = tmp_path / "libldm_hip_stale.so"
    data = open(lib_path, "rb").read().replace(want.encode(), b"0" * 64)
    stale.write_bytes(data)
    os.utime(stale, None)
    assert build.built_digest(str(stale)) == "0" * 64 != want
    assert build.built_digest(str(tmp_path / "missing.so")) is None


def test_binding_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of every struct of include/ldm_hip.h as gcc lays them out == the ctypes mirrors in
    layout_dm_amd/binding.py (the header is compiled, not transcribed)."""
    import subprocess

    from layout_dm_amd import binding

    structs = {"ldm_config": binding.LdmConfig, "ldm_sampler": binding.LdmSampler, "ldm_cond": binding.LdmCond,
               "ldm_relation": binding.LdmRelation}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "ldm_hip.h")}"',
             'int main(void) {', '  printf("abi %d\\n", LDM_ABI_VERSION);']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        parts = line.split()
        if parts[0] == "abi":
            assert int(parts[1]) == binding.ABI_VERSION
            continue
        cls = structs[parts[0]]
        if parts[1] == "size":
            assert ctypes.sizeof(cls) == int(parts[2]), line
        else:
            assert getattr(cls, parts[1]).offset == int(parts[2]), line
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())
    # every field of the header's structs is mirrored (a field added to the header only would shift nothing above)
    hdr = open(os.path.join(ROOT, "include", "ldm_hip.h")).read()
    assert hdr.count("int32_t lanes;") == 1 and len(binding.LdmConfig._fields_) == 15


def test_no_gpu_fails_loudly(lib_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from layout_dm_amd import binding

    with pytest.raises(RuntimeError, match="no CPU"):
        binding.Engine(n_category=25)
    # and the C entry point itself refuses too
    lib = binding.load_library()
    cfg = binding.LdmConfig(binding.ABI_VERSION, 25, 32, 25, 5, 464, 8, 1856, 4, 100, 0, 4, 0, 0, 0)
    h = ctypes.c_void_p()
    rc = lib.ldm_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc != 0 and b"no HIP device" in lib.ldm_last_error(None)


@pytest.mark.parametrize("field,value,msg", [
    ("abi_version", 1, b"abi_version"),
    ("n_attr", 4, b"5 attribute"),
    ("n_bin", 64, b"192 classes"),                 # 25 + 4 * 64 + 2 classes
    ("d_model", 460, b"multiples of 16"),
    ("n_head", 5, b"divisible by n_head"),
    ("n_head", 4, b"head_dim > 64"),
    ("precision", 5, b"precision"),
    ("max_batch", 0, b"max_batch"),
    ("max_elem", 26, b"at most 128 tokens"),       # fast mode: one 128 x 128 score tile per (layout, head)
    ("n_layer", 0, b">= 1"),
])
def test_create_rejects_unsupported_geometry_before_touching_a_device(lib_path, field, value, msg):
    """Argument validation of ldm_create (include/ldm_hip.h) runs before the device probe, names the offending field
    and never hands back a handle; the reference's own configurations (rico25 / publaynet x medium backbone) pass it
    (and then stop at the missing GPU here)."""
    from layout_dm_amd import binding

    lib = binding.load_library()
    names = [n for n, _ in binding.LdmConfig._fields_]
    good = dict(zip(names, (binding.ABI_VERSION, 25, 32, 25, 5, 464, 8, 1856, 4, 100, binding.PREC_FAST_F16, 4, 0, 0, 0)))
    cfg = binding.LdmConfig(*[value if n == field else good[n] for n in names])
    h = ctypes.c_void_p()
    rc = lib.ldm_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc == -1 and not h.value
    assert msg in lib.ldm_last_error(None), lib.ldm_last_error(None)


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under layout_dm_amd/ may reference it."""
    pkg = os.path.join(ROOT, "layout_dm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle." not in txt, f
