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


def test_binding_struct_layout_matches_header():
    from layout_dm_amd import binding

    assert ctypes.sizeof(binding.LdmConfig) == 14 * 4  # ABI 2: + q_type
    assert ctypes.sizeof(binding.LdmSampler) == 16
    assert ctypes.sizeof(binding.LdmCond) == 3 * 8 + 8  # three pointers + int32 (+pad)
    # ldm_relation: 5 pointers, int32[4], float, 2 x int32 (+4 pad)
    assert ctypes.sizeof(binding.LdmRelation) == 5 * 8 + 16 + 4 + 4 + 4 + 4
    assert binding.LdmRelation.canvas_bins.offset == 40 and binding.LdmRelation.relation_lambda.offset == 56


def test_no_gpu_fails_loudly(lib_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from layout_dm_amd import binding

    with pytest.raises(RuntimeError, match="no CPU"):
        binding.Engine(n_category=25)
    # and the C entry point itself refuses too
    lib = binding.load_library()
    cfg = binding.LdmConfig(binding.ABI_VERSION, 25, 32, 25, 5, 464, 8, 1856, 4, 100, 0, 4, 0)
    h = ctypes.c_void_p()
    rc = lib.ldm_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc != 0 and b"no HIP device" in lib.ldm_last_error(None)


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under layout_dm_amd/ may reference it."""
    pkg = os.path.join(ROOT, "layout_dm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle." not in txt, f
