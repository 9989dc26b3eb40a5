"""CPU: the reference-side binding INTEGRATION.md shows a maintainer (section 3, the fenced Python block) must match the
C-ABI that is actually shipped — a stale stub fails at `ldm_create` (abi_version) or, worse, shifts every argument of a
call.  Checked against include/ldm_hip.h (prototypes, LDM_ABI_VERSION) and layout_dm_amd/binding.py (struct layout):
  * the `LdmConfig` field list of the stub == binding.LdmConfig._fields_ == the fields of `ldm_config` in the header;
  * the stub constructs `LdmConfig` with one value per field and abi_version == LDM_ABI_VERSION;
  * every `lib.ldm_*(...)` call passes exactly as many arguments as the header's prototype declares."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):text.index("### 3a.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1, "section 3 must hold exactly one python block"
    return ast.parse(blocks[0])


def _eval_stub():
    """Section 3a (the evaluation side): its python blocks, the FFI call last."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("### 3a."):text.index("## 4.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 2
    return [ast.parse(b) for b in blocks]


def _header():
    return open(os.path.join(ROOT, "include", "ldm_hip.h")).read()


def _prototypes(h):
    """name -> number of parameters, for every `ldm_*` function declared in the header."""
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(ldm_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def _struct_fields(h, name):
    body = dict((n, b) for b, n in re.findall(r"typedef struct \{([^{}]*)\}\s*(\w+);", h))[name]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [m.group(1) for m in re.finditer(r"\b(\w+)\s*(?:\[\d+\])?\s*;", body)]


def test_stub_struct_matches_header_and_binding():
    from layout_dm_amd import binding

    tree, h = _stub(), _header()
    cls = [n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == "LdmConfig"]
    assert cls, "the stub must define LdmConfig"
    names = [c.value for n in ast.walk(cls[0]) if isinstance(n, ast.Tuple) for c in n.elts
             if isinstance(c, ast.Constant) and isinstance(c.value, str)]
    assert names == [f[0] for f in binding.LdmConfig._fields_]
    assert names == _struct_fields(h, "ldm_config")
    samp = [n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == "LdmSampler"][0]
    snames = [t.elts[0].value for n in ast.walk(samp) if isinstance(n, ast.List) for t in n.elts if isinstance(t, ast.Tuple)]
    assert snames == [f[0] for f in binding.LdmSampler._fields_] == _struct_fields(h, "ldm_sampler")


def test_stub_abi_version_and_config_arity():
    from layout_dm_amd import binding

    tree, h = _stub(), _header()
    abi = int(re.search(r"#define LDM_ABI_VERSION (\d+)", h).group(1))
    assert abi == binding.ABI_VERSION
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "LdmConfig"]
    assert len(calls) == 1
    args = calls[0].args
    assert len(args) == len(binding.LdmConfig._fields_), "one value per ldm_config field"
    assert isinstance(args[0], ast.Constant) and args[0].value == abi, "abi_version of the stub is stale"


def test_stub_call_arity_matches_prototypes():
    tree, protos = _stub(), _prototypes(_header())
    assert {"ldm_create", "ldm_sample_loop", "ldm_decode_layouts", "ldm_load_weight", "ldm_set_tie_report"} <= set(protos)
    seen = set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and isinstance(n.func.value, ast.Name) \
                and n.func.value.id == "lib" and n.func.attr.startswith("ldm_"):
            name = n.func.attr
            assert name in protos, f"{name} is not declared in include/ldm_hip.h"
            assert len(n.args) == protos[name], f"{name}: the stub passes {len(n.args)} arguments, the header declares {protos[name]}"
            seen.add(name)
    assert {"ldm_create", "ldm_load_weight", "ldm_finalize_weights", "ldm_sample_loop", "ldm_decode_layouts"} <= seen


def test_eval_side_stub_matches_header_and_package():
    """Section 3a: the names it imports exist with the reference's signatures, and its FFI call has the header's arity."""
    imports, ffi = _eval_stub()
    protos = _prototypes(_header())
    calls = [n for n in ast.walk(ffi) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)
             and isinstance(n.func.value, ast.Name) and n.func.value.id == "lib"]
    assert [c.func.attr for c in calls] == ["ldm_layout_metrics"]
    assert len(calls[0].args) == protos["ldm_layout_metrics"]
    import importlib
    import inspect

    for node in imports.body:
        assert isinstance(node, ast.ImportFrom)
        mod = importlib.import_module(node.module)
        for a in node.names:
            assert hasattr(mod, a.name), f"{node.module}.{a.name}"
    from layout_dm_amd import metrics

    for fn in (metrics.compute_alignment, metrics.compute_overlap):
        assert list(inspect.signature(fn).parameters) == ["bbox", "mask"]     # helpers/metric.py:98,152
