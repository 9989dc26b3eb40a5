"""CPU: the bench lines the GPU box produced in the evidence runs (profiles/r03_final*_bench.json, written by `python bench.py`
with no flags) against the driver's contract — the keys, their types and the arithmetic that ties them together.  A change of
bench.py that breaks the contract shows up here as soon as a new line is committed."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines():
    files = sorted(set(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_final*_bench.json"))))
    if not files:
        pytest.skip("no committed bench line")
    out = []
    for f in files:
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        assert len(lines) == 1, f"{f}: bench.py must print ONE JSON line"
        out.append(pytest.param(json.loads(lines[0]), id=os.path.basename(f)))
    return out


@pytest.mark.parametrize("d", _lines() if glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_final*_bench.json")) else [])
def test_bench_line_contract(d):
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(",")[0] in base["metric"] and d["unit"] == "layouts/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong") and "synthetic" in d["data"]
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    # value = layouts of one step / time of one step (512 layouts per GPU on config 2)
    assert "batch=512" in d["config"]["workload"]
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic work per launch / that kernel's average launch duration
    assert abs(r["achieved"] - r["algorithmic_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 2e-3 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == "layouts/s" and c["value"] > 0 and c["cores"] >= 1
    assert isinstance(c["sample"], str) and c["sample"]
    # the extras of the N = 1 line
    assert set(d["modes"]) >= {"exact", "fast", "fast_verified"} and d["modes"]["fast_verified"]["tokens_equal_exact_mode"]
    assert set(d["configs"]) >= {"3", "4", "refinement", "relation"}
    assert len(d["tokens_sha256"]["sha256"]) == 64
    if "config2_verbatim" in d:   # r04 lines
        v = d["config2_verbatim"]
        assert v["tokens_equal_exact_mode"] is True and v["steps"] >= 10 and "greedy" in v["workload"]
        assert "split" in d["modes"] and d["modes"]["exact"]["steps"] >= 10 and d["modes"]["split"]["steps"] >= 10
        nd = d["modes"]["fast_verified"]["nondegenerate"]
        assert all(nd[k]["tokens_equal_exact_mode"] for k in nd if k.startswith("from_step_"))
        assert set(d["configs"]) >= {"5_refinement_T200", "5_relation_T200"}
        assert d["config"]["cpu_baseline_kind"] == d["cpu_baseline"]["kind"] and d["config"]["library"]["knobs"] == ""
        assert d["world_size_seen"] == 1 and d["scaling_point"]["layouts_per_s_per_gpu"] == d["configs"]["4"]["value"]
    if "batch_shapes" in d:       # r05 lines: what a user gets — the reference-precision figures as scalars, the reference as the CPU baseline
        assert d["cpu_baseline"]["kind"] == "reference" and "oracle/_ref" in d["cpu_baseline"]["source"]
        assert d["reference_precision_layouts_per_s"] == d["modes"]["split"]["value"] == d["config"]["reference_precision_layouts_per_s"]
        assert d["fp32_mfma_layouts_per_s"] == d["modes"]["exact"]["value"]
        for point in ("mid", "wide"):
            assert d["config"][f"auto_selected_{point}"] in ("fast_verified", "hybrid_verified", "mixed_verified", "split", "exact")
            assert d["config"][f"auto_layouts_per_s_{point}"] == d["weight_sensitivity"][point]["value"]
            assert d["weight_sensitivity"][point]["fast_engine_err_rel_measured_at_load"] > d["weight_sensitivity"][point]["tolerance"] \
                or d["config"][f"auto_selected_{point}"] == "fast_verified"
        assert set(d["batch_shapes"]) >= {"rule", "300", "488", "640"} and d["layout_metrics"]["value"] > 0
