"""`python -m layout_dm_amd.test_entry cond=... job_dir=... result_dir=... [key=value ...]`

Drop-in for the reference's hydra entry point `python -m src.trainer.trainer.test` (trainer/test.py:57-283) with
`model.sample` running in libldm_hip.so.  Same CLI keys (hydra_configs.py:12-50 TestConfig), same job_dir layout
(`config.yaml` + `best_model.pt`, multi-seed `0/ 1/ ...`), same cond= plumbing, same result pickles
(`<result_dir>/<cond>_<key>/seed_<n>.pkl`, consumed by eval.py).

Two ways to run:
  * the reference package and its dependencies (hydra, omegaconf, torch_geometric, ...) are importable: the
    reference's OWN `main` runs, with `trainer.models.layoutdm.LayoutDM` swapped for the MI355X class — every cond type;
  * they are not (this image): a built-in runner with a tiny `key=value` parser (SURVEY §5 "config/flag system") and a
    plain-YAML reader covers what needs no dataset: `cond=unconditional`.  Conditional tasks need the reference's
    datasets / `get_cond` (torch_geometric) and say so.
"""
from __future__ import annotations

import os
import pickle
import random
import sys
import time
from typing import Any, Dict, List, Optional

# hydra_configs.py:12-50 (TestConfig) — same names, same defaults
TEST_DEFAULTS: Dict[str, Any] = dict(
    job_dir=None, result_dir=None, dataset_dir=None, max_batch_size=512, num_run=1, cond="unconditional",
    num_timesteps=100, is_validation=False, debug=False, debug_num_samples=-1, sampling="random", temperature=1.0,
    top_p=0.9, top_k=5.0, num_uncond_samples=1000, time_difference=0.0, refine_lambda=3.0, refine_mode="uniform",
    refine_offset_ratio=0.1, relation_lambda=3e6, relation_mode="average", relation_tau=1.0, relation_num_update=3,
    use_ddim=False)
# helpers/sampling.py:13-59 (SAMPLING_CONFIG_DICT; "top_k" resolves to the top_k_top_p dataclass there)
SAMPLING_DEFAULTS = {
    "deterministic": dict(name="deterministic"),
    "random": dict(temperature=1.0, name="random"),
    "gumbel": dict(temperature=1.0, name="gumbel"),
    "top_p": dict(temperature=1.0, name="top_p", top_p=0.9),
    "top_k": dict(temperature=1.0, name="top_k_top_p", top_k=5, top_p=0.9),
}
N_CATEGORY = {"Rico25Dataset": 25, "Rico13Dataset": 13, "Rico5Dataset": 5, "PubLayNetDataset": 5}


class AttrDict(dict):
    """DictConfig stand-in: attribute + item access, `in`, `.get`, `.items()`."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_attr(v) for v in obj]
    return obj


def _coerce(text: str, like):
    if isinstance(like, bool):
        if text.lower() in ("true", "1", "yes"):
            return True
        if text.lower() in ("false", "0", "no"):
            return False
        raise ValueError(f"not a boolean: {text}")
    if isinstance(like, int) and not isinstance(like, bool):
        return int(text)
    if isinstance(like, float):
        return float(text)
    return None if text in ("null", "None") else text


def parse_cli(argv: List[str]) -> AttrDict:
    """hydra-style `key=value` overrides of TestConfig; unknown keys are an error (as in a structured config)."""
    cfg = AttrDict(TEST_DEFAULTS)
    for arg in argv:
        if "=" not in arg:
            raise SystemExit(f"expected key=value, got '{arg}'")
        k, v = arg.split("=", 1)
        k = k.lstrip("+")
        if k not in TEST_DEFAULTS:
            raise SystemExit(f"unknown key '{k}' (TestConfig keys: {', '.join(TEST_DEFAULTS)})")
        cfg[k] = _coerce(v, TEST_DEFAULTS[k]) if TEST_DEFAULTS[k] is not None else (None if v in ("null", "None") else v)
    for k in ("job_dir", "result_dir"):
        if not cfg[k]:
            raise SystemExit(f"missing mandatory value: {k}=...")
    return cfg


class GeometryTokenizer:
    """What `LayoutDM` and the device-side decode read from the reference's LayoutSequenceTokenizer
    (helpers/layout_tokenizer.py:123-186) for the LayoutDM data configuration — c-x-y-w-h, stacked x-y-w-h bbox
    vocabulary, [pad, mask] — built from config.yaml alone.  encode() needs real layouts and therefore the reference."""

    def __init__(self, data_cfg, dataset_cfg, clustering_dir: Optional[str] = None):
        target = str(dataset_cfg["_target_"]).rsplit(".", 1)[-1]
        if target not in N_CATEGORY:
            raise NotImplementedError(f"dataset {target}")
        if data_cfg.get("var_order", "c-x-y-w-h") != "c-x-y-w-h" or data_cfg.get("shared_bbox_vocab") != "x-y-w-h":
            raise NotImplementedError("built-in runner: var_order c-x-y-w-h with shared_bbox_vocab x-y-w-h (LayoutDM)")
        if list(data_cfg.get("special_tokens", [])) != ["pad", "mask"]:
            raise NotImplementedError("built-in runner: special_tokens [pad, mask] (LayoutDM)")
        quant = data_cfg.get("bbox_quantization", "linear")
        self.N_category = N_CATEGORY[target]
        self.N_bbox_per_var = int(data_cfg.get("num_bin_bboxes", 32))
        self.max_seq_length = int(dataset_cfg["max_seq_length"])
        self.N_var_per_element = 5
        self.var_names = ["c", "x", "y", "w", "h"]
        self.special_tokens = ["pad", "mask"]
        self.N_total = self.N_category + 4 * self.N_bbox_per_var + 2
        self.max_token_length = self.max_seq_length * 5
        self.bbox_tokenizer = AttrDict(shared_bbox_vocab="x-y-w-h", bbox_quantization=quant,
                                       var_names=["x", "y", "w", "h"], _var_order=["x", "y", "w", "h"])
        if quant in ("kmeans", "percentile"):
            # bbox_tokenizer.py:51-68: <clustering_weights>/<dataset>_max<N>_<quant>_train_clusters.pkl, a dict of fitted
            # sklearn models keyed "<var>-<bins>"; 1-D centres are sorted.  Only .cluster_centers_ is needed here.
            name = {"Rico25Dataset": "rico25", "Rico13Dataset": "rico13", "Rico5Dataset": "rico5",
                    "PubLayNetDataset": "publaynet"}[target]
            fname = f"{name}_max{self.max_seq_length}_{quant}_train_clusters.pkl"
            path = _find_clustering_file(fname, clustering_dir)
            import numpy as np

            models = {}
            with open(path, "rb") as f:
                for key, model in pickle.load(f).items():
                    if key in [f"{k}-{self.N_bbox_per_var}" for k in "xywh"]:
                        centres = np.sort(np.asarray(model.cluster_centers_, dtype=np.float64), axis=0)
                        models[key] = AttrDict(cluster_centers_=centres)
            if len(models) != 4:
                raise FileNotFoundError(f"{path}: no {self.N_bbox_per_var}-bin models for x, y, w, h")
            self.bbox_tokenizer["clustering_models"] = models
        elif quant != "linear":
            raise NotImplementedError(f"bbox_quantization={quant}")

    def name_to_id(self, name):
        return {"pad": self.N_total - 2, "mask": self.N_total - 1}[name]

    def id_to_name(self, i):
        return {self.N_total - 2: "pad", self.N_total - 1: "mask"}[i]

    def decode(self, ids):
        raise RuntimeError("the device-side decode handles this tokenizer configuration")


def _find_clustering_file(fname: str, clustering_dir: Optional[str]) -> str:
    """KMEANS_WEIGHT_ROOT of the reference is <repo>/download/clustering_weights (global_configs.py:3-4); here:
    $LDM_CLUSTERING_WEIGHTS, an explicit directory, or ./download/clustering_weights."""
    cands = [clustering_dir, os.environ.get("LDM_CLUSTERING_WEIGHTS"), os.path.join("download", "clustering_weights")]
    for d in cands:
        if d and os.path.exists(os.path.join(d, fname)):
            return os.path.join(d, fname)
    raise FileNotFoundError(f"{fname} not found in {[c for c in cands if c]} (set LDM_CLUSTERING_WEIGHTS)")


def _find_ckpt_dirs(job_dir: str):
    """test.py:64-89: single job (config.yaml in job_dir) or multi-seed (job_dir/0, job_dir/1, ...)."""
    import yaml

    cfg_path = os.path.join(job_dir, "config.yaml")
    if os.path.exists(cfg_path):
        return to_attr(yaml.safe_load(open(cfg_path))), [job_dir]
    dirs, train_cfg, seed = [], None, 0
    while os.path.exists(os.path.join(job_dir, str(seed), "config.yaml")):
        if seed == 0:
            train_cfg = to_attr(yaml.safe_load(open(os.path.join(job_dir, "0", "config.yaml"))))
        dirs.append(os.path.join(job_dir, str(seed)))
        seed += 1
    if not dirs:
        raise FileNotFoundError(cfg_path)
    return train_cfg, dirs


def _filter_invalid(layouts):  # test.py:42-49
    out = []
    for b in range(layouts["bbox"].size(0)):
        m = layouts["mask"][b].numpy()
        out.append((layouts["bbox"][b].numpy()[m], layouts["label"][b].numpy()[m]))
    return out


def run_builtin(test_cfg: AttrDict) -> Dict[str, Any]:
    """trainer/test.py:57-283 for cond=unconditional without hydra / the reference package."""
    import numpy as np
    import torch

    from .layoutdm import LayoutDM

    if not os.path.isdir(test_cfg.job_dir):
        raise FileNotFoundError(test_cfg.job_dir)
    if test_cfg.cond != "unconditional":
        raise SystemExit(
            f"cond={test_cfg.cond} needs the reference's datasets and get_cond (trainer/helpers/task.py, torch_geometric): "
            "install the layout-dm package (poetry install) — this entry point then drives its own main(); the built-in "
            "runner covers cond=unconditional")
    train_cfg, ckpt_dirs = _find_ckpt_dirs(test_cfg.job_dir)
    if test_cfg.debug:
        ckpt_dirs = ckpt_dirs[:1]
    if test_cfg.sampling not in SAMPLING_DEFAULTS:
        raise SystemExit(f"sampling={test_cfg.sampling}: one of {sorted(SAMPLING_DEFAULTS)}")
    sampling_cfg = AttrDict(SAMPLING_DEFAULTS[test_cfg.sampling])
    if "temperature" in test_cfg and "temperature" in sampling_cfg:
        sampling_cfg.temperature = test_cfg.temperature
    if sampling_cfg.name == "top_p":
        sampling_cfg.top_p = test_cfg.top_p
    if sampling_cfg.name == "top_k_top_p":
        raise NotImplementedError("sampling=top_k resolves to top_k_top_p in the reference (sampling.py:52-54), which its "
                                  "own sample() does not implement either (sampling.py:117-118)")
    model_cfg = dict(train_cfg.model)
    target = str(model_cfg.pop("_target_"))
    model_cfg.pop("_partial_", None)
    if target.rsplit(".", 1)[-1] != "LayoutDM":
        raise NotImplementedError(f"model {target}: only LayoutDM is accelerated")
    data_cfg = train_cfg.data
    data_cfg["pad_until_max"] = True
    clustering_dir = os.path.join(test_cfg.dataset_dir, "..", "clustering_weights") if test_cfg.dataset_dir else None
    tokenizer = GeometryTokenizer(data_cfg, train_cfg.dataset, clustering_dir)
    model = LayoutDM(backbone_cfg=train_cfg.backbone, tokenizer=tokenizer,
                     max_batch=max(1, min(int(test_cfg.max_batch_size), 2048)), **model_cfg)
    sampling_cfg = model.aggregate_sampling_settings(sampling_cfg, test_cfg)
    key = "_".join(f"{k}_{v}" for k, v in sampling_cfg.items())  # test.py:120-128
    if test_cfg.is_validation:
        key += "_validation"
    if test_cfg.debug:
        key += "_debug"
    if test_cfg.debug_num_samples > 0:
        key += f"_only_{test_cfg.debug_num_samples}_samples"
    result_dir = os.path.join(test_cfg.result_dir, f"{test_cfg.cond}_{key}")
    os.makedirs(result_dir, exist_ok=True)
    print(f"Results saved to {result_dir}", file=sys.stderr)

    summary = {"result_dir": result_dir, "pickles": [], "ms_per_sample": []}
    for seed_no, ckpt_dir in enumerate(ckpt_dirs):
        random.seed(seed_no)          # set_seed, helpers/util.py:10-13
        np.random.seed(seed_no)
        torch.manual_seed(seed_no)
        sd = torch.load(os.path.join(ckpt_dir, "best_model.pt"), map_location="cpu")   # load_model, common/util.py:47-57
        model.load_state_dict(sd)
        model.eval()
        n, bs = int(test_cfg.num_uncond_samples), int(test_cfg.max_batch_size)
        batches = (n // bs) * [bs] + ([n % bs] if n % bs else [])   # split_num_samples, data/util.py:301-307
        t_total, n_total, results = 0.0, 0, []
        for batch_size in batches:
            t0 = time.time()
            layouts = model.sample(batch_size=batch_size, cond=None, sampling_cfg=sampling_cfg, cond_type=test_cfg.cond)
            t_total += time.time() - t0
            n_total += batch_size
            results.extend(_filter_invalid(layouts))
        dummy_cfg = AttrDict(train_cfg)
        dummy_cfg["sampling"] = sampling_cfg
        data = {"results": results, "train_cfg": _plain(dummy_cfg), "test_cfg": _plain(test_cfg)}
        pkl = os.path.join(result_dir, f"seed_{seed_no}.pkl")
        with open(pkl, "wb") as f:
            pickle.dump(data, f)
        print(n_total)
        print(f"ms per sample: {1e3 * t_total / max(n_total, 1)}")
        summary["pickles"].append(pkl)
        summary["ms_per_sample"].append(1e3 * t_total / max(n_total, 1))
    return summary


def _plain(obj):
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


def main(argv: Optional[List[str]] = None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        import hydra  # noqa: F401
        import trainer.models.layoutdm as ref_layoutdm
        import trainer.test as ref_test
    except Exception:
        return run_builtin(parse_cli(argv))
    from .layoutdm import LayoutDM

    ref_layoutdm.LayoutDM = LayoutDM  # hydra resolves `_target_: trainer.models.layoutdm.LayoutDM` to this
    sys.argv = [sys.argv[0]] + list(argv)
    ref_test.filter_args_for_ai_platform()
    return ref_test.main()


if __name__ == "__main__":
    main()
