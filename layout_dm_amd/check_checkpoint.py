"""`python -m layout_dm_amd.check_checkpoint job_dir=<dir> [max_batch_size=512] [precision=auto]`

What engine will a checkpoint get, and why — without sampling anything.  Loads `<job_dir>/config.yaml` +
`best_model.pt` (or the multi-seed layout `<job_dir>/0/ ...`) exactly as the entry point does (layout_dm_amd/test_entry.py, the
reference's trainer/test.py:64-89), builds the drop-in `LayoutDM` with `precision="auto"` (its default), lets `load_state_dict`
measure the fp16 engine (and, where that one is outside the tolerance, the hybrid and then the mixed engine) against the reference-precision
engine on the checkpoint, and prints ONE JSON object per checkpoint:

    {"checkpoint": ".../best_model.pt", "engine_selected": "hybrid_verified", "fast_logits_err_rel": 2.6e-3, "hybrid_logits_err_rel": 7.3e-4,
     "tolerance": 1e-3, "verifier": "split", "verifier_check": {...}, "expected_throughput": "~2 200 layouts/s ...", "library": {...}}

The same record goes to the `layout_dm_amd` logger at INFO when a job loads the checkpoint.  Needs the MI355X (the measurement
IS a handful of denoiser passes on it); a few seconds.
"""
from __future__ import annotations

import json
import logging
import os
import sys
from typing import Any, Dict, List, Optional


def check(job_dir: str, max_batch: int = 512, precision: str = "auto", dataset_dir: Optional[str] = None) -> List[Dict[str, Any]]:
    import torch

    from .layoutdm import LayoutDM
    from .test_entry import GeometryTokenizer, _find_ckpt_dirs

    train_cfg, ckpt_dirs = _find_ckpt_dirs(job_dir)
    model_cfg = dict(train_cfg.model)
    target = str(model_cfg.pop("_target_"))
    model_cfg.pop("_partial_", None)
    if target.rsplit(".", 1)[-1] != "LayoutDM":
        raise NotImplementedError(f"model {target}: only LayoutDM is accelerated")
    data_cfg = train_cfg.data
    data_cfg["pad_until_max"] = True
    clustering_dir = os.path.join(dataset_dir, "..", "clustering_weights") if dataset_dir else None
    tokenizer = GeometryTokenizer(data_cfg, train_cfg.dataset, clustering_dir)
    model = LayoutDM(backbone_cfg=train_cfg.backbone, tokenizer=tokenizer, max_batch=max(1, int(max_batch)),
                     precision=precision, **model_cfg)
    out = []
    for d in ckpt_dirs:
        path = os.path.join(d, "best_model.pt")
        model.load_state_dict(torch.load(path, map_location="cpu"))
        rep = dict(model.model.module.selection_report)
        rep["checkpoint"] = path
        try:
            rep["library"] = model.model.module.engine.describe()
        except Exception:  # (introspection only)
            pass
        out.append(rep)
    return out


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    kv = {}
    for a in argv:
        if "=" not in a:
            raise SystemExit(f"expected key=value, got '{a}' (keys: job_dir, max_batch_size, precision, dataset_dir)")
        k, v = a.split("=", 1)
        kv[k.lstrip("+")] = v
    unknown = set(kv) - {"job_dir", "max_batch_size", "precision", "dataset_dir"}
    if unknown or "job_dir" not in kv:
        raise SystemExit("usage: python -m layout_dm_amd.check_checkpoint job_dir=<dir> [max_batch_size=512] [precision=auto] "
                         "[dataset_dir=<dir>]" + (f"   (unknown: {sorted(unknown)})" if unknown else ""))
    logging.basicConfig(level=logging.INFO, stream=sys.stderr)
    for rep in check(kv["job_dir"], int(kv.get("max_batch_size", 512)), kv.get("precision", "auto"), kv.get("dataset_dir")):
        print(json.dumps(rep, default=str))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
