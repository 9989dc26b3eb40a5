"""TEST INFRASTRUCTURE ONLY — wall-clock of the REAL reference's CPU sampling path in the build container
(SURVEY §8d "CPU baseline timing"): `model.sample(batch_size, cond=None, sampling_cfg)` of the reference's own
ConstrainedMaskAndReplaceDiffusion (imported from /root/reference through oracle/ref_harness.py), timed with
time.time() around the call exactly as trainer/test.py:194-203 does, 1 warm-up + 3 timed runs.

    python -m oracle.time_reference > profiles/r02_reference_cpu_timing.json

The GPU box has no /root/reference, so bench.py's cpu_baseline leg times the oracle restatement there (kind "port");
this file is the committed timing of the reference itself (kind "reference"), on THIS container's host."""
from __future__ import annotations

import json
import os
import platform
import time

import torch

from . import ref_harness as rh
from . import spec as SP
from . import synth


def main():
    ncpu = os.cpu_count() or 1
    out = {"host": {"logical_cpus": ncpu, "machine": platform.machine(), "torch": torch.__version__,
                    "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if "model name" in l), "?")},
           "workload": "Rico25 cond=unconditional T=100 sampling=random, random-init weights (reference init)",
           "runs": []}
    m, _tok = rh.build_reference_model("rico25", seed=0)
    spec = SP.SPECS["rico25"]
    ssd = synth.synth_state_dict(spec, seed=0, perturb=False, prefix="")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
    cfg = rh.sampling_cfg("random")
    for batch, threads in ((4, ncpu), (64, ncpu)):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        reps = 3 if batch <= 4 else 1
        m.sample(batch_size=batch, cond=None, sampling_cfg=cfg) if batch <= 4 else None  # warm-up (small case only)
        times = []
        for _ in range(reps):
            t0 = time.time()
            ids = m.sample(batch_size=batch, cond=None, sampling_cfg=cfg)
            times.append(time.time() - t0)
        assert ids.shape == (batch, spec.seq_len)
        best = min(times)
        out["runs"].append({"batch": batch, "threads": threads, "seconds": [round(t, 2) for t in times],
                            "layouts_per_s": round(batch / best, 3), "kind": "reference"})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
