"""TEST INFRASTRUCTURE ONLY — sample sets of the REAL reference for the "FID vs reference" acceptance check of BASELINE
config 5: the reference's own `sample()` (categorical_diffusion/base.py:293-371, CPU, sampling=random,
torch.multinomial) on the trained-like "mid" synthetic checkpoint (oracle/synth.py), two independent seeds.

    python -m oracle.make_reference_samples            # ~20 min on 8 vCPU; writes tests/golden/rico25_mid_reference_samples.npz
    python -m oracle.make_reference_samples --chunks 1  # what tests/test_oracle_vs_reference.py re-generates (chunk 0 of each seed)

Layouts are drawn in chunks of 64 with torch.manual_seed(seed * 1000 + chunk) so that any chunk can be re-generated
alone and compared bit for bit with the committed file.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import ref_harness as rh
from . import spec as SP
from . import synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "rico25_mid_reference_samples.npz")
SEEDS, CHUNK, N_CHUNK, POINT, WEIGHT_SEED = (101, 202), 64, 16, "mid", 2


def generate(n_chunk: int = N_CHUNK):
    spec = SP.SPECS["rico25"]
    m, _ = rh.build_reference_model("rico25", seed=0)
    ssd = synth.trained_like_state_dict(spec, POINT, seed=WEIGHT_SEED, prefix="")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ssd.items()})
    cfg = rh.sampling_cfg("random")
    out = np.zeros((len(SEEDS), n_chunk * CHUNK, spec.seq_len), np.int16)
    for si, seed in enumerate(SEEDS):
        for c in range(n_chunk):
            torch.manual_seed(seed * 1000 + c)
            ids = m.sample(batch_size=CHUNK, cond=None, sampling_cfg=cfg)
            out[si, c * CHUNK:(c + 1) * CHUNK] = ids.numpy().astype(np.int16)
            print(f"seed {seed} chunk {c + 1}/{n_chunk}", flush=True)
    return {"tokens": out, "seeds": np.array(SEEDS), "chunk": np.int32(CHUNK), "point": np.array(POINT),
            "weight_seed": np.int32(WEIGHT_SEED)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=N_CHUNK)
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 2))
    np.savez_compressed(a.out, **generate(a.chunks))
