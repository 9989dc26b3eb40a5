"""GPU: the RCCL code path of the multi-GPU bench inside the driver-run suite.

No 8-GPU node is available to the builder, so the 1 -> 8 curve is the driver's to measure; what CAN be shown on the one-GPU box is
that everything a rank does at N > 1 — `init_process_group("nccl")` (RCCL), shard by global layout index, the single
`all_gather_into_tensor` of the final tokens (layout_dm_amd/distributed.py), barrier, max-over-ranks all_reduce, one JSON line from
rank 0 — runs on the real backend and returns the same tokens as the plain one-process run: bench.py under
`torch.distributed.run --nproc-per-node 1` with LDM_BENCH_FORCE_DIST=1."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--config", "4", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-traffic", "--modes", "none"]


def _one_line(p):
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, f"bench.py must print ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_under_torchrun_takes_the_rccl_path_with_one_rank():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    plain = _one_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *ARGS], env=env, cwd=ROOT,
                                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env_d = dict(env, LDM_BENCH_FORCE_DIST="1")
    dist = _one_line(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                     "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                                     "--gpus", "1", *ARGS], env=env_d, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                    text=True, timeout=600))
    assert plain["dist_backend"] is None and plain["world_size_seen"] == 1
    assert dist["dist_backend"] == "nccl" and dist["world_size_seen"] == 1 and dist["n_gpus"] == 1
    assert dist["value"] > 0 and dist["steps"] == 2 and "batch=1024/GPU" in dist["config"]["workload"]
    # Philox is keyed by the GLOBAL layout index: the gathered tokens of the sharded run are the plain run's
    assert dist["tokens_sha256"]["sha256"] == plain["tokens_sha256"]["sha256"]
    assert "per_rank_layouts_per_s" in dist and dist["per_rank_layouts_per_s"]["ranks"] == 1
