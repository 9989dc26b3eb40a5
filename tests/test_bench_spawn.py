"""CPU: `python bench.py --gpus N` launches ITSELF.  `--dry-run` takes the whole N-rank path —
re-execution under torch.distributed.run on 127.0.0.1, one process per rank, shard by global layout index,
distributed.sample_sharded's single all_gather, barrier + max-over-ranks timing, per-rank statistics, ONE JSON line
from rank 0 — on the gloo backend with a fake sampler that is a function of the GLOBAL layout index, exactly what the
Philox-keyed HIP sampler guarantees on the GPU.  The digest of the gathered tokens must not depend on N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"bench.py must print ONE line, got {len(lines)}: {p.stdout[-500:]}"
    return json.loads(lines[0])


def test_gpus_2_spawns_its_own_ranks_and_reports_once():
    one = _run("--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "1")
    two = _run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert one["n_gpus"] == 1 and one["world_size_seen"] == 1
    assert two["n_gpus"] == 2 and two["world_size_seen"] == 2 and two["dry_run"] is True
    assert two["per_rank_layouts_per_s"]["ranks"] == 2 and two["per_rank_layouts_per_s"]["min"] > 0
    # the same per-GPU workload at every N > 1 (config 4: 1024 layouts per GPU); N = 1 keeps config 2 as the headline
    assert "batch=1024/GPU" in two["config"]["workload"] and "batch=512/GPU" in one["config"]["workload"]
    assert two["tokens_sha256"]["sha256"] == one["tokens_sha256"]["sha256"]
    assert two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "weak" and two["value"] > 0


def test_strong_scaling_ragged_shards():
    d = _run("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--total", "37")
    assert d["scaling"] == "strong" and d["n_gpus"] == 2
