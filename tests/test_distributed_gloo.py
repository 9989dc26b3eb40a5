"""CPU, world_size 2 over gloo: the shard + single-all_gather plumbing of the multi-GPU path.
(The HIP sampler itself cannot run without a GPU; its invariance to the shard split — Philox keyed by
global layout index — is asserted on the GPU in test_hip_parity.py::test_full_batch_512_properties
and mirrored here with the oracle's Philox uniforms.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from layout_dm_amd.distributed import sample_sharded, shard_range
from oracle import restatement as R

S = 125


def fake_sampler(first, count):
    """Deterministic function of the GLOBAL layout index, built from the same Philox uniforms the
    kernel consumes."""
    u = R.token_uniforms(seed=9, first_layout=first, B=count, S=S, step=3)[..., 0]
    return torch.from_numpy((u * 150).astype(np.int32))


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = sample_sharded(fake_sampler, total)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37])
def test_two_rank_gather_equals_single_process(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got, fake_sampler(0, total).numpy())


def test_shard_range_partition():
    for total in (1, 7, 512, 8192, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
