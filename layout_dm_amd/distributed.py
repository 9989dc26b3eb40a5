"""Multi-GPU plumbing of the sampling path: one process per GPU, layouts sharded by global index,
ONE collective (all_gather of the final int32 tokens) at the end.  backend "nccl" is RCCL over xGMI
on MI355X; the same code runs on gloo for the CPU tests.

The reference has no distributed path at all (nn.DataParallel is bypassed by sample():
models/common/nn_lib.py:17-23, README.md:49); layouts are independent (no cross-batch op in
categorical_diffusion/base.py:293-371) so the path shards with no data-path collective.
Results do not depend on the shard count because the sampler's Philox stream is keyed by the
GLOBAL layout index (kernels_post.hip)."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) range of global layout indices owned by `rank`;
    the first (total % world) ranks get one extra layout."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sample_sharded(sample_fn: Callable[[int, int], torch.Tensor], total: int, group=None,
                   device: Optional[torch.device] = None) -> torch.Tensor:
    """Runs sample_fn(first_layout, count) -> (count, S) int32 tokens on this rank's shard and
    all_gathers the shards in global-index order.  With world_size 1 no collective is issued."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sample_fn(0, total)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    start, stop = shard_range(total, rank, world)
    local = sample_fn(start, stop - start).to(torch.int32).contiguous()
    S = local.shape[1]
    if total % world == 0:
        out = torch.empty((total, S), dtype=torch.int32, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # ragged shards: pad to the largest shard, gather, trim
    mx = -(-total // world)
    pad = torch.zeros((mx, S), dtype=torch.int32, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * mx, S), dtype=torch.int32, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(total, r, world)
        parts.append(buf[r * mx: r * mx + (b - a)])
    return torch.cat(parts)
