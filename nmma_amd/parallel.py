"""Multi-GPU evaluation: the live-point batch is sharded row-wise across ranks (one
process per GPU), every rank evaluates its shard with no data-path collective, and the
shards' logL are exchanged with ONE all-gather per batch (RCCL over xGMI when the
backend is ``nccl``; ``gloo`` on CPU hosts for tests).

The reference's analogue is the schwimmbad MPIPool task farm
(``nmma/core/mpi_setup.py:651-667``): N ranks x 1 point becomes G ranks x B/G points.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced [lo, hi) of ``n`` rows for ``rank`` (first n % world ranks get +1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedEvaluator:
    """``evaluate(theta_full)`` -> logL for ALL rows on every rank.

    ``local_fn(theta_shard) -> logL_shard`` is the per-rank evaluation (e.g.
    ``likelihood.log_likelihood_batch``); tensors stay on the rank's device.
    """

    def __init__(self, local_fn, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.local_fn = local_fn
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def evaluate(self, theta):
        import torch
        n = theta.shape[0]
        lo, hi = shard_bounds(n, self.world, self.rank)
        local = self.local_fn(theta[lo:hi])
        if not isinstance(local, torch.Tensor):
            local = torch.as_tensor(np.asarray(local))
        if self.world == 1:
            return local
        # equal-sized slots so a single all_gather_into_tensor suffices; pad the short shards
        slot = (n + self.world - 1) // self.world
        buf = torch.full((slot,), float("nan"), dtype=local.dtype, device=local.device)
        buf[: hi - lo] = local
        out = torch.empty(self.world * slot, dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)
        pieces = []
        for r in range(self.world):
            a, b = shard_bounds(n, self.world, r)
            pieces.append(out[r * slot: r * slot + (b - a)])
        return torch.cat(pieces)
