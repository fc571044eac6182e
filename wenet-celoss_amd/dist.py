"""Data-parallel helpers.

The loss path shards by utterance with no data-path collective: every rank (one
process per GPU, `torch.distributed` over RCCL/xGMI) owns a slice of the batch,
runs the joiner / RNN-T / CTC kernels on it, and only scalars cross ranks.  The
gradient all-reduce of a training step stays DistributedDataParallel's -- the
reference's wenet/bin/train.py:227-240 already wraps the model in DDP and is
unchanged.  The reference averages per-rank *means* even when ranks hold
different batch sizes (dynamic batching, SURVEY.md section 7 "DDP equivalence");
`global_mean_of_rank_means` reproduces exactly that.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `n_items` utterances for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shards(costs: Sequence[int], world_size: int) -> List[List[int]]:
    """Deal utterances to ranks so that the summed lattice sizes T_b*(U_b+1) are balanced:
    sort descending, give each to the currently lightest rank.  Returns index lists per rank."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(x) for x in out]


def global_mean_of_rank_means(local_mean: torch.Tensor) -> torch.Tensor:
    """What DDP's gradient averaging implies for the loss value: mean over ranks of each rank's batch mean."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_mean
    t = local_mean.detach().clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / dist.get_world_size()


def global_utterance_mean(local_costs: torch.Tensor) -> torch.Tensor:
    """True mean over all utterances of all ranks (sum of costs / number of utterances)."""
    s = torch.stack([local_costs.detach().sum(), torch.tensor(float(local_costs.numel()), device=local_costs.device,
                                                              dtype=local_costs.dtype)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s[0] / s[1]
