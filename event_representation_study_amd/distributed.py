"""Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm, "gloo" on CPU).

Builders shard naturally -- a window's output depends only on its own events -- so windows are
dealt to ranks and there is NO data-path collective.  The GWD score list is the one exchange:
every (representation, sample, quadrant) solve is independent, each rank computes its share, and a
single all_gather of float64 scalars assembles the vector the reference averages
(compute_otmi.py:211, gen1_compute.py:141).  The message is O(100 B): latency-bound, xGMI
bandwidth irrelevant."""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n_items, rank=None, world=None):
    """Indices of the items rank `rank` owns: round-robin (item i -> rank i mod world)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return list(range(rank, n_items, world))


def gather_scalars(local_values, n_items, device=None):
    """local_values: {item index: float} computed by this rank -> full float64 vector on every rank
    (one all_gather of a dense vector with zeros at foreign slots, then a sum)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (
            dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")
    vec = torch.zeros(n_items, dtype=torch.float64, device=device)
    for i, v in local_values.items():
        vec[i] = float(v)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return vec
    parts = [torch.zeros_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, vec)
    return torch.stack(parts).sum(0)


def gather_vector(local_vec, indices, n_items):
    """local_vec: DEVICE (or CPU) float64 tensor of this rank's scores for the items `indices` -> the full vector on every
    rank: ONE all_gather, no host round trip per score."""
    vec = torch.zeros(n_items, dtype=torch.float64, device=local_vec.device)
    if len(indices):
        vec[torch.as_tensor(list(indices), dtype=torch.int64, device=local_vec.device)] = local_vec.to(torch.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return vec
    parts = [torch.zeros_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, vec)
    return torch.stack(parts).sum(0)


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if (
        dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")


def gather_rows(local_values, device=None):
    """A short list of float64 figures per rank (timings, counts) -> (world, k) tensor on the CPU, the same on every rank:
    ONE all_gather.  The control-plane exchange of the per-rank legs (precompute, sweeps): bytes, not data."""
    if device is None:
        device = _default_device()
    row = torch.as_tensor([float(v) for v in local_values], dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return row[None].cpu()
    parts = [torch.zeros_like(row) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, row)
    return torch.stack(parts).cpu()


def sharded_scores(score_fn, items):
    """Evaluate score_fn(item) for this rank's share of `items` and all-gather the scalars."""
    mine = shard_indices(len(items))
    local = {i: score_fn(items[i]) for i in mine}
    return gather_scalars(local, len(items))


def mean_cp(scores, per_sample=3):
    """C_p as the reference aggregates it: mean over quadrants per sample, then over samples."""
    s = scores.reshape(-1, per_sample).mean(dim=1)
    return float(s.mean().item())
