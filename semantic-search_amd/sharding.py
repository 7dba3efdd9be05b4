"""Environment sharding across the GPUs of one node and the final metric reduce.

The reference runs one environment per worker process and assigns GPUs round-robin
(`gpu = env_gpus[rank % len(env_gpus)]`, eval_objnav.py:100); environments never exchange
data, so the fusion path shards by environment with no collective in the inner loop.  The
only exchange is the end-of-run aggregation of per-episode metrics, which the reference does
with numpy means on one host (eval_objnav.py:333-345); here it is one all-reduce(sum) of a
small float64 vector (RCCL on GPUs, gloo on CPU).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

METRIC_FIELDS = ("n_episodes", "success", "spl", "length", "fp_found", "fn_found",
                 "fp_detections", "fn_detections")
N_GOALS = 6


def shard_envs(n_envs_total: int, rank: int, world: int) -> List[int]:
    """Global env indices owned by `rank`: contiguous blocks, sizes differing by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_envs_total, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def pack_metrics(episodes: Sequence[Dict]) -> torch.Tensor:
    """Sums of the per-episode fields plus per-goal episode and success counts."""
    vec = torch.zeros(len(METRIC_FIELDS) + 2 * N_GOALS, dtype=torch.float64)
    for ep in episodes:
        vec[0] += 1.0
        for k, name in enumerate(METRIC_FIELDS[1:], start=1):
            vec[k] += float(ep.get(name, 0.0))
        g = int(ep.get("goal", 0))
        vec[len(METRIC_FIELDS) + g] += 1.0
        vec[len(METRIC_FIELDS) + N_GOALS + g] += float(ep.get("success", 0.0))
    return vec


def reduce_metrics(vec: torch.Tensor, device=None) -> Dict[str, float]:
    """all_reduce(sum) over the default process group (no-op without one), then the means the
    reference reports (eval_objnav.py:333-345, 353-354)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        buf = vec.to(device) if device is not None else vec.clone()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        vec = buf.cpu()
    n = max(float(vec[0]), 1.0)
    out = {"n_episodes": float(vec[0])}
    for k, name in enumerate(METRIC_FIELDS[1:], start=1):
        out[name] = float(vec[k]) / n
    m = len(METRIC_FIELDS)
    out["per_goal_success"] = [
        (float(vec[m + N_GOALS + g]) / float(vec[m + g])) if vec[m + g] > 0 else float("nan")
        for g in range(N_GOALS)]
    return out


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
