"""Problem-parallel sharding across the GPUs of one node.

Planning problems never exchange state, so rank r simply owns the contiguous global problem ids
[r * per_gpu, (r + 1) * per_gpu) (the id is also the problem's ChaCha stream id, so a problem's
result does not depend on how the batch is sharded).  The only collective is one all-gather of
a few counters per rank for the throughput report (RCCL over xGMI when the backend is nccl).
"""
import numpy as np


def problem_range(rank, per_gpu):
    """global problem ids owned by `rank`"""
    return rank * per_gpu, (rank + 1) * per_gpu


def gather_stats(stats, device=None):
    """all-gather a 1-D list of float64 counters; returns array [world, len(stats)].
    Works without an initialised process group (world = 1)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(stats), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()
    return t.cpu().numpy()[None, :]


def aggregate(all_stats, i_time=0, i_units=1):
    """whole-job throughput: units of every rank / the slowest rank's time"""
    all_stats = np.asarray(all_stats, dtype=np.float64)
    t_max = float(all_stats[:, i_time].max())
    total = float(all_stats[:, i_units].sum())
    return dict(value=total / t_max, t_max=t_max, total_units=total)
