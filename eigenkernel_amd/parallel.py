"""Multi-GPU host logic of this round: independent eigenproblems are sharded over ranks
(one process per GPU, torch.distributed; `nccl` = RCCL on the GPU box, `gloo` in CPU tests)
with no data-path collective; only the timing is reduced (max over ranks).

The reference's own decomposition is the 2-D block-cyclic grid of processes.f90:17-36 /
distribute_matrix.f90:92-148; mapping that grid onto the 8 GPUs of a node with RCCL
row/column communicators is the next row of SURVEY.md 8(e).
"""


def shard_problems(n_problems, rank, world):
    """Round-robin ownership of independent problems."""
    return list(range(rank, n_problems, world))


def aggregate_throughput(units_local, seconds_local, dist=None, device=None):
    """Whole-job throughput = all ranks' units / max-over-ranks seconds."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_local / seconds_local, seconds_local
    import torch
    t = torch.tensor([seconds_local], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item()), float(t.item())
