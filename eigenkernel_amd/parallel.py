"""Multi-GPU host logic of bench.py (one process per GPU, torch.distributed; `nccl` = RCCL on the GPU
box, `gloo` in the CPU tests): which independent problems a rank owns in the replicas mode, and the
whole-job throughput from all ranks' units and the slowest rank's time.  The process-grid helpers live in
descriptor.py (`make_process_grid`, `local_indices`); the data path of a grid solve is inside the library
(DESIGN.md section 6)."""


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
