"""Multi-GPU host logic (one process per GPU, torch.distributed; `nccl` = RCCL on the GPU box,
`gloo` in CPU tests).  Two ways to use N GPUs, neither needs a data-path collective:

  * independent eigenproblems are sharded over ranks (`shard_problems`): bench.py's default;
  * ONE eigenproblem on a process grid in replicated-input mode (ek_hip_solve_replicated):
    every rank holds the replicated matrices (main.f90:84-86), computes the reduction and
    the tridiagonal eigenproblem redundantly and back-transforms only the eigenvector
    columns its grid cell owns (`grid_cell`, `owned_eigenvector_columns`); a 1 x P grid
    shards PDORMTR / PDTRTRS P ways and leaves Z in the reference's block-cyclic layout.

Only the timing is reduced (max over ranks).  Distributing the reduction itself over the
reference's 2-D grid (processes.f90:17-36, distribute_matrix.f90:92-148) with RCCL row/column
communicators is the remaining part of SURVEY.md 8(e).
"""
from . import descriptor as _d


def grid_cell(rank, world, columns_only=False):
    """(nprow, npcol, myrow, mycol) of a rank: the reference's layout_procs grid, or the
    1 x P grid that shards the eigenvector columns P ways."""
    if columns_only:
        return _d.make_process_grid(rank, world, 1, world)
    return _d.make_process_grid(rank, world)


def owned_eigenvector_columns(n_vec, nb, mycol, npcol):
    """Global (0-based) eigenvector columns < n_vec that process column `mycol` back-transforms."""
    return _d.local_indices(n_vec, nb, mycol, npcol)


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
