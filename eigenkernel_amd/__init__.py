"""eigenkernel_amd -- MI355X-native drop-in for EigenKernel's `scalapack` /
`general_scalapack` / `*_select` solver path (reference: src/solver_main.f90:52-99).

The numerics live in csrc/ (hand-written HIP for gfx950) behind the C-ABI declared in
include/ek_hip.h.  This Python package is only the host-side mirror of the reference's
operator interface (descriptors, MatrixMarket I/O, verifier, solver dispatch) used by the
tests and bench; there is no CPU fallback: every solver entry raises if libek_hip.so is
missing.
"""
from .descriptor import (DESC_SIZE, descinit, layout_procs, numroc, setup_distributed_matrix,
                         g_block_size)
from .matrix_io import (read_matrix_file, write_matrix_file, write_eigenvalues, write_eigenvectors,
                        parse_printed_vecs_ranges, SparseMat)
from .solver import (eigen_solver, EigenpairsBlacs, Process, load_library, LibraryMissing,
                     SOLVERS)

__all__ = [
    "DESC_SIZE", "descinit", "layout_procs", "numroc", "setup_distributed_matrix",
    "g_block_size", "read_matrix_file", "write_matrix_file", "write_eigenvalues", "write_eigenvectors",
    "parse_printed_vecs_ranges", "SparseMat",
    "eigen_solver", "EigenpairsBlacs", "Process", "load_library", "LibraryMissing", "SOLVERS",
]
