"""ScaLAPACK-style array descriptors and process-grid rules of the reference host.

Mirrors (behaviour, not code):
  src/descriptor_parameters.f90:2-4   field indices of the 9-int descriptor
  src/processes.f90:56-65             layout_procs: near-square grid factorisation
  src/distribute_matrix.f90:92-148    setup_distributed_matrix incl. the block-shrink rule
  src/global_variables.f90:5          g_block_size = 64
"""
import math

import numpy as np

DESC_SIZE = 9
# 0-based positions of the reference's 1-based named indices (descriptor_parameters.f90:2-4)
DTYPE_, CONTEXT_, ROWS_, COLS_, BLOCK_ROW_, BLOCK_COL_, RSRC_, CSRC_, LOCAL_ROWS_ = range(9)

g_block_size = 64


def numroc(n, nb, iproc, isrcproc, nprocs):
    """Number of rows/cols of a block-cyclically distributed dimension owned by iproc."""
    mydist = (nprocs + iproc - isrcproc) % nprocs
    nblocks = n // nb
    num = (nblocks // nprocs) * nb
    extra = nblocks % nprocs
    if mydist < extra:
        num += nb
    elif mydist == extra:
        num += n % nb
    return num


def layout_procs(n_procs):
    """processes.f90:56-65: P_r = floor(sqrt(P+1)) decremented until it divides P."""
    pr = int(math.sqrt(float(n_procs + 1)))
    while n_procs % pr != 0:
        pr -= 1
    return pr, n_procs // pr


def descinit(m, n, mb, nb, irsrc, icsrc, ctxt, lld):
    return np.array([1, ctxt, m, n, mb, nb, irsrc, icsrc, lld], dtype=np.int32)


def setup_distributed_matrix(rows, cols, nprow=1, npcol=1, myrow=0, mycol=0, block_size=None,
                             ctxt=0):
    """distribute_matrix.f90:92-148: returns (desc, zero-filled local array, column-major).

    The block size is shrunk to max(min(rows/P_r, cols/P_c), 1) when the requested one
    would leave a process without entries (:114-120).
    """
    nb = g_block_size if block_size is None else block_size
    max_nb = max(min(rows // nprow, cols // npcol), 1)
    if nb > max_nb:
        nb = max_nb
    local_rows = max(1, numroc(rows, nb, myrow, 0, nprow))
    local_cols = max(1, numroc(cols, nb, mycol, 0, npcol))
    desc = descinit(rows, cols, nb, nb, 0, 0, ctxt, local_rows)
    mat = np.zeros((local_rows, local_cols), dtype=np.float64, order="F")
    return desc, mat


def local_to_global(l, nb, iproc, nprocs):
    """0-based INDXL2G with source process 0: global index of local index l on process iproc."""
    l = np.asarray(l)
    return ((l // nb) * nprocs + iproc) * nb + l % nb


def local_indices(n, nb, iproc, nprocs):
    """Global indices (ascending) owned by process iproc along one block-cyclic dimension."""
    return local_to_global(np.arange(numroc(n, nb, iproc, 0, nprocs)), nb, iproc, nprocs)


def make_process_grid(rank, n_procs, nprow=None, npcol=None):
    """processes.f90:17-36: row-major rank -> (myrow, mycol) on the layout_procs grid (or a given one)."""
    if nprow is None or npcol is None:
        nprow, npcol = layout_procs(n_procs)
    if nprow * npcol != n_procs:
        raise ValueError("grid %dx%d does not hold %d processes" % (nprow, npcol, n_procs))
    return nprow, npcol, rank // npcol, rank % npcol


def assemble_global(pieces, n_rows, n_cols, nb, nprow, npcol):
    """Global matrix from the local block-cyclic pieces {(myrow, mycol): array} (test helper)."""
    G = np.zeros((n_rows, n_cols), order="F")
    for (pr, pc), loc in pieces.items():
        ri = local_indices(n_rows, nb, pr, nprow)
        ci = local_indices(n_cols, nb, pc, npcol)
        if len(ri) and len(ci):
            G[np.ix_(ri, ci)] = loc[:len(ri), :len(ci)]
    return G
