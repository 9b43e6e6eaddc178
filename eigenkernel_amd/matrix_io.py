"""MatrixMarket coordinate I/O and result files, as the reference host reads/writes them.

Mirrors src/matrix_io.f90:22-144 (read: `coordinate real symmetric`, one triangle, 1-based
`i j value`, `%` comments after the banner) and src/main.f90:113-118 (eigenvalues.dat,
format `(I8, " ", E26.16e3)`).  Host-side only.
"""
from dataclasses import dataclass

import numpy as np

BANNER = "%%MatrixMarket matrix coordinate real symmetric"


@dataclass
class SparseMat:
    """ek_sparse_mat_t (matrix_io.f90:11-15): replicated triplets, 1-based suffix."""
    size: int
    num_non_zeros: int
    suffix: np.ndarray  # (2, nnz) int32, 1-based
    value: np.ndarray   # (nnz,) float64

    def to_dense(self):
        """Mirrored dense matrix, as distribute_global_sparse_matrix does
        (distribute_matrix.f90:411-418)."""
        A = np.zeros((self.size, self.size), dtype=np.float64, order="F")
        i = self.suffix[0] - 1
        j = self.suffix[1] - 1
        A[i, j] = self.value
        A[j, i] = self.value
        return A


def read_matrix_file(path):
    with open(path, "r") as f:
        banner = f.readline().strip()
        if banner.lower() != BANNER.lower():
            raise ValueError("read_matrix_file: unsupported MatrixMarket banner: %r" % banner)
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        rows, cols, nnz = (int(t) for t in line.split())
        if rows != cols:
            raise ValueError("read_matrix_file: matrix must be square")
        data = np.loadtxt(f, dtype=np.float64, ndmin=2, max_rows=nnz)
    if data.shape[0] != nnz:
        raise ValueError("read_matrix_file: expected %d entries, found %d" % (nnz, data.shape[0]))
    suffix = np.ascontiguousarray(data[:, :2].T.astype(np.int32))
    return SparseMat(rows, nnz, suffix, np.ascontiguousarray(data[:, 2]))


def write_matrix_file(path, A):
    """Lower triangle, `%.17e`."""
    n = A.shape[0]
    ii, jj = np.tril_indices(n)
    vals = A[ii, jj]
    keep = vals != 0.0
    ii, jj, vals = ii[keep], jj[keep], vals[keep]
    with open(path, "w") as f:
        f.write(BANNER + "\n")
        f.write("%d %d %d\n" % (n, n, len(vals)))
        for i, j, v in zip(ii, jj, vals):
            f.write("%d %d %.17e\n" % (i + 1, j + 1, v))


def _fortran_e26(v):
    """Fortran E26.16E3: 0.dddddddddddddddde+xxx, right-justified in 26 columns."""
    if v == 0.0 or not np.isfinite(v):
        mant, exp = (0.0, 0) if v == 0.0 else (v, 0)
    else:
        exp = int(np.floor(np.log10(abs(v)))) + 1
        mant = v / 10.0 ** exp
        if abs(round(mant, 16)) >= 1.0:
            mant /= 10.0
            exp += 1
    s = "%.16f" % mant
    return ("%sE%+04d" % (s, exp)).rjust(26)


def write_eigenvalues(path, values):
    with open(path, "w") as f:
        for k, v in enumerate(values):
            f.write("%8d %s\n" % (k + 1, _fortran_e26(float(v))))


def parse_printed_vecs_ranges(spec):
    """`-p` syntax of the reference (command_argument.f90:271-316): "1-3,7" -> [(1, 3), (7, 7)]."""
    out = []
    for part in spec.split(","):
        if not part or part.startswith("-") or part.endswith("-"):
            raise ValueError("invalid range %r" % part)
        a, _, b = part.partition("-")
        out.append((int(a), int(b) if b else int(a)))
    return out


def write_eigenvectors(directory, Z, ranges, binary=False):
    """print_eigenvectors (matrix_io.f90:173-285): one file `<dir>/%08d.dat` per index; text lines
    `(I8,' ',I8,' ',E26.16e3)` = i j value, or with binary=True one Fortran sequential
    unformatted record of N doubles (4-byte record markers)."""
    import os
    import struct
    n = Z.shape[0]
    for a, b in ranges:
        for j in range(a, b + 1):
            path = os.path.join(directory, "%08d.dat" % j)
            col = np.ascontiguousarray(Z[:, j - 1], dtype=np.float64)
            if binary:
                with open(path, "wb") as f:
                    f.write(struct.pack("<i", 8 * n)); f.write(col.tobytes()); f.write(struct.pack("<i", 8 * n))
            else:
                with open(path, "w") as f:
                    for i in range(n):
                        f.write("%8d %8d %s\n" % (i + 1, j, _fortran_e26(float(col[i]))))
