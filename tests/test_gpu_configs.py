"""Every BASELINE.json configuration at FULL size on the GPU, through the C-ABI.

    C2  N=4096  standard EVP, full spectrum, 1x1 grid
    C3  N=16384 generalized EVP, full spectrum, 1x1 grid          (the headline of bench.py)
    C4  N=32768 generalized EVP: 1x1 grid and one rank of the 2x4 grid the reference lays 8 ranks out on
    C5  N=16384 generalized EVP, lowest 1024 pairs (`*_select` arm): 1x1 and one rank of the 2x4 grid

Eigenvalues of C2, C3, C4 and C5 are held to the reference's own library path (the six ScaLAPACK calls of
solver_scalapack_all.f90:59-115 / generalized_to_standard.f90:24-103, oracle/scalapack_path.c on
oneMKL ScaLAPACK, 2x4 grid, NB=64; fixtures from tests/golden/make_scalapack_goldens.sh) within the
SURVEY.md 8(c) bound  max|l - l_ref| <= N eps max|l|.  Every configuration is also held to the
size-independent acceptance quantities of the reference's own verifier (verifier.f90:75-204, 233-330)
evaluated on the GPU against fresh copies of the inputs -- N B-orthonormal vectors with residuals at
rounding level ARE the full spectrum -- and, for the 8-rank layouts C4 and C5, to bit-identity between
one cell of the 2 x 4 grid and the 1x1 result.
"""
import ctypes
import os

import numpy as np
import pytest

from eigenkernel_amd import descriptor as _d

pytestmark = pytest.mark.gpu
EPS = 2.220446049250313e-16
_dp = ctypes.POINTER(ctypes.c_double)


class _Dev:
    """A handful of device buffers released on exit."""

    def __init__(self, lib):
        self.lib, self.ptrs = lib, []

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        assert self.lib.ek_hip_malloc(ctypes.byref(p), max(int(nbytes), 8)) == 0
        self.ptrs.append(p)
        return p

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for p in self.ptrs:
            self.lib.ek_hip_free(p)
        self.lib.ek_hip_finalize()      # give the cached workspaces back between the big cases


def _d2h(lib, dptr, shape):
    out = np.zeros(shape, order="F")
    assert lib.ek_hip_memcpy_d2h(out.ctypes.data, dptr, out.nbytes) == 0
    return out


def _acceptance(lib, gep, n, n_vec, dA0, dB0, dw, dZ):
    """The reference's -c / -t quantities on the GPU against the ORIGINAL matrices."""
    an, ave, mx, orth = (ctypes.c_double(0) for _ in range(4))
    assert lib.ek_hip_residual_device(1 if gep else 0, n, n_vec, dA0, n, dB0 if gep else None, n, dw, dZ, n,
                                      ctypes.byref(an), ctypes.byref(ave), ctypes.byref(mx)) == 0
    assert lib.ek_hip_orthogonality_device(1 if gep else 0, n, 1, n_vec, dB0 if gep else None, n, dZ, n,
                                           ctypes.byref(orth)) == 0
    assert mx.value <= 1e-14 * max(1.0, np.sqrt(n / 1024.0)), mx.value
    assert orth.value <= 1e-11, orth.value
    return mx.value, orth.value


def _solve_1x1(lib, dev, gep, n, n_vec):
    nn = n * n * 8
    dA, dA0, dZ, dw = dev.alloc(nn), dev.alloc(nn), dev.alloc(nn), dev.alloc(n * 8)
    dB, dB0 = (dev.alloc(nn), dev.alloc(nn)) if gep else (None, None)
    for dst in (dA, dA0):
        assert lib.ek_hip_synth_matrix_device(n, 1, dst, n) == 0
    if gep:
        for dst in (dB, dB0):
            assert lib.ek_hip_synth_matrix_device(n, 2, dst, n) == 0
    st = np.zeros(8)
    info = lib.ek_hip_solve_device(1 if gep else 0, n, n_vec, dA, n, dB, n, dw, dZ, n, st.ctypes.data_as(_dp), 8)
    assert info == 0
    w = _d2h(lib, dw, (n,))
    assert np.all(np.diff(w[:n_vec]) >= 0)
    return dict(dA=dA, dB=dB, dA0=dA0, dB0=dB0, dZ=dZ, dw=dw, w=w, stages=st)


def _grid_piece_is_bit_identical(lib, dev, r, gep, n, n_vec, grid, cell, nb=64, dc_team=False):
    """One rank of an nprow x npcol grid (replicated-input mode, no collective: a rank's work does not
    depend on the others, so the one GPU can play it): its block-cyclic piece of Z equals the 1x1 result.
    dc_team: with the divide & conquer's team form rehearsed by the cell (a team of npcol ranks, the library's own number
    of sharded heights for the order: ek_stedc.hip StedcTeam) -- still the 1x1 result bit for bit."""
    nprow, npcol = grid
    myrow, mycol = cell
    assert lib.ek_hip_debug_stedc_team(npcol if dc_team else 0, -1, 0) == 0
    assert lib.ek_hip_synth_matrix_device(n, 1, r["dA"], n) == 0     # the solve destroyed A and B
    if gep:
        assert lib.ek_hip_synth_matrix_device(n, 2, r["dB"], n) == 0
    lr = _d.numroc(n, nb, myrow, 0, nprow)
    lc = _d.numroc(n_vec, nb, mycol, 0, npcol)
    dZl, dw2 = dev.alloc(max(lr, 1) * max(lc, 1) * 8), dev.alloc(n * 8)
    info = lib.ek_hip_solve_device_grid(1 if gep else 0, n, n_vec, r["dA"], n, r["dB"], n, dw2, dZl, max(lr, 1),
                                        nb, nprow, npcol, myrow, mycol, None, 0)
    assert info == 0
    w2 = _d2h(lib, dw2, (n,))
    assert np.array_equal(w2[:n_vec], r["w"][:n_vec])
    Zl = _d2h(lib, dZl, (max(lr, 1), max(lc, 1)))[:lr, :lc]
    ri = _d.local_indices(n, nb, myrow, nprow)
    ci = _d.local_indices(n_vec, nb, mycol, npcol)
    # the 1x1 eigenvector matrix, column block by column block (keeps the host copy small)
    for c0 in range(0, len(ci), 512):
        cols = ci[c0:c0 + 512]
        blk = np.zeros((n, len(cols)), order="F")
        for k, c in enumerate(cols):      # columns of Z are contiguous in HBM
            assert lib.ek_hip_memcpy_d2h(blk[:, k:k + 1].ctypes.data, ctypes.c_void_p(r["dZ"].value + int(c) * n * 8),
                                         n * 8) == 0
        assert np.array_equal(Zl[:, c0:c0 + len(cols)], blk[ri, :])
    assert lib.ek_hip_debug_stedc_team(0, -1, 0) == 0


def test_c2_n4096_standard_full_spectrum(hip, golden_dir):
    lib = hip.load_library()
    n = 4096
    w_ref = np.loadtxt(os.path.join(golden_dir, "scalapack_synth_sep_n4096_np8.txt"))
    with _Dev(lib) as dev:
        r = _solve_1x1(lib, dev, False, n, n)
        assert np.abs(r["w"] - w_ref).max() <= n * EPS * np.abs(w_ref).max()
        _acceptance(lib, False, n, n, r["dA0"], None, r["dw"], r["dZ"])


def test_c3_n16384_generalized_full_spectrum(hip, golden_dir):
    """The headline configuration of bench.py, held to the reference's library path."""
    lib = hip.load_library()
    n = 16384
    w_ref = np.loadtxt(os.path.join(golden_dir, "scalapack_synth_gep_n16384_np8.txt"))
    with _Dev(lib) as dev:
        r = _solve_1x1(lib, dev, True, n, n)
        assert np.abs(r["w"] - w_ref).max() <= n * EPS * np.abs(w_ref).max()
        _acceptance(lib, True, n, n, r["dA0"], r["dB0"], r["dw"], r["dZ"])
        for name in range(7):
            assert r["stages"][name] >= 0.0
        assert r["stages"][:7].sum() > 0.0


def test_c5_n16384_generalized_lowest_1024(hip, golden_dir):
    lib = hip.load_library()
    n, n_vec = 16384, 1024
    w_ref = np.loadtxt(os.path.join(golden_dir, "scalapack_synth_gep_n16384_np8.txt"))
    with _Dev(lib) as dev:
        r = _solve_1x1(lib, dev, True, n, n_vec)
        assert np.abs(r["w"][:n_vec] - w_ref[:n_vec]).max() <= n * EPS * np.abs(w_ref).max()
        _acceptance(lib, True, n, n_vec, r["dA0"], r["dB0"], r["dw"], r["dZ"])
        # rank (0, 2) of the 2 x 4 grid layout_procs gives 8 ranks (processes.f90:56-65)
        _grid_piece_is_bit_identical(lib, dev, r, True, n, n_vec, (2, 4), (0, 2))
        _grid_piece_is_bit_identical(lib, dev, r, True, n, n_vec, (2, 4), (1, 1), dc_team=True)


def test_c4_n32768_generalized_full_spectrum(hip, golden_dir):
    """BASELINE.json configs[3] at full size on one GPU, eigenvalues held to the reference's library path (the same
    six ScaLAPACK calls on the 2 x 4 grid, tests/golden/make_scalapack_goldens.sh) within N eps max|lambda|."""
    lib = hip.load_library()
    n = 32768
    w_ref = np.loadtxt(os.path.join(golden_dir, "scalapack_synth_gep_n32768_np8.txt"))
    with _Dev(lib) as dev:
        r = _solve_1x1(lib, dev, True, n, n)
        assert np.abs(r["w"] - w_ref).max() <= n * EPS * np.abs(w_ref).max()
        _acceptance(lib, True, n, n, r["dA0"], r["dB0"], r["dw"], r["dZ"])
        w = r["w"]
        # the generator's spectrum: GEP eigenvalues inside [0.38, 2.63]
        assert 0.3 < w[0] < 0.5 and 2.4 < w[-1] < 2.8
        _grid_piece_is_bit_identical(lib, dev, r, True, n, n, (2, 4), (1, 3))
        _grid_piece_is_bit_identical(lib, dev, r, True, n, n, (1, 8), (0, 5), dc_team=True)


@pytest.mark.parametrize("gep,n,n_vec", [(True, 1024, 1024), (False, 1280, 1280), (True, 1024, 200)])
def test_caller_arrays_used_in_place_change_no_bit(hip, monkeypatch, gep, n, n_vec):
    """Device arrays that already have the internal layout (order a multiple of 128, ld = order, 256-byte aligned) ARE
    the work arrays of ek_hip_solve_device (ek_solve.hip, EK_HIP_ALIAS).  The same call with copies forced
    (EK_HIP_ALIAS=0), with padded leading dimensions (ld = order + 8) and with arrays that start 16 bytes off the
    alignment must give the same eigenvalues, eigenvectors and in-place lower triangles of A and B bit for bit, and must
    not write outside the n x n (n x n_vec) windows."""
    lib = hip.load_library()
    assert lib.ek_hip_init(0) == 0
    prob = 1 if gep else 0

    def run(pad, shift, alias):
        monkeypatch.setenv("EK_HIP_ALIAS", "1" if alias else "0")
        ld = n + pad
        with _Dev(lib) as dev:
            nb = ld * n * 8 + 256
            bufs = {k: dev.alloc(nb) for k in (("A", "B", "Z") if gep else ("A", "Z"))}
            dw = dev.alloc(n * 8)
            host = {}
            for k, p in bufs.items():
                h = np.asfortranarray(np.full((ld, n), 9.75))
                if k != "Z":
                    src = dev.alloc(n * n * 8)
                    assert lib.ek_hip_synth_matrix_device(n, 1 if k == "A" else 2, src, n) == 0
                    h[:n, :] = _d2h(lib, src, (n, n))
                flat = np.full(nb // 8, 4.5)
                flat[shift // 8: shift // 8 + ld * n] = h.reshape(-1, order="F")
                assert lib.ek_hip_memcpy_h2d(p, flat.ctypes.data, flat.nbytes) == 0
                host[k] = flat
            ptr = {k: ctypes.c_void_p(p.value + shift) for k, p in bufs.items()}
            st = np.zeros(8)
            info = lib.ek_hip_solve_device(prob, n, n_vec, ptr["A"], ld, ptr.get("B"), ld, dw, ptr["Z"], ld,
                                           st.ctypes.data_as(_dp), 8)
            assert info == 0
            out = {}
            for k, p in bufs.items():
                flat = np.zeros(nb // 8)
                assert lib.ek_hip_memcpy_d2h(flat.ctypes.data, p, flat.nbytes) == 0
                body = flat[shift // 8: shift // 8 + ld * n].reshape((ld, n), order="F")
                assert (flat[:shift // 8] == 4.5).all() and (flat[shift // 8 + ld * n:] == 4.5).all(), k
                assert (body[n:] == 9.75).all(), k                     # padding rows
                if k == "Z":
                    assert (body[:n, n_vec:] == 9.75).all()            # columns nobody asked for
                out[k] = body[:n].copy()
            w = np.zeros(n)
            assert lib.ek_hip_memcpy_d2h(w.ctypes.data, dw, n * 8) == 0
            return w, out

    w0, o0 = run(0, 0, True)          # in place
    for pad, shift, alias in ((0, 0, False), (8, 0, True), (0, 16, True)):
        w1, o1 = run(pad, shift, alias)
        assert np.array_equal(w0[:n_vec], w1[:n_vec]), (pad, shift, alias)
        assert np.array_equal(o0["Z"][:, :n_vec], o1["Z"][:, :n_vec]), (pad, shift, alias)
        assert np.array_equal(np.tril(o0["A"]), np.tril(o1["A"])), (pad, shift, alias)
        if gep:
            assert np.array_equal(np.tril(o0["B"]), np.tril(o1["B"])), (pad, shift, alias)


@pytest.mark.parametrize("gep,n,n_vec", [(False, 5699, 300), (True, 6401, 6401)])
def test_ragged_orders_where_the_panels_go_in_pairs(hip, gep, n, n_vec):
    """Odd orders just above 5120 + 512: the first panels of the dense -> band stage go in pairs (rank-256 updates, the second
    panel's SYMM on the not yet updated matrix), the trailing matrices are no multiples of 64, and the flow switches to single
    panels on the way: the reference's acceptance quantities on the GPU (tools/sanity_ragged.py is the manual sweep up to
    20011)."""
    lib = hip.load_library()
    with _Dev(lib) as dev:
        r = _solve_1x1(lib, dev, gep, n, n_vec)
        assert (np.diff(r["w"][:n_vec]) >= 0).all()
        _acceptance(lib, gep, n, n_vec, r["dA0"], r["dB0"] if gep else None, r["dw"], r["dZ"])
