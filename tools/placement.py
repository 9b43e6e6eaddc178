#!/usr/bin/env python3
"""Experiments behind DESIGN.md's "run-to-run spread, explained" (round 1): what decides whether the
one-stage tridiagonalisation runs in its fast or its slow placement mode (the relative position of its
scratch and the matrix in HBM).  One script, one sub-command per experiment:

    python tools/placement.py <addr|bench|explore|factorial|gemm|probe|scratch|split|which> [args]
"""
import sys


def cmd_addr(argv):
    sys.argv = ["placement.py addr"] + list(argv)
    """Addresses vs placement mode (tools): prints the device addresses of the matrix and scratch and the
    time of the first 64 columns for many pairs of allocations."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    n = 16384
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    GiB, MiB = 1 << 30, 1 << 20
    wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
    sec = ctypes.c_double(0)
    def alloc(nbytes):
        p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p.value
    V = alloc(MiB)
    def run(a, w):
        assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(a), ctypes.c_void_p(w), ctypes.c_void_p(V), ctypes.byref(sec)) == 0
        return sec.value * 1e3
    As = [("A%d" % i, alloc(sz * GiB + MiB)) for i, sz in enumerate((2, 2, 16, 2, 3, 16, 2))]
    Ws = [("W%d" % i, alloc(sz)) for i, sz in enumerate((wb + MiB, wb + MiB, GiB, wb + MiB, 16 * GiB, wb + MiB))]
    print("addresses:", " ".join("%s=%#x" % (k, v) for k, v in As + Ws))
    print("%-22s" % "A \\ scratch", " ".join("%-8s" % k for k, _ in Ws), " | scratch inside other A blocks: A2+4G A5+4G")
    for ka, a in As:
        row = ["%-8.2f" % run(a, w) for _, w in Ws]
        extra = ["%-8.2f" % run(a, As[2][1] + 4 * GiB), "%-8.2f" % run(a, As[5][1] + 4 * GiB)]
        print("%-22s" % ("%s=%#x" % (ka, a)), " ".join(row), " | ", " ".join(extra), flush=True)


def cmd_bench(argv):
    sys.argv = ["placement.py bench"] + list(argv)
    """Effect of the placement-aware workspace (EK_HIP_PLACEMENT) on the N=16384 GEP solve: several fresh
    allocations of the workspace in one process, stage time of the tridiagonalisation of each."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from eigenkernel_amd import solver
    n = 16384
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    dev = torch.device("cuda", 0)
    dA = torch.empty((n, n), dtype=torch.float64, device=dev); dB = torch.empty_like(dA); dZ = torch.empty_like(dA)
    dw = torch.empty((n,), dtype=torch.float64, device=dev)
    stage = (ctypes.c_double * 8)()
    hold, res, tot = [], [], []
    for rnd in range(rounds):
        for rep in range(2):
            lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n); lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n)
            assert lib.ek_hip_solve_device(1, n, n, dA.data_ptr(), n, dB.data_ptr(), n, dw.data_ptr(), dZ.data_ptr(), n, stage, 8) == 0
        res.append(round(stage[2], 4)); tot.append(round(sum(stage[i] for i in range(7)), 4))
        lib.ek_hip_finalize()
        hold.append(torch.empty(((rnd * 7 % 5 + 1) << 26,), dtype=torch.float64, device=dev))
    print("EK_HIP_PLACEMENT=%s sytrd by allocation: %s  solve: %s" % (os.environ.get("EK_HIP_PLACEMENT", "1"), res, tot), flush=True)


def cmd_explore(argv):
    sys.argv = ["placement.py explore"] + list(argv)
    import ctypes, os, sys
    sys.path.insert(0, os.getcwd())
    import torch
    from eigenkernel_amd import solver
    n = 16384
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    dev = torch.device("cuda", 0)
    dA = torch.empty((n, n), dtype=torch.float64, device=dev); dB = torch.empty_like(dA); dZ = torch.empty_like(dA)
    dw = torch.empty((n,), dtype=torch.float64, device=dev)
    stage = (ctypes.c_double * 8)()
    for rep in range(2):
        lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n); lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n)
        assert lib.ek_hip_solve_device(1, n, n, dA.data_ptr(), n, dB.data_ptr(), n, dw.data_ptr(), dZ.data_ptr(), n, stage, 8) == 0
    print("sytrd %.4f" % stage[2])


def cmd_factorial(argv):
    sys.argv = ["placement.py factorial"] + list(argv)
    """Factorial look at what decides the placement mode (tools): the matrix and the scratch of the
    tridiagonalisation in their own allocations of various sizes or inside big blocks."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    n = 16384
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    GiB, MiB = 1 << 30, 1 << 20
    wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
    sec = ctypes.c_double(0)
    def alloc(nbytes):
        p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p.value
    V = alloc(MiB)
    def run(a, w):
        assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(a), ctypes.c_void_p(w), ctypes.c_void_p(V), ctypes.byref(sec)) == 0
        return sec.value * 1e3
    A_own = alloc(2 * GiB + MiB)
    X = alloc(16 * GiB); Y = alloc(16 * GiB)
    print("A own 2 GiB | scratch own, size 64 MB .. 8 GiB:", " ".join("%s:%.2f" % (lbl, run(A_own, alloc(sz)))
          for lbl, sz in (("64M", wb + MiB), ("128M", 128 * MiB), ("256M", 256 * MiB), ("512M", 512 * MiB), ("1G", GiB), ("2G", 2 * GiB), ("4G", 4 * GiB), ("8G", 8 * GiB))), flush=True)
    print("A own 2 GiB | scratch in block X (0, 7, 15 GiB):", " ".join("%.2f" % run(A_own, X + o * GiB) for o in (0, 7, 15)), flush=True)
    print("A in X (0) | scratch in X (4, 8, 15 GiB):       ", " ".join("%.2f" % run(X, X + o * GiB) for o in (4, 8, 15)), flush=True)
    print("A in X (0) | scratch in Y (0, 8, 15 GiB):       ", " ".join("%.2f" % run(X, Y + o * GiB) for o in (0, 8, 15)), flush=True)
    print("A in X (8 GiB) | scratch in X (0, 4, 15 GiB):   ", " ".join("%.2f" % run(X + 8 * GiB, X + o * GiB) for o in (0, 4, 15)), flush=True)
    print("A in X (0) | scratch own 64 MB, own 1 GiB:      ", "%.2f %.2f" % (run(X, alloc(wb + MiB)), run(X, alloc(GiB))), flush=True)
    for sz in (3, 4, 6, 8):
        a = alloc(sz * GiB)
        print("A own %d GiB | scratch own 64 MB, in Y:         " % sz, "%.2f %.2f" % (run(a, alloc(wb + MiB)), run(a, Y + 3 * GiB)), flush=True)


def cmd_gemm(argv):
    sys.argv = ["placement.py gemm"] + list(argv)
    """Do the MFMA-bound stages care where their operands lie relative to each other (tools)?
    Shapes of the path at N=16384 with A, B, C in separate allocations; prints time per call for several
    allocation triples (a bimodal spread would mean yes)."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    GiB, MiB = 1 << 30, 1 << 20
    n = 16384
    sec = ctypes.c_double(0)
    def alloc(nbytes):
        p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p.value
    bufs = [alloc(2 * GiB + MiB) for _ in range(6)]
    for b in bufs:
        assert lib.ek_hip_synth_matrix_device(n, 1, ctypes.c_void_p(b), n) == 0
    def run(ta, tb, m, nn, k, a, b, c, beta, lower, reps=3):
        assert lib.ek_hip_debug_gemm_at(ta, tb, m, nn, k, ctypes.c_void_p(a), n, ctypes.c_void_p(b), n, beta, ctypes.c_void_p(c), n,
                                        lower, reps, ctypes.byref(sec)) == 0
        return sec.value * 1e3
    shapes = [("back-transform block  C(16384x16384) -= V(16384x512) T", 0, 0, n, n, 512, 1.0, 0),
              ("rank-128 trailing update, lower (SYR2K)", 0, 1, n, n, 128, 1.0, 1),
              ("half-size product (8192^3), beta=0", 0, 0, 8192, 8192, 8192, 0.0, 0),
              ("solve update C(8192x16384) -= L21 X", 0, 0, 8192, n, 8192, 1.0, 0)]
    for name, ta, tb, m, nn, k, beta, lower in shapes:
        row = []
        for (ia, ib, ic) in ((0, 1, 2), (0, 1, 3), (0, 1, 4), (0, 1, 5), (2, 3, 0), (2, 3, 1), (4, 5, 0), (0, 0, 1), (0, 1, 1)):
            row.append("%.3f" % run(ta, tb, m, nn, k, bufs[ia], bufs[ib], bufs[ic], beta, lower))
        print("%-60s %s ms" % (name, " ".join(row)), flush=True)


def cmd_probe(argv):
    sys.argv = ["placement.py probe"] + list(argv)
    """Does a short probe (the first panels of the tridiagonalisation) predict which of the two
    placement modes (DESIGN.md: 'run-to-run spread, explained') a workspace allocation is in?
    For several placements of the library's workspace: probe time (first 128 columns, 5 reps) and the
    full N=16384 tridiagonalisation time."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    n = 16384
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    sec = ctypes.c_double(0)
    hold = []
    for rnd in range(rounds):
        lib.ek_hip_debug_set_sytrd_maxcols(-1)
        lib.ek_hip_debug_sytrd(n, 0, 1, ctypes.byref(sec))          # allocates the workspace, warms up
        lib.ek_hip_debug_set_sytrd_maxcols(128)
        probes = []
        for _ in range(3):
            lib.ek_hip_debug_sytrd(n, 0, 5, ctypes.byref(sec)); probes.append(sec.value * 1e3)
        lib.ek_hip_debug_set_sytrd_maxcols(-1)
        lib.ek_hip_debug_sytrd(n, 0, 2, ctypes.byref(sec)); full = sec.value
        print("placement %d: probe (128 columns) %s ms, full sytrd %.4f s" % (rnd, ["%.3f" % p for p in probes], full), flush=True)
        lib.ek_hip_finalize()
        p = ctypes.c_void_p()
        assert lib.ek_hip_malloc(ctypes.byref(p), ((rnd * 7 % 5 + 1) << 29)) == 0     # 0.5 .. 2.5 GiB ballast, kept
        hold.append(p)


def cmd_scratch(argv):
    sys.argv = ["placement.py scratch"] + list(argv)
    """Placement of the tridiagonalisation's SCRATCH (x, panel, partial sums: 64 MB at N=16384) with the
    matrix fixed: time of the first 64 columns for scratch positions spread over a 16 GiB block, and for
    separately allocated scratch buffers (tools)."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    n = 16384
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    GiB, MiB = 1 << 30, 1 << 20
    wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
    sec = ctypes.c_double(0)
    def alloc(nbytes):
        p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), nbytes) == 0; return p
    A = alloc(2 * GiB + MiB); V = alloc(MiB)
    def run(work_addr):
        assert lib.ek_hip_debug_sytrd_at(n, 64, 3, A, ctypes.c_void_p(work_addr), V, ctypes.byref(sec)) == 0
        return sec.value * 1e3
    blk = alloc(16 * GiB)
    row = []
    for k in range(0, 64):
        off = k * 256 * MiB
        if off + wb > 16 * GiB: break
        row.append(run(blk.value + off))
    print("scratch at 256 MiB steps inside one 16 GiB block:", " ".join("%.2f" % t for t in row), flush=True)
    row = []
    keep = []
    for k in range(24):
        p = alloc(wb + MiB); keep.append(p)
        row.append(run(p.value))
    print("24 separately allocated scratch buffers:          ", " ".join("%.2f" % t for t in row), flush=True)
    row = []
    for k in range(16):
        row.append(run(blk.value + 16 * GiB - wb - (k * 4 + 1) * MiB))
    print("scratch 1, 5, 9, ... MiB below the top of the block:", " ".join("%.2f" % t for t in row), flush=True)


def cmd_split(argv):
    sys.argv = ["placement.py split"] + list(argv)
    """Which part of the tridiagonalisation's scratch decides the placement mode (tools)?  Finds a
    same-colour (slow) and a different-colour (fast) scratch for one matrix, then moves sub-buffers one by
    one from the slow scratch into the fast one (mask: 1 x, 2 panel, 4 row-part sums, 8 column-part sums,
    16 the rest) and times the first 64 columns."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    n = 16384
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    GiB, MiB = 1 << 30, 1 << 20
    wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
    sec = ctypes.c_double(0)
    def alloc(nbytes):
        p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p.value
    V = alloc(MiB); A = alloc(2 * GiB + MiB)
    def run(w):
        assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(A), ctypes.c_void_p(w), ctypes.c_void_p(V), ctypes.byref(sec)) == 0
        return sec.value * 1e3
    slow = fast = None
    for _ in range(12):
        w = alloc(wb + MiB); t = run(w)
        if t > 13.3 and slow is None: slow = w
        if t < 13.15 and fast is None: fast = w
        if slow and fast: break
    assert slow and fast, "both colours not found"
    print("slow scratch %.3f ms, fast scratch %.3f ms" % (run(slow), run(fast)))
    for mask, name in ((1, "x"), (2, "panel"), (4, "row-part sums"), (8, "column-part sums"), (16, "rest"),
                       (12, "both partial sums"), (3, "x + panel"), (31, "all")):
        lib.ek_hip_debug_sytrd_split(ctypes.c_void_p(fast), mask)
        a = run(slow)
        lib.ek_hip_debug_sytrd_split(ctypes.c_void_p(slow), mask)
        b = run(fast)
        print("%-20s moved slow->fast: %.3f ms   moved fast->slow: %.3f ms" % (name, a, b), flush=True)
    lib.ek_hip_debug_sytrd_split(None, 0)


def cmd_which(argv):
    sys.argv = ["placement.py which"] + list(argv)
    """Which array's position decides the placement mode of the tridiagonalisation (tools)?
    Blocks of 21 GiB (the library's arena of ~13 GiB + 8 GiB of slack); the matrix (2 GiB at N=16384),
    the stage scratch and the vectors are put at chosen offsets; time of the first 64 columns."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from eigenkernel_amd import solver
    n = 16384
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    GiB, MiB = 1 << 30, 1 << 20
    wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
    arena = 13 * GiB + 64 * MiB
    work_off = 8 * GiB + 32 * MiB
    slack = 8 * GiB
    total = arena + slack
    sec = ctypes.c_double(0)
    hold = []
    for b in range(3):
        blk = ctypes.c_void_p()
        assert lib.ek_hip_malloc(ctypes.byref(blk), total) == 0
        hold.append(blk)
        base = blk.value
        def run(offA, offW, offV):
            assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(base + offA), ctypes.c_void_p(base + offW),
                                             ctypes.c_void_p(base + offV), ctypes.byref(sec)) == 0
            return sec.value * 1e3
        rows = [("library layout, shift 0", 0, work_off, arena - MiB),
                ("library layout, shift 8 GiB (top)", slack, slack + work_off, slack + arena - MiB),
                ("A as at top, work+vecs as at shift 0", slack, work_off, arena - MiB),
                ("A as at shift 0, work+vecs as at top", 0, slack + work_off, slack + arena - MiB),
                ("A as at shift 0, work at shift 0, vecs at top", 0, work_off, slack + arena - MiB),
                ("A as at shift 0, work at top, vecs at shift 0", 0, slack + work_off, arena - MiB)]
        for name, a, w, v in rows:
            print("block %d: %-48s %.3f ms" % (b, name, run(a, w, v)), flush=True)


COMMANDS = {"addr": cmd_addr, "bench": cmd_bench, "explore": cmd_explore, "factorial": cmd_factorial, "gemm": cmd_gemm, "probe": cmd_probe, "scratch": cmd_scratch, "split": cmd_split, "which": cmd_which}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
