#!/usr/bin/env python3
"""What a 16 MiB device <-> host copy costs WHILE the library's kernels keep the GPU busy, by API and by the HIP runtime
that serves the process (--torch: import and initialise PyTorch first, whose bundled runtime then serves everything):
1-D and 2-D asynchronous copies to / from pinned memory, synchronous copies to / from pageable memory.
    python tools/copy_beside_kernels.py [--torch]"""
import ctypes
import os
import sys
import threading
import time

if "--torch" in sys.argv:
    import torch
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from eigenkernel_amd import solver  # noqa: E402

lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
hip = None
with open("/proc/self/maps") as f:
    for l in f:
        if "libamdhip64" in l:
            hip = ctypes.CDLL(l.split()[-1]); print("runtime:", l.split()[-1]); break
vp, sz = ctypes.c_void_p, ctypes.c_size_t
hip.hipHostMalloc.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(vp), sz]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(vp), ctypes.c_uint]
hip.hipMemcpyAsync.argtypes = [vp, vp, sz, ctypes.c_int, vp]
hip.hipMemcpy.argtypes = [vp, vp, sz, ctypes.c_int]
hip.hipMemcpy2DAsync.argtypes = [vp, sz, vp, sz, sz, sz, ctypes.c_int, vp]
hip.hipMemcpy2D.argtypes = [vp, sz, vp, sz, sz, sz, ctypes.c_int]
hip.hipStreamSynchronize.argtypes = [vp]
H2D, D2H = 1, 2
CH = 16 << 20
pin, dev, st = vp(), vp(), vp()
assert hip.hipHostMalloc(ctypes.byref(pin), 2 * CH, 0) == 0
assert hip.hipMalloc(ctypes.byref(dev), 4 * CH) == 0
assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
page = np.ones(2 * CH // 8)
rows, cols = 16384, CH // (16384 * 8)            # a 16 MiB slab of 128 columns of 16384 doubles; device pitch 2x


def measure(label, fn, seconds=0.6):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn(); n += 1
    dt = time.perf_counter() - t0
    print("  %-58s %6.1f GB/s  (%.2f ms per 16 MiB)" % (label, n * CH / dt / 1e9, 1e3 * dt / n), flush=True)


tests = [
    ("1-D async device -> pinned", lambda: (hip.hipMemcpyAsync(pin, dev, CH, D2H, st), hip.hipStreamSynchronize(st))),
    ("2-D async device (pitch 2x) -> pinned", lambda: (hip.hipMemcpy2DAsync(pin, rows * 8, dev, 2 * rows * 8, rows * 8, cols, D2H, st), hip.hipStreamSynchronize(st))),
    ("1-D sync  device -> pageable", lambda: hip.hipMemcpy(page.ctypes.data, dev, CH, D2H)),
    ("2-D sync  device (pitch 2x) -> pageable", lambda: hip.hipMemcpy2D(page.ctypes.data, rows * 8, dev, 2 * rows * 8, rows * 8, cols, D2H)),
    ("2-D async device (pitch 2x) -> pageable", lambda: (hip.hipMemcpy2DAsync(page.ctypes.data, rows * 8, dev, 2 * rows * 8, rows * 8, cols, D2H, st), hip.hipStreamSynchronize(st))),
    ("1-D async pinned -> device", lambda: (hip.hipMemcpyAsync(dev, pin, CH, H2D, st), hip.hipStreamSynchronize(st))),
    ("2-D async pinned -> device (pitch 2x)", lambda: (hip.hipMemcpy2DAsync(dev, 2 * rows * 8, pin, rows * 8, rows * 8, cols, H2D, st), hip.hipStreamSynchronize(st))),
    ("2-D async pageable -> device (pitch 2x)", lambda: (hip.hipMemcpy2DAsync(dev, 2 * rows * 8, page.ctypes.data, rows * 8, rows * 8, cols, H2D, st), hip.hipStreamSynchronize(st))),
]
print("GPU idle:")
for label, fn in tests: measure(label, fn, 0.3)

# the library's headline solve, over and over, in a thread (ctypes releases the GIL)
n = 16384
dp = ctypes.POINTER(ctypes.c_double)
bufs = []
for _ in range(4):
    p = vp(); assert lib.ek_hip_malloc(ctypes.byref(p), n * n * 8) == 0; bufs.append(p)
dw = vp(); assert lib.ek_hip_malloc(ctypes.byref(dw), n * 8) == 0
stop = False


def busy():
    while not stop:
        lib.ek_hip_synth_matrix_device(n, 1, bufs[0], n); lib.ek_hip_synth_matrix_device(n, 2, bufs[1], n)
        lib.ek_hip_solve_device(1, n, n, bufs[0], n, bufs[1], n, dw, bufs[2], n, None, 0)


th = threading.Thread(target=busy); th.start()
time.sleep(1.5)
print("beside the library's N = 16384 solves:")
for label, fn in tests: measure(label, fn, 1.7)
stop = True; th.join()
