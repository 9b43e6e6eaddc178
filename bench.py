#!/usr/bin/env python3
"""bench.py -- headline measurement of the hot path on MI355X.

A "step" is one full-spectrum generalized eigensolve (Cholesky + reduction +
tridiagonalisation + divide & conquer + back-transformation + recovery) of the synthetic
dense SPD pair of SURVEY.md 8(d) (A = seed 1, B = seed 2) at N = 16384 (BASELINE.json
configs[2], the configuration the metric is quoted on), inputs resident in HBM when the
timed region starts (they are regenerated on the device before every step, outside the
timed region, because the solve overwrites A and B as the reference does).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--order 16384] [--problem gep|sep]

For N > 1 there is one rank per GPU: either the driver launches them with torch.distributed.run, or --
when no RANK is in the environment -- `python bench.py --gpus N` starts them itself as fresh child
processes (self_launch(): the parent makes no GPU call and relays the ranks' line and exit code).  By default
(`--distribution auto`) the ranks first solve one problem each (replicas: no data-path collective;
the safe measurement, kept in the line as "replicas"), then ONE problem on the 1 x N process grid
with the library's RCCL communicator attached -- Cholesky factor, reduction and the dense -> band stage of
the tridiagonalisation distributed over the ranks (per panel one broadcast and one all-reduce), band ->
tridiagonal replicated, eigenvector stages sharded by columns -- to the full contract (W warm-up solves,
exactly K solves between barriers, max over ranks, parity checked on every rank).  If it passed it becomes the headline
("scaling": "strong", value = n_vec * K / time: the eigenpairs of the one problem all ranks worked
on); if neither passes (or an exchange hangs: a watchdog abandons it) the replicas line is the
headline ("scaling": "weak").  `--distribution replicas | columns | grid` force one mode.

`--config c2|c3|c4|c5` selects a BASELINE.json configuration by name (c3, the default, is the one
the metric is quoted on; c4 and c5 name 8-GPU layouts: with fewer ranks the same problem is run on
the ranks there are).

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline"        : the dominant kernel, timed live with HIP events on its launch stream in the
                      timed region: with the two-stage tridiagonalisation (orders >= 512) the
                      MFMA-bound application of the bulge-chasing reflectors (q2_apply_nb_kernel<4>),
                      else the HBM-bound symv of the one-stage reduction; "traffic" from the committed PMC
                      record only while the kernel's source file is the one it was measured with;
  "roofline_stages" : per stage of the path, algorithmic flops / device seconds
                      against the fp64 matrix peak, and the flops the path executed where that differs
                      (divide & conquer after deflation, counted on the device; both back-transformations);
  "value_incl_copies": the same solve through ek_hip_solve on HOST arrays (PCIe staging of A, B in
                      and Z, A, B, w out included -- SURVEY.md 8(d)'s t_solve; the copies overlap the stages),
                      the second of two calls, outside the timed region, in a child process that loads the library
                      alone (the system's HIP runtime, as a host of the reference's shape links it) with the same
                      measurement inside this process (PyTorch's bundled HIP runtime) beside it;
  "other_configs"   : the other BASELINE configurations at full size on this GPU (c2 x10, c5 x5, c4 x2 steps:
                      ms_per_step, parity, the dominant kernel's fraction), after the headline region;
  "cpu_baseline"    : the reference's ScaLAPACK call sequence on the host cores on a bounded sample
                      (smaller N) of the same generator, rank 0 at N=1 only, with the GPU path timed
                      at that SAME order beside it ("gpu_same_order").
Every extra runs after the headline exists and is caught on its own: none of them can take the line down.
"""
import argparse
import numpy as np
import ctypes
import json
import os
import signal
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6   # fp64 matrix peak per GPU


# BASELINE.json configs by name: (order, problem, n_vec; 0 = full spectrum).  c1 (BNZ30, N = 30) is the
# reference's own CPU-runnable plumbing case: a parity test (tests/test_gpu_path.py), not a bench line.
CONFIGS = {"c2": (4096, "sep", 0), "c3": (16384, "gep", 0), "c4": (32768, "gep", 0), "c5": (16384, "gep", 1024)}


def flops(problem, n, n_vec):
    """Algorithmic flops F(N) of SURVEY.md 8(d)."""
    n3 = float(n) ** 3
    if n_vec == n:
        return 7.0 * n3 if problem == 1 else 14.0 / 3.0 * n3
    k = float(n_vec)
    base = 4.0 * n3 / 3.0 + 2.0 * n * n * k
    return base + (n3 / 3.0 + n3 + n * n * k if problem == 1 else 0.0)


def cpu_baseline(problem, sample_n):
    """The CPU oracle (oracle/ek_oracle.c, scalar C port of the reference's call sequence)
    timed on one host core on a smaller instance of the same synthetic workload."""
    import numpy as np  # noqa: F401
    from oracle import ek_oracle
    A = ek_oracle.synth_matrix(sample_n, 1)
    B = ek_oracle.synth_matrix(sample_n, 2) if problem == 1 else None
    t0 = time.perf_counter()
    w, Z, info, _ = ek_oracle.solve(A, B)
    dt = time.perf_counter() - t0
    assert info == 0
    return {"value": sample_n / dt, "unit": "eigenpairs/s", "cores": 1, "kind": "port",
            "seconds": dt,
            "sample": "oracle/ek_oracle.c ok_solve (unblocked DPOTF2/DSYGS2/DSYTD2/D&C/ORM2L/TRSV "
                      "restatement), %s N=%d of the same generator, 1 thread"
                      % ("GEP" if problem == 1 else "SEP", sample_n),
            "gflops_equiv": flops(problem, sample_n, sample_n) / dt / 1e9}


def cpu_baseline_scalapack(problem, sample_n, limit=600.0):
    """The reference's own CPU path: the same six ScaLAPACK calls in the same order
    (oracle/scalapack_path.c, oneMKL ScaLAPACK + MPICH from /opt/conda, NB=64, near-square
    grid of processes.f90:56-65), one rank per physical core (max 64), 1 BLAS thread per rank.
    Returns None when the driver or mpiexec is missing on this box."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "scalapack_path")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        return None
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        cores = os.cpu_count() or 1
    np_ = max(1, min(cores, 64))
    # the ranks are CPU-only children: nothing of a profiler wrapped around bench.py (its preloaded
    # tool library would initialise the GPU in every rank and across every exec hop of mpiexec) and
    # no GPU may reach them
    env = {k: v for k, v in os.environ.items()
           if not (k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE", "ROCP_TOOL_LIBRARIES")
                   or k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX", "HSA_TOOLS")))}
    env.update(MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    global _cpu_child
    try:
        # a child we wait for with a progress line every half minute (a run at the headline order is silent for minutes,
        # and a silent command is taken for a hung one) and end ourselves when it overruns `limit`.  It leads a session of
        # its own (mpiexec and its ranks are one process group that this bench can end as a whole), so nobody else's
        # process-group kill reaches it: the handle is kept where _on_term finds it, and every way out of this function
        # ends the group.
        t0 = time.perf_counter()
        child = subprocess.Popen([mpiexec, "-np", str(np_), exe, str(sample_n), str(problem)], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, start_new_session=True)
        _cpu_child = child
        try:
            nxt = 30.0
            while child.poll() is None:
                time.sleep(0.5)
                el = time.perf_counter() - t0
                if el > limit:
                    raise TimeoutError("ScaLAPACK baseline at N=%d exceeded %.0f s" % (sample_n, limit))
                if el > nxt:
                    sys.stderr.write("[bench] ScaLAPACK baseline N=%d np=%d: %.0f s so far\n" % (sample_n, np_, el))
                    sys.stderr.flush()
                    nxt += 30.0
            stdout = child.stdout.read()
        finally:
            _kill_cpu_child()
        line = [l for l in stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
    except Exception as exc:   # pragma: no cover - depends on the box
        sys.stderr.write("scalapack baseline unavailable: %r\n" % (exc,))
        return None
    solve = sum(j["stages"].values())
    return {"value": sample_n / solve, "unit": "eigenpairs/s", "cores": np_, "kind": "port",
            "seconds": solve, "stages": j["stages"], "grid": j["grid"],
            "sample": "reference call sequence (PDPOTRF, PDSYGST, PDSYTRD, gather, PDSTEDC, PDORMTR, PDTRTRS; "
                      "oracle/scalapack_path.c) on oneMKL ScaLAPACK + MPICH, %s N=%d of the same generator, "
                      "np=%d (grid %dx%d), NB=64, 1 BLAS thread/rank" % (
                          "GEP" if problem == 1 else "SEP", sample_n, np_, j["grid"][0], j["grid"][1]),
            "gflops_equiv": flops(problem, sample_n, sample_n) / solve / 1e9}


def gpu_step_at(lib, torch, dev, problem, n, steps=3):
    """The GPU path at the order the CPU baseline was run on (device-resident inputs, like the headline)."""
    dA = torch.empty((n, n), dtype=torch.float64, device=dev)
    dB = torch.empty((n, n), dtype=torch.float64, device=dev) if problem == 1 else None
    dZ = torch.empty((n, n), dtype=torch.float64, device=dev)
    dw = torch.empty((n,), dtype=torch.float64, device=dev)
    tot = 0.0
    for it in range(steps + 1):
        if lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n) != 0:
            raise RuntimeError("synth")
        if problem == 1 and lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n) != 0:
            raise RuntimeError("synth")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = lib.ek_hip_solve_device(problem, n, n, dA.data_ptr(), n, dB.data_ptr() if problem == 1 else None, n,
                                       dw.data_ptr(), dZ.data_ptr(), n, None, 0)
        torch.cuda.synchronize()
        if info != 0:
            raise RuntimeError("ek_hip_solve_device info=%d" % info)
        if it > 0:
            tot += time.perf_counter() - t0
    sec = tot / steps
    return {"n": n, "seconds": sec, "value": n / sec, "unit": "eigenpairs/s",
            "gflops_equiv": flops(problem, n, n) / sec / 1e9}


def host_path_step(lib, solver, problem, n, n_vec):
    """One solve through ek_hip_solve on host arrays: what a host of the reference's shape sees, PCIe
    staging included.  Inputs are generated on the device and copied out first."""
    import numpy as np
    from eigenkernel_amd import descriptor as dsc
    tmp = ctypes.c_void_p()
    try:
        A = np.zeros((n, n), order="F"); B = np.zeros((n, n), order="F") if problem == 1 else None
        Z = np.zeros((n, n), order="F"); w = np.zeros(n)
        Z.fill(0.0)      # mapped before the call, like A and B (np.zeros leaves the pages to the first write: 2 GB of
        #                  page faults inside the timed call, 0.1 - 0.25 s depending on the box)
        if lib.ek_hip_malloc(ctypes.byref(tmp), n * n * 8) != 0:
            return {"error": "ek_hip_malloc"}
        desc = dsc.descinit(n, n, 64, 64, 0, 0, 0, n)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        st = (ctypes.c_double * 8)()
        # Two calls at full size, inputs regenerated in front of each (the solve overwrites A and B): the first sets up
        # what the library keeps between calls (device images of the caller's arrays and the worker
        # threads of the staging pipeline) and is reported as first_call_seconds; the second is the figure.  (Until round
        # 4's last day the first call was a small one, order 2048: the device images then grew inside the timed call --
        # 6 GiB of hipMalloc, 0.05 - 0.12 s depending on the box.)
        secs = []
        for _call in range(2):
            for seed, M in ((1, A), (2, B)):
                if M is not None:
                    if lib.ek_hip_synth_matrix_device(n, seed, tmp, n) != 0 or lib.ek_hip_memcpy_d2h(M.ctypes.data, tmp, n * n * 8) != 0:
                        return {"error": "input generation"}
            t0 = time.perf_counter()
            info = lib.ek_hip_solve(problem, n, n_vec, A.ctypes.data_as(dp), desc.ctypes.data_as(ip),
                                    B.ctypes.data_as(dp) if problem == 1 else None, desc.ctypes.data_as(ip) if problem == 1 else None,
                                    w.ctypes.data_as(dp), Z.ctypes.data_as(dp), desc.ctypes.data_as(ip), 1, 1, 0, 0, st, 8)
            secs.append(time.perf_counter() - t0)
            if info != 0:
                break
        sec = secs[-1]
        lib.ek_hip_free(tmp); tmp = ctypes.c_void_p()
        if info != 0:
            return {"error": "ek_hip_solve info=%d" % info}
        ps = (ctypes.c_double * 12)()
        lib.ek_hip_debug_last_pipe_stats(ps, 12)
        pipe = {"bytes_in": ps[0], "in_span_seconds": ps[1], "in_gb_per_s": ps[0] / ps[1] / 1e9 if ps[1] > 0 else None,
                "bytes_out": ps[3], "out_busy_seconds": ps[5], "main_thread_waited_for_inputs_seconds": ps[6],
                "main_thread_waited_for_the_drain_seconds": ps[7], "workers_in": int(ps[8]) // 100, "workers_out": int(ps[8]) % 100,
                "pipeline_seconds": ps[10], "stage_seconds_sum": sum(st[q] for q in range(7))}
        return {"value": n_vec / sec, "unit": "eigenpairs/s", "seconds": sec, "first_call_seconds": secs[0],
                "host_device_copies_seconds": st[7], "pipeline": pipe,
                "note": "the second of two ek_hip_solve calls on pageable host arrays, all mapped before the calls (A, B in; "
                        "Z, A, B, w out); the first one allocates what the library keeps between calls"}
    except Exception as exc:      # an optional extra never takes the line down
        return {"error": repr(exc)}
    finally:
        if tmp:
            lib.ek_hip_free(tmp)


def host_path_child(problem, n, n_vec, timeout=300.0):
    """host_path_step in a child process that loads the library alone.  This process has PyTorch in it, and PyTorch ships
    its own HIP runtime (torch/lib/libamdhip64.so, libhsa-runtime64.so) which, loaded first, serves the library's calls
    too: its copies into pageable host memory run at 0.8 GB/s per thread beside the kernels where the system's runtime
    (/opt/rocm: what a host of the reference's shape links -- host/eigenkernel_hip_app does) moves 10 GB/s per thread.
    The child is the reference-shaped host; the figure measured in this process is reported beside it."""
    import subprocess
    code = ("import json, sys; sys.path.insert(0, %r); import bench; from eigenkernel_amd import solver; "
            "lib = solver.load_library(); assert lib.ek_hip_init(0) == 0; "
            "print('HOSTPATH ' + json.dumps(bench.host_path_step(lib, solver, %d, %d, %d)))"
            % (os.path.dirname(os.path.abspath(__file__)), problem, n, n_vec))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
    except Exception as exc:
        return {"error": repr(exc)}
    for line in r.stdout.splitlines():
        if line.startswith("HOSTPATH "):
            return json.loads(line[len("HOSTPATH "):])
    return {"error": "child rc=%s: %s" % (r.returncode, (r.stderr or "")[-300:])}


def run_other_config(lib, torch, dev, name, steps, warmup):
    """One more BASELINE configuration on this GPU after the headline region: `steps` timed solves (each between
    synchronisations, inputs regenerated outside the timed part), parity of the last output through the reference's
    acceptance quantities, and the fraction of the fp64 matrix peak its largest kernel (q2_apply_nb_kernel) reached."""
    n, prob, nv = CONFIGS[name]
    problem = 1 if prob == "gep" else 0
    n_vec = nv if 0 < nv < n else n
    dA = torch.empty((n, n), dtype=torch.float64, device=dev)
    dB = torch.empty((n, n), dtype=torch.float64, device=dev) if problem == 1 else None
    dZ = torch.empty((n, n), dtype=torch.float64, device=dev)
    dw = torch.empty((n,), dtype=torch.float64, device=dev)
    stage = (ctypes.c_double * 8)()
    ssum = [0.0] * 8

    def regen():
        if lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n) != 0:
            raise RuntimeError("synth")
        if problem == 1 and lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n) != 0:
            raise RuntimeError("synth")

    tot = 0.0
    kp_s, kp_l = (ctypes.c_double * 4)(), (ctypes.c_longlong * 4)()
    for it in range(warmup + steps):
        regen()
        if it == warmup:
            lib.ek_hip_profile_kernels(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = lib.ek_hip_solve_device(problem, n, n_vec, dA.data_ptr(), n, dB.data_ptr() if problem == 1 else None, n,
                                       dw.data_ptr(), dZ.data_ptr(), n, stage, 8)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if info != 0:
            lib.ek_hip_profile_kernels(0)
            raise RuntimeError("ek_hip_solve_device info=%d" % info)
        if it >= warmup:
            tot += dt
            for q in range(8):
                ssum[q] += stage[q]
    lib.ek_hip_profile_kernels_get(kp_s, kp_l)
    lib.ek_hip_profile_kernels(0)
    sec = tot / steps
    res = {"config": {"workload": "synthetic (SURVEY 8(d)) N=%d %s, %s" % (
               n, "generalized EVP" if problem == 1 else "standard EVP",
               "full spectrum" if n_vec == n else "lowest %d eigenpairs" % n_vec), "n": n, "problem": prob, "n_vec": n_vec},
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * sec, "value": n_vec / sec, "unit": "eigenpairs/s",
           "tflops_equiv": flops(problem, n, n_vec) / sec / 1e12,
           "stage_seconds_per_step": {lib.ek_hip_stage_name(q).decode(): ssum[q] / steps for q in range(8)}}
    stats = (ctypes.c_double * 8)()
    lib.ek_hip_debug_last_solve_stats(stats, 8)
    if kp_l[0] > 0 and kp_s[0] > 0 and stats[1] > 0.5:      # (only a solve that stayed on the two-stage path has this kernel)
        dur = kp_s[0] / kp_l[0]
        fl = 2.0 * n * n * n_vec
        res["dominant_kernel"] = {"kernel": "q2_apply_nb_kernel", "avg_launch_us": 1e6 * dur, "achieved": fl / dur / 1e12,
                                  "unit": "TFLOP/s", "frac": fl / dur / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                  "chase_avg_launch_us": 1e6 * kp_s[1] / max(kp_l[1], 1)}
    w = dw.cpu().numpy()
    ok = bool((w[1:n_vec] >= w[:n_vec - 1]).all())
    regen()
    an, ave, mx, orth = (ctypes.c_double(0) for _ in range(4))
    rc1 = lib.ek_hip_residual_device(problem, n, n_vec, dA.data_ptr(), n, dB.data_ptr() if problem == 1 else None, n,
                                     dw.data_ptr(), dZ.data_ptr(), n, ctypes.byref(an), ctypes.byref(ave), ctypes.byref(mx))
    rc2 = lib.ek_hip_orthogonality_device(problem, n, 1, n_vec, dB.data_ptr() if problem == 1 else None, n,
                                          dZ.data_ptr(), n, ctypes.byref(orth))
    bound = 1e-14 * max(1.0, (n / 1024.0) ** 0.5)
    res["parity"] = {"residual_norm_max": mx.value, "orthogonality": orth.value,
                     "bounds": {"residual_norm_max": bound, "orthogonality": 1e-11},
                     "ok": bool(ok and rc1 == 0 and rc2 == 0 and mx.value <= bound and orth.value <= 1e-11)}
    return res


def gemm_ceiling(lib, torch, dev, dA, n):
    """What the library's own large GEMM sustains on this box right now: C(n/2, n/2) -= A B with K = n/2, operands taken from
    a device array that is spent at this point, C its own scratch (three repetitions, HIP events; ek_hip_debug_gemm_at) --
    the practical fp64 MFMA ceiling the dominant kernel's issued rate is compared with (the datasheet's 78.6 is the
    `peak` of the roofline object)."""
    try:
        h = n // 2
        dC = torch.zeros((h, h), dtype=torch.float64, device=dev)
        sec = ctypes.c_double(0)
        a = ctypes.c_void_p(dA.data_ptr())
        rc = lib.ek_hip_debug_gemm_at(0, 0, h, h, h, a, n, ctypes.c_void_p(dA.data_ptr() + 8 * h * n), n,
                                      1.0, ctypes.c_void_p(dC.data_ptr()), h, 0, 3, ctypes.byref(sec))
        del dC
        if rc != 0 or sec.value <= 0:
            return None
        return 2.0 * h * h * h / sec.value / 1e12
    except Exception:
        return None


def q2_traffic_record(n, ncols):
    """HBM bytes per q2_apply_nb_kernel launch from the committed PMC measurement -- only while the kernel's source is
    the file the measurement was taken from (the record carries its sha256); a stale record is not reported."""
    import hashlib
    src = os.path.join(ROOT, "eigenkernel_amd", "csrc", "ek_sb2st.hip")
    why = "no PMC record"
    for name in ("r06_q2_apply_traffic.json", "r05_q2_apply_traffic.json"):      # the newest record that is still valid
        tpath = os.path.join(ROOT, "profiles", name)
        try:
            tj = json.load(open(tpath))
            sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
            if tj.get("n") != n or tj.get("ncols") != int(ncols):
                why = "profiles/%s is for another shape" % name
                continue
            if tj.get("source_sha256") != sha:
                why = "profiles/%s is stale: ek_sb2st.hip has changed since it was measured" % name
                continue
            return tj.get("hbm_bytes_per_launch"), ("profiles/%s (rocprofv3 --pmc at git %s, not live)" % (name, tj.get("git", "?")))
        except Exception as exc:
            why = "no PMC record (%r)" % (exc,)
    return None, why


_cpu_child = None    # the CPU baseline's mpiexec (leader of its own session) while it runs


def _kill_cpu_child():
    """Ends the CPU baseline's process group (mpiexec and its ranks), if one is still there."""
    global _cpu_child
    child, _cpu_child = _cpu_child, None
    if child is None:
        return
    try:
        if child.poll() is None:
            os.killpg(child.pid, signal.SIGKILL)
        child.wait(timeout=10)
    except Exception:
        pass


_emitted = False
_pending = None      # the main JSON line as soon as it exists (the watchdog prints it if the probe hangs)
T_START = time.perf_counter()


def _on_term(signum, frame):
    """A run that is told to end (the driver's limit) still prints the line it has: the headline exists long before the extras."""
    _kill_cpu_child()            # (64 MPI ranks in a session of their own would otherwise outlive this run by minutes)
    if _pending is not None:
        _pending["ended_by_signal"] = int(signum)
        emit(_pending)
    sys.stdout.flush()
    os._exit(124)


def emit(out):
    """Prints THE one JSON line, once."""
    global _emitted
    if not _emitted:
        _emitted = True
        print(json.dumps(out), flush=True)


def attach_communicator(solver, dist, rank, world, rehearse):
    """The library's communicator: RCCL from an id made on rank 0 and carried by torch.distributed,
    or (one-GPU rehearsal) the host communicator over torch.distributed itself."""
    if rehearse:
        solver.set_allgatherv(solver.torch_allgatherv(dist))
        solver.comm_attach_host(world, rank)
        return
    uid = [solver.comm_unique_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(uid, src=0)
    solver.comm_init(uid[0], world, rank)


def grid_probe(args, lib, solver, dist, torch, dev, cdev, rank, world, n, problem, n_vec, dAs, dBs, dZ, dw, regenerate):
    """ONE problem on the 1 x world grid with the library's communicator attached (Cholesky factor and
    reduction from three ranks on, tridiagonalisation always: distributed; eigenvector stages
    column-sharded), measured to the same contract as the headline (W warm-up solves, then exactly K
    solves between barriers, max over ranks) in the library's one distributed form ("two_stage": dense -> band over the
    team, per panel one broadcast and one all-reduce, the next panel's chain and broadcast on a second stream beside the
    rest of the update).  Runs after the replicas region; a watchdog abandons it (os._exit after printing the line that
    exists by then) if an exchange never returns: a pool box has a single GPU, so this path could
    only be rehearsed there (team rehearsal, RCCL with one rank, processes sharing the GPU)."""
    import threading
    from eigenkernel_amd import descriptor as dsc
    NB = 64
    K, W = min(args.steps, 8), max(1, min(args.warmup, 2))      # a probe, not the headline: bounded whatever --steps says
    res = {"distribution": "1 x %d process grid, replicated inputs; PDPOTRF/PDSYGST (from 3 ranks on) and PDSYTRD "
                           "distributed over %s, eigenvector columns sharded"
                           % (world, "the HOST communicator (one-GPU rehearsal: timings mean nothing)"
                              if args.rehearse_on_one_gpu else "RCCL / xGMI"),
           "steps": K, "warmup": W, "modes": {}}
    done = threading.Event()

    def watchdog():
        if not done.wait(args.grid_probe_timeout):
            sys.stderr.write("[bench] grid probe abandoned after %.0f s on rank %d\n" % (args.grid_probe_timeout, rank))
            if rank == 0 and _pending is not None:
                _pending["grid_probe"] = dict(res, error="timeout")
                if args.distribution == "auto":      # a mode that had finished and passed still counts
                    try:
                        promote_grid_mode(_pending, res, world)
                    except Exception:
                        pass
                emit(_pending)
            sys.stdout.flush()
            os._exit(3)          # an abandoned exchange is a failure of the run, whatever line could be saved

    def sync():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    def allreduce(value, op):
        t = torch.tensor([value], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    my_cols = dsc.local_indices(n_vec, NB, rank, world)
    nc_loc = len(my_cols)
    stage = (ctypes.c_double * 8)()

    def solve(i):
        info = lib.ek_hip_solve_device_grid(problem, n, n_vec, dAs[i].data_ptr(), n,
                                            dBs[i].data_ptr() if dBs is not None else None, n,
                                            dw.data_ptr(), dZ.data_ptr(), n, NB, 1, world, 0, rank, stage, 8)
        if info != 0:
            raise RuntimeError("ek_hip_solve_device_grid info=%d" % info)

    def measure(mode):
        m = {}
        # "two_stage": the library's distributed form (dense -> band over the team's column strips: per panel one
        # broadcast of [V | T | tau] and one all-reduce of Y; one all-gather of the band; bulge chasing replicated;
        # back-transformations sharded by columns)
        for _ in range(W):
            regenerate(0)
            solve(0)
        for i in range(K):
            regenerate(i)
        stage_sum = [0.0] * 8
        if not args.no_symv_events:
            lib.ek_hip_profile_symv(max(1, args.symv_events_stride))
        sync()
        t0 = time.perf_counter()
        for i in range(K):
            solve(i)
            for q in range(8):
                stage_sum[q] += stage[q]
        sync()
        total = allreduce(time.perf_counter() - t0, dist.ReduceOp.MAX)
        sv_s, sv_l, sv_b = ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_double(0)
        if not args.no_symv_events:
            lib.ek_hip_profile_symv_get(ctypes.byref(sv_s), ctypes.byref(sv_l), ctypes.byref(sv_b))
            lib.ek_hip_profile_symv(0)
        m.update({"ms_per_step": 1e3 * total / K, "value": n_vec * K / total, "unit": "eigenpairs/s",
                  "tflops_equiv": flops(problem, n, n_vec) * K / total / 1e12, "scaling": "strong",
                  "stage_seconds_per_step_rank0": {lib.ek_hip_stage_name(q).decode(): stage_sum[q] / K for q in range(8)}})
        if sv_l.value > 0 and sv_s.value > 0:
            ach = sv_b.value / sv_s.value / 1e9
            m["roofline"] = {"kernel": "symv_kernel<DIST> on rank 0 (its 1/%d share of the lower triangle per column)" % world,
                             "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "traffic": None, "launches": sv_l.value,
                             "timed_every_kth_column": max(1, args.symv_events_stride),
                             "avg_launch_us": 1e6 * sv_s.value / sv_l.value,
                             "algorithmic_bytes_per_launch": sv_b.value / sv_l.value}
        # parity of every rank's eigenpairs (the reference's acceptance quantities, on the GPU)
        w = dw.cpu().numpy()
        ok = bool((w[1:n_vec] >= w[:n_vec - 1]).all())
        if nc_loc > 0:
            regenerate(0)
            an, ave, mx, orth = (ctypes.c_double(0) for _ in range(4))
            dwc = torch.from_numpy(np.ascontiguousarray(w[my_cols])).to(dev)
            rc1 = lib.ek_hip_residual_device(problem, n, nc_loc, dAs[0].data_ptr(), n,
                                             dBs[0].data_ptr() if dBs is not None else None, n, dwc.data_ptr(),
                                             dZ.data_ptr(), n, ctypes.byref(an), ctypes.byref(ave), ctypes.byref(mx))
            rc2 = lib.ek_hip_orthogonality_device(problem, n, 1, nc_loc, dBs[0].data_ptr() if dBs is not None else None,
                                                  n, dZ.data_ptr(), n, ctypes.byref(orth))
            bound = 1e-14 * max(1.0, (n / 1024.0) ** 0.5)
            ok = ok and rc1 == 0 and rc2 == 0 and mx.value <= bound and orth.value <= 1e-11
            m["parity_rank0"] = {"A_norm": an.value, "residual_norm_average": ave.value, "residual_norm_max": mx.value,
                                 "orthogonality": orth.value,
                                 "bounds": {"residual_norm_max": bound, "orthogonality": 1e-11}}
        m["parity_ok_all_ranks"] = allreduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5
        wt = torch.from_numpy(w.copy()).to(cdev)
        wmax = wt.clone(); wmin = wt.clone()
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX); dist.all_reduce(wmin, op=dist.ReduceOp.MIN)
        m["eigenvalues_identical_on_all_ranks"] = bool(torch.equal(wmax, wmin))
        return m, w.copy()

    try:
        threading.Thread(target=watchdog, daemon=True).start()
        attach_communicator(solver, dist, rank, world, args.rehearse_on_one_gpu)
        for mode in ("two_stage",):
            try:
                m, w = measure(mode)
                res["modes"][mode] = m
            except Exception as exc:          # ranks fail alike (collective calls): record and go on
                res["modes"][mode] = {"error": repr(exc)}
        solver.comm_destroy()
    except Exception as exc:   # the probe never takes the line down
        res["error"] = repr(exc)
    done.set()
    return res


def promote_grid_mode(out, probe, world):
    """--distribution auto: the faster distributed mode that passed parity on every rank becomes the
    headline; the replicas measurement stays in the line as "replicas"."""
    good = {k: m for k, m in probe.get("modes", {}).items()
            if "error" not in m and m.get("parity_ok_all_ranks") and m.get("eigenvalues_identical_on_all_ranks")}
    if not good:
        return
    mode = max(good, key=lambda k: good[k]["value"])
    m = good[mode]
    out["replicas"] = {k: out[k] for k in ("value", "unit", "ms_per_step", "scaling", "tflops_equiv",
                                           "stage_seconds_per_step", "parity", "roofline") if k in out}
    out["replicas"]["config_parallelism"] = out["config"]["parallelism"]
    out["value"] = m["value"]
    out["ms_per_step"] = m["ms_per_step"]
    out["scaling"] = "strong"
    out["tflops_equiv"] = m["tflops_equiv"]
    out["stage_seconds_per_step"] = m["stage_seconds_per_step_rank0"]
    out["parity"] = m.get("parity_rank0")
    out["roofline"] = m.get("roofline")
    out["config"]["workload"] = out["config"]["workload"].replace("1 problem per GPU", "ONE problem on all GPUs")
    out["config"]["parallelism"] = ("1 x %d process grid over RCCL/xGMI, replicated inputs: Cholesky factor and reduction "
                                    "distributed (1 x P block-cyclic, 128-wide blocks), tridiagonalisation: %s, eigenvector "
                                    "stages sharded by columns" % (world, mode))
    out["headline_mode"] = mode


def self_launch(n_ranks):
    """`python bench.py --gpus N` with no launcher around it: start N ranks (one per GPU) with
    torch.distributed.run as FRESH child processes -- this parent has made no GPU call and makes none,
    and nothing is re-executed in place -- pass the ranks' output through and return their exit code.
    Under an existing launcher (RANK in the environment) this function is not reached."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n_ranks, " ".join(cmd)))
    sys.stderr.flush()
    # stdout and stderr are inherited: rank 0's JSON line is this run's line.  cwd = the repo root: `python -m`
    # puts the working directory first on sys.path, and a stray module there must not shadow the launcher's imports
    return subprocess.call(cmd, env=env, cwd=ROOT)


def dry_launch(rank, world):
    """The launch path without a GPU: every rank joins a gloo group, the ranks are counted by an
    all-reduce, rank 0 prints a line whose n_gpus is that count."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    one = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(one)
    assert int(one) == world == dist.get_world_size()
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry launch (no measurement)", "value": None, "n_gpus": int(one),
                          "dry_launch": True}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default=None,
                    help="a BASELINE.json configuration by name: c2 N=4096 standard, c3 N=16384 generalized "
                         "(default), c4 N=32768 generalized, c5 N=16384 generalized lowest 1024 pairs")
    ap.add_argument("--order", dest="n", type=int, default=16384, help="matrix order N")
    ap.add_argument("--problem", choices=["gep", "sep"], default="gep")
    ap.add_argument("--n-vec", type=int, default=0,
                    help="lowest n_vec eigenpairs only (the *_select arms, BASELINE.json configs[4]); 0 = all")
    ap.add_argument("--cpu-sample-n", type=int, default=1536,
                    help="order of the CPU-oracle sample (scalar C port, 1 core)")
    ap.add_argument("--scalapack-sample-n", type=int, default=8192,
                    help="order of the ScaLAPACK-path sample (all physical cores)")
    ap.add_argument("--distribution", choices=["auto", "replicas", "columns", "grid"], default="auto",
                    help="N>1 GPUs: 'auto' (default) = measure replicas first (one independent problem per rank: "
                         "the safe line), then ONE problem distributed over the 1 x N grid "
                         "(the 'grid_probe'), and report it, if it passed the parity check on every rank, as the "
                         "headline (\"scaling\": \"strong\"), the replicas numbers beside it; if it did not, the "
                         "replicas line is the headline.  "
                         "'replicas' = one independent problem per rank (weak scaling) as the headline; "
                         "'columns' = ONE problem on a 1 x N process grid in replicated-input mode "
                         "(ek_hip_solve_device_grid: reduction replicated, eigenvector columns sharded; strong scaling); "
                         "'grid' = the same with the library's RCCL communicator attached: the dense -> band stage of the "
                         "tridiagonalisation is distributed over the N ranks as well (per panel one broadcast and one all-reduce)")
    ap.add_argument("--virtual-grid", type=int, default=0,
                    help="with --distribution columns on ONE GPU: play rank --virtual-rank of a 1 x P grid "
                         "(the mode has no collective, so a rank's time does not depend on the others)")
    ap.add_argument("--virtual-rank", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the ScaLAPACK baseline (mandatory under a profiler: 64 MPI ranks are not to be profiled)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive step (value_incl_copies)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other BASELINE configurations (c2, c5, c4 at full size, a few steps each) that the "
                         "default single-GPU run appends as \"other_configs\"")
    ap.add_argument("--no-symv-events", action="store_true")
    ap.add_argument("--symv-events-stride", type=int, default=8,
                    help="time the symv launch of every k-th column with HIP events (1 = all launches; "
                         "a pair of event records costs ~5 us of host time, 90 ms per solve at k = 1)")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-grid-probe", action="store_true",
                    help="N>1 with --distribution replicas: skip the extra measurement of ONE problem on the "
                         "1 x N grid with the tridiagonalisation distributed over RCCL (reported as \"grid_probe\")")
    ap.add_argument("--force-grid-probe", action="store_true",
                    help="run the grid probe at world size 1 too (rehearsal of the code path on a one-GPU box; "
                         "needs a torch.distributed launch)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 ranks that all use GPU 0 (a pool box has one): gloo process group, collective "
                         "tensors on the CPU, the library's HOST communicator instead of RCCL (which refuses two "
                         "ranks on one device).  Exercises the whole N>1 control flow; the timings mean nothing")
    ap.add_argument("--grid-probe-timeout", type=float, default=240.0,
                    help="seconds after which a stuck grid probe is abandoned (the main line is still printed)")
    ap.add_argument("--budget-seconds", type=int, default=480,
                    help="wall-clock budget of the whole run: the CPU baseline at the headline order (minutes on 64 cores) is "
                         "only started if its projection fits what is left")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, join them in a gloo group, count them and print a line with n_gpus = the "
                         "number of ranks that answered -- no GPU call anywhere (the CPU test of the self-launch)")
    args = ap.parse_args()
    if args.config:
        args.n, args.problem, args.n_vec = CONFIGS[args.config]

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process makes no GPU call (it has not even imported torch);
        # it starts the N ranks as fresh children and leaves with their exit code
        sys.exit(self_launch(args.gpus))
    import signal
    signal.signal(signal.SIGTERM, _on_term)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks: reporting n_gpus = %d\n"
                         % (args.gpus, world, world))
    if args.dry_launch:
        return dry_launch(rank, world)
    n, problem = args.n, (1 if args.problem == "gep" else 0)
    n_vec = args.n_vec if 0 < args.n_vec < n else n

    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the product path has no CPU fallback")
    rehearse = args.rehearse_on_one_gpu
    if world > 1 and not rehearse and torch.cuda.device_count() < world:
        raise RuntimeError("bench.py: %d ranks but %d GPUs visible (one rank per GPU; --rehearse-on-one-gpu shares GPU 0 "
                           "through the host communicator)" % (world, torch.cuda.device_count()))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:   # launched by torch.distributed.run
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world

    from eigenkernel_amd import solver
    lib = solver.load_library()
    rc = lib.ek_hip_init(local_rank)
    if rc != 0:
        raise RuntimeError("ek_hip_init(%d) failed: %d" % (local_rank, rc))

    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev      # where the collectives' tensors live
    # Device-resident operands (torch only owns the memory; the library does all the work).
    # The solve overwrites A and B (as PDSYTRD / PDPOTRF do), so every timed step gets its own
    # pre-generated copy of the inputs: 288 GB of HBM holds them easily and the timed region
    # stays one contiguous run of K solves.
    K = args.steps
    dAs = [torch.empty((n, n), dtype=torch.float64, device=dev) for _ in range(K)]
    dBs = [torch.empty((n, n), dtype=torch.float64, device=dev) for _ in range(K)] if problem == 1 else None
    columns = args.distribution in ("columns", "grid")     # "auto" and "replicas": one problem per rank first
    if args.distribution == "grid":
        attach_communicator(solver, dist, rank, world, rehearse)
    NB = 64                                        # g_block_size (global_variables.f90:5)
    if columns:
        from eigenkernel_amd import descriptor as dsc
        npcol, mycol = (args.virtual_grid, args.virtual_rank) if (world == 1 and args.virtual_grid > 0) else (world, rank)
        my_cols = dsc.local_indices(n_vec, NB, mycol, npcol)
        nc_loc = len(my_cols)
    else:
        npcol, mycol, nc_loc = 1, 0, n_vec
    dZ = torch.empty((max(nc_loc, 1), n) if columns else (n, n), dtype=torch.float64, device=dev)  # column-major n x nc
    dw = torch.empty((n,), dtype=torch.float64, device=dev)
    stage = (ctypes.c_double * 8)()
    stage_sum = [0.0] * 8

    def regenerate(i):
        assert lib.ek_hip_synth_matrix_device(n, 1, dAs[i].data_ptr(), n) == 0
        if problem == 1:
            assert lib.ek_hip_synth_matrix_device(n, 2, dBs[i].data_ptr(), n) == 0

    def step(i, collect):
        if columns:
            info = lib.ek_hip_solve_device_grid(problem, n, n_vec, dAs[i].data_ptr(), n,
                                                dBs[i].data_ptr() if problem == 1 else None, n,
                                                dw.data_ptr(), dZ.data_ptr(), n, NB, 1, npcol, 0, mycol, stage, 8)
        else:
            info = lib.ek_hip_solve_device(problem, n, n_vec, dAs[i].data_ptr(), n,
                                           dBs[i].data_ptr() if problem == 1 else None, n,
                                           dw.data_ptr(), dZ.data_ptr(), n, stage, 8)
        if info != 0:
            raise RuntimeError("ek_hip_solve_device info=%d" % info)
        if collect:
            for q in range(8):
                stage_sum[q] += stage[q]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        regenerate(0)
        step(0, False)
    for i in range(K):
        regenerate(i)
    events = not args.no_symv_events
    if events:
        lib.ek_hip_profile_symv(max(1, args.symv_events_stride))
        lib.ek_hip_profile_kernels(1)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(i, True)
    barrier()
    total = time.perf_counter() - t0
    # whole-job figure: every rank's eigenpairs over the slowest rank's time (eigenkernel_amd/parallel.py);
    # ONE problem on a grid counts its eigenpairs once
    from eigenkernel_amd.parallel import aggregate_throughput
    units_local = float(n_vec * K) if not columns else (float(n_vec * K) if rank == 0 else 0.0)
    value_all, total = aggregate_throughput(units_local, total, dist, cdev)

    symv_s, symv_l, symv_b = ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_double(0)
    kp_s, kp_l = (ctypes.c_double * 4)(), (ctypes.c_longlong * 4)()
    if events:
        lib.ek_hip_profile_symv_get(ctypes.byref(symv_s), ctypes.byref(symv_l), ctypes.byref(symv_b))
        lib.ek_hip_profile_symv(0)
        lib.ek_hip_profile_kernels_get(kp_s, kp_l)
        lib.ek_hip_profile_kernels(0)

    # Parity guard on the output of the LAST timed step, at the full benchmark size, through the
    # reference's own acceptance quantities (verifier.f90:75-204, 233-330) evaluated on the GPU
    # against freshly regenerated inputs (the solve destroyed its copies): outside the timed region.
    w = dw.cpu().numpy()
    assert (w[1:n_vec] >= w[:n_vec - 1]).all() and abs(w[:n_vec]).max() < 1e6
    parity = None
    if not args.no_parity_check and nc_loc > 0:
        regenerate(0)
        an, ave, mx, orth = (ctypes.c_double(0) for _ in range(4))
        dwc = dw
        if columns:      # this rank's eigenpairs: its block-cyclic columns and their eigenvalues
            dwc = torch.from_numpy(np.ascontiguousarray(w[my_cols])).to(dev)
        rc = lib.ek_hip_residual_device(problem, n, nc_loc, dAs[0].data_ptr(), n,
                                        dBs[0].data_ptr() if problem == 1 else None, n, dwc.data_ptr(),
                                        dZ.data_ptr(), n, ctypes.byref(an), ctypes.byref(ave), ctypes.byref(mx))
        assert rc == 0, rc
        rc = lib.ek_hip_orthogonality_device(problem, n, 1, nc_loc, dBs[0].data_ptr() if problem == 1 else None, n,
                                             dZ.data_ptr(), n, ctypes.byref(orth))
        assert rc == 0, rc
        parity = {"A_norm": an.value, "residual_norm_average": ave.value, "residual_norm_max": mx.value,
                  "orthogonality": orth.value,
                  "bounds": {"residual_norm_max": 1e-14 * max(1.0, (n / 1024.0) ** 0.5), "orthogonality": 1e-11}}
        assert mx.value <= parity["bounds"]["residual_norm_max"], parity
        assert orth.value <= parity["bounds"]["orthogonality"], parity

    global _pending
    out = None
    if rank == 0:
        value = value_all
        out = {
            "metric": ("eigenpairs/s (full spectrum) + achieved fp64 TFLOP/s vs roofline, N=16384 GEP"
                       if (n == 16384 and problem == 1 and n_vec == n) else
                       ("eigenpairs/s (full spectrum)" if n_vec == n else "eigenpairs/s (lowest n_vec)")),
            "value": value, "unit": "eigenpairs/s",
            # the measurement contract of this task: inputs resident in HBM when the timed region starts; SURVEY.md 8(d)'s
            # t_solve (host arrays in and out through ek_hip_solve, PCIe copies included) is "value_incl_copies" below
            "value_device_resident": value,
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * total / K, "higher_is_better": True,
            "scaling": "strong" if columns else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic dense SPD pair (SURVEY 8(d), seeds 1,2) N=%d %s, "
                                   "full spectrum, 1 problem per GPU" % (n, "generalized EVP (Cholesky+reduce+SEP)"
                                                                         if problem == 1 else "standard EVP"),
                       "n": n, "problem": args.problem, "n_vec": n_vec,
                       "parallelism": ("1 x %d process grid, replicated inputs, %seigenvector columns sharded "
                                       "(rank %d%s)" % (npcol, "tridiagonalisation distributed over RCCL, "
                                                        if args.distribution == "grid" else "",
                                                        mycol, ", played on one GPU" if world == 1 else "")
                                       if columns else "replicas x%d" % world if world > 1 else "1 GPU")},
            "tflops_equiv": (1 if columns else world) * flops(problem, n, n_vec) * K / total / 1e12,
            "fp64_mfma_peak_tflops": FP64_MFMA_PEAK_TFLOPS,
            "stage_seconds_per_step": {lib.ek_hip_stage_name(i).decode(): stage_sum[i] / K for i in range(8)},
            "parity": parity,
        }
        n3 = float(n) ** 3
        k = float(nc_loc if columns else n_vec)
        if events and kp_l[0] > 0 and kp_s[0] > 0:
            # two-stage path: the application of the bulge-chasing reflectors Q2 to the eigenvectors is
            # the largest kernel.  Algorithmic flops: n^2 / (2*64) reflectors of length 64 applied to k
            # columns at 4 flops per entry = 2 n^2 k per launch (the compact-WY form the kernel executes
            # issues ~1.3x that on the matrix cores; that surplus is not counted).
            fl = 2.0 * n * n * k
            dur = kp_s[0] / kp_l[0]
            ach = fl / dur / 1e12
            traffic, tsrc = q2_traffic_record(n, k)
            out["roofline"] = {
                "kernel": "q2_apply_nb_kernel<%d> (Z <- Q2 Z: reflectors of the band->tridiagonal stage, compact-WY blocks of 32 "
                          "sweeps applied %s blocks per pass, window of Z resident in MFMA accumulator registers)"
                          % ((4, "four") if n >= 8192 else (3, "three")),
                "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                "launches": int(kp_l[0]), "avg_launch_us": 1e6 * dur, "algorithmic_flops_per_launch": fl,
                "measured_mfma_ceiling_tflops": gemm_ceiling(lib, torch, dev, dAs[0], n),   # live, see gemm_ceiling()
                "other_kernels": {
                    "chase_pos_kernel (band -> tridiagonal: positions of the band in registers, sweeps pass through by mail)":
                        {"launches": int(kp_l[1]), "avg_launch_us": 1e6 * kp_s[1] / max(kp_l[1], 1)},
                    "symm_lower_kernel (Y = A22 V of every 8th panel)":
                        {"launches": int(kp_l[2]), "avg_launch_us": 1e6 * kp_s[2] / max(kp_l[2], 1)}},
            }
        elif events and symv_l.value > 0 and symv_s.value > 0:
            ach = symv_b.value / symv_s.value / 1e9
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, "profiles", "symv_traffic.json")
            if os.path.exists(tpath):   # PMC pass (rocprofv3 --pmc FETCH_SIZE, corrected) if committed
                try:
                    tj = json.load(open(tpath))
                    if tj.get("n") == n:
                        traffic, tsrc = tj.get("hbm_bytes_per_launch"), "profiles/symv_traffic.json (rocprofv3 --pmc, round 1, not live)"
                except Exception:
                    traffic = None
            out["roofline"] = {
                "kernel": "symv_kernel (tridiagonalisation panel: y = A22 v, lower triangle read once)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                "launches": symv_l.value, "timed_every_kth_column": max(1, args.symv_events_stride),
                "avg_launch_us": 1e6 * symv_s.value / symv_l.value,
                "algorithmic_bytes_per_launch": symv_b.value / symv_l.value,
            }
        else:
            out["roofline"] = None
        # per stage: algorithmic flops of SURVEY.md 8(d) (K1 n^3/3, K2 n^3, K3 4n^3/3, K5 4n^3/3 NOMINAL, K6 2 n^2 k,
        # K7 n^2 k) over the stage's device seconds, against the fp64 matrix peak -- and, where the path executes a
        # different count, that count beside it: the divide & conquer runs what deflation leaves (counted on the
        # device from the merge products' dimensions; the nominal figure is an upper bound, so ITS fraction is the
        # executed one), the two-stage back-transformation applies two sets of reflectors (Q2 and Q1: 4 n^2 k).
        stats = (ctypes.c_double * 8)()
        lib.ek_hip_debug_last_solve_stats(stats, 8)
        dc_exec, two_stage, band_shortcut = float(stats[0]), stats[1] > 0.5, stats[3] > 0.5
        st_fl = {"reduce_generalized:pdpotrf": (n3 / 3 if problem == 1 else 0.0, None),
                 "reduce_generalized:pdsygst": (n3 if problem == 1 else 0.0, None),
                 "eigen_solver_scalapack_all:pdsytrd": (4 * n3 / 3, None),
                 "eigen_solver_scalapack_all:pdstedc": (4 * n3 / 3 if k == n else 2 * n3 / 3 + 2 * n * n * k / 3,
                                                        dc_exec if dc_exec > 0 else None),
                 # (two-stage: Q2 and Q1, 4 n^2 k; a band on entry skips the first stage, Q1 = I: 2 n^2 k as counted)
                 "eigen_solver_scalapack_all:pdormtr": (2.0 * n * n * k, 4.0 * n * n * k if (two_stage and not band_shortcut) else None),
                 "recovery_generalized": (n * n * k if problem == 1 else 0.0, None)}
        out["roofline_stages"] = {}
        fl_exec_total = 0.0
        for name, (fl, fx) in st_fl.items():
            sec = out["stage_seconds_per_step"].get(name, 0.0)
            fl_exec_total += fx if fx is not None else fl
            if fl > 0 and sec > 0:
                e = {"algorithmic_flops": fl, "seconds": sec, "tflops": fl / sec / 1e12}
                if fx is not None:
                    e["executed_flops"] = fx
                    e["tflops_executed"] = fx / sec / 1e12
                if name.endswith("pdstedc") and fx is not None:
                    e["algorithmic_flops_note"] = "LAPACK's nominal count without deflation: an upper bound, not priced"
                    e["frac_of_fp64_mfma_peak"] = fx / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS
                else:
                    e["frac_of_fp64_mfma_peak"] = fl / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS
                out["roofline_stages"][name] = e
        out["tflops_executed"] = (1 if columns else world) * fl_exec_total * K / total / 1e12
        out["tflops_equiv_note"] = ("tflops_equiv prices the solve with SURVEY.md 8(d)'s nominal F(N) (BASELINE.json's metric); "
                                    "tflops_executed with what the path ran: D&C after deflation, both back-transformations of "
                                    "the two-stage form")
        _pending = out      # the headline exists from here on: none of the extras below can take it down
        if world == 1 and not columns and not args.no_host_path:
            try:
                here = host_path_step(lib, solver, problem, n, n_vec)
                child = host_path_child(problem, n, n_vec)
                if "error" in child:           # the figure of this process stands alone
                    here["child_process_error"] = child["error"]
                    here["process"] = "this process (PyTorch's bundled HIP runtime)"
                    out["value_incl_copies"] = here
                else:
                    child["process"] = ("a child process that loads the library alone: the system's HIP runtime, as a host of "
                                        "the reference's shape links it")
                    child["in_this_process_with_pytorchs_bundled_hip_runtime"] = {
                        k: here.get(k) for k in ("value", "seconds", "first_call_seconds", "host_device_copies_seconds", "error") if k in here}
                    out["value_incl_copies"] = child
            except Exception as exc:
                out["value_incl_copies"] = {"error": repr(exc)}
        if world == 1 and not columns and not args.no_other_configs and (n, args.problem, n_vec) == (16384, "gep", 16384):
            # every other BASELINE configuration at full size on this GPU, driver-timed in the same run
            del dAs[:]
            if dBs is not None:
                del dBs[:]
            torch.cuda.empty_cache()
            out["other_configs"] = {}
            for name, st, wu in (("c2", 10, 2), ("c5", 5, 1), ("c4", 2, 1)):
                try:
                    out["other_configs"][name] = run_other_config(lib, torch, dev, name, st, wu)
                except Exception as exc:
                    out["other_configs"][name] = {"error": repr(exc)}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            try:
                base = cpu_baseline_scalapack(problem, args.scalapack_sample_n)
                out["cpu_baseline"] = base if base is not None else cpu_baseline(problem, args.cpu_sample_n)
                sn = args.scalapack_sample_n if base is not None else args.cpu_sample_n
                try:
                    out["cpu_baseline"]["gpu_same_order"] = gpu_step_at(lib, torch, dev, problem, sn)
                except Exception as exc:
                    out["cpu_baseline"]["gpu_same_order"] = {"error": repr(exc)}
                # The same call sequence at the HEADLINE order on this box's cores, when it fits what is left of this run's
                # budget (--budget-seconds; the work grows with N^3: 8.5 x the sample's time is the projection): then THAT
                # is the baseline of the line and the sample stays beside it.  Otherwise the sample stands and the line
                # says so; the committed anchors (this round's, measured on a GPU box's 64 cores; round 4's, in the
                # 8-core build container) are attached either way.
                if base is not None and (n, problem, n_vec) == (16384, 1, 16384) and sn < n:
                    left = args.budget_seconds - (time.perf_counter() - T_START)
                    proj = 8.5 * base["seconds"] * (float(n) / sn / 2.0) ** 3 + 15.0
                    if proj < left - 20.0:
                        full = cpu_baseline_scalapack(problem, n, limit=left - 15.0)
                        if full is not None:
                            full["gpu_same_order"] = {"seconds": total / K, "value": n_vec * K / total,
                                                      "note": "the headline measurement of this line"}
                            full["sample_at_half_the_order"] = out["cpu_baseline"]
                            out["cpu_baseline"] = full
                        else:
                            out["cpu_baseline"]["headline_order"] = "attempted, failed or overran what was left of the budget"
                    else:
                        out["cpu_baseline"]["headline_order"] = ("not run: projected %.0f s, %.0f s left of --budget-seconds %d"
                                                                 % (proj, left, args.budget_seconds))
                # BASELINE.md 3 asks for N = 4096 ... 32768: the small order is run as well (seconds), the largest is a
                # LABELLED PROJECTION (the call sequence is O(N^3); ~30 min on 64 cores fits no budget of this run)
                try:
                    if base is not None and (n, problem, n_vec) == (16384, 1, 16384):
                        cb = out["cpu_baseline"]
                        orders = {}
                        small = cpu_baseline_scalapack(problem, 4096, limit=120.0)
                        if small is not None:
                            try:
                                small["gpu_same_order"] = gpu_step_at(lib, torch, dev, problem, 4096)
                            except Exception as exc:
                                small["gpu_same_order"] = {"error": repr(exc)}
                            orders["4096"] = {k: small[k] for k in ("value", "seconds", "cores", "stages", "gpu_same_order") if k in small}
                        half = cb.get("sample_at_half_the_order", cb if sn == 8192 else None)
                        if half is not None:
                            orders["8192"] = {k: half[k] for k in ("value", "seconds", "cores", "stages", "gpu_same_order") if k in half}
                        measured_full = "sample_at_half_the_order" in cb
                        if measured_full:
                            orders["16384"] = {k: cb[k] for k in ("value", "seconds", "cores", "stages", "gpu_same_order") if k in cb}
                        src_n, src_s = (16384, cb["seconds"]) if measured_full else (sn, base["seconds"])
                        proj_s = src_s * (32768.0 / src_n) ** 3
                        orders["32768"] = {"projection": True, "seconds": proj_s, "value": 32768.0 / proj_s, "cores": base["cores"],
                                           "how": "NOT measured: the N=%d time of this run x (32768/%d)^3 (the call sequence is O(N^3))"
                                                  % (src_n, src_n)}
                        c4 = (out.get("other_configs") or {}).get("c4") or {}
                        if "ms_per_step" in c4:
                            orders["32768"]["gpu_same_order"] = {"seconds": 1e-3 * c4["ms_per_step"],
                                                                 "note": "C4 on one GPU, other_configs of this line"}
                        cb["orders"] = orders
                        cb["what_it_is"] = ("the builder's C driver (oracle/scalapack_path.c) making the reference's six ScaLAPACK "
                                            "calls in the reference's order on oneMKL -- kind 'port', not the reference's Fortran "
                                            "binary; a stated baseline, never the target")
                        # (vs_baseline stays null: BASELINE.json publishes no number for this metric; the same-box ratio is here)
                        out["vs_cpu_baseline"] = {"ratio": out["value"] / cb["value"], "at_order": src_n if measured_full else sn,
                                                  "baseline": "cpu_baseline (kind 'port', %d cores)" % cb["cores"],
                                                  "note": ("GPU headline eigenpairs/s / CPU eigenpairs/s at the same order"
                                                           if measured_full else
                                                           "the CPU figure is the N=%d sample (eigenpairs/s fall with N): not like for like" % sn)}
                except Exception as exc:
                    out["cpu_baseline"]["orders_error"] = repr(exc)
                try:
                    if (n, problem, n_vec) == (16384, 1, 16384):
                        anchors = {}
                        for tag, fn in (("gpu_box_64_cores_round6", "r06_cpu_anchor_n16384_np64.json"),
                                        ("gpu_box_64_cores_round5", "r05_cpu_anchor_n16384_np64.json"),
                                        ("build_container_8_cores_round4", "r04_cpu_anchor_n16384_np8.json")):
                            fp = os.path.join(ROOT, "profiles", fn)
                            if os.path.exists(fp):
                                anchors[tag] = json.load(open(fp))
                        out["cpu_baseline"]["anchor_headline_order"] = anchors
                except Exception:
                    pass
            except Exception as exc:
                out["cpu_baseline"] = {"error": repr(exc)}
        _pending = out
    if ((world > 1 or args.force_grid_probe) and dist is not None and args.distribution in ("auto", "replicas")
            and not args.no_grid_probe):
        probe = grid_probe(args, lib, solver, dist, torch, dev, cdev, rank, world, n, problem, n_vec,
                           dAs, dBs, dZ, dw, regenerate)
        if out is not None:
            out["grid_probe"] = probe
            if args.distribution == "auto":
                promote_grid_mode(out, probe, world)
    if out is not None:
        emit(out)
    if world > 1:
        # the line is out: no closing barrier (a rank whose probe failed differently from its
        # peers' must not wait for them), no tear-down of the GPU runtime and the process-group threads
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    if args.distribution == "grid":
        solver.comm_destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
    # multi-rank runs: the line is out; leave without the interpreter's tear-down of the GPU
    # runtime and the process-group threads, so that nothing after the measurement can hold a rank
    # up (a single rank exits normally: a profiler attached to it writes its output at exit)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
