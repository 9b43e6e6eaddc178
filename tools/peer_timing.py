#!/usr/bin/env python3
"""Cost of the peer-window exchange, as far as ONE GPU can show it (tools, not product).

  python tools/peer_timing.py N P

P processes share GPU 0, each is one rank of the distributed tridiagonalisation of order N
(ek_hip_debug_sytrd_team(n, 0, ...)): handles travel over gloo (host communicator), the per-column
exchange goes through peer windows.  The ranks' kernels time-share the one GPU, so the wall time
is to be compared with the rehearsal of the same team inside one process (tools/team_timing.py N P,
"total"): what the P-process run takes on top of it is what P exchanges per column cost here.
"""
import ctypes
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, n, peer, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigenkernel_amd import solver as sv
    lib = sv.load_library()
    assert lib.ek_hip_init(0) == 0
    sv.set_allgatherv(sv.torch_allgatherv(dist))
    sv.comm_attach_host(world, rank)
    if peer:
        assert lib.ek_hip_comm_peer_enable(n) == 0
    sec = ctypes.c_double(0)
    assert lib.ek_hip_debug_sytrd_team(n, 0, 1, ctypes.byref(sec)) == 0
    dist.barrier()
    t0 = time.perf_counter()
    assert lib.ek_hip_debug_sytrd_team(n, 0, 2, ctypes.byref(sec)) == 0
    dist.barrier()
    wall = (time.perf_counter() - t0) / 2
    q.put((rank, sec.value, wall))
    sv.comm_destroy()
    dist.barrier()
    q.close(); q.join_thread()
    os._exit(0)


if __name__ == "__main__":
    n, world = int(sys.argv[1]), int(sys.argv[2])
    modes = [int(x) for x in sys.argv[3:]] or [1]
    for peer in modes:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 39000 + os.getpid() % 1000 + peer
        procs = [ctx.Process(target=worker, args=(r, world, port, n, peer, q), daemon=True) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(30)
        print("n=%d, %d processes on one GPU, %s: per rank device time %s s, wall %.4f s"
              % (n, world, "peer windows" if peer else "host-hook exchange",
                 ["%.4f" % r[1] for r in res], max(r[2] for r in res)), flush=True)
