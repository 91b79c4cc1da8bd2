"""Latency of the collectives of a Krylov iteration -- one plane exchange, one scalar all-reduce -- per transport, on
ONE GPU: RCCL in a one-rank world (its ring neighbours are the rank itself: the launch + kernel cost of a grouped
ncclSend/ncclRecv and of ncclAllReduce, no wire), and the peer transport between TWO PROCESSES sharing the GPU in its
host-ordered and device-ordered flavours (csrc/halo.hip).  Wall time per call over back-to-back calls.

    python tools/comm_latency.py [doubles per message ...]          (default: 8 1024 131072 1048576)
    python tools/comm_latency.py --two-gpus [doubles ...]           two processes on GPU 0 and GPU 1 of a multi-GPU node: RCCL
                                                                    send/recv + all-reduce over xGMI next to both peer flavours
                                                                    (the numbers DESIGN.md 5's budget could only estimate)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 200


def worker():
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    rank, uid, counts = int(sys.argv[2]), bytes.fromhex(sys.argv[3]), [int(a) for a in sys.argv[4:]]
    dev = rank if os.environ.get("PIB_LATENCY_TWO_GPUS") == "1" else 0
    s = LinSolverHIP("poisson", config_text="solver(s)=PCG\ns:preconditioner(p)=NOSOLVER\n", rank=rank, nranks=2, uid=uid, device=dev)
    us = (ctypes.c_double * 2)()
    for c in counts:
        capi.check(capi.load().pib_comm_latency(s._h, c, REPS, us))
        if rank == 0:
            print(f"RESULT {c} {us[0]:.2f} {us[1]:.2f}", flush=True)
    s.destroy()


def main():
    from petibm_amd import capi
    two = "--two-gpus" in sys.argv
    counts = [int(a) for a in sys.argv[1:] if a != "--two-gpus"] or [8, 1024, 131072, 1048576]
    lib = capi.load()
    rows = {}
    us = (ctypes.c_double * 2)()
    for c in counts:
        capi.check(lib.pib_comm_latency(None, c, REPS, us))
        rows.setdefault(c, {})["rccl (1 rank, self ring)"] = (us[0], us[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PIB_PEER_TIMEOUT_S="120", PIB_LATENCY_TWO_GPUS="1" if two else "0")
    where = "2 processes, GPU 0 and GPU 1" if two else "2 processes"
    flavours = [(f"peer, host-ordered ({where})", 0), (f"peer, device-ordered ({where})", 1)]
    if two:
        flavours.insert(0, (f"rccl ({where})", -1))
    for label, dev in flavours:
        uid = ctypes.create_string_buffer(capi.UID_BYTES)
        capi.check(lib.pib_comm_unique_id(uid) if dev < 0 else lib.pib_comm_peer_id_ordered(uid, dev))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), uid.raw.hex()] + [str(c) for c in counts],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        for ln in outs[0].splitlines():
            if ln.startswith("RESULT"):
                _, c, a, b = ln.split()
                rows[int(c)][label] = (float(a), float(b))
    print(f"# wall time per call, {REPS} back-to-back calls, " + ("two MI355X (one process each)" if two else "one MI355X (the two peer processes share it)"))
    print(f"{'transport':38s} " + " ".join(f"{'exch ' + str(8 * c) + ' B':>16s}" for c in counts) + f" {'all-reduce 64 B':>16s}")
    for label in rows[counts[0]]:
        print(f"{label:38s} " + " ".join(f"{rows[c][label][0]:13.1f} us" for c in counts) + f" {rows[counts[0]][label][1]:13.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker()
    else:
        main()
