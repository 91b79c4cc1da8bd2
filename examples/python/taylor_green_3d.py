#!/usr/bin/env python3
"""The reference's examples/navierstokes/taylorgreenvortex3dRe1600_GPU on one MI355X: runs the case directory
(examples/cases/taylorgreenvortex3dRe1600: 256^3 cells, 2000 steps of dt = 0.01), samples the mean kinetic energy every
`--every` steps (the quantity of the reference's mean-kinetic-energy.png) and prints it next to the pseudo-spectral
512^3 data the reference ships (tests/golden/reference_test_vectors.json).

    python examples/python/taylor_green_3d.py [--cells 256] [--nt 2000] [--every 100]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from petibm_amd import navierstokes  # noqa: E402


def mean_kinetic_energy(s, dist=None, dev="cpu"):
    """uniform periodic mesh: every point carries the same volume.  On several ranks U is this rank's packed
    [u-slab | v-slab | w-slab]; the sums of squares add up over the ranks."""
    U, _ = s.getState()
    k0, k1 = s._slab() if s.nranks > 1 else (0, s.n[-1])
    sums, counts, off = [], [], 0
    for _, shape in s._field_shapes()[: s.dim]:
        own = int(np.prod(shape[1:])) * (min(k1, shape[0]) - k0)
        sums.append(float(np.sum(U[off:off + own] ** 2)))
        counts.append(float(np.prod(shape)))
        off += own
    if dist is not None:
        import torch
        t = torch.tensor(sums, dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        sums = t.tolist()
    return sum(0.5 * a / c for a, c in zip(sums, counts))


def self_launch(a):
    """--gpus N without a launcher: start the N ranks here (python -m torch.distributed.run on 127.0.0.1), as bench.py does"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--nt", type=int, default=None)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--dt", type=float, default=None, help="time step (the case's 0.01 is the 256^3 one: CFL 0.4; halve it at 512^3)")
    ap.add_argument("--poisson-extra", default="", help="lines appended to the Poisson solver's configuration (';' separated)")
    ap.add_argument("--velocity-extra", default="", help="lines appended to the velocity solver's configuration (';' separated)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks = z-slabs of the periodic box (the reference's README runs this case on 4 GPUs)")
    ap.add_argument("--transport", default=os.environ.get("PIB_TRANSPORT", "rccl"), choices=["rccl", "peer"],
                    help="several ranks: RCCL, or the HIP-IPC window transport (which also takes several ranks per GPU: PIB_BENCH_SHARE_GPU=1)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    rank, world, local = (int(os.environ.get(k, v)) for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    dist, uid, red_dev = None, None, "cpu"
    if world > 1:
        import ctypes

        import torch
        import torch.distributed as dist
        from petibm_amd import capi
        share = os.environ.get("PIB_BENCH_SHARE_GPU", "0") == "1"
        if share:
            local = 0
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
            red_dev = "cuda"
        buf = ctypes.create_string_buffer(capi.UID_BYTES)
        if rank == 0:
            capi.check((capi.load().pib_comm_peer_id if a.transport == "peer" else capi.load().pib_comm_unique_id)(buf))
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(red_dev)
        dist.broadcast(t, src=0)
        uid = bytes(t.cpu().numpy().tobytes())
    say = print if rank == 0 else (lambda *x, **k: None)
    d = os.path.join(ROOT, "examples", "cases", "taylorgreenvortex3dRe1600")
    cfg = yaml.safe_load(open(os.path.join(d, "config.yaml")))
    if a.cells:
        for ax in cfg["mesh"]:
            ax["subDomains"][0]["cells"] = a.cells
    if a.dt:
        cfg["parameters"]["dt"] = a.dt
    nt = a.nt if a.nt is not None else int(cfg["parameters"]["nt"])
    texts = {k: open(os.path.join(d, cfg["parameters"][k]["config"])).read() for k in ("velocitySolver", "poissonSolver")}
    texts["poissonSolver"] += "".join(ln + "\n" for ln in a.poisson_extra.split(";") if ln)
    texts["velocitySolver"] += "".join(ln + "\n" for ln in a.velocity_extra.split(";") if ln)
    ref = {round(r[0], 6): r[1] for r in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_test_vectors.json")))[
        "taylor_green_vortex_3d_re1600_spectral_512"]["rows"]}
    t0 = time.perf_counter()
    s = navierstokes.NavierStokesSolver(cfg, velocity_cfg=texts["velocitySolver"], poisson_cfg=texts["poissonSolver"], rank=rank,
                                        nranks=world, uid=uid, device=local if world > 1 else -1)
    say(f"{s.n} cells on {world} rank(s), {s.UN} velocity + {s.pN} pressure unknowns on rank 0; set-up {time.perf_counter() - t0:.1f} s", flush=True)
    say("  step      t     E_k         spectral 512^3   v_its p_its", flush=True)
    wall = 0.0
    worst = 0.0
    done = 0
    while done < nt:
        k = min(a.every, nt - done)
        t1 = time.perf_counter()
        s.advance(k)
        ite, vi, vr, pi, pr = s.linSolversInfo()  # synchronises
        wall += time.perf_counter() - t1
        done += k
        e = mean_kinetic_energy(s, dist, red_dev)
        r = ref.get(round(s.t, 6))
        if r is not None:
            worst = max(worst, abs(e - r) / r)
        say(f"{done:6d} {s.t:6.2f} {e:.8f}  {'' if r is None else f'{r:.8f}':>14}   {vi:4d} {pi:4d}", flush=True)
    say(f"{nt} steps in {wall:.1f} s = {1e3 * wall / nt:.1f} ms/step; largest relative deviation of E_k from the "
        f"spectral data {worst:.3%}")
    s.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
