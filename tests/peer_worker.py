"""One rank (= one process) of a peer-transport job: tests/test_gpu_peer_transport.py starts P of these on the one GPU.
usage: python tests/peer_worker.py <job.pkl> <rank>        (the job file carries the id, the system and the solver text)"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def poisson(job, r):
    from petibm_amd import capi
    import slab_plans as partition
    from petibm_amd.linsolver import LinSolverHIP
    n, P = job["n"], job["P"]
    pl = partition.all_plans(n, P)[r]
    s = LinSolverHIP("poisson", config_text=job["cfg"], rank=r, nranks=P, uid=job["uid"], device=0)
    if job.get("periodic"):
        s.setPeriodic(job["periodic"])
    s.assemblePoisson(list(n), job["w"], job["dt"], capi.NULLSPACE_CONSTANT)
    assert s.n_local == pl.n_local
    y = np.empty(pl.n_local)
    s.matMult(np.ascontiguousarray(job["xs"][pl.row0:pl.row0 + pl.n_local]), y)
    x = np.zeros(pl.n_local)
    s.solve(x, np.ascontiguousarray(job["b"][pl.row0:pl.row0 + pl.n_local]))
    import ctypes
    import time
    rep = ctypes.c_int64(0)
    capi.check(capi.load().pib_get_graph_replays(s._h, ctypes.byref(rep)))
    t0 = time.perf_counter()
    for _ in range(int(job.get("timed_solves", 0))):
        x2 = np.zeros(pl.n_local)
        s.solve(x2, np.ascontiguousarray(job["b"][pl.row0:pl.row0 + pl.n_local]))
    dt = (time.perf_counter() - t0) / max(1, int(job.get("timed_solves", 0)))
    out = dict(y=y, x=x, its=s.getIters(), hist=np.asarray(s.getResidualHistory()), counters=np.asarray(s.counters()),
               graph_replays=rep.value, seconds_per_solve=dt)
    s.destroy()
    return out


def boxes(job, r):
    """rows in DMDA boxes / the per-rank-packed velocity ordering (tests/test_gpu_dmda_boxes.py): setMatrix only"""
    from oracle import operators as oops
    from petibm_amd.linsolver import LinSolverHIP
    P = job["P"]
    s = LinSolverHIP(job["name"], config_text=job["cfg"], rank=r, nranks=P, uid=job["uid"], device=0)
    rp, cl, vl = job["parts"][r]
    r0, r1 = int(job["offsets"][r]), int(job["offsets"][r + 1])
    s.setMatrix(oops.CSR(r1 - r0, job["n_global"], rp, cl, vl), row0=r0, n_global=job["n_global"])
    st = s.gridStructure()
    y = np.empty(r1 - r0)
    s.matMult(np.ascontiguousarray(job["xs"][r0:r1]), y)
    x = np.zeros(r1 - r0)
    s.solve(x, np.ascontiguousarray(job["b"][r0:r1]))
    out = dict(y=y, x=x, its=s.getIters(), reason=s.getReason(), counters=np.asarray(s.counters()),
               detected=bool(st is not None and st["detected"]))
    s.destroy()
    return out


def navierstokes(job, r):
    """cases of test_gpu_navierstokes_slabs.py (plain time step or immersed bodies) on this rank's slab; the single-rank
    fields of the job are cut to the owned points here (the solver knows its slab)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from petibm_amd.navierstokes import DecoupledIBPMSolver, NavierStokesSolver
    import test_gpu_navierstokes_slabs as T
    P = job["P"]
    kw = dict(velocity_cfg=T.VEL, rank=r, nranks=P, uid=job["uid"], device=0)
    out = {}
    if job["bodies"]:
        cfg, bodies, pose = T._ib_case(job["case"])
        s = DecoupledIBPMSolver(cfg, bodies=bodies, poisson_cfg=T.KSP_P, forces_cfg=T.FORCES, **kw)
        dt = cfg["parameters"]["dt"]
        for step in range(1, job["steps"] + 1):
            if pose is not None:
                x, v = pose(step * dt)
                s.moveBodies([x], [v])
            s.advance()
        f, avg = s.getForces()
        out["forces"] = np.asarray(f)
    else:
        make, pinned = T.CASES[job["case"]]
        s = NavierStokesSolver(make(), poisson_cfg=T.AMGX_P if pinned else T.KSP_P, **kw)
        s.setState(s.ownedVelocity(job["U0"]), s.ownedPressure(job["p0"]))
        s.advance(job["steps"])
    U, p = s.getState()
    out.update(U=U, p=p, U_ref=s.ownedVelocity(job["U_ref"]), p_ref=s.ownedPressure(job["p_ref"]))
    s.destroy()
    return out


def main():
    job = pickle.load(open(sys.argv[1], "rb"))
    r = int(sys.argv[2])
    out = {"poisson": poisson, "navierstokes": navierstokes, "boxes": boxes}[job["kind"]](job, r)
    np.savez(os.path.join(os.path.dirname(sys.argv[1]), f"rank{r}.npz"), **out)


if __name__ == "__main__":
    main()
