"""BASELINE config 5 on ONE MI355X: a 384 x 256 x 256 three-block stretched mesh (2.5e7 cells, 7.5e7 velocity unknowns) with
a rigid body in prescribed motion -- a heaving flat plate, RigidKinematicsSolver::moveBodies
(applications/rigidkinematics/rigidkinematics.cpp:118-160): the immersed-boundary operators are re-assembled and the force
system re-factorised every step -- with the solver files of the reference's flatplate3dRe100_GPU case
(examples/cases/flatplate3dRe100AoA30/config).  Prints the time per step, split into moving the body and advancing.
The 8-slab run of this case is a test (tests/test_gpu_navierstokes_slabs.py); this is the single-GPU timing.

    python tools/config5_heaving_plate.py [--steps 40]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petibm_amd import cases  # noqa: E402
from petibm_amd.navierstokes import DecoupledIBPMSolver  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def axis(name, half, cells_core, cells_out, ratio):
    """uniform block [-half/2, half/2] of `cells_core` cells, stretched outwards by `ratio` over `cells_out` cells on either
    side -- the outer blocks as long as it takes for their first cell to continue the core's width (the way the reference's
    example meshes are laid out: examples/decoupledibpm/flatplate3dRe100_GPU/config.yaml)"""
    core = half / 2.0
    h = half / cells_core
    out = h * ratio * (ratio ** cells_out - 1.0) / (ratio - 1.0)
    return {"direction": name, "start": -core - out,
            "subDomains": [{"end": -core, "cells": cells_out, "stretchRatio": 1.0 / ratio},
                           {"end": core, "cells": cells_core, "stretchRatio": 1.0},
                           {"end": core + out, "cells": cells_out, "stretchRatio": ratio}]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--chebyshev", action="store_true", help="velocity_solver.info with solver=CHEBYSHEV instead of PBICGSTAB (krylov.hip: solve_chebyshev)")
    a = ap.parse_args()
    cfg = cases.cavity((384, 256, 256), lid=0.0)
    cfg["mesh"] = [axis("x", 1.5, 192, 96, 1.04), axis("y", 1.0, 128, 64, 1.04), axis("z", 1.0, 128, 64, 1.04)]
    for bc in cfg["flow"]["boundaryConditions"]:
        for c in "uvw":
            free = 1.0 if c == "u" else 0.0
            bc[c] = ["CONVECTIVE", free] if bc["location"] == "xPlus" else ["DIRICHLET", free]
    cfg["flow"]["nu"] = 0.01
    cfg["flow"]["initialVelocity"] = [1.0, 0.0, 0.0]
    cfg["parameters"] = {"dt": 0.004, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    d = os.path.join(ROOT, "examples", "cases", "flatplate3dRe100AoA30", "config")
    text = {k: open(os.path.join(d, k + "_solver.info")).read() for k in ("velocity", "poisson", "forces")}
    if a.chebyshev:
        text["velocity"] = text["velocity"].replace("solver(solv)=PBICGSTAB", "solver(solv)=CHEBYSHEV")
        print("velocity solver: CHEBYSHEV", flush=True)
    h = 3.0 / 192  # the uniform block's cell: 0.75 x 0.375 plate of 48 x 24 = 1152 points
    xs, zs = np.meshgrid(-0.375 + h * np.arange(48), -0.19 + h * np.arange(24), indexing="ij")
    plate = np.stack([xs.ravel(), np.zeros(xs.size), zs.ravel()], axis=1)
    amp, om, dt = 0.1, 2.0 * np.pi, cfg["parameters"]["dt"]

    def pose(t):
        return plate + np.array([0.0, amp * np.sin(om * t), 0.0]), np.tile([0.0, amp * om * np.cos(om * t), 0.0], (plate.shape[0], 1))

    t0 = time.perf_counter()
    s = DecoupledIBPMSolver(cfg, bodies=[plate], velocity_cfg=text["velocity"], poisson_cfg=text["poisson"], forces_cfg=text["forces"])
    print(f"{s.pN} cells, {s.UN} velocity unknowns, {3 * plate.shape[0]} force unknowns; set-up {time.perf_counter() - t0:.1f} s", flush=True)
    try:  # the pressure solver's multigrid hierarchy (pib_get_multigrid_levels)
        lv = s.poissonSolver.multigridLevels() if hasattr(s, "poissonSolver") else []
        if lv:
            print("multigrid levels:", " ".join("x".join(str(v) for v in t) for t in lv[:6]), "...",
                  f"cells of all levels / fine = {sum(t[0] * t[1] * t[2] for t in lv) / s.pN:.2f}", flush=True)
    except Exception:  # noqa: BLE001
        pass
    tm = ta = 0.0
    for step in range(1, a.steps + 1):
        x, v = pose(step * dt)
        t0 = time.perf_counter()
        s.moveBodies([x], [v])  # returns with the force system factorised (host-synchronous)
        t1 = time.perf_counter()
        s.advance()
        s.getForces()
        t2 = time.perf_counter()
        if step == 5:
            s.enableStageTimers()  # the reference's PetscLogStage names (pib_ns_get_stage_times), HIP events on the engine's stream
        if step > 5:  # the first steps carry the start-up transient (and lazy allocations)
            tm += t1 - t0
            ta += t2 - t1
        if step % 10 == 0 or step == a.steps:
            info = s.linSolversInfo()
            f, avg = s.getForces()
            print(f"step {step}: velocity {info[1]} its, Poisson {info[3]} its, forces {info[5]} its; C_y = {2 * avg[0][1] / (0.75 * 0.375):+.4f}", flush=True)
    n = a.steps - 5
    print(f"{n} steps: {1e3 * (tm + ta) / n:.1f} ms per step = {1e3 * tm / n:.1f} ms moving the body (operators + factorisation) "
          f"+ {1e3 * ta / n:.1f} ms advancing", flush=True)
    st = s.stageTimes()
    k = max(st.pop("steps"), 1)
    print("stages of advance(), ms per step over", k, "steps:", "  ".join(f"{name} {ms / k:.2f}" for name, ms in st.items()), flush=True)
    s.destroy()


if __name__ == "__main__":
    main()
