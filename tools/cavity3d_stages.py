"""BASELINE configs 2 / 3 as a TIME STEP on one MI355X: the 3-D lid-driven cavity (Re = 1000, dt = 0.001 as in the
reference's examples/navierstokes/liddrivencavity3dRe1000_GPU) at N^3 cells through the device-resident step, with the
stages of a step timed under the reference's PetscLogStage names (applications/navierstokes/navierstokes.cpp:186-199:
what `-log_view` prints per stage).  The solver files are the reference-shaped ones of examples/cases.

    python tools/cavity3d_stages.py [256] [--steps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from petibm_amd import cases  # noqa: E402
from petibm_amd.navierstokes import NavierStokesSolver  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int, nargs="?", default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--chebyshev", action="store_true", help="velocity_solver.info with solver=CHEBYSHEV instead of PBICGSTAB (krylov.hip: solve_chebyshev)")
    a = ap.parse_args()
    n = a.n
    cfg = cases.cavity((n, n, n), lid=1.0)
    cfg["flow"]["nu"] = 0.001
    cfg["parameters"] = {"dt": 0.001, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    d = os.path.join(ROOT, "examples", "cases", "flatplate3dRe100AoA30", "config")  # the reference's 3-D GPU solver files
    vel, poi = (open(os.path.join(d, k + "_solver.info")).read() for k in ("velocity", "poisson"))
    if a.chebyshev:
        vel = vel.replace("solver(solv)=PBICGSTAB", "solver(solv)=CHEBYSHEV")
        print("velocity solver: CHEBYSHEV", flush=True)
    t0 = time.perf_counter()
    s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
    print(f"{n}^3 cavity: {s.pN} cells, {s.UN} velocity unknowns; set-up {time.perf_counter() - t0:.1f} s", flush=True)
    s.advance(5)  # start-up transient, lazy allocations
    s.getState()
    s.enableStageTimers()
    t0 = time.perf_counter()
    s.advance(a.steps)
    st = s.stageTimes()  # (waits for the last stage's event: the steps are complete)
    wall = time.perf_counter() - t0
    info = s.linSolversInfo()
    k = max(st.pop("steps"), 1)
    print(f"{a.steps} steps: {1e3 * wall / a.steps:.1f} ms per step (wall, state left on the device); last step: velocity {info[1]} its, "
          f"Poisson {info[3]} its", flush=True)
    print("stages, ms per step over", k, "steps:", "  ".join(f"{name} {ms / k:.2f}" for name, ms in st.items()), flush=True)
    s.destroy()


if __name__ == "__main__":
    main()
