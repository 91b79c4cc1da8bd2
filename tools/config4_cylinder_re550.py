"""BASELINE config 4 on one MI355X: the reference's examples/decoupledibpm/cylinder2dRe550_GPU verbatim (450 x 450 stretched
mesh, 315 Lagrangian points, nu = 1/550, dt = 0.0025, convective outlet; velocity BiCGStab + Jacobi 1e-6, Poisson PCG + AMG
1e-6, direct forces solve).  Prints the time per step; the drag against Koumoutsakos & Leonard is a test
(tests/test_gpu_ibm.py::test_cylinder_re550_baseline_config4_drag).

    python tools/config4_cylinder_re550.py [--nt 1200]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petibm_amd import cases  # noqa: E402
from petibm_amd.navierstokes import DecoupledIBPMSolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=1200)
    ap.add_argument("--poisson-extra", default="", help="lines appended to the Poisson solver file (';' separated)")
    a = ap.parse_args()
    sub = [{"end": -0.54, "cells": 171, "stretchRatio": 0.980392156}, {"end": 0.54, "cells": 108, "stretchRatio": 1.0},
           {"end": 15.0, "cells": 171, "stretchRatio": 1.02}]
    cfg = cases.cavity((450, 450), lid=0.0)
    cfg["mesh"] = [{"direction": d, "start": -15.0, "subDomains": [dict(s) for s in sub]} for d in "xy"]
    for bc in cfg["flow"]["boundaryConditions"]:
        for c in "uv":
            free = 1.0 if c == "u" else 0.0
            bc[c] = ["CONVECTIVE", free] if bc["location"] == "xPlus" else ["DIRICHLET", free]
    cfg["flow"]["nu"] = 0.00181818181818
    cfg["flow"]["initialVelocity"] = [1.0, 0.0]
    cfg["parameters"] = {"dt": 0.0025, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    vel = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n"
           "-velocity_pc_type jacobi\n-velocity_pc_jacobi_type diagonal\n")
    poi = ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
           "solv:tolerance=1.0E-06\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=AMG\nprec:cycle=V\n"
           "prec:presweeps=1\nprec:postsweeps=1\nprec:coarsest_sweeps=2\nprec:smoother(smooth)=BLOCK_JACOBI\n"
           "smooth:relaxation_factor=0.9\n") + a.poisson_extra.replace(";", "\n") + "\n"
    frc = "-forces_ksp_type preonly\n-forces_pc_type lu\n-forces_pc_factor_mat_solver_type superlu_dist\n"
    s = DecoupledIBPMSolver(cfg, bodies=[cases.circle(315)], velocity_cfg=vel, poisson_cfg=poi, forces_cfg=frc)
    s.advance(50)  # start-up transient (and lazy allocations) not timed
    s.getForces()
    t0 = time.perf_counter()
    s.advance(a.nt - 50)
    _, avg = s.getForces()
    el = time.perf_counter() - t0
    info = s.linSolversInfo()
    print(f"{s.pN} cells, {s.nf} force unknowns: {a.nt - 50} steps in {el:.2f} s = {1e3 * el / (a.nt - 50):.3f} ms per step; "
          f"last step: velocity {info[1]} its, Poisson {info[3]} its; C_D = {2.0 * avg[0][0]:.4f}")
    s.destroy()


if __name__ == "__main__":
    main()
