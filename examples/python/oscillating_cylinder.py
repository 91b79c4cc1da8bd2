#!/usr/bin/env python3
"""In-line oscillating cylinder in a fluid at rest (Re = 100, KC = 5) -- the reference's API example
examples/api_examples/oscillatingcylinder2dRe100_GPU (a solver derived from RigidKinematicsSolver that overrides
setCoordinatesBodies / setVelocityBodies), on the GPU: 512 x 512 uniform mesh on [-4, 4]^2, dt = 0.002, Peskin's kernel,
10000 steps = 4 periods.  Every step moves the Lagrangian points, re-assembles the immersed-boundary operators on the device
and re-factorises the force system.

    python examples/python/oscillating_cylinder.py [--nt 10000]

Prints the extrema of the in-line force coefficient computed as the reference's plotDragCoefficient.py does
(fx + rho V a_x, normalised by rho Um^2 D / 2)."""
import argparse
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from petibm_amd import navierstokes  # noqa: E402


class OscillatingCylinderSolver(navierstokes.DecoupledIBPMSolver):
    """OscillatingCylinderSolver (oscillatingcylinder.cpp:30-110): Xd = -Am sin(2 pi f t), Ux = -Um cos(2 pi f t)."""

    def __init__(self, config, coords0, **kw):
        kin = config["bodies"][0]["kinematics"]
        self.f = float(kin.get("f", 0.0))
        D, KC = float(kin.get("D", 1.0)), float(kin.get("KC", 0.0))
        self.Am = D * KC / (2.0 * math.pi)
        self.Um = 2.0 * math.pi * self.f * self.Am
        self.coords0 = np.array(coords0, dtype=np.float64)
        super().__init__(config, bodies=[self.coords0], **kw)

    def advance(self):
        ti = self.t + self.dt  # RigidKinematicsSolver::advance: moveBodies(t + dt) (rigidkinematics.cpp:75-79)
        x = self.coords0 + np.array([-self.Am * math.sin(2.0 * math.pi * self.f * ti), 0.0])
        ub = np.tile([-self.Um * math.cos(2.0 * math.pi * self.f * ti), 0.0], (self.coords0.shape[0], 1))
        self.moveBodies([x], [ub])
        super().advance()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=10000)
    a = ap.parse_args()
    n, dt, D, KC, f = 512, 0.002, 1.0, 5.0, 0.2
    axis = lambda d: {"direction": d, "start": -4.0, "subDomains": [{"end": 4.0, "cells": n, "stretchRatio": 1.0}]}  # noqa: E731
    wall = lambda loc: {"location": loc, "u": ["DIRICHLET", 0.0], "v": ["DIRICHLET", 0.0]}  # noqa: E731
    cfg = {"mesh": [axis("x"), axis("y")],
           "flow": {"nu": 0.01, "initialVelocity": [0.0, 0.0],
                    "boundaryConditions": [wall(w) for w in ("xMinus", "xPlus", "yMinus", "yPlus")]},
           "parameters": {"dt": dt, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON", "delta": "PESKIN_2002"},
           "bodies": [{"type": "points", "kinematics": {"KC": KC, "D": D, "f": f, "center": [0.0, 0.0]}}]}
    npts = int(math.ceil(math.pi * D / (8.0 / n)))  # one point per cell width along the circumference
    ang = 2.0 * math.pi * np.arange(npts) / npts
    circle = np.stack([0.5 * D * np.cos(ang), 0.5 * D * np.sin(ang)], axis=1)
    vel = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n"
           "-velocity_pc_type jacobi\n")
    poi = ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=1000\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
           "solv:tolerance=1.0E-06\nsolv:norm=L2\nsolv:preconditioner(prec)=AMG\nprec:cycle=V\nprec:presweeps=1\n"
           "prec:postsweeps=1\nprec:smoother(smooth)=BLOCK_JACOBI\nsmooth:relaxation_factor=0.9\n")
    s = OscillatingCylinderSolver(cfg, circle, velocity_cfg=vel, poisson_cfg=poi)
    t, fx = np.empty(a.nt), np.empty(a.nt)
    t0 = time.perf_counter()
    for k in range(a.nt):
        s.advance()
        t[k], fx[k] = s.t, s.getForces()[1][0][0]
    wall_s = time.perf_counter() - t0
    w = 2.0 * math.pi * f
    Am = KC * D / (2.0 * math.pi)
    Um = w * Am
    ax = w ** 2 * Am * np.sin(w * t)
    cd = (fx + 1.0 * (math.pi * D ** 2 / 4.0) * ax) / (0.5 * Um ** 2 * D)
    print(f"{a.nt} steps in {wall_s:.1f} s ({1e3 * wall_s / a.nt:.2f} ms/step), {npts} Lagrangian points; last step {s.linSolversInfo()}")
    half = cd[t >= (t[-1] - 2.0 / f)] if a.nt * dt >= 2.0 / f else cd
    print(f"in-line force coefficient over the last two periods: max {half.max():.3f}, min {half.min():.3f}")
    s.destroy()


if __name__ == "__main__":
    main()
