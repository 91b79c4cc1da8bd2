"""Builders of case descriptions in the shape of the reference's config.yaml (a dict with `mesh`, `flow`,
`parameters`), for the tools and examples of this package -- the host-side mirrors (`NavierStokesSolver`,
`DecoupledIBPMSolver`) take exactly this dict.  Layouts follow the reference's example cases:
examples/navierstokes/liddrivencavity2dRe1000_GPU/config.yaml (cavity), taylorgreenvortex2dRe100/config.yaml (periodic
box), examples/decoupledibpm/cylinder2dRe40_GPU/config.yaml (uniform block around the body, stretched outwards)."""
from __future__ import annotations

from typing import Sequence

import numpy as np

_NAMES = "xyz"
_LOCS = ["xMinus", "xPlus", "yMinus", "yPlus", "zMinus", "zPlus"]


def _walls(dim: int, periodic: Sequence[bool], lid: float):
    out = []
    for loc in _LOCS[: 2 * dim]:
        axis = _LOCS.index(loc) // 2
        kind = "PERIODIC" if periodic[axis] else "DIRICHLET"
        bc = {"location": loc}
        for c in "uvw"[:dim]:
            bc[c] = [kind, 0.0]
        if loc == "yPlus" and not periodic[axis]:
            bc["u"] = ["DIRICHLET", lid]
        out.append(bc)
    return out


def cavity(n: Sequence[int], lo: float = 0.0, hi: float = 1.0, lid: float = 1.0) -> dict:
    """lid-driven cavity: uniform mesh, no-slip walls, u = lid on yPlus"""
    dim = len(n)
    mesh = [{"direction": _NAMES[d], "start": lo, "subDomains": [{"end": hi, "cells": int(n[d]), "stretchRatio": 1.0}]}
            for d in range(dim)]
    return {"mesh": mesh, "flow": {"boundaryConditions": _walls(dim, [False] * dim, lid)}}


def periodic_box(n: Sequence[int], periodic: Sequence[bool], lo: float = 0.0, hi: float = 1.0) -> dict:
    """PERIODIC at both ends of the flagged directions for every component, no-slip walls elsewhere"""
    dim = len(n)
    mesh = [{"direction": _NAMES[d], "start": lo, "subDomains": [{"end": hi, "cells": int(n[d]), "stretchRatio": 1.0}]}
            for d in range(dim)]
    return {"mesh": mesh, "flow": {"boundaryConditions": _walls(dim, list(periodic), 0.0)}}


def body_block(cells=(12, 16, 12), ratio: float = 1.2, span: float = 3.0, core: float = 0.8, dim: int = 2) -> dict:
    """uniform block [-core, core] around the body, stretched towards +-span (every cylinder example of the reference)"""
    a, c, e = cells
    sub = [{"end": -core, "cells": a, "stretchRatio": 1.0 / ratio}, {"end": core, "cells": c, "stretchRatio": 1.0},
           {"end": span, "cells": e, "stretchRatio": ratio}]
    cfg = cavity((a + c + e,) * dim, lid=0.0)
    cfg["mesh"] = [{"direction": _NAMES[d], "start": -span, "subDomains": [dict(s) for s in sub]} for d in range(dim)]
    return cfg


def circle(npts: int, radius: float = 0.5, centre=(0.0, 0.0)) -> np.ndarray:
    """Lagrangian points of a cylinder section (the .body files of the reference's cylinder cases)"""
    a = 2.0 * np.pi * np.arange(npts) / npts
    return np.stack([centre[0] + radius * np.cos(a), centre[1] + radius * np.sin(a)], axis=1)


# Solver files of the reference's immersed-boundary examples (examples/decoupledibpm/cylinder2dRe40_GPU/config/*.info): the Poisson
# solver as an AmgX PCG + AMG V(1,1) text with the tolerance left open, the direct forces solve.
AMGX_POISSON = ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
                "solv:tolerance={tol}\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=AMG\nprec:cycle=V\n"
                "prec:presweeps=1\nprec:postsweeps=1\nprec:coarsest_sweeps=2\nprec:smoother(smooth)=BLOCK_JACOBI\n"
                "smooth:relaxation_factor=0.9\n")
DIRECT_FORCES = "-forces_ksp_type preonly\n-forces_pc_type lu\n-forces_pc_factor_mat_solver_type superlu_dist\n"


def uniform_stream(cfg: dict, nu: float = 0.025, dt: float = 0.01, kernel: str = None) -> dict:
    """free stream u = 1 through the box of `cfg`: DIRICHLET inflow and sides, CONVECTIVE outlet on xPlus, AB2 + Crank-Nicolson
    (the boundary set of the reference's cylinder and flat-plate cases)"""
    cfg = dict(cfg)
    dim = len(cfg["mesh"])
    for bc in cfg["flow"]["boundaryConditions"]:
        for c in "uvw"[:dim]:
            free = 1.0 if c == "u" else 0.0
            bc[c] = ["CONVECTIVE", 1.0] if bc["location"] == "xPlus" else ["DIRICHLET", free]
    cfg["flow"]["nu"] = nu
    cfg["flow"]["initialVelocity"] = [1.0, 0.0, 0.0][:dim]
    cfg["parameters"] = {"dt": dt, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    if kernel:
        cfg["parameters"]["delta"] = kernel
    return cfg
