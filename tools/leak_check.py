import sys
sys.path.insert(0, ".")
import numpy as np, torch
from petibm_amd import cases
from petibm_amd.navierstokes import NavierStokesSolver, DecoupledIBPMSolver
from petibm_amd.cases import circle, body_block as body_mesh
FORCES = "-forces_ksp_type preonly\n-forces_pc_type lu\n"


def flow_config(cfg, nu=0.025, dt=0.01):
    """uniform stream, convective outlet on xPlus (the boundary set of the reference's cylinder cases)"""
    dim = len(cfg["mesh"])
    for bc in cfg["flow"]["boundaryConditions"]:
        for c in "uvw"[:dim]:
            free_stream = 1.0 if c == "u" else 0.0
            bc[c] = ["CONVECTIVE", 1.0] if bc["location"] == "xPlus" else ["DIRICHLET", free_stream]
    cfg["flow"]["nu"] = nu
    cfg["flow"]["initialVelocity"] = [1.0, 0.0, 0.0][:dim]
    cfg["parameters"] = {"dt": dt, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    return cfg


def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
cfgp = cases.periodic_box((64, 64, 64), (True, True, True)); cfgp["flow"]["nu"] = 0.01
cfgp["parameters"] = {"dt": 0.01, "BN": 2}
cfgi = flow_config(body_mesh(cells=(8, 16, 8), ratio=1.25, span=3.0, core=0.8), dt=0.01)
cfgi["parameters"].update(convection="EULER_EXPLICIT", diffusion="EULER_IMPLICIT")
base = None
for it in range(12):
    s = NavierStokesSolver(cfgp); s.advance(2); s.destroy()
    s = DecoupledIBPMSolver(cfgi, bodies=[circle(32)], forces_cfg=FORCES); s.advance(2); s.destroy()
    f = free()
    if it == 1: base = f
    if it >= 1: print(it, (base - f) / 1e6, "MB lost since iteration 1")
