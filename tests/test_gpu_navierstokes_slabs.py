"""The device time step on z-slabs (y-slabs in 2-D): NavierStokesSolver::advance on the DMDA decomposition
(applications/navierstokes/navierstokes.cpp:240-266, src/mesh/cartesianmesh.cpp:492-538), P ranks = P host threads on the
one test GPU through the loopback transport (RCCL refuses several ranks per device).

Bars: the explicit part of a step -- rhs1 = -G p + u/dt + N(u) terms + diffusion + boundary corrections, and rhs2 = D u* +
Dbc -- is the single-rank engine's BIT FOR BIT on every rank's owned points (same kernels on the extended slab); after
three steps the fields agree with the single rank to the solver tolerance (the Krylov reductions sum in another order).
"""
import numpy as np
import pytest

from test_gpu_navierstokes import cavity, moving_walls_3d, convective_outlet, neumann_outlet, AMGX_P, KSP_P, VEL
from test_gpu_multirank_loopback import _run_ranks

pytestmark = pytest.mark.gpu


CASES = {
    "3d_cavity": (lambda: cavity((12, 10, 12), nu=0.02, dt=0.005, stretched=True), False),
    "3d_moving_walls_neumann": (moving_walls_3d, False),
    "3d_convective_outlet": (lambda: convective_outlet((10, 8, 9)), True),
    "2d_convective_outlet": (lambda: convective_outlet((16, 12)), True),
    "2d_cavity": (lambda: cavity((14, 13), stretched=True), False),
    # a periodic slab axis: the wrap goes through the ring of plane exchanges (rank 0 <-> rank P - 1)
    "3d_periodic_box": (lambda: _periodic((10, 8, 12), (True, True, True)), False),
    "3d_channel_periodic_z": (lambda: _periodic((8, 7, 16), (False, False, True)), True),
    "2d_periodic_y": (lambda: _periodic((12, 12), (True, True)), False),
    # round 5: a NEUMANN outlet on the NORMAL component (the divergence's ghost fold; the Poisson operator from the window chain
    # of sparse products, bn.hip, BiCGStab in place of the file's CG); outlet next to the pinned cell: non-singular system
    "3d_neumann_outlet": (lambda: neumann_outlet((10, 8, 9), outlet="xMinus"), True),
    "2d_neumann_outlet": (lambda: neumann_outlet((16, 12), outlet="xMinus"), True),
}


def _periodic(n, per):
    from oracle import mesh as omesh
    cfg = omesh.periodic_config(n, per, lo=0.0, hi=2.0)
    cfg["flow"]["nu"] = 0.02
    cfg["parameters"] = {"dt": 0.005, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    return cfg


@pytest.mark.parametrize("case,P", [("3d_cavity", 2), ("3d_cavity", 3), ("3d_moving_walls_neumann", 2),
                                    ("3d_convective_outlet", 3), ("2d_convective_outlet", 2), ("2d_cavity", 3),
                                    ("3d_periodic_box", 2), ("3d_periodic_box", 3), ("3d_channel_periodic_z", 4),
                                    ("2d_periodic_y", 2), ("3d_neumann_outlet", 2), ("2d_neumann_outlet", 3)])
def test_time_step_on_slabs_reproduces_the_single_rank(case, P):
    from petibm_amd.navierstokes import NavierStokesSolver
    make, pinned = CASES[case]
    cfg = make()
    pcfg = AMGX_P if pinned else KSP_P
    one = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=pcfg)
    rng = np.random.default_rng(11)
    U0 = 0.1 * rng.uniform(-1, 1, one.UN)
    p0 = 0.1 * rng.uniform(-1, 1, one.pN)
    if "convective" in case:
        U0[: int(np.prod(one._field_shape(0)))] += 1.0  # perturbed free stream
    if "neumann_outlet" in case:
        U0[: int(np.prod(one._field_shape(0)))] -= 1.0  # ... towards xMinus
    one.setState(U0, p0)
    one.advance(1)
    U1, p1, rhs1, rhs2 = one.getState(rhs=True)
    one.advance(2)
    U3, p3 = one.getState()

    def rank_fn(r, uid):
        s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=pcfg, device=0, rank=r, nranks=P, uid=uid)
        s.setState(s.ownedVelocity(U0), s.ownedPressure(p0))
        s.advance(1)
        a = s.getState(rhs=True)
        s.advance(2)
        b = s.getState()
        cut = [s.ownedVelocity(x) for x in (U1, rhs1, U3)] + [s.ownedPressure(x) for x in (p1, rhs2, p3)]
        s.destroy()
        return a, b, cut

    res = _run_ranks(P, rank_fn)
    for (Ua, pa, r1, r2), (Ub, pb), (cU1, crhs1, cU3, cp1, crhs2, cp3) in res:
        if "periodic" in case:
            # across the periodic seam the single-rank engine sums a row in the wrapped neighbour's sorted place, the slab
            # engine in the natural order of its extended slab: the outer planes agree to rounding
            assert np.abs(r1 - crhs1).max() <= 1e-13 * np.abs(crhs1).max()
        else:
            assert np.array_equal(r1, crhs1), "rhs1 of the first step differs from the single-rank engine's"
        assert np.allclose(r2, crhs2, rtol=0, atol=1e-11 * max(1.0, np.abs(rhs2).max()))  # D u* after a solve to 1e-14
        assert np.allclose(Ua, cU1, rtol=0, atol=1e-10) and np.allclose(Ub, cU3, rtol=0, atol=1e-9)
        if pinned:
            assert np.allclose(pa, cp1, rtol=0, atol=1e-8) and np.allclose(pb, cp3, rtol=0, atol=1e-8)
    # the pressure of a constant-null-space solve is defined up to a constant: compare after removing the global mean
    if not pinned:
        for idx, ref in ((1, p3),):
            pg = np.concatenate([r[idx][1] for r in res])
            assert np.allclose(pg - pg.mean(), ref - ref.mean(), rtol=0, atol=1e-8)
    one.destroy()


def test_slab_engine_limits_and_sizes():
    from petibm_amd import capi
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = cavity((8, 8, 8), nu=0.05, dt=0.01)

    def rank_fn(r, uid):
        s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=KSP_P, device=0, rank=r, nranks=2, uid=uid)
        sizes = (s.UN, s.pN)
        lib = capi.load()
        codes = (lib.pib_ns_set_bn_order(s._h, 0),)  # "The order of Bn can not be smaller than 1." (createbn.cpp:27-29)
        s.destroy()
        return sizes, codes

    res = _run_ranks(2, rank_fn)
    # u, v: 4 planes of 7*8 / 8*7 points each; w: 4 planes on rank 0, 3 on the last rank (7 faces in all)
    assert res[0][0] == (4 * 56 + 4 * 56 + 4 * 64, 4 * 64) and res[1][0] == (4 * 56 + 4 * 56 + 3 * 64, 4 * 64)
    assert all(c == capi.ERR_SUP for r in res for c in r[1])


# ---- immersed bodies on slabs (DecoupledIBPMSolver / RigidKinematicsSolver on the DMDA decomposition) -----------------
# Every rank assembles Delta / E / H on the velocity points it owns; the sums over velocity points -- E u and the force
# system E BN H -- add up over the ranks (all-reduce), the small force system is factorised on every rank.  The order of
# those sums differs from the single rank's, so the bar is the solver tolerance, not bits.
FORCES = "-forces_ksp_type preonly\n-forces_pc_type lu\n-forces_pc_factor_mat_solver_type superlu_dist\n"


def _ib_case(kind):
    from petibm_amd import cases
    from test_gpu_ibm import flow_config, sphere_points
    if kind == "2d_cylinder":
        # the body sits across the cut of two y-slabs and inside the middle one of three
        return flow_config(cases.body_block(cells=(8, 16, 8), ratio=1.25, span=3.0, core=0.8)), [cases.circle(40, 0.5)], None
    if kind == "2d_cylinder_periodic_y":
        # the cylinder array of examples/decoupledibpm/multicylinders2dRe100_GPU in small: free stream along x, periodic y --
        # in 2-D the slab axis.  A window point beyond the periodic seam is a domain length away from the body and carries
        # nothing (createdelta.cpp:171-208), on one rank and on slabs alike.
        cfg = flow_config(cases.body_block(cells=(8, 16, 8), ratio=1.25, span=3.0, core=0.8))
        for bc in cfg["flow"]["boundaryConditions"]:
            if bc["location"] in ("yMinus", "yPlus"):
                for c in "uv":
                    bc[c] = ["PERIODIC", 0.0]
        return cfg, [cases.circle(40, 0.5) + np.array([0.0, 2.3])], None   # its kernel windows reach beyond the seam of the last slab
    if kind == "3d_sphere":
        return flow_config(cases.body_block(cells=(4, 10, 4), ratio=1.4, span=2.0, core=0.7, dim=3), nu=0.05), [sphere_points(50, 0.4)], None
    if kind == "moving_sphere_3d":
        cfg = flow_config(cases.body_block(cells=(4, 12, 4), ratio=1.4, span=2.0, core=0.8, dim=3), nu=0.05)
        base3 = sphere_points(40, 0.3)

        def pose3(t):
            return base3 + np.array([0.0, 0.0, 0.2 * np.sin(2.0 * np.pi * t)]), np.tile([0.0, 0.0, 0.4 * np.pi * np.cos(2.0 * np.pi * t)], (40, 1))

        return cfg, [base3], pose3
    # a cylinder oscillating in a closed box of fluid at rest (applications/rigidkinematics)
    cfg = cases.body_block(cells=(6, 20, 6), ratio=1.3, span=2.0, core=1.0)
    cfg["flow"]["nu"] = 0.02
    cfg["parameters"] = {"dt": 0.01, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    base = cases.circle(36, 0.3)
    amp, om = 0.25, 2.0 * np.pi

    def pose(t):
        return base + np.array([0.0, amp * np.sin(om * t)]), np.tile([0.0, amp * om * np.cos(om * t)], (base.shape[0], 1))

    return cfg, [base], pose


@pytest.mark.parametrize("kind,P,bn", [("2d_cylinder", 2, 1), ("2d_cylinder", 3, 1), ("3d_sphere", 2, 1), ("moving_cylinder", 2, 1),
                                       ("moving_cylinder", 3, 1), ("moving_sphere_3d", 2, 1), ("moving_sphere_3d", 4, 1),
                                       ("2d_cylinder_periodic_y", 2, 1), ("2d_cylinder_periodic_y", 3, 1),
                                       # parameters.BN > 1 WITH bodies on several ranks (round 4; decoupledibpm.cpp:194-205 builds BNH = BN H,
                                       # EBNH = E BNH on any communicator): BN term by term through the engine's halo exchanges, EBNH dense
                                       ("2d_cylinder", 2, 2), ("2d_cylinder", 3, 3), ("3d_sphere", 2, 2), ("moving_cylinder", 2, 2)])
def test_immersed_bodies_on_slabs_reproduce_the_single_rank(kind, P, bn):
    from petibm_amd.navierstokes import DecoupledIBPMSolver
    cfg, bodies, pose = _ib_case(kind)
    if bn > 1:
        cfg["parameters"]["BN"] = bn
    dt = cfg["parameters"]["dt"]
    nsteps = 4

    def run(s):
        out = []
        for step in range(1, nsteps + 1):
            if pose is not None:
                x, v = pose(step * dt)  # moveBodies(t + dt) precedes the step (rigidkinematics.cpp:75-79)
                s.moveBodies([x], [v])
            s.advance()
            U, p = s.getState()
            f, avg = s.getForces()
            out.append((U, p, f.copy(), avg.copy()))
        return out

    one = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=FORCES)
    ref = run(one)

    def rank_fn(r, uid):
        s = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=FORCES, device=0, rank=r,
                                nranks=P, uid=uid)
        got = run(s)
        cut = [(s.ownedVelocity(U), s.ownedPressure(p)) for U, p, _, _ in ref]
        s.destroy()
        return got, cut

    res = _run_ranks(P, rank_fn)
    for got, cut in res:
        for (U, p, f, avg), (cU, cp), (_, _, rf, ravg) in zip(got, cut, ref):
            assert np.abs(U - cU).max() <= 1e-8 * max(1.0, np.abs(cU).max())
            assert np.abs(f - rf).max() <= 1e-7 * np.abs(rf).max()
            assert np.abs(avg - ravg).max() <= 1e-7 * np.abs(ravg).max()
    # every rank holds the same forces, bit for bit (the replicated force solves must not drift apart)
    for step in range(nsteps):
        assert all(np.array_equal(res[0][0][step][2], r[0][step][2]) for r in res[1:])
    one.destroy()


KRYLOV_FORCES = {"cg": "-forces_ksp_type cg\n-forces_pc_type jacobi\n-forces_ksp_rtol 1.0E-13\n-forces_ksp_atol 1.0E-50\n-forces_ksp_max_it 2000\n",
                 "bcgs": "-forces_ksp_type bcgs\n-forces_pc_type none\n-forces_ksp_rtol 1.0E-13\n-forces_ksp_atol 1.0E-50\n-forces_ksp_max_it 2000\n",
                 "amgx": ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=2000\nsolv:monitor_residual=1\nsolv:convergence=RELATIVE_INI\n"
                          "solv:tolerance=1e-13\nsolv:norm=L2\nsolv:preconditioner(prec)=BLOCK_JACOBI\nprec:relaxation_factor=1.0\n")}


@pytest.mark.parametrize("kind,P,bn,forces", [("2d_cylinder", 2, 1, "cg"), ("2d_cylinder", 3, 1, "bcgs"), ("3d_sphere", 2, 1, "amgx"),
                                              ("moving_cylinder", 2, 1, "cg"), ("2d_cylinder", 2, 2, "cg")])
def test_immersed_bodies_on_slabs_with_a_krylov_forces_solver(kind, P, bn, forces):
    """decoupledibpm.cpp:75-80 hands whatever forces_solver.info says to createLinSolver, on any communicator.  On slabs the
    ranks' parts of E BN H are summed once at assembly (csrc/ibm.hip: ib_adopt_summed) and every rank iterates on the same
    replicated matrix: the single rank's forces to the solver tolerance, the same bits on every rank."""
    from petibm_amd.navierstokes import DecoupledIBPMSolver
    cfg, bodies, pose = _ib_case(kind)
    if bn > 1:
        cfg["parameters"]["BN"] = bn
    dt = cfg["parameters"]["dt"]
    nsteps = 3
    fcfg = KRYLOV_FORCES[forces]

    def run(s):
        out = []
        for step in range(1, nsteps + 1):
            if pose is not None:
                x, v = pose(step * dt)
                s.moveBodies([x], [v])
            s.advance()
            U, p = s.getState()
            f, avg = s.getForces()
            out.append((U, p, f.copy(), avg.copy()))
        return out

    one = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=fcfg)
    ref = run(one)
    direct = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=FORCES)
    refd = run(direct)
    direct.destroy()

    def rank_fn(r, uid):
        s = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=fcfg, device=0, rank=r, nranks=P,
                                uid=uid)
        got = run(s)
        cut = [(s.ownedVelocity(U), s.ownedPressure(p)) for U, p, _, _ in ref]
        s.destroy()
        return got, cut

    res = _run_ranks(P, rank_fn)
    for got, cut in res:
        for (U, p, f, avg), (cU, cp), (_, _, rf, ravg), (_, _, df, _) in zip(got, cut, ref, refd):
            assert np.abs(U - cU).max() <= 1e-8 * max(1.0, np.abs(cU).max())
            assert np.abs(f - rf).max() <= 1e-7 * np.abs(rf).max()
            assert np.abs(avg - ravg).max() <= 1e-7 * np.abs(ravg).max()
            assert np.abs(f - df).max() <= 1e-6 * np.abs(df).max()   # ... and the direct solver's
    for step in range(nsteps):
        assert all(np.array_equal(res[0][0][step][2], r[0][step][2]) for r in res[1:])
    one.destroy()


def _three_block_axis(name, core, cells_core, cells_out, ratio):
    """stretched / uniform / stretched with continuous widths (the layout of flatplate3dRe100AoA30_GPU/config.yaml:30-68)"""
    h = 2.0 * core / cells_core
    out = h * ratio * (ratio ** cells_out - 1.0) / (ratio - 1.0)
    return {"direction": name, "start": -core - out,
            "subDomains": [{"end": -core, "cells": cells_out, "stretchRatio": 1.0 / ratio},
                           {"end": core, "cells": cells_core, "stretchRatio": 1.0},
                           {"end": core + out, "cells": cells_out, "stretchRatio": ratio}]}


def test_config5_heaving_plate_384x256x256_on_8_slabs():
    """BASELINE config 5 at its size: a 384 x 256 x 256 mesh of three sub-domains per axis, a rigid body with prescribed
    motion (a heaving plate, RigidKinematicsSolver: operators re-assembled and the force system re-factorised every step),
    on 8 z-slabs -- through the loopback transport on the one test GPU -- against the single-rank engine on the same mesh:
    velocity to 1e-8, forces to 1e-7 after two steps; every rank holds the same forces bit for bit."""
    from petibm_amd.navierstokes import DecoupledIBPMSolver
    from test_gpu_ibm import flow_config
    from petibm_amd import cases
    cfg = cases.cavity((384, 256, 256), lid=0.0)
    cfg["mesh"] = [_three_block_axis("x", 1.5, 192, 96, 1.04), _three_block_axis("y", 1.0, 128, 64, 1.04),
                   _three_block_axis("z", 1.0, 128, 64, 1.04)]
    cfg = flow_config(cfg, nu=0.01, dt=0.004)
    h = 3.0 / 192
    xs, zs = np.meshgrid(-0.25 + h * np.arange(24), -0.19 + h * np.arange(24), indexing="ij")
    plate = np.stack([xs.ravel(), np.zeros(xs.size), zs.ravel()], axis=1)
    amp, om, dt = 0.1, 2.0 * np.pi, cfg["parameters"]["dt"]

    def pose(t):
        x = plate + np.array([0.0, amp * np.sin(om * t), 0.0])
        return x, np.tile([0.0, amp * om * np.cos(om * t), 0.0], (plate.shape[0], 1))

    def run(s, n=2):
        for step in range(1, n + 1):
            x, v = pose(step * dt)
            s.moveBodies([x], [v])
            s.advance()
        U, p = s.getState()
        f, avg = s.getForces()
        return U, f.copy(), avg.copy()

    one = DecoupledIBPMSolver(cfg, bodies=[plate], velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=FORCES)
    U1, f1, a1 = run(one)
    one.destroy()
    P = 8

    def rank_fn(r, uid):
        s = DecoupledIBPMSolver(cfg, bodies=[plate], velocity_cfg=VEL, poisson_cfg=KSP_P, forces_cfg=FORCES, device=0, rank=r,
                                nranks=P, uid=uid)
        U, f, a = run(s)
        cU = s.ownedVelocity(U1)
        s.destroy()
        return U, f, a, cU

    res = _run_ranks(P, rank_fn)
    for U, f, a, cU in res:
        assert np.abs(U - cU).max() <= 1e-8 * max(1.0, np.abs(cU).max())
        assert np.abs(f - f1).max() <= 1e-7 * np.abs(f1).max()
        assert np.array_equal(f, res[0][1])
    assert np.abs(res[0][2] - a1).max() <= 1e-7 * np.abs(a1).max()
