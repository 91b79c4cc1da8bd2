"""Seeded sweep over meshes and boundary sets the hand-written cases do not list ("edge cases the reference
tests" -- one to three sub-domains per direction with their own stretching ratios, three to a dozen cells, walls / sliding
walls / zero-gradient tangential components / a convective outlet / periodic directions in any combination, 2-D and 3-D):
the assembled Poisson and velocity operators bit for bit against the oracle's restatement of createDivergence /
createGradient / createLaplacian, and two time steps of NavierStokesSolver::advance against the oracle's."""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, navierstokes as ons, operators as oops
from test_gpu_navierstokes import AMGX_P, KSP_P, VEL
from test_gpu_parity import _a0_table, amgx_cfg

pytestmark = pytest.mark.gpu
NAMES = "xyz"


def random_config(seed):
    rng = np.random.default_rng(1000 + seed)
    dim = 2 if seed % 3 else 3
    comps = ["u", "v", "w"][:dim]
    mesh, periodic = [], []
    for d in range(dim):
        nsub = int(rng.integers(1, 4))
        cells = [int(rng.integers(1, 5)) for _ in range(nsub)]
        while sum(cells) < 4:
            cells[int(rng.integers(0, nsub))] += 1
        start = float(rng.uniform(-1.0, 0.5))
        end, subs = start, []
        for c in cells:
            end += float(rng.uniform(0.3, 1.2))
            subs.append({"end": end, "cells": c, "stretchRatio": float(rng.choice([1.0, 1.0, 1.07, 0.9, 1.3, 0.75]))})
        mesh.append({"direction": NAMES[d], "start": start, "subDomains": subs})
        periodic.append(bool(rng.uniform() < 0.3))
    stream = int(rng.integers(0, dim)) if rng.uniform() < 0.35 else -1
    if stream >= 0:
        periodic[stream] = False
    bcs = []
    for d in range(dim):
        for side, loc in enumerate((NAMES[d] + "Minus", NAMES[d] + "Plus")):
            bc = {"location": loc}
            kind = "periodic" if periodic[d] else str(rng.choice(["wall", "sliding", "zero_gradient"]))
            for f, c in enumerate(comps):
                if kind == "periodic":
                    bc[c] = ["PERIODIC", 0.0]
                elif stream >= 0 and d == stream and side == 1:
                    bc[c] = ["CONVECTIVE", 1.0]           # the outlet of the cylinder cases
                elif stream >= 0:
                    bc[c] = ["DIRICHLET", 1.0 if f == stream else 0.0]  # inflow and free stream
                elif f == d:
                    bc[c] = ["DIRICHLET", 0.0]            # no flow through a wall (a normal NEUMANN changes D: PIB_ERR_SUP)
                elif kind == "sliding":
                    bc[c] = ["DIRICHLET", float(rng.uniform(-0.5, 0.5))]
                elif kind == "zero_gradient":
                    bc[c] = ["NEUMANN", float(rng.uniform(-0.05, 0.05))]
                else:
                    bc[c] = ["DIRICHLET", 0.0]
            bcs.append(bc)
    cfg = {"mesh": mesh, "flow": {"nu": float(rng.choice([0.01, 0.05])), "boundaryConditions": bcs},
           "parameters": {"dt": float(rng.choice([0.004, 0.01])), "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}}
    return cfg, periodic, stream


SEEDS = list(range(24))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_mesh_operators_bit_exact(seed):
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    cfg, per, _ = random_config(seed)
    m = omesh.create_mesh(cfg)
    dt, cnu = cfg["parameters"]["dt"], 0.5 * cfg["flow"]["nu"]
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=1)
    pinned = bool(seed % 2)
    if pinned:
        A = oops.pin_row0(A)
    s = LinSolverHIP("poisson", config_text=amgx_cfg())
    s.setPeriodic(per)
    s.assemblePoisson(n, w, dt, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, A.rowptr) and np.array_equal(cl, A.col) and np.array_equal(vl, A.val)
    x = np.random.default_rng(seed).uniform(-1, 1, A.n_rows)
    y = np.empty_like(x)
    s.matMult(x, y)
    assert np.array_equal(y, clib.spmv(A, x))
    s.destroy()
    V = oops.create_velocity_operator(L, dt, cnu)
    s = LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=500))
    s.setPeriodic(per)
    s.assembleVelocity(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, V.rowptr) and np.array_equal(cl, V.col) and np.array_equal(vl, V.val)
    us = np.random.default_rng(seed + 1).uniform(-1, 1, V.n_rows)
    b = clib.spmv(V, us)
    for free in (1, 0):  # the products of the solve from the mesh tables, then from the CSR: same iterates
        t = LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=500)
                         + f"pib_matrix_free_velocity={free}\npib_bicgstab_form=0\n")
        t.setPeriodic(per)
        t.assembleVelocity(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu)
        xv = np.zeros(V.n_rows)
        t.solve(xv, b)
        if free:
            x_free, it_free = xv, t.getIters()
        else:
            assert t.getIters() == it_free and np.array_equal(xv, x_free)
        t.destroy()
    assert np.linalg.norm(x_free - us) <= 1e-9 * np.linalg.norm(us)
    s.destroy()


@pytest.mark.parametrize("seed", SEEDS[:16])
def test_random_mesh_divergence_fold_bit_exact(seed):
    """createDivergence's ghost fold (createdivergence.cpp:231-242; SURVEY 8 a-6, on the device since round 5): on the random
    meshes, one to three wall-bounded faces get a NEUMANN condition on their NORMAL component (a0 = 1: the boundary cells' rows of
    D lose that face), and D (BN) G from the device's chain of sparse products -- BN order 1 and 2, pinned or not -- is the
    oracle's bit for bit; so is its product with a random vector.  (Whether the pinned matrix is then non-singular depends on
    where the faces are: operators only here, the time step has its own case.)"""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    cfg, per, _ = random_config(seed)
    rng = np.random.default_rng(5000 + seed)
    dim = len(cfg["mesh"])
    faces = [(d, side) for d in range(dim) for side in (0, 1) if not per[d]]
    if not faces:
        pytest.skip("every direction periodic: no ghost point to fold")
    for k in rng.choice(len(faces), size=min(len(faces), int(rng.integers(1, 4))), replace=False):
        d, side = faces[int(k)]
        cfg["flow"]["boundaryConditions"][2 * d + side]["uvw"[d]] = ["NEUMANN", float(rng.uniform(-0.05, 0.05))]
    m = omesh.create_mesh(cfg)
    dt, cnu = cfg["parameters"]["dt"], 0.5 * cfg["flow"]["nu"]
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    a0 = _a0_table(m)
    assert any(a0[f][2 * f + e] == 1.0 for f in range(m.dim) for e in (0, 1))
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    for order in (1, 2):
        _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=order)
        pinned = bool((seed + order) % 2)
        if pinned:
            A = oops.pin_row0(A)
        s = LinSolverHIP("poisson", config_text=amgx_cfg())
        s.setPeriodic(per)
        s.assemblePoissonBN(n, w, m.min[: m.dim], m.max[: m.dim], a0, dt, cnu, order, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
        rp, cl, vl = s.getCSR()
        assert np.array_equal(rp, A.rowptr) and np.array_equal(cl, A.col) and np.array_equal(vl, A.val)
        x = np.random.default_rng(seed).uniform(-1, 1, A.n_rows)
        y = np.empty_like(x)
        s.matMult(x, y)
        assert np.array_equal(y, clib.spmv(A, x))
        s.destroy()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_mesh_time_steps_match_oracle(seed):
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg, per, stream = random_config(seed)
    m = omesh.create_mesh(cfg)
    dt, nu = cfg["parameters"]["dt"], cfg["flow"]["nu"]
    pinned = bool(seed % 2) or stream >= 0  # (an outlet's net flux makes rhs2 incompatible with the constant null space)
    ref = ons.NavierStokes(m, dt, nu, pinned=pinned, vtol=1e-14, ptol=1e-13)
    rng = np.random.default_rng(77 + seed)
    U0, p0 = 0.1 * rng.uniform(-1, 1, m.UN), 0.1 * rng.uniform(-1, 1, m.pN)
    if stream >= 0:
        off = sum(int(np.prod(m.n[f])) for f in range(stream))
        U0[off: off + int(np.prod(m.n[stream]))] += 1.0
    if pinned:
        p0[0] = 0.0
    ref.set_state(U0, p0)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=AMGX_P if pinned else KSP_P)
    assert (s.UN, s.pN) == (m.UN, m.pN)
    s.setState(U0, p0)
    for step in range(2):
        ref.advance()
        s.advance()
        U, p, r1, r2 = s.getState(rhs=True)
        scale = np.abs(ref.last_rhs1).max()
        if step == 0 and not any(per):
            assert np.array_equal(r1, ref.last_rhs1)
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-9 * scale
        assert np.abs(r2 - ref.last_rhs2).max() <= 1e-9 * max(np.abs(ref.last_rhs2).max(), 1e-30) + 1e-14
        assert np.abs(U - ref.U).max() <= 1e-9 * np.abs(ref.U).max()
        dp = (p - p.mean()) - (ref.p - ref.p.mean())
        assert np.abs(dp).max() <= 1e-8 * max(np.abs(ref.p - ref.p.mean()).max(), 1e-30)
    s.destroy()


@pytest.mark.parametrize("seed", [0, 3, 6, 9, 12, 15, 18, 21])
def test_random_wide_3d_meshes_first_step_rhs_through_the_march(seed):
    """The three-component march of the explicit terms (navierstokes.hip k_ns_rhs_march: 3-D meshes of 32 cells and more along x)
    on the sweep's boundary sets -- walls, sliding walls, zero-gradient components, a convective outlet, periodic directions -- and
    sub-domain layouts, refined 12 x 3 x 3 so that the march's tiles (partial ones included) cover them: rhs1 of the first step is
    the oracle's bit for bit (across a periodic seam a shell row sums its wrapped neighbour in another place: 1e-13 there)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg, per, stream = random_config(seed)
    assert len(cfg["mesh"]) == 3
    for d, factor in enumerate((12, 3, 3)):
        for sub in cfg["mesh"][d]["subDomains"]:
            sub["cells"] *= factor
            sub["stretchRatio"] = float(sub["stretchRatio"] ** (1.0 / factor))
    m = omesh.create_mesh(cfg)
    assert m.n[3][0] >= 48
    dt, nu = cfg["parameters"]["dt"], cfg["flow"]["nu"]
    pinned = bool(seed % 2) or stream >= 0
    ref = ons.NavierStokes(m, dt, nu, pinned=pinned)
    rng = np.random.default_rng(177 + seed)
    U0, p0 = 0.1 * rng.uniform(-1, 1, m.UN), 0.1 * rng.uniform(-1, 1, m.pN)
    if stream >= 0:
        off = sum(int(np.prod(m.n[f])) for f in range(stream))
        U0[off: off + int(np.prod(m.n[stream]))] += 1.0
    if pinned:
        p0[0] = 0.0
    ref.set_state(U0, p0)
    ref.advance(velocity_rhs_only=True)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=AMGX_P if pinned else KSP_P)
    s.setState(U0, p0)
    s.advance()
    r1 = s.getState(rhs=True)[2]
    if any(per):
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-13 * np.abs(ref.last_rhs1).max()
        assert (r1 != ref.last_rhs1).mean() <= 0.2  # (the seam's shell rows only)
    else:
        assert np.array_equal(r1, ref.last_rhs1)
    s.destroy()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_mesh_time_steps_on_slabs(seed):
    """The same configurations on 2 or 3 slabs (loopback ranks on the one GPU) against the single-rank engine: rhs1 of the
    first step bit for bit (to rounding with a periodic direction), the fields after two steps to the solver tolerance."""
    from petibm_amd.navierstokes import NavierStokesSolver
    from test_gpu_multirank_loopback import _run_ranks
    cfg, per, stream = random_config(seed)
    m = omesh.create_mesh(cfg)
    P = 2 if int(m.n[3][m.dim - 1]) < 6 or seed % 2 else 3
    pinned = bool(seed % 2) or stream >= 0
    pcfg = AMGX_P if pinned else KSP_P
    one = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=pcfg)
    rng = np.random.default_rng(77 + seed)
    U0, p0 = 0.1 * rng.uniform(-1, 1, m.UN), 0.1 * rng.uniform(-1, 1, m.pN)
    if stream >= 0:
        off = sum(int(np.prod(m.n[f])) for f in range(stream))
        U0[off: off + int(np.prod(m.n[stream]))] += 1.0
    if pinned:
        p0[0] = 0.0
    one.setState(U0, p0)
    one.advance(1)
    _, _, rhs1, rhs2 = one.getState(rhs=True)
    one.advance(1)
    U2, p2 = one.getState()

    def rank_fn(r, uid):
        s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=pcfg, device=0, rank=r, nranks=P, uid=uid)
        s.setState(s.ownedVelocity(U0), s.ownedPressure(p0))
        s.advance(1)
        a = s.getState(rhs=True)
        s.advance(1)
        b = s.getState()
        cut = [s.ownedVelocity(x) for x in (rhs1, U2)] + [s.ownedPressure(x) for x in (rhs2, p2)]
        s.destroy()
        return a, b, cut

    res = _run_ranks(P, rank_fn)
    for (_, _, r1, r2), (Ub, pb), (crhs1, cU2, crhs2, cp2) in res:
        if any(per):
            assert np.abs(r1 - crhs1).max() <= 1e-13 * np.abs(rhs1).max()
        else:
            assert np.array_equal(r1, crhs1)
        assert np.allclose(r2, crhs2, rtol=0, atol=1e-11 * max(1.0, np.abs(rhs2).max()))
        assert np.allclose(Ub, cU2, rtol=0, atol=1e-9 * max(1.0, np.abs(U2).max()))
        if pinned:
            assert np.allclose(pb, cp2, rtol=0, atol=1e-8 * max(1.0, np.abs(p2).max()))
    if not pinned:
        pg = np.concatenate([r[1][1] for r in res])
        assert np.allclose(pg - pg.mean(), p2 - p2.mean(), rtol=0, atol=1e-8 * max(1.0, np.abs(p2).max()))
    one.destroy()


@pytest.mark.parametrize("seed", SEEDS[:16])
def test_random_bodies_ib_operators_bit_identical(seed):
    """Immersed-boundary operators (createDelta / E / H / E BN H, src/operators/createdelta.cpp:34-208) for random
    Lagrangian points on the random meshes: anywhere in the box -- next to walls (clipped supports), across periodic
    seams, in the stretched regions -- with either kernel."""
    from oracle import ibm
    from petibm_amd.navierstokes import DecoupledIBPMSolver
    from test_gpu_ibm import AMGX_P as IB_P
    # (an iterative forces solver without a preconditioner: random points closer than a cell make E BN H singular -- a zero
    # pivot for the direct solver -- and a point whose support holds no point of a component has a zero diagonal entry)
    FORCES = "-forces_ksp_type cg\n-forces_pc_type none\n-forces_ksp_atol 1.0E-12\n-forces_ksp_rtol 0.0\n"
    cfg, per, stream = random_config(seed)
    rng = np.random.default_rng(500 + seed)
    kernel = "PESKIN_2002" if seed % 2 else "ROMA_ET_AL_1999"
    cfg["parameters"]["delta"] = kernel
    m = omesh.create_mesh(cfg)
    lo, hi = np.array(m.min[: m.dim]), np.array(m.max[: m.dim])
    bodies = []
    for _ in range(int(rng.integers(1, 3))):
        npts = int(rng.integers(3, 20))
        bodies.append(lo + (hi - lo) * rng.uniform(0.02, 0.98, (npts, m.dim)))
    dt = cfg["parameters"]["dt"]
    ref = ibm.create_ib_operators(m, bodies, dt, kernel)
    s = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=IB_P.format(tol=1e-12), forces_cfg=FORCES)
    assert s.nf == ref["E"].n_rows
    for name in ("delta", "E", "EBNH"):
        nr, rp, cl, vl = s.getOperator(name)
        r = ref[name]
        assert nr == r.n_rows and np.array_equal(rp, r.rowptr) and np.array_equal(cl, r.col), name
        assert np.array_equal(vl, r.val), name
    nr, rp, cl, vl, ids = s.getOperator("H")
    H = ref["H"]
    live = np.flatnonzero(np.diff(H.rowptr))
    assert np.array_equal(ids, live) and np.array_equal(rp, H.rowptr[np.append(live, H.n_rows)])
    assert np.array_equal(cl, H.col) and np.array_equal(vl, H.val)
    s.destroy()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_mesh_drop_in_route_recovers_the_structure(seed):
    """Nothing but setMatrix (an unchanged PetIBM: linsolveramgx.cpp:84, navierstokes.cpp:345, 357): the mesh structure of
    the oracle's Poisson matrix -- sizes, periodic directions, null-space convention -- and of its velocity matrix come out
    of the entries, or the matrix is left to the CSR kernels; either way the solves meet the residual contract."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    cfg, per, _ = random_config(seed)
    m = omesh.create_mesh(cfg)
    dt, cnu = cfg["parameters"]["dt"], 0.5 * cfg["flow"]["nu"]
    n = tuple(int(v) for v in m.n[3][: m.dim])
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=1)
    pinned = bool(seed % 2)
    if pinned:
        A = oops.pin_row0(A)
    xs = np.random.default_rng(seed).uniform(-1, 1, A.n_rows)
    if pinned:
        xs[0] = 0.0
    else:
        xs -= xs.mean()
    b = clib.spmv(A, xs)
    amg = "prec:cycle=V\nprec:presweeps=1\nprec:postsweeps=1\nprec:smoother(smooth)=BLOCK_JACOBI\nsmooth:relaxation_factor=0.9\npib_sweep_pairs=0\n"
    s = LinSolverHIP("poisson", config_text=amgx_cfg(pc="AMG", tol=1e-10, extra=amg + "pib_initial_guess_nonzero=0\n"))
    s.setMatrix(A)
    st = s.gridStructure()
    assert st is not None and st["detected"], "an operator assembled by createDivergence / createGradient must be recognised"
    assert st["dim"] == m.dim and tuple(st["n"]) == n
    assert st["nullspace"] == (capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    assert s.getReason() > 0 and s.getIters() <= 40
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    s.destroy()
    V = oops.create_velocity_operator(L, dt, cnu)
    bv = np.random.default_rng(seed + 3).uniform(-1, 1, V.n_rows)
    out = []
    for mf in (1, 0):
        t = LinSolverHIP("velocity", config_text=amgx_cfg(solver="PBICGSTAB", pc="BLOCK_JACOBI", tol=1e-12, conv="ABSOLUTE", maxit=500,
                                                          extra=f"pib_matrix_free_velocity={mf}\n"))
        t.setMatrix(V)
        sv = t.velocityStructure()
        if mf and sv is not None:
            assert sv["detected"] and sv["dim"] == m.dim and sv["n"] == n and sv["periodic"] == tuple(per)
        xv = np.zeros(V.n_rows)
        t.solve(xv, bv)
        out.append((xv, t.getIters(), sv is not None))
        t.destroy()
    assert out[0][1] == out[1][1]
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * np.abs(out[1][0]).max()
    assert np.linalg.norm(bv - clib.spmv(V, out[0][0])) <= 1e-11 * np.linalg.norm(bv)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_mesh_drop_in_route_on_slabs(seed):
    """The same on 2 or 3 ranks, rows cut into z-slabs of the natural ordering -- what a (1, 1, P) process grid gives
    MatMPIAIJGetLocalMat (global columns, across the seam of a periodic slab axis too); PETSC_DECIDE's boxes from 4 ranks
    up are tests/test_gpu_dmda_boxes.py.  The structure is gathered from the ranks' lines of entries."""
    from petibm_amd import capi
    import slab_plans as partition
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_multirank_loopback import _cfg, _run_ranks
    cfg, per, _ = random_config(seed)
    m = omesh.create_mesh(cfg)
    dt, cnu = cfg["parameters"]["dt"], 0.5 * cfg["flow"]["nu"]
    n = tuple(int(v) for v in m.n[3][: m.dim])
    P = 2 if n[-1] < 6 or seed % 2 else 3
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=1)
    pinned = bool(seed % 2)
    if pinned:
        A = oops.pin_row0(A)
    xs = np.random.default_rng(seed).uniform(-1, 1, A.n_rows)
    if pinned:
        xs[0] = 0.0
    else:
        xs -= xs.mean()
    b = clib.spmv(A, xs)
    plans = partition.all_plans(n, P)

    def rank_fn(r, uid):
        pl = plans[r]
        s = LinSolverHIP("poisson", config_text=_cfg("AMG", tol=1e-11), rank=r, nranks=P, uid=uid, device=0)
        r0, r1 = pl.row0, pl.row0 + pl.n_local
        p0, p1 = A.rowptr[r0], A.rowptr[r1]
        local = oops.CSR(pl.n_local, A.n_cols, A.rowptr[r0:r1 + 1] - p0, A.col[p0:p1], A.val[p0:p1])
        s.setMatrix(local, row0=r0, n_global=A.n_rows)
        st = s.gridStructure()
        x = np.zeros(pl.n_local)
        s.solve(x, np.ascontiguousarray(b[r0:r1]))
        out = x, s.getIters(), st, s.getReason()
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    for r in res:
        assert r[3] > 0 and r[2] is not None and r[2]["detected"] and tuple(r[2]["n"]) == n
        assert r[2]["nullspace"] == (capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    x = np.concatenate([r[0] for r in res])
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-11 * np.linalg.norm(b)
    assert len({r[1] for r in res}) == 1 and res[0][1] < 60


def _multigrid_mesh(seed):
    """Pressure grids for the multigrid's forms: 2-D and 3-D, one to three sub-domains per direction with their own
    stretching, sizes that leave several levels (odd ones included), periodic directions at random."""
    rng = np.random.default_rng(7000 + seed)
    dim = 2 if seed % 2 else 3
    lo_hi = (24, 90) if dim == 2 else (10, 40)
    n, widths, per = [], [], []
    for d in range(dim):
        nsub = int(rng.integers(1, 4))
        total = int(rng.integers(*lo_hi))
        cuts = np.sort(rng.choice(np.arange(1, total), size=nsub - 1, replace=False)) if nsub > 1 else np.array([], dtype=int)
        cells = np.diff(np.concatenate(([0], cuts, [total])))
        w = []
        for c in cells:
            r = float(rng.choice([1.0, 1.0, 1.04, 0.95, 1.1]))
            first = float(rng.uniform(0.5, 1.5))
            w.extend(first * r ** np.arange(int(c)))
        w = np.asarray(w) / np.sum(w)
        p = bool(rng.uniform() < 0.3)
        if p:
            w = np.full(total, 1.0 / total)  # (a periodic direction: uniform, as the reference's periodic cases are)
        n.append(total)
        widths.append(w)
        per.append(p)
    return dim, n, widths, per


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("PIB_FUZZ_SEEDS", "16")))))
def test_random_mesh_multigrid_forms_are_bit_identical(seed):
    """The V-cycle's latency-bound bottom has several forms -- one launch per phase, the fused small-level kernels
    (k_small_down / k_small_up), the single-workgroup tail in HBM or in LDS, one or several cells per thread: on random
    meshes every form must give the bits of the per-phase launches, and the solve must reach its tolerance."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_parity import gmg_cfg
    dim, n, w, per = _multigrid_mesh(seed)
    rng = np.random.default_rng(seed)
    pre, post = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    dt = 0.01
    N = int(np.prod(n))
    b = rng.uniform(-1, 1, N)
    b -= b.mean()
    out = []
    res = None
    forms = ((0, 0, 0), (1, 0, 0), (1, 1024, 0), (1, 1024, 1), (0, 1024, 1), (1, 4096, 1), (1, 200, 1))
    for fuse, tail, lds in forms:
        s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=f"pib_fuse_small_levels={fuse}\npib_coarse_tail={tail}\n"
                                                                                 f"pib_coarse_tail_lds={lds}\n"))
        if any(per):
            s.setPeriodic(per)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(N)
        s.solve(x, b)
        assert s.getReason() > 0, (seed, fuse, tail, lds)
        out.append((x, s.getIters(), s.getResidualHistory().copy()))
        if res is None:  # the true residual, with the assembled operator (the device's own CSR product)
            Ax = np.zeros(N)
            s.matMult(x, Ax)
            res = np.linalg.norm(b - Ax) / np.linalg.norm(b)
        s.destroy()
    assert res <= 2e-10
    for form, (x, its, h) in zip(forms[1:], out[1:]):
        assert its == out[0][1], (seed, form)
        assert np.array_equal(h, out[0][2]), (seed, form)
        assert np.array_equal(x, out[0][0]), (seed, form)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("PIB_FUZZ_SEEDS", "12")))))
def test_random_mesh_general_restriction_march_is_bit_identical(seed):
    """gmg.hip k_restrict_zmarch (restriction of ANY aggregation -- lone cells among the pairs of a stretched mesh -- as a z-march:
    a wave per coarse row, a fine plane's rows loaded once) against k_restrict_rows on random 3-D meshes, stretched and periodic in
    x / y at random: the same bits in the iterates and the residual histories."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_parity import gmg_cfg
    rng = np.random.default_rng(1000 + seed)
    n = [int(rng.integers(20, 70)), int(rng.integers(10, 40)), int(rng.integers(9, 40))]
    w = []
    for d in range(3):
        r = float(rng.choice([1.0, 1.04, 1.1, 0.93]))
        ww = r ** np.arange(n[d])
        if rng.random() < 0.5:  # a refined block in the middle, stretched far field on both sides
            m = n[d] // 3
            ww = np.concatenate([1.08 ** np.arange(m, 0, -1), np.ones(n[d] - 2 * m), 1.08 ** np.arange(1, m + 1)])
        w.append(ww / ww.sum())
    per = [bool(rng.random() < 0.3), bool(rng.random() < 0.3), False]
    pre, post = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    dt = 0.01
    N = int(np.prod(n))
    b = rng.uniform(-1, 1, N)
    b -= b.mean()
    out = []
    for march in (0, 1):
        s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra=f"pib_march_min_cells=0\npib_march={march}\n"
                                                                                 "pib_fuse_small_levels=0\npib_coarse_tail=0\n"))
        if any(per):
            s.setPeriodic(per)
        s.assemblePoisson(n, w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(N)
        s.solve(x, b)
        assert s.getReason() > 0, (seed, march)
        out.append((x, s.getIters(), s.getResidualHistory().copy()))
        s.destroy()
    assert out[0][1] == out[1][1], seed
    assert np.array_equal(out[0][2], out[1][2]), seed
    assert np.array_equal(out[0][0], out[1][0]), seed


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("PIB_FUZZ_SEEDS", "12")))))
def test_random_paired_levels_marching_kernels_are_bit_identical(seed):
    """The LDS-tiled marching kernels of the large levels (pre-smoothing pair, residual + restriction in one march,
    prolongation + post-smoothing in one march, the XCD bands) on random fully paired meshes -- mild stretching, walls and
    periodic directions at random, V(1,1) ... V(2,2) -- against the streaming / row kernels: the same bits."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_parity import gmg_cfg
    rng = np.random.default_rng(9000 + seed)
    n = [(128, 32, 24), (128, 16, 40), (256, 16, 16), (128, 48, 32), (128, 32, 64)][int(rng.integers(0, 5))]
    per = [bool(rng.uniform() < 0.35) for _ in range(3)]
    ratios = [1.0 if per[d] else float(rng.choice([1.0, 1.002, 0.997, 1.006])) for d in range(3)]
    pre, post = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    m = omesh.create_mesh(omesh.periodic_config(n, per, ratios=ratios))
    w = [m.dL[3][d].true for d in range(m.dim)]
    dt = 0.01
    N = int(np.prod(n))
    b = rng.uniform(-1, 1, N)
    b -= b.mean()
    out = []
    off = "pib_march=0\npib_fuse_presmooth=0\npib_fuse_prolong=0\n"
    for extra in ("", "pib_fuse_post_pair=0\n", "pib_fuse_residual_restrict=0\n", off):
        s = LinSolverHIP("poisson", config_text=gmg_cfg(pre=pre, post=post, extra="pib_march_min_cells=0\n" + extra))
        if any(per):
            s.setPeriodic(per)
        s.assemblePoisson(list(n), w, dt, capi.NULLSPACE_CONSTANT)
        x = np.zeros(N)
        s.solve(x, b)
        assert s.getReason() > 0, (seed, extra)
        out.append((x, s.getIters(), s.getResidualHistory().copy()))
        s.destroy()
    # the fused marches against the kernels they replace: bit for bit; against the streaming kernels (which group the Krylov
    # sums differently): the same iteration count and the iterates to rounding
    for x, its, h in out[1:3]:
        assert its == out[0][1] and np.array_equal(h, out[0][2]) and np.array_equal(x, out[0][0]), seed
    assert out[3][1] == out[0][1], seed
    assert np.allclose(out[3][2], out[0][2], rtol=1e-9), seed
    assert np.abs(out[3][0] - out[0][0]).max() <= 1e-10 * np.abs(out[0][0]).max(), seed
