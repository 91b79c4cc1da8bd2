"""Device-resident time step against the oracle restatement of NavierStokesSolver::advance
and against the physics data the reference ships (Ghia et al. 1982, examples/data)."""
import copy
import json
import os

import numpy as np
import pytest

from oracle import mesh as omesh, navierstokes as ons

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_test_vectors.json")))


def cavity(n, nu=0.01, dt=0.01, lid=1.0, stretched=False):
    cfg = omesh.uniform_config(n, lid=lid)
    if stretched:
        for d, ax in enumerate(cfg["mesh"]):
            h = n[d] // 2
            ax["subDomains"] = [{"end": 0.5, "cells": h, "stretchRatio": 1.0 / (1.05 + 0.02 * d)},
                                {"end": 1.0, "cells": n[d] - h, "stretchRatio": 1.05 + 0.02 * d}]
    cfg["flow"]["nu"] = nu
    cfg["parameters"] = {"dt": dt, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON"}
    return cfg


def moving_walls_3d():
    cfg = cavity((10, 9, 8), nu=0.02, dt=0.005, stretched=True)
    bcs = cfg["flow"]["boundaryConditions"]
    bcs[0]["v"] = ["DIRICHLET", 0.3]       # xMinus: tangential motion
    bcs[2]["w"] = ["DIRICHLET", -0.2]      # yMinus
    bcs[4]["u"] = ["DIRICHLET", 0.25]      # zMinus
    bcs[0]["u"] = ["DIRICHLET", 0.1]       # inflow through xMinus ...
    bcs[1]["u"] = ["DIRICHLET", 0.1]       # ... leaves through xPlus (same area: compatible with the Neumann Poisson problem)
    bcs[1]["v"] = ["NEUMANN", 0.0]         # zero-gradient tangential components at the outlet
    bcs[1]["w"] = ["NEUMANN", 0.05]
    return cfg


def convective_outlet(n, stretched=True):
    """Uniform stream through the box, convective outlet on xPlus (the boundary set of the reference's cylinder
    cases: examples/ibpm/cylinder2dRe40/config.yaml -- inflow / free-stream DIRICHLET, outlet CONVECTIVE)."""
    cfg = cavity(n, nu=0.02, dt=0.004, lid=0.0, stretched=stretched)
    names = ["u", "v", "w"][: len(n)]
    for bc in cfg["flow"]["boundaryConditions"]:
        for c in names:
            free = 1.0 if c == "u" else 0.0
            bc[c] = ["CONVECTIVE", 1.0] if bc["location"] == "xPlus" else ["DIRICHLET", free]
    return cfg


AMGX_P = ("config_version=2\nsolver(solv)=PCG\nsolv:max_iters=500\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
          "solv:tolerance=1e-13\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=AMG\nprec:cycle=V\n"
          "prec:presweeps=1\nprec:postsweeps=1\nprec:coarsest_sweeps=2\nprec:smoother(smooth)=BLOCK_JACOBI\n"
          "smooth:relaxation_factor=0.9\n")
KSP_P = ("-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-13\n-poisson_ksp_rtol 0.0\n-poisson_ksp_max_it 500\n"
         "-poisson_ksp_norm_type unpreconditioned\n-poisson_pc_type gamg\n-poisson_pib_smoother JACOBI\n")
VEL = ("config_version=2\nsolver(solv)=PBICGSTAB\nsolv:max_iters=1000\nsolv:monitor_residual=1\nsolv:convergence=ABSOLUTE\n"
       "solv:tolerance=1e-14\nsolv:norm=L2\nsolv:store_res_history=1\nsolv:preconditioner(prec)=BLOCK_JACOBI\n"
       "prec:relaxation_factor=1.0\n")


@pytest.mark.parametrize("case,pinned", [("2d_stretched", False), ("2d_stretched", True), ("3d_moving_walls", False),
                                         ("3d_uniform", True), ("2d_convective_outlet", True),
                                         ("3d_convective_outlet", True)])
def test_time_step_matches_oracle(case, pinned):
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = {"2d_stretched": cavity((14, 12), stretched=True), "3d_moving_walls": moving_walls_3d(),
           "3d_uniform": cavity((8, 8, 8), nu=0.05, dt=0.01),
           "2d_convective_outlet": convective_outlet((16, 12)),
           "3d_convective_outlet": convective_outlet((10, 8, 6))}[case]
    m = omesh.create_mesh(cfg)
    dt, nu = cfg["parameters"]["dt"], cfg["flow"]["nu"]
    ref = ons.NavierStokes(m, dt, nu, pinned=pinned, vtol=1e-14, ptol=1e-13)
    rng = np.random.default_rng(9)
    U0 = 0.1 * rng.uniform(-1, 1, m.UN)
    p0 = 0.1 * rng.uniform(-1, 1, m.pN)
    if "convective" in case:
        U0[: int(np.prod(m.n[0]))] += 1.0  # perturbed free stream
    if pinned:
        p0[0] = 0.0
    ref.set_state(U0, p0)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=AMGX_P if pinned else KSP_P)
    assert (s.UN, s.pN) == (m.UN, m.pN)
    s.setState(U0, p0)
    for step in range(3):
        ref.advance()
        s.advance()
        U, p, r1, r2 = s.getState(rhs=True)
        if step == 0:
            # explicit part, first step: same operations in the same order -> identical bits
            assert np.array_equal(r1, ref.last_rhs1)
        scale = np.abs(ref.last_rhs1).max()
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-9 * scale
        assert np.abs(r2 - ref.last_rhs2).max() <= 1e-9 * max(np.abs(ref.last_rhs2).max(), 1e-30) + 1e-14
        assert np.abs(U - ref.U).max() <= 1e-9 * np.abs(ref.U).max()
        dp = (p - p.mean()) - (ref.p - ref.p.mean())
        assert np.abs(dp).max() <= 1e-8 * max(np.abs(ref.p - ref.p.mean()).max(), 1e-30)
    ite, vi, vr, pi, pr = s.linSolversInfo()
    assert ite == 3 and 0 < vi < 50 and 0 < pi < 60 and vr <= 1e-14 and pr <= 1e-13
    # discrete continuity after the projection: D u + Dbc = 0
    from oracle import clib
    div = clib.spmv(ref.D, U) + ons.divergence_correction(m, ref.ghosts)
    if pinned:
        div[0] = 0.0  # the pinned cell absorbs any net flux through the boundaries (rhs2[0] = 0, :553-558)
    assert np.abs(div).max() <= 1e-10 * np.abs(ref.D.val).max()
    s.destroy()


@pytest.mark.parametrize("n", [(300, 70, 6), (130, 150, 9), (70, 20, 40), (30, 150, 9)])
def test_first_step_rhs_on_wide_3d_meshes_is_bit_identical(n):
    """navierstokes.hip where the launch geometry of the explicit terms has something to get wrong.  The three-component march
    (k_ns_rhs_march, 32 cells and more along x): partial 64 x 8 tiles in x and y, tile rows dealt to the XCDs in bands, one and two
    z chunks of 32 planes, the last cell plane on which only u and v have interior points.  The per-component kernel
    (k_ns_rhs_velocity, the (30, 150, 9) mesh): rows in bands, two planes per workgroup with an odd plane count.  Every interior
    point exactly once, the bits of the oracle's explicit terms (navierstokes.cpp:432-521)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = cavity(n, nu=0.02, dt=0.005)
    m = omesh.create_mesh(cfg)
    ref = ons.NavierStokes(m, 0.005, 0.02, pinned=False)
    rng = np.random.default_rng(11)
    U0 = 0.1 * rng.uniform(-1, 1, m.UN)
    p0 = 0.1 * rng.uniform(-1, 1, m.pN)
    ref.set_state(U0, p0)
    ref.advance(velocity_rhs_only=True)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=KSP_P)
    s.setState(U0, p0)
    s.advance()
    r1 = s.getState(rhs=True)[2]
    assert np.array_equal(r1, ref.last_rhs1)
    s.destroy()


def neumann_outlet(n, stretched=True, outlet="xPlus"):
    """A uniform stream towards `outlet`, which is a zero-gradient (NEUMANN) boundary for EVERY component -- the normal one
    included: a0 = 1 then folds into D (createdivergence.cpp:231-242, singleboundaryneumann.cpp:27-28), the outlet cells' rows of
    D lose their x faces and DBNG its symmetry there; Dirichlet free stream elsewhere."""
    cfg = cavity(n, nu=0.02, dt=0.004, lid=0.0, stretched=stretched)
    names = ["u", "v", "w"][: len(n)]
    stream = 1.0 if outlet == "xPlus" else -1.0
    for bc in cfg["flow"]["boundaryConditions"]:
        for c in names:
            free = stream if c == "u" else 0.0
            bc[c] = ["NEUMANN", 0.0] if bc["location"] == outlet else ["DIRICHLET", free]
    return cfg


@pytest.mark.parametrize("outlet", ["xPlus", "xMinus"])
@pytest.mark.parametrize("n", [(16, 12), (10, 8, 6)])
@pytest.mark.parametrize("order,pinned", [(1, True), (1, False), (2, True)])
def test_divergence_with_the_neumann_fold_gives_the_oracles_poisson_operator(n, order, pinned, outlet):
    """SURVEY 8 a-6 (round 5; a PIB_ERR_SUP until then): createDivergence's ghost fold on the device.  D itself is a factor of
    the chain of sparse products (bn.hip k_bn_divergence): D (BN) G with the folded D is the oracle's, entry for entry and bit for
    bit, for BN order 1 (BN = dt I) and 2; the matrix is NOT symmetric at the Neumann face."""
    from petibm_amd import capi
    from petibm_amd.linsolver import LinSolverHIP
    from test_gpu_parity import _a0_table, amgx_cfg
    from oracle import operators as oops
    cfg = neumann_outlet(n, outlet=outlet)
    m = omesh.create_mesh(cfg)
    dt, cnu = cfg["parameters"]["dt"], 0.5 * cfg["flow"]["nu"]
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=order)
    if pinned:
        A = oops.pin_row0(A)
    a0 = _a0_table(m)
    assert a0[0][1 if outlet == "xPlus" else 0] == 1.0  # u at the outlet: the fold
    s = LinSolverHIP("poisson", config_text=amgx_cfg())
    nn = [int(v) for v in m.n[3][: m.dim]]
    s.assemblePoissonBN(nn, [m.dL[3][d].true for d in range(m.dim)], m.min[: m.dim], m.max[: m.dim], a0, dt, cnu, order,
                        capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, A.rowptr) and np.array_equal(cl, A.col)
    assert np.array_equal(vl, A.val)
    # not symmetric: the last cell of the first grid line couples to its -x neighbour, the neighbour's row holds another value
    dense = np.zeros((m.pN, m.pN))
    for r in range(m.pN):
        dense[r, cl[rp[r]:rp[r + 1]]] = vl[rp[r]:rp[r + 1]]
    if not pinned:
        assert np.abs(dense - dense.T).max() > 1e-3 * np.abs(dense).max()
    s.destroy()


@pytest.mark.parametrize("n", [(16, 12), (10, 8, 6)])
def test_time_step_with_a_neumann_outlet_on_the_normal_component_matches_oracle(n, capfd):
    """... and the time step: k_ns_rhs_poisson applies the folded row of D, the Poisson operator is the chain's, and the solver
    file's CG is replaced by BiCGStab with the same (multigrid) preconditioner -- the matrix is not symmetric -- with a note on
    stderr.  Three steps against the oracle (which solves the same system with Jacobi-BiCGStab).
    The outlet is the xMinus face ON PURPOSE: with the fold, the LEFT null vector of D (dt I) G lives on the outlet cells alone, so
    pinning row 0 (navierstokes.cpp:414-420) only removes the singularity when cell 0 is one of them -- with the outlet on xPlus
    the pinned matrix keeps a singular value of 4e-19 (numpy SVD of the oracle's matrix) and the pressure correction, hence the
    projected velocity, is not unique: a property of the reference's discretisation, not of either implementation (the operator
    itself is compared bit for bit for both faces above)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = neumann_outlet(n, outlet="xMinus")
    m = omesh.create_mesh(cfg)
    dt, nu = cfg["parameters"]["dt"], cfg["flow"]["nu"]
    ref = ons.NavierStokes(m, dt, nu, pinned=True, vtol=1e-14, ptol=1e-13)
    assert ref.nonsymmetric
    rng = np.random.default_rng(9)
    U0 = 0.1 * rng.uniform(-1, 1, m.UN)
    U0[: int(np.prod(m.n[0]))] -= 1.0
    p0 = 0.1 * rng.uniform(-1, 1, m.pN)
    p0[0] = 0.0
    ref.set_state(U0, p0)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=AMGX_P)
    assert "replaced by BiCGStab" in capfd.readouterr().err
    # ... and through the API, not on stderr only (pib_describe): the method that runs, and the departure from the file
    d = s.describeSolver("poisson")
    assert "method=bicgstab" in d.splitlines()[0] and 'type="NVIDIA AmgX"' in d.splitlines()[0]
    assert any(ln.startswith("departure: method: the file's cg runs as bicgstab") for ln in d.splitlines()[1:])
    assert "method=bicgstab" in s.describeSolver("velocity").splitlines()[0] and "departure: method" not in s.describeSolver("velocity")
    s.setState(U0, p0)
    for step in range(3):
        ref.advance()
        s.advance()
        U, p, r1, r2 = s.getState(rhs=True)
        if step == 0:
            assert np.array_equal(r1, ref.last_rhs1)
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-9 * np.abs(ref.last_rhs1).max()
        assert np.abs(r2 - ref.last_rhs2).max() <= 1e-9 * max(np.abs(ref.last_rhs2).max(), 1e-30) + 1e-14
        assert np.abs(U - ref.U).max() <= 1e-8 * np.abs(ref.U).max()
        assert np.abs(p - ref.p).max() <= 1e-7 * max(np.abs(ref.p).max(), 1e-30)
    ite, vi, vr, pi, pr = s.linSolversInfo()
    assert ite == 3 and 0 < pi < 200 and pr <= 1e-13
    s.destroy()


def test_unsupported_boundary_conditions_are_errors():
    from petibm_amd.capi import PibError, ERR_ARG_WRONG
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = cavity((8, 8))
    cfg["flow"]["boundaryConditions"][1]["u"] = ["PERIODIC", 0.0]  # one end / one component only (misc.cpp:32-82)
    with pytest.raises(PibError) as ei:
        NavierStokesSolver(cfg)
    assert ei.value.code == ERR_ARG_WRONG


def test_lid_driven_cavity_re100_matches_ghia():
    """The reference's own validation case (examples/navierstokes/liddrivencavity2dRe100: 32x32, nu = 0.01,
    dt = 0.01, 1000 steps, README compares the centre-line velocities with Ghia et al. 1982), run on the GPU."""
    from petibm_amd.navierstokes import NavierStokesSolver
    n = 32
    cfg = cavity((n, n), nu=0.01, dt=0.01)
    s = NavierStokesSolver(cfg)
    s.advance(1000)
    U, p = s.getState()
    u = U[: (n - 1) * n].reshape(n, n - 1)
    v = U[(n - 1) * n:].reshape(n - 1, n)
    yc = (np.arange(n) + 0.5) / n
    g = G["ghia_1982_re100_u_centerline"]
    ui = np.interp(g["y"][1:-1], yc, u[:, n // 2 - 1])       # u along the vertical centre line x = 0.5
    vi = np.interp(g["x"][1:-1], yc, v[n // 2 - 1, :])       # v along the horizontal centre line y = 0.5
    assert np.abs(ui - np.array(g["u"][1:-1])).max() < 0.006
    assert np.abs(vi - np.array(g["v"][1:-1])).max() < 0.008
    # and against the oracle stepping the same 40 steps from rest
    m = omesh.create_mesh(cfg)
    ref = ons.NavierStokes(m, 0.01, 0.01, pinned=False, vtol=1e-13, ptol=1e-12)
    t = NavierStokesSolver(cfg)
    for _ in range(40):
        ref.advance()
    t.advance(40)
    Ut, _ = t.getState()
    assert np.abs(Ut - ref.U).max() <= 1e-8
    s.destroy()
    t.destroy()


def test_baseline_config1_cavity_128_re100_matches_ghia():
    """BASELINE config 1 as stated: the 2-D lid-driven cavity on 128 x 128 cells at Re = 100 (the reference's
    liddrivencavity2dRe100 case -- nu = 0.01, dt = 0.01, CG + BiCGStab from its PETSc options files -- on the finer mesh),
    2000 steps to t = 20; centre-line velocities against Ghia et al. (1982)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    n = 128
    cfg = cavity((n, n), nu=0.01, dt=0.01)
    vel = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n"
           "-velocity_pc_type jacobi\n")
    poi = ("-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-06\n-poisson_ksp_rtol 0.0\n-poisson_ksp_max_it 1000\n"
           "-poisson_pc_type gamg\n")
    s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
    s.advance(2000)
    U, p = s.getState()
    u = U[: (n - 1) * n].reshape(n, n - 1)
    v = U[(n - 1) * n:].reshape(n - 1, n)
    yc = (np.arange(n) + 0.5) / n
    g = G["ghia_1982_re100_u_centerline"]
    ui = np.interp(g["y"][1:-1], yc, u[:, n // 2 - 1])
    vi = np.interp(g["x"][1:-1], yc, v[n // 2 - 1, :])
    assert np.abs(ui - np.array(g["u"][1:-1])).max() < 0.006
    assert np.abs(vi - np.array(g["v"][1:-1])).max() < 0.01  # Ghia's tabulated v at Re = 100 is no better than this (32 x 32: 0.008)
    s.destroy()


def test_solution_grid_and_restart_files(tmp_path):
    """The reference's on-disk formats (PetscViewerHDF5): grid.h5 (cartesianmesh.cpp:798-823), <step>.h5 with u, v, p and the
    `time` attribute of /p (navierstokes.cpp:618-634), restart data /convection/<i>, /diffusion/0 (:637-746).  A run
    restarted from its own file continues exactly like the uninterrupted one."""
    h5io = pytest.importorskip("petibm_amd.h5io")
    try:
        h5io.lib()
    except ImportError as e:
        pytest.skip(str(e))
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = cavity((16, 12), nu=0.02, dt=0.01, stretched=True)
    a = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=KSP_P)
    a.advance(3)
    rst = str(tmp_path / "0000003.h5")
    a.write(rst)
    a.writeRestartData(rst)
    a.writeGrid(str(tmp_path / "grid.h5"))
    a.advance(3)
    Ua, pa = a.getState()
    b = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=KSP_P)
    b.readRestartData(rst)
    assert b.ite == 3 and abs(b.t - 0.03) < 1e-15
    b.advance(3)
    Ub, pb = b.getState()
    assert np.abs(Ub - Ua).max() <= 1e-13 and np.abs((pb - pb.mean()) - (pa - pa.mean())).max() <= 1e-11
    with h5io.File(rst, "r") as f:
        assert f.read("u").shape == (12, 15) and f.read("v").shape == (11, 16) and f.read("p").shape == (12, 16)
        assert f.read_attr("p", "time") == pytest.approx(0.03) and f.read("diffusion/0").shape == (a.UN,)
    with h5io.File(str(tmp_path / "grid.h5"), "r") as f:
        m = omesh.create_mesh(cfg)
        for fi, name in enumerate(("u", "v")):
            for d, ax in enumerate("xy"):
                want = np.array([m.coord[fi][d][i] for i in range(int(m.n[fi][d]))])  # n[f][d] entries from index 0
                assert np.allclose(f.read(f"{name}/{ax}"), want, rtol=0, atol=1e-15)
        assert np.allclose(f.read("p/x"), m.coord[3][0].true) and f.read("vertex/y").shape == (13,) and f.read("w/z").shape == (1,)
    a.destroy()
    b.destroy()


def test_lid_driven_cavity_re1000_matches_ghia():
    """examples/navierstokes/liddrivencavity2dRe1000 verbatim: 128 x 128, nu = 0.001, dt = 0.004, 10000 steps (t = 40);
    its plotCenterlineVelocities.py compares with Ghia et al. (1982) at Re = 1000."""
    from petibm_amd.navierstokes import NavierStokesSolver
    n = 128
    cfg = cavity((n, n), nu=0.001, dt=0.004)
    vel = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n"
           "-velocity_pc_type jacobi\n")
    poi = ("-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-06\n-poisson_ksp_rtol 0.0\n-poisson_ksp_max_it 1000\n"
           "-poisson_pc_type gamg\n")
    s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
    s.advance(10000)
    U, p = s.getState()
    u = U[: (n - 1) * n].reshape(n, n - 1)
    v = U[(n - 1) * n:].reshape(n - 1, n)
    yc = (np.arange(n) + 0.5) / n
    g = G["ghia_1982_re1000_centerlines"]
    ui = np.interp(g["y"][1:-1], yc, u[:, n // 2 - 1])
    vi = np.interp(g["x"][1:-1], yc, v[n // 2 - 1, :])
    assert np.abs(ui - np.array(g["u"][1:-1])).max() < 0.02
    assert np.abs(vi - np.array(g["v"][1:-1])).max() < 0.02
    ite, vi_, vr, pi_, pr = s.linSolversInfo()
    assert ite == 10000 and vi_ < 20 and pi_ < 20
    s.destroy()


def test_lid_driven_cavity_re3200_matches_ghia():
    """examples/navierstokes/liddrivencavity2dRe3200 verbatim: 192 x 192, nu = 1/3200, dt = 0.002, 25000 steps (t = 50);
    centre-line velocities against Ghia et al. (1982) as in its plotCenterlineVelocities.py.  (Re = 5000, 60000 steps:
    tools/cavity_ghia.py 5000, 65 s, within 0.027.)"""
    from petibm_amd.navierstokes import NavierStokesSolver
    n = 192
    cfg = cavity((n, n), nu=0.0003125, dt=0.002)
    vel = ("-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n-velocity_ksp_max_it 1000\n"
           "-velocity_pc_type jacobi\n")
    poi = ("-poisson_ksp_type cg\n-poisson_ksp_atol 1.0E-06\n-poisson_ksp_rtol 0.0\n-poisson_ksp_max_it 1000\n"
           "-poisson_pc_type gamg\n")
    s = NavierStokesSolver(cfg, velocity_cfg=vel, poisson_cfg=poi)
    s.advance(25000)
    U, p = s.getState()
    u = U[: (n - 1) * n].reshape(n, n - 1)
    v = U[(n - 1) * n:].reshape(n - 1, n)
    yc = (np.arange(n) + 0.5) / n
    g = G["ghia_1982_re3200_centerlines"]
    keep = np.array([abs(y - 0.4531) > 1e-6 for y in g["y"][1:-1]])  # the table's misprint (-0.86636 for -0.08664)
    ui = np.interp(g["y"][1:-1], yc, u[:, n // 2 - 1])
    vi = np.interp(g["x"][1:-1], yc, v[n // 2 - 1, :])
    assert np.abs(ui - np.array(g["u"][1:-1]))[keep].max() < 0.04   # t = 50 is not quite steady at this Reynolds number
    assert abs(ui[~keep][0] + 0.08664) < 0.02
    assert np.abs(vi - np.array(g["v"][1:-1])).max() < 0.04
    s.destroy()


@pytest.mark.parametrize("flavour", ["amgx_pinned", "ksp_constant"])
def test_matrix_free_poisson_products_in_the_time_step(flavour):
    """On grids of >= 2^20 cells the time step's Poisson solves take their Krylov products from the stencil twin
    (gmg.hip stencil_matmult; `pib_matrix_free_poisson`): the same operator to rounding, so the steps agree with the CSR
    products to solver tolerance -- also with the pinned pressure, whose identity row the twin patches."""
    from petibm_amd.navierstokes import NavierStokesSolver
    n = 1024
    cfg = cavity((n, n), nu=0.01, dt=0.0005)
    cfg["flow"]["initialVelocity"] = ["0.3*sin(3*x)*cos(2*y)", "-0.2*cos(2*x)*sin(3*y)"]
    poi = AMGX_P if flavour == "amgx_pinned" else KSP_P
    out = []
    for mf in (1, 0):
        extra = f"pib_matrix_free_poisson={mf}\n" if flavour == "amgx_pinned" else f"-poisson_pib_matrix_free_poisson {mf}\n"
        s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=poi + extra)
        s.advance(3)
        U, p = s.getState()
        out.append((U, p - p.mean(), s.linSolversInfo()))
        s.destroy()
    assert abs(out[0][2][3] - out[1][2][3]) <= 1 and out[0][2][3] > 0
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-10 * np.abs(out[1][0]).max()
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-8 * np.abs(out[1][1]).max()


@pytest.mark.parametrize("case", ["stretched_cavity", "periodic"])
def test_matrix_free_poisson_products_march_in_3d(case):
    """3-D grids of whole 128 x 8 tiles take the twin's product from the LDS-tiled march (gmg.hip k_level_march<0>), the
    p.w partials summed by the same kernel: the steps agree with the CSR products to solver tolerance -- a stretched
    cavity (pinned pressure: the identity row is patched) and the all-periodic box (constant null space, z seam)."""
    from petibm_amd.navierstokes import NavierStokesSolver
    n = (128, 128, 64)
    if case == "stretched_cavity":
        cfg = cavity(n, nu=0.01, dt=0.002, stretched=True)
        poi = AMGX_P
        fmt = "pib_matrix_free_poisson={}\npib_march_min_cells=1\n"
    else:
        cfg = omesh.periodic_config(n, (True, True, True), lo=0.0, hi=2.0 * np.pi)
        cfg["flow"]["nu"] = 0.01
        cfg["parameters"] = {"dt": 0.002}
        poi = KSP_P
        fmt = "-poisson_pib_matrix_free_poisson {}\n-poisson_pib_march_min_cells 1\n"
    cfg["flow"]["initialVelocity"] = ["0.3*sin(x)*cos(y)*cos(z)", "-0.3*cos(x)*sin(y)*cos(z)", "0.0"]
    out = []
    for mf in (1, 0):
        s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=poi + fmt.format(mf))
        s.advance(3)
        U, p = s.getState()
        out.append((U, p - p.mean(), s.linSolversInfo()))
        s.destroy()
    assert abs(out[0][2][3] - out[1][2][3]) <= 1 and out[0][2][3] > 0
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-10 * np.abs(out[1][0]).max()
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-8 * np.abs(out[1][1]).max()


@pytest.mark.parametrize("case", ["2d", "3d", "2d_periodic_y"])
def test_vorticity_utility_matches_the_restatement(case, tmp_path):
    """petibm-vorticity (applications/vorticity/main.cpp) on the device, against the oracle's index-for-index
    restatement; and the datasets it appends to the solution / grid files."""
    from petibm_amd import h5io
    from petibm_amd.navierstokes import NavierStokesSolver
    if case == "2d":
        cfg = cavity((14, 12), stretched=True)
    elif case == "3d":
        cfg = moving_walls_3d()
    else:
        cfg = omesh.periodic_config((12, 10), (False, True), lo=-1.0, hi=1.0)
        cfg["flow"]["nu"] = 0.01
        cfg["parameters"] = {"dt": 0.01}
    m = omesh.create_mesh(cfg)
    U0 = 0.3 * np.random.default_rng(1).uniform(-1, 1, m.UN)
    s = NavierStokesSolver(cfg)
    s.setState(U0, None)
    ghosts = ons.make_ghosts(m)
    ons.set_ghost_ics(m, ghosts, U0)
    ref = ons.vorticity(m, U0, ghosts)
    w = s.vorticity()
    assert set(w) == set(ref)
    for name in ref:
        assert w[name].shape == ref[name].shape
        assert np.array_equal(w[name], ref[name]), name
    f = str(tmp_path / "0000000.h5")
    g = str(tmp_path / "grid.h5")
    s.write(f)
    s.writeGrid(g)
    s.writeVorticity(f, g)
    with h5io.File(f) as h:
        assert np.array_equal(h.read("wz"), w["wz"])
    with h5io.File(g) as h:
        assert len(h.read("wz/x")) == m.n[3][0] + 1
    s.destroy()


def test_vorticity_of_a_rigid_rotation_is_twice_the_rate():
    from petibm_amd.navierstokes import NavierStokesSolver
    cfg = cavity((16, 16))
    cfg["flow"]["initialVelocity"] = ["-(y - 0.5)", "x - 0.5"]
    s = NavierStokesSolver(cfg)
    wz = s.vorticity()["wz"]
    assert np.allclose(wz[2:-2, 2:-2], 2.0, atol=1e-12)
    s.destroy()


def test_stage_timers_carry_the_reference_stage_names():
    """the reference brackets a step with PetscLogStages "rhsVelocity", "solveVelocity", "rhsPoisson",
    "solvePoisson", "update" (navierstokes.cpp:186-199; "rhsForces" / "solveForces" in decoupledibpm.cpp:93-97, one entry
    here).  Off by default; on, the per-stage sums cover the steps taken, add up to (a little less than) the wall time
    of those steps, and the two solves dominate a 3-D step as they do in the reference."""
    import ctypes
    import time
    from petibm_amd import capi
    from petibm_amd.navierstokes import NavierStokesSolver
    lib = capi.load()
    assert [lib.pib_ns_stage_name(k).decode() for k in range(7)] == ["rhsVelocity", "solveVelocity", "solveForces", "rhsPoisson",
                                                                       "solvePoisson", "update", ""]
    s = NavierStokesSolver(cavity((48, 48, 48), nu=0.01, dt=0.005))
    s.advance(2)
    st = s.stageTimes()
    assert st["steps"] == 0 and all(st[k] == 0.0 for k in st if k != "steps")   # never switched on: nothing recorded
    s.enableStageTimers()
    t0 = time.perf_counter()
    s.advance(3)
    s.advance(2)
    wall = 1e3 * (time.perf_counter() - t0)
    st = s.stageTimes()
    assert st.pop("steps") == 5
    assert list(st) == ["rhsVelocity", "solveVelocity", "solveForces", "rhsPoisson", "solvePoisson", "update"]
    assert all(v >= 0.0 for v in st.values()) and st["solveVelocity"] > 0 and st["solvePoisson"] > 0 and st["rhsVelocity"] > 0
    total = sum(st.values())
    assert 0.5 * wall <= total <= 1.02 * wall, (total, wall)
    assert st["solveVelocity"] + st["solvePoisson"] >= 0.5 * total
    assert st["solveForces"] <= 0.05 * total            # no bodies: nothing between the two marks
    # the state is what the untimed engine computes
    t = NavierStokesSolver(cavity((48, 48, 48), nu=0.01, dt=0.005))
    t.advance(7)
    assert np.array_equal(s.getState()[0], t.getState()[0])
    s.enableStageTimers(False)
    s.advance(1)
    assert s.stageTimes()["steps"] == 0
    assert lib.pib_ns_get_stage_times(None, None, None) != 0 and lib.pib_ns_stage_timers(None, 1) != 0
    s.destroy()
    t.destroy()
