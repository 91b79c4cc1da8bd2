"""BN order N > 1 (parameters.BN of the reference): the Poisson operator D * BN * G assembled in
HBM through the reference's chain of sparse products (createBnHead, src/operators/createbn.cpp:19-95;
navierstokes.cpp:349-356) against the oracle's restatement of the same chain -- bit-exact -- and its solve."""
import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops
from test_gpu_parity import STRETCHED_2D, _a0_table, amgx_cfg, gmg_cfg, iters_close, rhs_for, stretched_3d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lin():
    from petibm_amd import linsolver
    return linsolver


def make(case):
    cfg = {"2d_stretched": STRETCHED_2D, "3d_stretched": stretched_3d((8, 7, 6)),
           "2d_periodic_y": omesh.periodic_config((10, 9), (False, True), ratios=(1.07, 1.0)),
           "3d_periodic": omesh.periodic_config((6, 5, 7), (True, True, True))}[case]
    m = omesh.create_mesh(cfg)
    per = [bool(m.periodic[0][d]) for d in range(m.dim)]
    return m, per


@pytest.mark.parametrize("case", ["2d_stretched", "3d_stretched", "2d_periodic_y", "3d_periodic"])
@pytest.mark.parametrize("order,pinned", [(2, False), (3, True)])
def test_bn_poisson_operator_bit_exact(lin, case, order, pinned):
    from petibm_amd import capi
    m, per = make(case)
    dt, cnu = 0.0125, 0.5 * 0.02
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=order)
    if pinned:
        A = oops.pin_row0(A)
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg())
    s.setPeriodic(per)
    n = [int(v) for v in m.n[3][: m.dim]]
    s.assemblePoissonBN(n, [m.dL[3][d].true for d in range(m.dim)], m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu,
                        order, capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT)
    rp, cl, vl = s.getCSR()
    assert np.array_equal(rp, A.rowptr) and np.array_equal(cl, A.col)
    assert np.array_equal(vl, A.val)
    width = np.diff(rp).max()
    assert width > 2 * m.dim + 1  # wider than the 5/7-point stencil
    s.destroy()


@pytest.mark.parametrize("case,order", [("2d", 2), ("3d", 2), ("3d", 3)])
def test_bn_poisson_solve_with_the_order1_multigrid(lin, case, order):
    from petibm_amd import capi
    cfg = omesh.uniform_config((48, 40)) if case == "2d" else stretched_3d((20, 18, 14))
    m = omesh.create_mesh(cfg)
    dt, cnu = 0.01, 0.5 * 0.01
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=order)
    xs, b = rhs_for(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    s = lin.LinSolverHIP("poisson", config_text=gmg_cfg())
    s.assemblePoissonBN(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu, order, capi.NULLSPACE_CONSTANT)
    x = np.zeros(A.n_rows)
    s.solve(x, b)
    g = clib.GMG(n, w, dt, nullspace=1, pre=1, post=1, omega=0.9, coarsest_sweeps=32)
    ref = g.pcg(A, b, rtol=1e-10, maxit=200)
    assert ref["reason"] > 0 and s.getReason() > 0
    assert iters_close(s.getIters(), ref["iters"]) and s.getIters() <= 40
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 1.5e-10 * np.linalg.norm(b)
    s.destroy()


def test_bn_order_below_one_is_the_reference_error(lin):
    from petibm_amd import capi
    m, per = make("2d_stretched")
    s = lin.LinSolverHIP("poisson", config_text=amgx_cfg())
    with pytest.raises(capi.PibError) as ei:
        s.assemblePoissonBN([int(v) for v in m.n[3][:2]], [m.dL[3][d].true for d in range(2)], m.min[:2], m.max[:2],
                            _a0_table(m), 0.01, 0.005, 0, capi.NULLSPACE_CONSTANT)
    assert ei.value.code == capi.ERR_SUP  # createbn.cpp:27-29: error 56
    s.destroy()


@pytest.mark.parametrize("case,order", [("2d_cavity", 2), ("3d_periodic_xz", 2), ("2d_cavity", 3)])
def test_time_step_with_bn_order_matches_oracle(case, order):
    """parameters.BN > 1 in the flow solver: the Poisson operator D*BN*G and the projection u = u* - BN G dP
    (navierstokes.cpp:349-356,583-598)."""
    from oracle import navierstokes as ons
    from petibm_amd.navierstokes import NavierStokesSolver
    from test_gpu_periodic import VEL, KSP_P
    if case == "2d_cavity":
        cfg = omesh.uniform_config((14, 12), lid=1.0)
    else:
        cfg = omesh.periodic_config((8, 7, 6), (True, False, True))
    cfg["flow"]["nu"] = 0.02
    cfg["parameters"] = {"dt": 0.005, "convection": "ADAMS_BASHFORTH_2", "diffusion": "CRANK_NICOLSON", "BN": order}
    m = omesh.create_mesh(cfg)
    ref = ons.NavierStokes(m, 0.005, 0.02, vtol=1e-14, ptol=1e-13, bn_order=order)
    rng = np.random.default_rng(5)
    U0, p0 = 0.1 * rng.uniform(-1, 1, m.UN), 0.1 * rng.uniform(-1, 1, m.pN)
    ref.set_state(U0, p0)
    s = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=KSP_P)
    assert s.bn_order == order
    s.setState(U0, p0)
    for step in range(3):
        ref.advance()
        s.advance()
        U, p, r1, r2 = s.getState(rhs=True)
        assert np.abs(r1 - ref.last_rhs1).max() <= 1e-9 * np.abs(ref.last_rhs1).max()
        assert np.abs(r2 - ref.last_rhs2).max() <= 1e-9 * np.abs(ref.last_rhs2).max() + 1e-14
        assert np.abs(U - ref.U).max() <= 1e-9 * np.abs(ref.U).max()
        dp = (p - p.mean()) - (ref.p - ref.p.mean())
        assert np.abs(dp).max() <= 1e-8 * np.abs(ref.p - ref.p.mean()).max()
    s.destroy()


@pytest.mark.parametrize("P,case,order,pinned", [(2, "3d", 2, False), (3, "3d", 3, True), (2, "2d", 2, True), (4, "2d", 3, False),
                                                 (2, "3d_periodic", 2, False), (3, "3d_periodic", 2, True), (2, "2d_periodic_y", 3, False)])
def test_bn_poisson_operator_on_slabs(lin, P, case, order, pinned):
    """SURVEY.md 8e for BN order > 1: every rank runs the product chain on a window of the mesh around its planes and keeps
    its rows (bn.hip: assemble_poisson_bn_slab); the ghost columns reach `order` planes into the neighbours, across the seam
    of a periodic slab axis through the ring.  Rows against the oracle's one-rank chain -- bit for bit, to rounding next to a
    periodic seam -- and the solve on the ranks against the one-rank solve."""
    from petibm_amd import capi
    import slab_plans as partition
    from test_gpu_multirank_loopback import _run_ranks
    cfg = {"3d": stretched_3d((8, 7, 13)), "2d": omesh.uniform_config((14, 17)),
           "3d_periodic": omesh.periodic_config((6, 5, 16), (True, False, True)),
           "2d_periodic_y": omesh.periodic_config((10, 18), (False, True), ratios=(1.07, 1.0))}[case]
    m = omesh.create_mesh(cfg)
    per = [bool(m.periodic[0][d]) for d in range(m.dim)]
    dt, cnu = 0.0125, 0.5 * 0.02
    D, G, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, G, L, dt, cnu, bn_order=order)
    if pinned:
        A = oops.pin_row0(A)
    n = [int(v) for v in m.n[3][: m.dim]]
    w = [m.dL[3][d].true for d in range(m.dim)]
    null = capi.NULLSPACE_PINNED if pinned else capi.NULLSPACE_CONSTANT
    pl = int(np.prod(n[:-1]))
    xs, b = rhs_for(A)

    def rank_fn(r, uid):
        s = lin.LinSolverHIP("poisson", config_text=gmg_cfg(), rank=r, nranks=P, uid=uid, device=0)
        s.setPeriodic(per)
        s.assemblePoissonBN(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu, order, null)
        k0, k1 = partition.slab_range(n[-1], P, r)
        assert s.n_local == (k1 - k0) * pl
        csr = s.getCSR()
        x = np.zeros(s.n_local)
        s.solve(x, np.ascontiguousarray(b[k0 * pl: k1 * pl]))
        out = csr, x, s.getIters(), s.getReason()
        s.destroy()
        return out

    res = _run_ranks(P, rank_fn)
    seam = per[-1]
    x = np.concatenate([q[1] for q in res])
    row = 0
    for (rp, cl, vl), _, _, reason in res:
        assert reason > 0
        for i in range(rp.size - 1):
            c, v = cl[rp[i]: rp[i + 1]] % A.n_rows, vl[rp[i]: rp[i + 1]]
            o = np.argsort(c, kind="stable")
            a0_, a1_ = A.rowptr[row], A.rowptr[row + 1]
            assert np.array_equal(c[o], A.col[a0_:a1_])
            if seam:
                assert np.allclose(v[o], A.val[a0_:a1_], rtol=1e-13, atol=1e-13 * np.abs(A.val[a0_:a1_]).max())
            else:
                assert np.array_equal(v[o], A.val[a0_:a1_])
            row += 1
    assert row == A.n_rows
    assert len({q[2] for q in res}) == 1
    s1 = lin.LinSolverHIP("poisson", config_text=gmg_cfg())
    s1.setPeriodic(per)
    s1.assemblePoissonBN(n, w, m.min[: m.dim], m.max[: m.dim], _a0_table(m), dt, cnu, order, null)
    x1 = np.zeros(A.n_rows)
    s1.solve(x1, b)
    assert abs(res[0][2] - s1.getIters()) <= 3
    assert np.linalg.norm(b - clib.spmv(A, x)) <= 2e-10 * np.linalg.norm(b)
    e = (x - x.mean()) - (x1 - x1.mean()) if not pinned else x - x1
    assert np.linalg.norm(e) <= 1e-7 * np.linalg.norm(x1)
    s1.destroy()


@pytest.mark.parametrize("case,order", [("2d", 2), ("2d", 3), ("3d", 2), ("2d_moving", 2)])
def test_decoupled_ibpm_with_bn_order_above_one(case, order):
    """parameters.BN = N > 1 TOGETHER with immersed bodies (applications/decoupledibpm/decoupledibpm.cpp:194-205: BN =
    createBnHead(L, dt, c nu, N), BNH = BN H, EBNH = E BNH by MatMatMult): the device builds both through the product chain of
    bn.hip from the assembled BN -- bit-identical to the oracle's restatement of the same chain -- and the time step (forces
    system, u += BNH df, projection with BNG) follows the oracle's; a moving body re-assembles them every step."""
    from oracle import ibm
    from petibm_amd.navierstokes import DecoupledIBPMSolver
    from test_gpu_ibm import AMGX_P, FORCES, VEL, flow_config, sphere_points
    from test_oracle_ibm import body_mesh, circle
    moving = case == "2d_moving"
    if case == "3d":
        cfg = flow_config(body_mesh(cells=(4, 8, 4), ratio=1.4, span=2.0, core=0.6, dim=3), nu=0.05, dt=0.02)
        bodies = [sphere_points(40, r=0.35)]
    else:
        cfg = flow_config(body_mesh(cells=(8, 16, 8), ratio=1.25, span=3.0, core=0.8), dt=0.01)
        bodies = [circle(32)]
    cfg["parameters"]["BN"] = order
    m = omesh.create_mesh(cfg)
    dt, nu = cfg["parameters"]["dt"], cfg["flow"]["nu"]
    ref = ibm.DecoupledIBPM(m, dt, nu, bodies, pinned=True, vtol=1e-14, ptol=1e-13, bn_order=order)
    U0 = np.zeros(m.UN)
    U0[: int(np.prod(m.n[0]))] = 1.0
    U0 += 0.02 * np.random.default_rng(3).uniform(-1, 1, m.UN)
    ref.set_state(U0, np.zeros(m.pN))
    s = DecoupledIBPMSolver(cfg, bodies=bodies, velocity_cfg=VEL, poisson_cfg=AMGX_P.format(tol=1e-13), forces_cfg=FORCES)
    s.setState(U0, np.zeros(m.pN))

    def same_operators():
        for name in ("EBNH", "BNH"):
            nr, rp, cl, vl = s.getOperator(name)
            r = ref.ops[name]
            assert nr == r.n_rows and np.array_equal(rp, r.rowptr) and np.array_equal(cl, r.col), name
            assert np.array_equal(vl, r.val), name

    same_operators()
    assert ref.ops["BNH"].nnz > ref.ops["H"].nnz  # really wider than dt H
    for step in range(1, 4):
        if moving:
            x = bodies[0] + np.array([0.05 * np.sin(2 * np.pi * step * dt), 0.0])
            v = np.tile([0.05 * 2 * np.pi * np.cos(2 * np.pi * step * dt), 0.0], (x.shape[0], 1))
            ref.move_bodies([x], [v])
            s.moveBodies([x], [v])
            same_operators()
        ref.advance()
        s.advance()
        U, p = s.getState()
        f, avg = s.getForces()
        assert np.abs(U - ref.U).max() <= 1e-8 * np.abs(ref.U).max()
        assert np.abs(f - ref.f).max() <= 1e-7 * np.abs(ref.f).max()
    s.destroy()


@pytest.mark.parametrize("case,P,order", [("3d_cavity", 2, 2), ("3d_cavity", 3, 3), ("2d_cavity", 3, 2), ("3d_convective_outlet", 2, 2),
                                          ("3d_channel_periodic_z", 2, 2)])
def test_bn_order_above_one_in_the_time_step_on_slabs(case, P, order):
    """parameters.BN = N > 1 in the slab engine (BASELINE configs 3 and 5 are 8-GPU runs of the time step): the Poisson
    operator D BN G from the per-rank windows of the product chain, the projection u -= BN G dP by applying BN term by term
    with the velocity solver's matrix-free L (a row of the assembled BNG reaches N planes beyond the slab; one exchange per
    extra term instead).  2 / 3 loopback ranks against the single-rank engine (which multiplies by the assembled BNG): the
    first right-hand side bit for bit, three steps to the solver tolerance."""
    from petibm_amd.navierstokes import NavierStokesSolver
    from test_gpu_multirank_loopback import _run_ranks
    from test_gpu_navierstokes import AMGX_P, KSP_P, VEL
    from test_gpu_navierstokes_slabs import CASES
    make, pinned = CASES[case]
    cfg = make()
    cfg["parameters"]["BN"] = order
    pcfg = AMGX_P if pinned else KSP_P
    one = NavierStokesSolver(cfg, velocity_cfg=VEL, poisson_cfg=pcfg)
    rng = np.random.default_rng(11)
    U0 = 0.1 * rng.uniform(-1, 1, one.UN)
    p0 = 0.1 * rng.uniform(-1, 1, one.pN)
    if "convective" in case:
        U0[: int(np.prod(one._field_shape(0)))] += 1.0
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
            assert np.abs(r1 - crhs1).max() <= 1e-13 * np.abs(crhs1).max()
        else:
            assert np.array_equal(r1, crhs1)
        assert np.allclose(Ua, cU1, rtol=0, atol=1e-10) and np.allclose(Ub, cU3, rtol=0, atol=1e-9)
        if pinned:
            assert np.allclose(pa, cp1, rtol=0, atol=1e-8) and np.allclose(pb, cp3, rtol=0, atol=1e-8)
    if not pinned:
        pg = np.concatenate([r[1][1] for r in res])
        assert np.allclose(pg - pg.mean(), p3 - p3.mean(), rtol=0, atol=1e-8)
    # BN > 1 really changes the step (the same cases with BN = 1 end elsewhere)
    cfg1 = make()
    ref1 = NavierStokesSolver(cfg1, velocity_cfg=VEL, poisson_cfg=pcfg)
    ref1.setState(U0, p0)
    ref1.advance(3)
    assert np.abs(ref1.getState()[0] - U3).max() > 1e-7
    ref1.destroy()
    one.destroy()
