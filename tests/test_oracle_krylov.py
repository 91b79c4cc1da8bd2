"""The oracle's Krylov restatements (KSPCG / KSPBCGS recurrences, GMG-PCG) cross-checked against
scipy.sparse.linalg -- an independent implementation of the published algorithms."""
import json
import os

import numpy as np
import pytest

from oracle import clib, mesh as omesh, operators as oops

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_test_vectors.json")))


@pytest.fixture(scope="module")
def systems():
    m = omesh.create_mesh(G["cartesianmesh2d_dirichlet"]["config"])
    D, Gm, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    _, A = oops.create_poisson_operator(D, Gm, L, 0.01, 0.005)
    Av = oops.create_velocity_operator(L, 0.01, 0.005)
    return m, A, Av


def test_spmv_order_and_value(systems):
    m, A, _ = systems
    x = np.random.default_rng(1).uniform(-1, 1, A.n_cols)
    y = clib.spmv(A, x)
    # canonical order: rounded products added sequentially in CSR order
    for r in (0, 17, 131):
        s = 0.0
        for p in range(A.rowptr[r], A.rowptr[r + 1]):
            s = s + A.val[p] * x[A.col[p]]
        assert y[r] == s
    assert np.allclose(y, A.to_dense() @ x, rtol=1e-13, atol=1e-18)


def test_cg_matches_scipy_iterates(systems):
    spla = pytest.importorskip("scipy.sparse.linalg")
    m, A, _ = systems
    P = oops.pin_row0(A)
    rng = np.random.default_rng(2)
    b = rng.uniform(-1, 1, A.n_rows)
    b[0] = 0.0
    # negative definite (+1 at the decoupled pinned row): PETSc's CG flags sign CHANGES only; scipy needs SPD
    Ps = P.to_scipy().tolil()
    Ps = -Ps
    Ps[0, 0] = 1.0
    Ps = Ps.tocsr()
    its = []
    x_sp, info = spla.cg(Ps, -b, rtol=1e-12, atol=0.0, maxiter=2000, callback=lambda xk: its.append(1))
    assert info == 0
    r = clib.cg(P, b, pc="none", norm="unpreconditioned", rtol=1e-12, atol=0.0, dtol=1e300, maxit=2000)
    assert r["reason"] > 0
    assert abs(r["iters"] - len(its)) <= 2
    assert np.linalg.norm(r["x"] - x_sp) <= 1e-8 * np.linalg.norm(x_sp)
    # monotone-ish history, first entry = ||b||
    assert np.isclose(r["history"][0], np.linalg.norm(b))


def test_cg_constant_nullspace_and_jacobi(systems):
    m, A, _ = systems
    xs = np.random.default_rng(3).uniform(-1, 1, A.n_rows)
    xs -= xs.mean()
    b = clib.spmv(A, xs)
    for pc in ("none", "jacobi"):
        for norm in ("preconditioned", "unpreconditioned"):
            r = clib.cg(A, b, pc=pc, nullspace=1, norm=norm, rtol=1e-12, atol=0.0, dtol=1e300, maxit=3000)
            assert r["reason"] > 0
            e = r["x"] - r["x"].mean() - xs
            assert np.linalg.norm(e) <= 1e-8 * np.linalg.norm(xs)
    # KSPConvergedDefault: atol wins when larger; reason ATOL vs RTOL
    r = clib.cg(A, b, pc="jacobi", nullspace=1, rtol=0.0, atol=1e-6, maxit=3000)
    assert r["reason"] == 3 and r["rnorm"] < 1e-6
    # max_it -> DIVERGED_ITS (-3)
    r = clib.cg(A, b, pc="none", nullspace=1, rtol=1e-14, atol=0.0, maxit=3)
    assert r["reason"] == -3 and r["iters"] == 3


def test_bcgs_matches_scipy_solution(systems):
    spla = pytest.importorskip("scipy.sparse.linalg")
    m, _, Av = systems
    us = np.random.default_rng(4).uniform(-1, 1, Av.n_rows)
    b = clib.spmv(Av, us)
    assert np.abs(Av.to_dense() - Av.to_dense().T).max() > 0  # non-symmetric on the stretched mesh
    x_sp, info = spla.bicgstab(Av.to_scipy(), b, rtol=1e-13, atol=0.0, maxiter=500)
    assert info == 0
    for norm in ("preconditioned", "unpreconditioned"):
        r = clib.bcgs(Av, b, pc="jacobi", norm=norm, rtol=0.0, atol=1e-12, dtol=1e300, maxit=500)
        assert r["reason"] > 0 and r["iters"] < 20
        assert np.linalg.norm(r["x"] - us) <= 1e-10 * np.linalg.norm(us)
        assert np.linalg.norm(r["x"] - x_sp) <= 1e-9 * np.linalg.norm(us)


@pytest.mark.parametrize("n", [(32, 32, 32), (20, 17, 9), (48, 40)])
def test_gmg_oracle_properties(n):
    """The build's V-cycle: the level-0 stencil twin equals the CSR operator, the preconditioner is
    symmetric (<Mr,s> = <r,Ms> on zero-mean vectors), and PCG iteration counts are mesh-independent."""
    m = omesh.create_mesh(omesh.uniform_config(n))
    D, Gm, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    dt = 1e-3
    _, A = oops.create_poisson_operator(D, Gm, L, dt, 0.5e-3)
    w = [m.dL[3][d].true for d in range(m.dim)]
    g = clib.GMG(n, w, dt, nullspace=1)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, m.pN)
    ax = clib.spmv(A, x)
    assert np.abs(g.apply_operator(x) - ax).max() <= 1e-13 * np.abs(ax).max()
    r1 = rng.uniform(-1, 1, m.pN)
    r2 = rng.uniform(-1, 1, m.pN)
    r1 -= r1.mean()
    r2 -= r2.mean()
    z1, z2 = g.apply(r1), g.apply(r2)
    z1 -= z1.mean()
    z2 -= z2.mean()
    assert abs(z1 @ r2 - r1 @ z2) <= 1e-10 * abs(z1 @ r2)
    xs = x - x.mean()
    b = clib.spmv(A, xs)
    res = g.pcg(A, b, rtol=1e-10)
    assert res["reason"] > 0 and res["iters"] <= 25
    assert np.linalg.norm(b - clib.spmv(A, res["x"])) <= 1.2e-10 * np.linalg.norm(b)


def test_single_reduction_cg_is_the_same_iteration(systems):
    """KSPCGUseSingleReduction (oracle.c:orc_cg_single_reduction) is an algebraic rearrangement of KSPCG -- the matrix applied
    to z, w = A p and p'w by recurrence: in exact arithmetic the same iterates.  Checked against the standard restatement
    (itself checked against scipy above): the first 15 residual norms to 1e-9 (over the hundreds of iterations of the
    unpreconditioned solve the two rounding histories drift apart as any two CG runs do), counts within 10 %, the same
    solution -- every preconditioner / norm / null-space combination; the multigrid PCG (11-14 iterations) throughout."""
    m, A, _ = systems
    xs = np.random.default_rng(31).uniform(-1, 1, A.n_rows)
    xs -= xs.mean()
    b = clib.spmv(A, xs)
    for pc in ("none", "jacobi"):
        for norm in ("preconditioned", "unpreconditioned"):
            kw = dict(pc=pc, nullspace=1, norm=norm, rtol=1e-10, atol=0.0, dtol=1e300, maxit=3000)
            r0, r1 = clib.cg(A, b, **kw), clib.cg(A, b, single_reduction=True, **kw)
            assert r1["reason"] == r0["reason"] > 0 and abs(r1["iters"] - r0["iters"]) <= max(1, r0["iters"] // 10)
            assert np.abs(r1["history"][:15] - r0["history"][:15]).max() <= 1e-9 * r0["history"][0]
            e0, e1 = r0["x"] - r0["x"].mean() - xs, r1["x"] - r1["x"].mean() - xs
            assert np.linalg.norm(e1) <= 1e-7 * np.linalg.norm(xs) and np.linalg.norm(e0) <= 1e-7 * np.linalg.norm(xs)
            assert np.isclose(r1["history"][0], r0["history"][0], rtol=1e-14)  # (the set-up is the same code; OpenMP sums)
    # a pinned (non-singular, negative definite) system without null space, nonzero guess
    P = oops.pin_row0(A)
    bp = b.copy()
    bp[0] = 0.0
    x0 = np.random.default_rng(32).uniform(-1, 1, A.n_rows)
    x0[0] = 0.0  # (the decoupled +1 row stays out of the recurrences: r[0] = 0, as in the scipy test above)
    kw = dict(pc="jacobi", norm="unpreconditioned", rtol=1e-11, atol=0.0, dtol=1e300, maxit=3000, x0=x0)
    r0, r1 = clib.cg(P, bp, **kw), clib.cg(P, bp, single_reduction=True, **kw)
    assert r1["reason"] > 0 and abs(r1["iters"] - r0["iters"]) <= max(1, r0["iters"] // 10)
    assert np.abs(r1["history"][:15] - r0["history"][:15]).max() <= 1e-9 * r0["history"][0]
    assert np.linalg.norm(r1["x"] - r0["x"]) <= 1e-7 * np.linalg.norm(r0["x"])
    # max_it and the first iteration (where PETSc multiplies with p itself)
    r = clib.cg(A, b, single_reduction=True, pc="none", nullspace=1, rtol=1e-14, atol=0.0, maxit=3)
    assert r["reason"] == -3 and r["iters"] == 3
    r0, r1 = (clib.cg(A, b, single_reduction=sr, pc="jacobi", nullspace=1, rtol=1e-14, atol=0.0, maxit=1) for sr in (False, True))
    assert np.allclose(r0["x"], r1["x"], rtol=1e-12, atol=1e-14) and np.allclose(r0["history"], r1["history"], rtol=1e-13)
    # multigrid PCG
    mu = omesh.create_mesh(omesh.uniform_config((24, 20, 16)))
    D, Gm, L = oops.create_divergence(mu), oops.create_gradient(mu), oops.create_laplacian(mu)
    _, Au = oops.create_poisson_operator(D, Gm, L, 1e-3, 0.5e-3)
    g = clib.GMG((24, 20, 16), [mu.dL[3][d].true for d in range(3)], 1e-3, nullspace=1, pre=2, post=2)
    xu = np.random.default_rng(33).uniform(-1, 1, mu.pN)
    xu -= xu.mean()
    bu = clib.spmv(Au, xu)
    for norm in ("preconditioned", "unpreconditioned"):
        r0, r1 = g.pcg(Au, bu, norm=norm, rtol=1e-10), g.pcg(Au, bu, norm=norm, rtol=1e-10, single_reduction=True)
        assert r1["reason"] > 0 and r1["iters"] == r0["iters"]
        assert np.abs(r1["history"] - r0["history"]).max() <= 1e-8 * r0["history"][0]
        assert np.linalg.norm(r1["x"] - r0["x"]) <= 1e-9 * np.linalg.norm(r0["x"])


def test_chebyshev_restatement_against_scipy_and_numpy(systems):
    """orc_chebyshev (KSPSolve_Chebyshev restated: PETSc is not in /root/reference -- parity unpinned by the reference) on
    the velocity operator: the solution is scipy's, the residual history is that of the textbook three-term recurrence
    written with numpy, the Gershgorin bounds contain the spectrum, and the iteration count is KSP's (the verifying product
    counts: its = steps + 1 on convergence, = max_it when the limit ends the loop)."""
    spla = pytest.importorskip("scipy.sparse.linalg")
    sp = pytest.importorskip("scipy.sparse")
    m, _, Av = systems
    rng = np.random.default_rng(5)
    us = rng.uniform(-1, 1, Av.n_rows)
    b = clib.spmv(Av, us)
    S = sp.csr_matrix((Av.val, Av.col, Av.rowptr), shape=(Av.n_rows, Av.n_cols))
    rho = clib.gershgorin_jacobi(Av)
    d = S.diagonal()
    ev = np.linalg.eigvals((sp.diags(1.0 / d) @ S).toarray())
    assert 0.0 < rho < 1.0 and np.abs(ev.imag).max() < 1e-9 and ev.real.min() >= 1.0 - rho - 1e-12 and ev.real.max() <= 1.0 + rho + 1e-12
    for norm in ("preconditioned", "unpreconditioned"):
        r = clib.chebyshev(Av, b, pc="jacobi", norm=norm, rtol=1e-12, atol=0.0, maxit=500)
        assert r["reason"] == 2 and np.linalg.norm(r["x"] - spla.spsolve(S.tocsc(), b)) <= 1e-10 * np.linalg.norm(us)
        # the recurrence with numpy
        emin, emax = 1.0 - rho, 1.0 + rho
        scale = 2.0 / (emax + emin)
        alpha = 1.0 - scale * emin
        mu, omegaprod = 1.0 / alpha, 2.0 / alpha
        c = [1.0, mu]
        pm = np.zeros_like(b)
        z = b / d
        hist = [np.linalg.norm(z if norm == "preconditioned" else b)]
        pk = pm + scale * z
        for i in range(1, r["iters"]):
            res = b - S @ pk
            z = res / d
            hist.append(np.linalg.norm(z if norm == "preconditioned" else res))
            if i == r["iters"] - 1:
                break
            cn = 2.0 * mu * c[1] - c[0]
            om = omegaprod * c[1] / cn
            pm, pk = pk, (1.0 - om) * pm + om * pk + om * scale * z
            c = [c[1], cn]
        assert np.allclose(r["history"][: len(hist)], hist, rtol=1e-8, atol=1e-14 * hist[0])  # (the last entries sit at rounding level)
        assert r["history"][-1] == r["history"][-2] == r["rnorm"] and hist[-1] <= 1e-12 * hist[0] < hist[-2]
    lim = clib.chebyshev(Av, b, pc="jacobi", rtol=1e-12, atol=0.0, maxit=4)
    assert lim["reason"] == -3 and lim["iters"] == 4 and len(lim["history"]) == 5
    # no preconditioner, explicit bounds (-ksp_chebyshev_eigenvalues)
    evA = np.linalg.eigvals(S.toarray()).real
    r = clib.chebyshev(Av, b, emin=0.98 * evA.min(), emax=1.02 * evA.max(), pc="none", rtol=1e-12, atol=0.0, maxit=500)
    assert r["reason"] == 2 and np.linalg.norm(r["x"] - us) <= 1e-10 * np.linalg.norm(us)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("n,side", [((16, 12, 10), "right"), ((16, 12, 10), "left"), ((24, 20), "right")])
def test_bcgs_with_the_multigrid_is_the_published_recurrence(n, side, pinned):
    """oracle/csrc/oracle.c:orc_bcgs_gmg (BiCGStab around the V-cycle, the mean removed after every application on the singular
    system) against a numpy restatement of the published recurrences with `GMG.apply` as M^-1: right-preconditioned with the true
    residual's L2 norm (AmgX PBICGSTAB, what /root/reference/src/linsolver/linsolveramgx.cpp:62-72 configures from a solver file) and
    left-preconditioned with the preconditioned residual's norm (KSPBCGS, linsolverksp.cpp:62-66).  The C restatement is what the GPU
    path is compared with (tests/test_gpu_bicgstab_gmg.py); this pins it on the CPU.  pinned (round 5): row / column 0 replaced by
    the identity (navierstokes.cpp:414-420) -- the cycle's right-hand side made compatible inside `GMG.apply` (nullspace 2), its output
    shifted by its value at cell 0, the pinned unknown keeping the input's value (oracle.c pcapply)."""
    m = omesh.create_mesh(omesh.uniform_config(n))
    D, Gm, L = oops.create_divergence(m), oops.create_gradient(m), oops.create_laplacian(m)
    dt = 1e-2
    _, A = oops.create_poisson_operator(D, Gm, L, dt, 0.5e-2)
    if pinned:
        A = oops.pin_row0(A)
    w = [m.dL[3][d].true for d in range(m.dim)]
    g = clib.GMG(n, w, dt, nullspace=2 if pinned else 1, pre=1, post=1)
    rng = np.random.default_rng(17)
    xs = rng.uniform(-1, 1, m.pN)
    xs -= xs[0] if pinned else xs.mean()
    b = clib.spmv(A, xs)

    def minv(v):
        z = g.apply(v)
        if pinned:
            z = z - z[0]
            z[0] = v[0]
            return z
        return z - z.mean()

    amul = lambda v: clib.spmv(A, v)
    op = (lambda v: amul(minv(v))) if side == "right" else (lambda v: minv(amul(v)))
    x = np.zeros(m.pN)
    r = b.copy() if side == "right" else minv(b)
    rp = r.copy()
    p = np.zeros_like(r)
    v = np.zeros_like(r)
    rho_old = alpha = omega_old = 1.0
    hist = [np.linalg.norm(r)]
    for it in range(12):
        rho = r @ rp
        beta = (rho / rho_old) * (alpha / omega_old)
        p = r - (omega_old * beta) * v + beta * p
        v = op(p)
        alpha = rho / (v @ rp)
        s = r - alpha * v
        t = op(s)
        omega = (s @ t) / (t @ t)
        x = x + alpha * (minv(p) if side == "right" else p) + omega * (minv(s) if side == "right" else s)
        r = s - omega * t
        hist.append(np.linalg.norm(r))
        rho_old, omega_old = rho, omega
        if hist[-1] <= 1e-10 * hist[0]:
            break
    ref = g.bcgs(A, b, norm="unpreconditioned" if side == "right" else "preconditioned", rtol=1e-10, atol=1e-50, dtol=1e300, maxit=50)
    assert ref["reason"] > 0 and abs(ref["iters"] - (len(hist) - 1)) <= 1
    k = min(len(hist), len(ref["history"]), 6)
    assert np.allclose(ref["history"][:k], hist[:k], rtol=1e-8)
    e = (ref["x"] - ref["x"].mean()) - (x - x.mean())
    assert np.linalg.norm(e) <= 1e-6 * np.linalg.norm(xs)
    assert np.linalg.norm(b - clib.spmv(A, ref["x"])) <= (2e-10 if side == "right" else 1e-7) * np.linalg.norm(b)
    if pinned:
        assert abs(ref["x"][0]) <= 1e-12 * np.abs(xs).max() and np.linalg.norm(ref["x"] - xs) <= 1e-6 * np.linalg.norm(xs)


def test_merged_residual_update_identities():
    """csrc/krylov.hip k_finalize_post<7> (pib_bicgstab_form=3): with omega = s.t / t.t and r = s - omega t, the sums the iteration
    needs follow from five sums over s, t, r~ alone -- |r|^2 = s.s - omega (2 s.t - omega t.t), r.r~ = r~.s - omega r~.t -- so the
    pass that formed r only to sum it can go.  The relative error of |r|^2 by that formula is eps |s|^2 / |r|^2: checked here for
    half-steps that cut the residual by 3 x ... 1000 x."""
    rng = np.random.default_rng(23)
    n = 20000
    for cut in (3.0, 30.0, 1000.0):
        t = rng.standard_normal(n)
        perp = rng.standard_normal(n)
        perp -= (perp @ t) / (t @ t) * t
        s = 0.7 * t + (0.7 * np.linalg.norm(t) / np.sqrt(cut * cut - 1.0)) * perp / np.linalg.norm(perp)
        rp = rng.standard_normal(n)
        st, tt, ss, rps, rpt = s @ t, t @ t, s @ s, rp @ s, rp @ t
        om = st / tt
        r = s - om * t
        assert abs(np.linalg.norm(s) / np.linalg.norm(r) - cut) <= 1e-6 * cut
        r2 = ss - om * (2.0 * st - om * tt)
        assert abs(r2 - r @ r) <= 64 * np.finfo(float).eps * ss
        assert abs(np.sqrt(r2) - np.linalg.norm(r)) <= 1e-9 * np.linalg.norm(r)
        assert abs((rps - om * rpt) - rp @ r) <= 64 * np.finfo(float).eps * (abs(rps) + abs(om * rpt) + np.linalg.norm(rp) * np.linalg.norm(s))
