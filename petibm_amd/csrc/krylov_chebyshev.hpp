// krylov_chebyshev.hpp -- included by krylov.hip only, behind its helpers and the BiCGStab section.
#pragma once

namespace pib {
// ------------------------------------------------------------------ Chebyshev
// PETSc's KSPCHEBYSHEV (`-<name>_ksp_type chebyshev`, reached like every KSP type through KSPSetFromOptions at
// /root/reference/src/linsolver/linsolverksp.cpp:62-66; AmgX flavour: solver=CHEBYSHEV) -- oracle/csrc/oracle.c:orc_chebyshev
// has the recurrences.  The velocity operator I/dt - c nu L is strongly diagonally dominant, and an iteration that needs
// no inner products moves a third of BiCGStab's bytes per product: one product, one pass
//     r = b - A p[k] ; z = M^-1 r ; |r|^2, |z|^2 ; p[kp1] = (1 - omega) p[km1] + omega p[k] + omega scale z
// (p[kp1] overwrites p[km1]: two rotating vectors), one reduction for the monitored norm.  The coefficients of the next
// update are formed on the device by the scalar step, so a PAIR of iterations (in = P1, then in = P0) is a body that can
// be captured once and replayed.  The update of an iteration is applied before its norm is known: harmless, the iterate
// the norm speaks of -- p[k] -- is only read, and S->sol says which vector holds it.
struct OpChebInit {  // r = b - w (guess) or b ; z = M^-1 r ; |r|^2 (0), |z|^2 (1) ; p1 = p0 + scale z
    static constexpr int NRED = 2;
    const double *b, *w, *dinv, *p0;
    double *p1;
    double opc, scale;
    int guess;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[2]) const
    {
        Pack<W> vr = ld<W>(b, i), v0 = ld<W>(p0, i), v1;
        if (guess) {
            const Pack<W> vw = ld<W>(w, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vr.v[k] = vr.v[k] - vw.v[k];
        }
        Pack<W> vz = vr;
        if (dinv != nullptr) {
            const Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vz.v[k] = opc * (vd.v[k] * vr.v[k]);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[0] += vr.v[k] * vr.v[k];
            acc[1] += vz.v[k] * vz.v[k];
            v1.v[k] = v0.v[k] + scale * vz.v[k];
        }
        st<W>(p1, i, v1);
    }
};
struct OpChebStep {  // r = b - w ; z = M^-1 r ; |r|^2 (0), |z|^2 (1) ; pm = (a0 pm + omega pk) + cz z
    static constexpr int NRED = 2;
    const double *b, *w, *dinv, *pk;
    double *pm;
    double opc;
    double a0, om, cz;
    __device__ void prepare(const Scalars *S)
    {
        a0 = S->cheb_a0;
        om = S->omega;
        cz = S->cheb_cz;
    }
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[2]) const
    {
        Pack<W> vr = ld<W>(b, i), vm = ld<W>(pm, i);
        const Pack<W> vw = ld<W>(w, i), vk = ld<W>(pk, i);
#pragma unroll
        for (int k = 0; k < W; ++k) vr.v[k] = vr.v[k] - vw.v[k];
        Pack<W> vz = vr;
        if (dinv != nullptr) {
            const Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vz.v[k] = opc * (vd.v[k] * vr.v[k]);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[0] += vr.v[k] * vr.v[k];
            acc[1] += vz.v[k] * vz.v[k];
            vm.v[k] = (a0 * vm.v[k] + om * vk.v[k]) + cz * vz.v[k];
        }
        st<W>(pm, i, vm);
    }
};
// c[kp1] = 2 mu c[k] - c[km1] ; omega = omegaprod c[k] / c[kp1]: the coefficients of the update that follows
__device__ __forceinline__ void cheb_next(Scalars *S)
{
    const double ckp1 = 2.0 * S->cheb_mu * S->c_k - S->c_km1;
    S->omega = S->cheb_omegaprod * S->c_k / ckp1;
    S->cheb_a0 = 1.0 - S->omega;
    S->cheb_cz = S->omega * S->cheb_scale;
    S->c_km1 = S->c_k;
    S->c_k = ckp1;
    if (S->c_k > 0x1p900) {  // only the ratio of consecutive c's enters: an exact rescaling keeps a long run from overflowing
        S->c_km1 *= 0x1p-900;
        S->c_k *= 0x1p-900;
    }
}
__global__ void k_cheb_s_init(Scalars *S, double *hist, int monitor, double mu, double omegaprod, double scale)
{
    const double dp = sqrt(S->normtype == 0 ? S->red[1] : S->red[0]);
    S->dp = dp;
    S->rnorm0 = dp;
    S->ttol = monitor ? fmax(S->rtol * dp, S->atol) : -1.0;
    S->its = 0;
    S->reason = 0;
    S->done = 0;
    S->sol = 0;
    hist[0] = dp;
    converged_default(S, dp);
    if (!S->done && S->maxit <= 0) {
        S->reason = PIB_DIVERGED_ITS;
        S->done = 1;
    }
    if (S->done) return;
    S->its = 1;  // p[k] = p[km1] + scale M^-1 r is the first iteration
    S->sol = 1;
    S->cheb_mu = mu;
    S->cheb_omegaprod = omegaprod;
    S->cheb_scale = scale;
    S->c_km1 = 1.0;
    S->c_k = mu;
    cheb_next(S);
}
// the scalar step of loop pass i = its (KSPSolve_Chebyshev counts the pass before it tests), or -- its == maxit -- the closing
// residual of the last iterate
__global__ void k_cheb_s_step(Scalars *S, double *hist, int conv_is_its)
{
    if (S->done) return;
    const double dp = sqrt(S->normtype == 0 ? S->red[1] : S->red[0]);
    S->dp = dp;
    const int i = S->its;
    hist[i] = dp;
    if (i >= S->maxit) {  // the loop is over: this was the residual of the last iterate
        converged_default(S, dp);
        if (!S->done) {
            S->reason = conv_is_its ? PIB_CONVERGED_ITS : PIB_DIVERGED_ITS;
            S->done = 1;
        }
        return;
    }
    S->its = i + 1;
    converged_default(S, dp);
    if (S->done) {
        hist[S->its] = dp;  // (the entry pib_get_residual / getResidual(its) read)
        return;
    }
    S->sol ^= 1;  // the update this pass applied is the new iterate
    cheb_next(S);
}
__global__ __launch_bounds__(256) void k_cheb_flush(const Scalars *__restrict__ S, int64_t n, const double *__restrict__ p0,
                                                    const double *__restrict__ p1, double *__restrict__ x, int if_done)
{
    if (if_done && !S->done) return;
    const double *src = S->sol ? p1 : p0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = src[i];
}

int solve_chebyshev(pib_solver *s, double *x, const double *b)
{
    const DeviceCsr &A = s->A;
    const int64_t n = A.n;
    hipStream_t q = s->stream;
    const Precond pc = s->cfg.pc;
    if (pc != Precond::JACOBI && pc != Precond::NONE)
        return fail(PIB_ERR_SUP, "solver %s: the Chebyshev iteration runs with Jacobi or no preconditioner", s->name.c_str());
    if (s->nullspace != PIB_NULLSPACE_NONE)
        return fail(PIB_ERR_SUP, "solver %s: the Chebyshev iteration is for definite systems (the velocity operator), not for a singular one", s->name.c_str());
    if (s->post_matmult != nullptr)
        return fail(PIB_ERR_SUP, "solver %s: the Chebyshev iteration needs the operator to be the matrix alone", s->name.c_str());
    const bool jac = (pc == Precond::JACOBI);
    if (jac && A.dinv == nullptr) return fail(PIB_ERR_ORDER, "Jacobi preconditioner without a diagonal");
    const double opc = jac ? s->cfg.jacobi_relaxation : 1.0;
    double emin = s->cfg.cheb_emin, emax = s->cfg.cheb_emax;
    if (!(emax > 0.0)) {
        double lo = 0.0, hi = 0.0;
        PIB_CHK(gershgorin_bounds(s, jac, &lo, &hi));
        emin = jac ? opc * lo : lo;
        emax = jac ? opc * hi : hi;
        if (!(emin > 0.0) || !(emax >= emin))
            return fail(PIB_ERR_SUP,
                        "solver %s: the matrix is not strictly diagonally dominant (Gershgorin interval [%g, %g]): give the Chebyshev "
                        "iteration its bounds (-%s_ksp_chebyshev_eigenvalues emin,emax / cheby_min_lambda, cheby_max_lambda)",
                        s->name.c_str(), emin, emax, s->name.c_str());
    }
    const double scale = 2.0 / (emax + emin), alpha = 1.0 - scale * emin;
    if (!(alpha > 0.0)) return fail(PIB_ERR_ARG_OUTOFRANGE, "solver %s: Chebyshev bounds with emin = emax", s->name.c_str());
    const double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
    PIB_CHK(ensure_work(s, 3));
    double *P0 = s->vec(0), *P1 = s->vec(1), *Wv = s->vec(2);
    const bool guess = s->cfg.initial_guess_nonzero;
    const int monitor = s->cfg.monitor_residual ? 1 : 0;
    const int conv_is_its = monitor ? 0 : 1;
    const bool v2 = aligned16(x) && aligned16(b);
    const double *dv = jac ? A.dinv : nullptr;
    for (int k = 0; k < 8; ++k) s->counters[k] = 0;
    PIB_CHK(init_scalars(s));
    int nb = 0;
    if (guess) {
        OpCopy cp{x, P0};
        PIB_CHK(launch_vec(s, n, cp, v2, 0, nullptr, false, q));
        PIB_CHK(matmult(s, P0, Wv, nullptr, false, q));
    } else {
        OpFill z0{P0, 0.0};
        PIB_CHK(launch_vec(s, n, z0, true, 0, nullptr, false, q));
    }
    {
        OpChebInit op{b, Wv, dv, P0, P1, opc, scale, guess ? 1 : 0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    }
    PIB_CHK(finalize(s, 0, 2, nb, q));
    hipLaunchKernelGGL(k_cheb_s_init, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, monitor, mu, omegaprod, scale);
    PIB_HIP(hipGetLastError());
    const unsigned fb = (unsigned)std::min<int64_t>(VGRID_MAX, std::max<int64_t>(1, (n + 255) / 256));
    auto flush = [&](int if_done) { hipLaunchKernelGGL(k_cheb_flush, dim3(fb), dim3(256), 0, q, s->d_s, n, P0, P1, x, if_done); };
    // one loop pass: the residual of `in` and the update of `ot`
    // the matrix-free velocity operator in its one-launch form applies the update behind the product (velstencil.hip:
    // k_vel_product<2>): p[k] read once with its halos, b, 1 / a_ii and p[km1] read once, p[kp1] written -- five vector
    // passes per iteration where BiCGStab's lean form moves 23 for two products
    const bool fused = s->cfg.fuse_chebyshev_update && (s->comm.nranks == 1 || s->vel.slab_axis >= 0) && s->vel.valid &&
                       s->cfg.matrix_free_velocity && vel_stencil_fused_ok(s) &&
                       ((reinterpret_cast<uintptr_t>(P0) | reinterpret_cast<uintptr_t>(P1)) & 31u) == 0;
    auto pass = [&](double *in, double *ot) -> int {
        if (fused) {
            if (s->comm.nranks > 1) PIB_CHK(halo_exchange(s, in, q));
            PIB_CHK(vel_stencil_apply_cheb(s, in, ot, b, dv, opc, q, 0));
            nb = VEL_DOT_PARTIALS;
        } else {
            PIB_CHK(matmult(s, in, Wv, nullptr, true, q));
            OpChebStep op{b, Wv, dv, in, ot, opc, 0.0, 0.0, 0.0};
            PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, true, q));
        }
        PIB_CHK(finalize(s, 0, 2, nb, q));
        hipLaunchKernelGGL(k_cheb_s_step, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, conv_is_its);
        PIB_HIP(hipGetLastError());
        return 0;
    };
    auto body = [&]() -> int {
        PIB_CHK(pass(P1, P0));
        return pass(P0, P1);
    };
    const int maxit = s->cfg.max_iters;  // passes 1 .. maxit - 1 and the closing residual: maxit of them
    int enq = 0;
    if (!skip_first_poll(s)) {
        flush(1);
        PIB_CHK(fetch_results(s, 0));
        if (s->h_s->done) return 0;
    }
    const int batch0 = first_batch(s), batch1 = next_batch(s);
    while (!s->h_s->done && enq < maxit) {
        const int todo = (std::min(enq == 0 ? batch0 : batch1, maxit - enq) + 1) / 2;  // pairs
        PIB_CHK(run_iterations(s, todo, enq / 2, graph_key(3, x, b), q, body));
        const bool first = enq == 0;
        enq += 2 * todo;
        if (first) {
            flush(1);
            PIB_CHK(fetch_results(s, enq));
            if (s->h_s->done) return 0;
        } else
            PIB_CHK(poll(s));
    }
    flush(0);
    return fetch_results(s, enq);
}
}  // namespace pib
