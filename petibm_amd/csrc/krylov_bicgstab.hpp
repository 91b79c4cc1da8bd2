// krylov_bicgstab.hpp -- included by krylov.hip only, behind its helpers (launch_vec, finalize, run_iterations, fetch_results ...).
// ------------------------------------------------------------------ BiCGStab (K10)
// PETSc's KSPBCGS recurrences (oracle/csrc/oracle.c:orc_bcgs):
//   KSP flavour : left preconditioning, recurrences on the preconditioned residual
//   AmgX flavour: PBICGSTAB, right preconditioning, true-residual L2 norm
// used for the velocity system A = I/dt - c nu L (navierstokes.cpp:342-344), which is
// non-symmetric on stretched meshes (createlaplacian.cpp row scaling).
// reduction slots: 0 |r|^2  1 r.rp  2 v.rp  3 s.t  4 t.t
#pragma once

namespace pib {

template <int PCM>
struct OpBInit {  // r = M^-1 (b - w) (left) or b - w (right); rp = r; p = v = 0
    static constexpr int NRED = 1;
    const double *b, *w, *dinv;
    double *r, *rp, *p, *v;
    double omega_pc;
    int guess, left;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[1]) const
    {
        Pack<W> vb = ld<W>(b, i), vr, zero;
        if (guess) {
            Pack<W> vw = ld<W>(w, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vr.v[k] = vb.v[k] - vw.v[k];
        } else {
            vr = vb;
        }
        if (PCM == PCM_JACOBI && left) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vr.v[k] = omega_pc * (vd.v[k] * vr.v[k]);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            zero.v[k] = 0.0;
            acc[0] += vr.v[k] * vr.v[k];
        }
        st<W>(r, i, vr);
        st<W>(rp, i, vr);
        st<W>(p, i, zero);
        st<W>(v, i, zero);
    }
};

template <int PCM>
struct OpBUpdateP {  // p = r - (omegaold*beta) v + beta p ; right: ph = M^-1 p
    static constexpr int NRED = 0;
    const double *r, *v, *dinv;
    double *p, *ph;
    double omega_pc;
    int left;
    double beta, ob;
    __device__ void prepare(const Scalars *S)
    {
        beta = S->b;
        ob = S->omegaold * S->b;
    }
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        Pack<W> vr = ld<W>(r, i), vv = ld<W>(v, i), vp = ld<W>(p, i);
#pragma unroll
        for (int k = 0; k < W; ++k) vp.v[k] = (vr.v[k] - ob * vv.v[k]) + beta * vp.v[k];
        st<W>(p, i, vp);
        if (!left && PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vp.v[k] = omega_pc * (vd.v[k] * vp.v[k]);
            st<W>(ph, i, vp);
        }
    }
};

template <int PCM>
struct OpBPcDot {  // left: out = M^-1 in ; partial slot = out . other
    static constexpr int NRED = 1;
    const double *in, *dinv, *other;
    double *out;
    double omega_pc;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[1]) const
    {
        Pack<W> vi = ld<W>(in, i), vo = ld<W>(other, i);
        if (PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vi.v[k] = omega_pc * (vd.v[k] * vi.v[k]);
            st<W>(out, i, vi);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) acc[0] += vi.v[k] * vo.v[k];
    }
};

template <int PCM>
struct OpBUpdateS {  // s = r - alpha v ; right: sh = M^-1 s
    static constexpr int NRED = 0;
    const double *r, *v, *dinv;
    double *sv, *sh;
    double omega_pc;
    int left;
    double alpha;
    __device__ void prepare(const Scalars *S) { alpha = S->alpha; }
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        Pack<W> vr = ld<W>(r, i), vv = ld<W>(v, i);
#pragma unroll
        for (int k = 0; k < W; ++k) vr.v[k] = vr.v[k] - alpha * vv.v[k];
        st<W>(sv, i, vr);
        if (!left && PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vr.v[k] = omega_pc * (vd.v[k] * vr.v[k]);
            st<W>(sh, i, vr);
        }
    }
};

template <int PCM>
struct OpBPcDot2 {  // left: t = M^-1 in ; partials s.t (slot 3) t.t (slot 4)
    static constexpr int NRED = 2;
    const double *in, *dinv, *sv;
    double *t;
    double omega_pc;
    int left;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[2]) const
    {
        Pack<W> vi = ld<W>(in, i), vs = ld<W>(sv, i);
        if (left && PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vi.v[k] = omega_pc * (vd.v[k] * vi.v[k]);
            st<W>(t, i, vi);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[0] += vs.v[k] * vi.v[k];
            acc[1] += vi.v[k] * vi.v[k];
        }
    }
};

struct OpBUpdateX {  // x += alpha ph + omega sh ; r = s - omega t ; partials |r|^2 (0), r.rp (1)
    static constexpr int NRED = 2;
    const double *ph, *sh, *sv, *t, *rp;
    double *x, *r;
    double alpha, omega;
    int only_alpha;  // t == 0 exit of PETSc: x += alpha p, nothing else
    __device__ void prepare(const Scalars *S)
    {
        alpha = S->alpha;
        omega = S->omega;
    }
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[2]) const
    {
        Pack<W> vp = ld<W>(ph, i), vsh = ld<W>(sh, i), vs = ld<W>(sv, i), vt = ld<W>(t, i), vx = ld<W>(x, i),
                vrp = ld<W>(rp, i), vr;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            vx.v[k] = (vx.v[k] + alpha * vp.v[k]) + omega * vsh.v[k];
            vr.v[k] = vs.v[k] - omega * vt.v[k];
            acc[0] += vr.v[k] * vr.v[k];
            acc[1] += vr.v[k] * vrp.v[k];
        }
        st<W>(x, i, vx);
        st<W>(r, i, vr);
    }
};

// the multigrid under BiCGStab: z = M^-1 r is one V-cycle, and on a singular system (constant null space) its mean is removed
// after every application, as KSP_PCApply + KSP_RemoveNullSpace do (oracle: pcapply with PC_GMG)
struct OpSumV {  // partial: sum v
    static constexpr int NRED = 1;
    const double *v;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[1]) const
    {
        const Pack<W> vv = ld<W>(v, i);
#pragma unroll
        for (int k = 0; k < W; ++k) acc[0] += vv.v[k];
    }
};
struct OpShiftV {  // v -= red[slot] / n_global
    static constexpr int NRED = 0;
    double *v;
    double inv_n;
    int slot;
    double m;
    __device__ void prepare(const Scalars *S) { m = S->red[slot] * inv_n; }
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        Pack<W> vv = ld<W>(v, i);
#pragma unroll
        for (int k = 0; k < W; ++k) vv.v[k] = vv.v[k] - m;
        st<W>(v, i, vv);
    }
};

// ---- the same recurrences on the matrix-free velocity operator (right preconditioning, Jacobi, one rank, the one-launch
// product): M^-1 p and M^-1 s are never stored -- the products apply the sweep as they read their input
// (vel_stencil_apply's dinv / opc) -- and x += alpha M^-1 p + omega M^-1 s is applied by the NEXT iteration's p-update,
// which reads p anyway.  216 -> 200 B/row/iteration (27 -> 25 vector passes).  Every value is computed by the expression
// of the general path above: bit-identical iterates -- unless `pib_bicgstab_form` >= 2 (default 3) lets the products sum
// v.rp and s.t, t.t themselves (two passes less, 184 B/row/iteration): those sums are grouped by tile, so alpha and omega
// agree with the general path's to rounding only.
struct OpBFUpdateP {  // x += xalpha ph + xomega sh (owed) ; p = r - (omegaold*beta) v + beta p
    // y != nullptr: the owed update goes into y += xalpha p + xomega s instead -- the sum of the search directions BEFORE the
    // (stationary) Jacobi sweep, x = x0 + M^-1 y once at the end (k_b_flush_x) -- which takes the dinv and x streams out
    // of this pass: 56 instead of 64 B/row.  x then differs from the general path's by rounding (the residual recurrence
    // does not see x), so this rides with the fused sums (`pib_bicgstab_form` >= 2), not with the bit-identical route.
    // t != nullptr (`pib_bicgstab_form` 3, with y): the residual update the previous iteration owes, r = s - omega t, is formed
    // HERE (s is read for y anyway) and stored for the next s = r - alpha v: OpBFUpdateR's pass (s, t, rp in, r out) is gone, its two
    // sums come out of the second product's five (k_finalize_post<7>).  Same expression: r has the bits OpBFUpdateR would store.
    static constexpr int NRED = 0;
    const double *r, *v, *dinv, *sv;
    double *p, *x, *y;
    double omega_pc;
    double beta, ob, xa, xo;
    int pend;
    const double *t = nullptr;
    double *rw = nullptr;
    __device__ void prepare(const Scalars *S)
    {
        beta = S->b;
        ob = S->omegaold * S->b;
        pend = S->xpend;
        xa = S->xalpha;
        xo = S->xomega;
    }
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        Pack<W> vr, vv = ld<W>(v, i), vp = ld<W>(p, i);
        if (!(pend && t != nullptr)) vr = ld<W>(r, i);
        if (pend && y != nullptr) {
            Pack<W> vs = ld<W>(sv, i), vy = ld<W>(y, i);
            if (t != nullptr) {
                const Pack<W> vt = ld<W>(t, i);
#pragma unroll
                for (int k = 0; k < W; ++k) vr.v[k] = vs.v[k] - xo * vt.v[k];
                st<W>(rw, i, vr);
            }
#pragma unroll
            for (int k = 0; k < W; ++k) vy.v[k] = (vy.v[k] + xa * vp.v[k]) + xo * vs.v[k];
            st<W>(y, i, vy);
        } else if (pend && dinv == nullptr) {  // no preconditioner (NOSOLVER): ph = p, sh = s
            Pack<W> vs = ld<W>(sv, i), vx = ld<W>(x, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vx.v[k] = (vx.v[k] + xa * vp.v[k]) + xo * vs.v[k];
            st<W>(x, i, vx);
        } else if (pend) {
            Pack<W> vd = ld<W>(dinv, i), vs = ld<W>(sv, i), vx = ld<W>(x, i);
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const double ph = omega_pc * (vd.v[k] * vp.v[k]), sh = omega_pc * (vd.v[k] * vs.v[k]);
                vx.v[k] = (vx.v[k] + xa * ph) + xo * sh;
            }
            st<W>(x, i, vx);
        }
#pragma unroll
        for (int k = 0; k < W; ++k) vp.v[k] = (vr.v[k] - ob * vv.v[k]) + beta * vp.v[k];
        st<W>(p, i, vp);
    }
};
struct OpBFUpdateR {  // r = s - omega t ; partials |r|^2 (0), r.rp (1)
    static constexpr int NRED = 2;
    const double *sv, *t, *rp;
    double *r;
    double omega;
    __device__ void prepare(const Scalars *S) { omega = S->omega; }
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[2]) const
    {
        Pack<W> vs = ld<W>(sv, i), vt = ld<W>(t, i), vrp = ld<W>(rp, i), vr;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            vr.v[k] = vs.v[k] - omega * vt.v[k];
            acc[0] += vr.v[k] * vr.v[k];
            acc[1] += vr.v[k] * vrp.v[k];
        }
        st<W>(r, i, vr);
    }
};
// the x update still owed when the iteration stops
// (y != nullptr: x = x0 + M^-1 (y + what is owed), see OpBFUpdateP)
__global__ __launch_bounds__(256) void k_b_flush_x(Scalars *__restrict__ S, int64_t n, const double *__restrict__ p,
                                                   const double *__restrict__ sv, const double *__restrict__ dinv, double omega_pc,
                                                   double *__restrict__ x, const double *__restrict__ y, int if_done)
{
    if (if_done && !S->done) return;
    const int pend = S->xpend;
    if (!pend && y == nullptr) return;
    const double xa = S->xalpha, xo = S->xomega;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (y != nullptr) {
            const double acc = pend ? (y[i] + xa * p[i]) + xo * sv[i] : y[i];
            x[i] = x[i] + (dinv != nullptr ? omega_pc * (dinv[i] * acc) : acc);
        } else if (dinv == nullptr) {
            x[i] = (x[i] + xa * p[i]) + xo * sv[i];
        } else {
            const double ph = omega_pc * (dinv[i] * p[i]), sh = omega_pc * (dinv[i] * sv[i]);
            x[i] = (x[i] + xa * ph) + xo * sh;
        }
    }
}
__global__ void k_b_flush_done(Scalars *S, int if_done)
{
    if (if_done && !S->done) return;
    S->xpend = 0;
}

__global__ void k_b_s_init(Scalars *S, double *hist, int monitor)
{
    const double dp = sqrt(S->red[0]);
    S->dp = dp;
    S->rnorm0 = dp;
    S->ttol = monitor ? fmax(S->rtol * dp, S->atol) : -1.0;
    S->its = 0;
    S->reason = 0;
    S->done = 0;
    S->xpend = 0;
    hist[0] = dp;
    converged_default(S, dp);
    S->rho = S->red[0];  // rp = r  ->  <r,rp> = |r|^2
    S->rhoold = 1.0;
    S->alpha = 1.0;
    S->omega = 1.0;
    S->omegaold = 1.0;
    if (!S->done && S->maxit <= 0) {
        S->reason = PIB_DIVERGED_ITS;
        S->done = 1;
    }
    if (!S->done && S->rho == 0.0) {
        S->reason = PIB_DIVERGED_BREAKDOWN;
        S->done = 1;
    }
    S->b = (S->rho / S->rhoold) * (S->alpha / S->omegaold);  // beta of the first iteration
}

__device__ __forceinline__ void b_s_alpha(Scalars *S)
{
    const double d1 = S->red[2];
    if (d1 == 0.0 || d1 != d1) {
        S->reason = (d1 != d1) ? PIB_DIVERGED_NANORINF : PIB_DIVERGED_BREAKDOWN;
        S->done = 1;
        return;
    }
    S->alpha = S->rho / d1;
}
__global__ void k_b_s_alpha(Scalars *S)
{
    if (S->done) return;
    b_s_alpha(S);
}

__device__ __forceinline__ void b_s_omega(Scalars *S)
{
    const double d1 = S->red[3], d2 = S->red[4];
    if (d2 == 0.0) {
        // t = 0: PETSc accepts x += alpha p when s = 0 too; s.s is not available separately here, but
        // t = K s = 0 with a non-singular operator means s = 0.
        S->omega = 0.0;
        return;
    }
    S->omega = d1 / d2;
}
__global__ void k_b_s_omega(Scalars *S)
{
    if (S->done) return;
    b_s_omega(S);
}

__device__ __forceinline__ void b_s_end(Scalars *S, double *hist, int conv_is_its)
{
    const double dp = sqrt(S->red[0]);
    S->dp = dp;
    S->rhoold = S->rho;
    S->omegaold = S->omega;
    S->its += 1;
    hist[S->its] = dp;
    converged_default(S, dp);
    if (!S->done && S->its >= S->maxit) {
        S->reason = conv_is_its ? PIB_CONVERGED_ITS : PIB_DIVERGED_ITS;
        S->done = 1;
    }
    if (S->done) return;
    if (S->rhoold == 0.0 || S->omega == 0.0) {
        S->reason = PIB_DIVERGED_BREAKDOWN;
        S->done = 1;
        return;
    }
    S->rho = S->red[1];
    if (S->rho == 0.0) {
        S->reason = PIB_DIVERGED_BREAKDOWN;
        S->done = 1;
        return;
    }
    S->b = (S->rho / S->rhoold) * (S->alpha / S->omegaold);
}
__global__ void k_b_s_end(Scalars *S, double *hist, int conv_is_its)
{
    if (S->done) return;
    b_s_end(S, hist, conv_is_its);
}

// The reduction of k_finalize (same order, slot after slot) followed by the scalar step that consumes it, in one launch:
// on one rank nothing sits between the two (no all-reduce), and a small problem's Krylov iteration is a chain of ~5 us
// launches.  POST: 1 BiCGStab alpha, 2 omega, 3 end of iteration, 4 CG alpha, 5 / 6: 1 / 3 with the deferred x update.
template <int POST>
__global__ __launch_bounds__(256) void k_finalize_post(Scalars *__restrict__ S, const double *__restrict__ part, int slot0, int nslots,
                                                       int count, double *hist, int conv_is_its, PinRowDev pr = PinRowDev{nullptr, 0, {}, {}})
{
    if (S->done) return;
    __shared__ double sh[4];
    for (int q = 0; q < nslots; ++q) {
        const int slot = slot0 + q;
        const double *p = part + (int64_t)slot * PIB_MAXPART;
        double v = 0.0;
        for (int i = threadIdx.x; i < count; i += 256) v += p[i];
        v = wsum(v);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) S->red[slot] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (POST == 1) b_s_alpha(S);
        if (POST == 2) b_s_omega(S);
        if (POST == 3) b_s_end(S, hist, conv_is_its);
        if (POST == 4) {
            cg_s1(S);
            if (!S->done) cg_pin_sigma(S, pr);
        }
        if (POST == 5) {  // matrix-free BiCGStab: this iteration's p-update has applied what the previous one owed
            S->xpend = 0;
            b_s_alpha(S);
        }
        if (POST == 6) {  // ... and its x update is owed from here on
            S->xalpha = S->alpha;
            S->xomega = S->omega;
            S->xpend = 1;
            b_s_end(S, hist, conv_is_its);
        }
        if (POST == 8) cg_s2(S, hist, 0.0, 0, 1, 0, conv_is_its);  // CG: the monitored norm |r| and its convergence test (no shift, no beta)
        if (POST == 7) {  // omega and the end of the iteration at once: r = s - omega t is owed too (OpBFUpdateP::t), its sums
            // follow from the second product's five -- red[3..7] = s.t, t.t, s.s, rp.s, rp.t:
            // |r|^2 = s.s - omega (2 s.t - omega t.t), r.rp = rp.s - omega rp.t
            b_s_omega(S);
            const double om = S->omega;
            const double r2 = S->red[5] - om * (2.0 * S->red[3] - om * S->red[4]);
            S->red[0] = r2 > 0.0 ? r2 : (r2 != r2 ? r2 : 0.0);
            S->red[1] = S->red[6] - om * S->red[7];
            S->xalpha = S->alpha;
            S->xomega = S->omega;
            S->xpend = 1;
            b_s_end(S, hist, conv_is_its);
        }
    }
}
template <int POST>
static int finalize_post(pib_solver *s, int slot0, int nslots, int count, double *hist, int conv_is_its, hipStream_t q, const PinRowDev &pr)
{
    hipLaunchKernelGGL((k_finalize_post<POST>), dim3(1), dim3(256), 0, q, s->d_s, s->d_part, slot0, nslots, count, hist, conv_is_its, pr);
    PIB_HIP(hipGetLastError());
    return 0;
}

int solve_bicgstab(pib_solver *s, double *x, const double *b)
{
    const DeviceCsr &A = s->A;
    const int64_t n = A.n;
    hipStream_t q = s->stream;
    const Precond pc = s->cfg.pc;
    // BiCGStab with the multigrid (round 4): AmgX takes any solver x preconditioner pair of a solver file
    // (/root/reference/src/linsolver/linsolveramgx.cpp:62-72), PETSc any -ksp_type / -pc_type; the V-cycle is applied where the
    // general path applies the Jacobi sweep, as a call of its own (gmg_apply), followed by the projection on a singular system
    const bool gmg = (pc == Precond::GMG);
    if (gmg && !s->has_grid)
        return fail(PIB_ERR_ORDER,
                    "solver %s: a multigrid (AMG/GMG) preconditioner needs the grid structure: call "
                    "pib_set_grid_hint or pib_assemble_poisson before pib_solve", s->name.c_str());
    if (pc == Precond::JACOBI && A.dinv == nullptr) return fail(PIB_ERR_ORDER, "Jacobi preconditioner without a diagonal");
    PIB_CHK(ensure_work(s, 9));
    double *R = s->vec(0), *RP = s->vec(1), *P = s->vec(2), *V = s->vec(3), *S = s->vec(4), *T = s->vec(5),
           *T2 = s->vec(6), *PH = s->vec(7), *SH = s->vec(8);
    const bool left = (s->cfg.norm == NormType::PRECONDITIONED);
    const bool jac = (pc == Precond::JACOBI);
    const bool guess = s->cfg.initial_guess_nonzero;
    const double opc = jac ? s->cfg.jacobi_relaxation : 1.0;
    const int monitor = s->cfg.monitor_residual ? 1 : 0;
    const int conv_is_its = monitor ? 0 : 1;
    const bool v2 = aligned16(x) && aligned16(b);
    if ((!jac && !gmg) || left) {  // no separate preconditioned copies needed
        PH = P;
        SH = S;
    }
    for (int k = 0; k < 8; ++k) s->counters[k] = 0;
    PIB_CHK(init_scalars(s));
    int nb = 0;
    const bool project = gmg && s->nullspace == PIB_NULLSPACE_CONSTANT;
    // a pinned pressure row (round 5; the oracle's pcapply, nullspace 2): the cycle's right-hand side is made compatible with the
    // sum of ITS input (slot 5, where gmg_apply reads it), its output is shifted by its value at cell 0, and the pinned unknown
    // keeps the input's value -- the preconditioner of the pinned system, as the CG path applies it
    const bool pinned = gmg && s->nullspace == PIB_NULLSPACE_PINNED;
    auto apply_gmg = [&](const double *in, double *out, bool guarded) -> int {
        s->gmg_guarded = guarded;
        s->gmg_want_dots = false;
        if (pinned) {
            int nbs = 0;
            OpSumV sv{in};
            PIB_CHK(launch_vec(s, n, sv, true, 5, &nbs, guarded, q));
            PIB_CHK(finalize(s, 5, 1, nbs, q));
        }
        PIB_CHK(gmg_apply(s, in, out, q));
        s->counters[1]++;
        if (pinned) {
            hipLaunchKernelGGL(k_fetch_z0, dim3(1), dim3(1), 0, q, s->d_s, out, (A.row0 == 0) ? 1 : 0, 5);
            PIB_HIP(hipGetLastError());
            PIB_CHK(allreduce_slots(s, 5, 1, q));
            OpPinShift<0> sh{out, in, (A.row0 == 0) ? 1 : 0, 5, 0.0};
            PIB_CHK(launch_vec(s, n, sh, true, 0, nullptr, true, q));  // (reads the shift from the scalars: always given them)
        }
        if (project) {
            int nbs = 0;
            OpSumV sv{out};
            PIB_CHK(launch_vec(s, n, sv, true, 5, &nbs, guarded, q));
            PIB_CHK(finalize(s, 5, 1, nbs, q));
            OpShiftV sh{out, 1.0 / (double)A.n_global, 5, 0.0};
            PIB_CHK(launch_vec(s, n, sh, true, 0, nullptr, true, q));  // (reads the sum from the scalars: always given them)
        }
        return 0;
    };
    if (guess) {
        OpCopy cp{x, PH == P ? PH : P};
        // use T2 as the ghost-padded SpMV input so P stays free
        OpCopy cp2{x, T2};
        (void)cp;
        PIB_CHK(launch_vec(s, n, cp2, v2, 0, nullptr, false, q));
        PIB_CHK(matmult(s, T2, T, nullptr, false, q));
    } else {
        OpFill z0{x, 0.0};
        PIB_CHK(launch_vec(s, n, z0, v2, 0, nullptr, false, q));
    }
    if (jac) {
        OpBInit<PCM_JACOBI> op{b, T, A.dinv, R, RP, P, V, opc, guess ? 1 : 0, left ? 1 : 0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    } else if (gmg && left) {  // r = M^-1 (b - w): the raw residual (in T2), the V-cycle, then the pass that sets rp, p, v and |r|^2
        OpBInit<PCM_NONE> raw{b, T, nullptr, T2, RP, P, V, 1.0, guess ? 1 : 0, 0};
        PIB_CHK(launch_vec(s, n, raw, v2, 0, &nb, false, q));
        PIB_CHK(apply_gmg(T2, T, false));
        OpBInit<PCM_NONE> op{T, T, nullptr, R, RP, P, V, 1.0, 0, 0};
        PIB_CHK(launch_vec(s, n, op, true, 0, &nb, false, q));
    } else {
        OpBInit<PCM_NONE> op{b, T, nullptr, R, RP, P, V, 1.0, guess ? 1 : 0, left ? 1 : 0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    }
    PIB_CHK(finalize(s, 0, 1, nb, q));
    hipLaunchKernelGGL(k_b_s_init, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, monitor);
    PIB_HIP(hipGetLastError());

    const int batch0 = first_batch(s), batch1 = next_batch(s);
    const int maxit = s->cfg.max_iters;
    const bool one_rank = s->comm.nranks == 1;
    // the matrix-free velocity operator in its one-launch form: no stored M^-1 p / M^-1 s, deferred x update (OpBFUpdateP)
    // (also without a preconditioner -- NOSOLVER, the velocity solver file of flatplate3dRe100_GPU and multicylinders2dRe100_GPU:
    // the sweep drops out, dv == nullptr)
    const double *dv = jac ? A.dinv : nullptr;  // (on slabs: its ghost-padded copy, below)
    // On slabs the products exchange their input's boundary planes first and the sums go through the all-reduce before their
    // scalar step; the sweep needs the neighbours' diagonal on the ghost planes: a ghost-padded copy of 1 / a_ii, exchanged
    // once per solve, in the vector the general path keeps M^-1 s in.
    const bool lean = (jac || pc == Precond::NONE) && !left && (one_rank || s->vel.slab_axis >= 0) &&
                      s->cfg.bicgstab_form >= 1 && s->vel.valid && s->cfg.matrix_free_velocity &&
                      s->post_matmult == nullptr && vel_stencil_fused_ok(s) && aligned16(x) &&
                      ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(V) |
                        reinterpret_cast<uintptr_t>(T)) & 31u) == 0;
    const bool fused_dots = lean && s->cfg.bicgstab_form >= 2;
    // ... and x accumulated before the Jacobi sweep (OpBFUpdateP::y) in the vector the general path keeps M^-1 p in
    double *Y = fused_dots ? s->vec(7) : nullptr;
    // ... and the residual update merged into the next p-update, |r|^2 and r.rp out of the second product's sums
    const bool merge_r = Y != nullptr && s->cfg.bicgstab_form >= 3;
    if (lean && jac && !one_rank) {
        double *D = s->vec(8);
        PIB_HIP(hipMemcpyAsync(D, A.dinv, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, q));
        PIB_CHK(halo_exchange(s, D, q));
        dv = D;
    }
    if (Y != nullptr) {
        OpFill y0{Y, 0.0};
        PIB_CHK(launch_vec(s, n, y0, v2, 0, nullptr, false, q));
    }
    int enq = 0;
    // the x update the last iteration owes (lean recurrences only)
    auto flush = [&](int if_done) {
        if (!lean) return;
        hipLaunchKernelGGL(k_b_flush_x, dim3((unsigned)std::min<int64_t>(VGRID_MAX, std::max<int64_t>(1, (n + 255) / 256))), dim3(256), 0, q,
                           s->d_s, n, P, S, dv, opc, x, (const double *)Y, if_done);
        hipLaunchKernelGGL(k_b_flush_done, dim3(1), dim3(1), 0, q, s->d_s, if_done);
    };
    if (!skip_first_poll(s)) {  // (as in solve_cg)
        flush(1);
        PIB_CHK(fetch_results(s, 0));
        if (s->h_s->done) return 0;
    }
    while (!s->h_s->done && enq < maxit) {
        const int todo = std::min(enq == 0 ? batch0 : batch1, maxit - enq);
        // reduce the partial sums of `nslots` slots and run scalar step POST: one launch on one rank, with the all-reduce
        // in between on several (k_finalize_post with no slots left to reduce is the scalar step alone)
        auto reduce_then = [&](auto post_tag, int slot0, int nslots, int count, double *hist, int cis) -> int {
            constexpr int POST = decltype(post_tag)::value;
            if (one_rank) return finalize_post<POST>(s, slot0, nslots, count, hist, cis, q);
            PIB_CHK(finalize(s, slot0, nslots, count, q));
            return finalize_post<POST>(s, slot0, 0, 0, hist, cis, q);
        };
        auto body_lean = [&]() -> int {
            OpBFUpdateP up{R, V, dv, S, P, x, Y, opc, 0.0, 0.0, 0.0, 0.0, 0};
            if (merge_r) {
                up.t = T;
                up.rw = R;
            }
            PIB_CHK(launch_vec(s, n, up, true, 0, nullptr, true, q));
            if (!one_rank) PIB_CHK(halo_exchange(s, P, q));
            if (fused_dots) {  // v = K M^-1 p and v.rp by the same kernel
                PIB_CHK(vel_stencil_apply(s, P, V, true, q, dv, opc, 1, RP, 2));
                nb = VEL_DOT_PARTIALS;
            } else {
                PIB_CHK(vel_stencil_apply(s, P, V, true, q, dv, opc));  // v = K M^-1 p
                OpBPcDot<PCM_NONE> d1{V, nullptr, RP, V, 1.0};
                PIB_CHK(launch_vec(s, n, d1, true, 2, &nb, true, q));
            }
            PIB_CHK(reduce_then(std::integral_constant<int, 5>(), 2, 1, nb, nullptr, 0));
            OpBUpdateS<PCM_NONE> us{R, V, nullptr, S, S, 1.0, 0, 0.0};  // s = r - alpha v
            PIB_CHK(launch_vec(s, n, us, true, 0, nullptr, true, q));
            if (!one_rank) PIB_CHK(halo_exchange(s, S, q));
            if (merge_r) {  // t = K M^-1 s with s.t, t.t, s.s, rp.s, rp.t; omega and the end of the iteration in one scalar step
                PIB_CHK(vel_stencil_apply(s, S, T, true, q, dv, opc, 4, RP, 3));
                PIB_CHK(reduce_then(std::integral_constant<int, 7>(), 3, 5, VEL_DOT_PARTIALS, s->d_hist, conv_is_its));
                PIB_HIP(hipGetLastError());
                return 0;
            }
            if (fused_dots) {  // t = K M^-1 s with s.t and t.t
                PIB_CHK(vel_stencil_apply(s, S, T, true, q, dv, opc, 2, nullptr, 3));
                nb = VEL_DOT_PARTIALS;
            } else {
                PIB_CHK(vel_stencil_apply(s, S, T, true, q, dv, opc));  // t = K M^-1 s
                OpBPcDot2<PCM_NONE> d2{T, nullptr, S, T, 1.0, 0};
                PIB_CHK(launch_vec(s, n, d2, true, 3, &nb, true, q));
            }
            PIB_CHK(reduce_then(std::integral_constant<int, 2>(), 3, 2, nb, nullptr, 0));
            OpBFUpdateR ur{S, T, RP, R, 0.0};
            PIB_CHK(launch_vec(s, n, ur, true, 0, &nb, true, q));
            PIB_CHK(reduce_then(std::integral_constant<int, 6>(), 0, 2, nb, s->d_hist, conv_is_its));
            PIB_HIP(hipGetLastError());
            return 0;
        };
        auto body = [&]() -> int {
            if (lean) return body_lean();
            // p = r - omegaold*beta*v + beta*p  (+ ph = M^-1 p)
            if (jac) {
                OpBUpdateP<PCM_JACOBI> op{R, V, A.dinv, P, PH, opc, left ? 1 : 0, 0.0, 0.0};
                PIB_CHK(launch_vec(s, n, op, true, 0, nullptr, true, q));
            } else {
                OpBUpdateP<PCM_NONE> op{R, V, nullptr, P, PH, 1.0, left ? 1 : 0, 0.0, 0.0};
                PIB_CHK(launch_vec(s, n, op, true, 0, nullptr, true, q));
                if (gmg && !left) PIB_CHK(apply_gmg(P, PH, true));  // ph = M^-1 p
            }
            // v = K p ; d1 = v.rp
            if (left && gmg) {  // v = M^-1 (K p)
                PIB_CHK(matmult(s, P, T2, nullptr, true, q));
                PIB_CHK(apply_gmg(T2, V, true));
                OpBPcDot<PCM_NONE> op{V, nullptr, RP, V, 1.0};
                PIB_CHK(launch_vec(s, n, op, true, 2, &nb, true, q));
            } else if (left && jac) {
                PIB_CHK(matmult(s, P, T2, nullptr, true, q));
                OpBPcDot<PCM_JACOBI> op{T2, A.dinv, RP, V, opc};
                PIB_CHK(launch_vec(s, n, op, true, 2, &nb, true, q));
            } else {
                PIB_CHK(matmult(s, PH, V, nullptr, true, q));
                OpBPcDot<PCM_NONE> op{V, nullptr, RP, V, 1.0};
                PIB_CHK(launch_vec(s, n, op, true, 2, &nb, true, q));
            }
            if (one_rank)
                PIB_CHK(finalize_post<1>(s, 2, 1, nb, nullptr, 0, q));
            else {
                PIB_CHK(finalize(s, 2, 1, nb, q));
                hipLaunchKernelGGL(k_b_s_alpha, dim3(1), dim3(1), 0, q, s->d_s);
            }
            // s = r - alpha v (+ sh = M^-1 s)
            if (jac) {
                OpBUpdateS<PCM_JACOBI> op{R, V, A.dinv, S, SH, opc, left ? 1 : 0, 0.0};
                PIB_CHK(launch_vec(s, n, op, true, 0, nullptr, true, q));
            } else {
                OpBUpdateS<PCM_NONE> op{R, V, nullptr, S, SH, 1.0, left ? 1 : 0, 0.0};
                PIB_CHK(launch_vec(s, n, op, true, 0, nullptr, true, q));
                if (gmg && !left) PIB_CHK(apply_gmg(S, SH, true));  // sh = M^-1 s
            }
            // t = K s ; s.t, t.t
            if (left && gmg) {  // t = M^-1 (K s)
                PIB_CHK(matmult(s, S, T2, nullptr, true, q));
                PIB_CHK(apply_gmg(T2, T, true));
                OpBPcDot2<PCM_NONE> op{T, nullptr, S, T, 1.0, 0};
                PIB_CHK(launch_vec(s, n, op, true, 3, &nb, true, q));
            } else if (left && jac) {
                PIB_CHK(matmult(s, S, T2, nullptr, true, q));
                OpBPcDot2<PCM_JACOBI> op{T2, A.dinv, S, T, opc, 1};
                PIB_CHK(launch_vec(s, n, op, true, 3, &nb, true, q));
            } else {
                PIB_CHK(matmult(s, SH, T, nullptr, true, q));
                OpBPcDot2<PCM_NONE> op{T, nullptr, S, T, 1.0, 0};
                PIB_CHK(launch_vec(s, n, op, true, 3, &nb, true, q));
            }
            if (one_rank)
                PIB_CHK(finalize_post<2>(s, 3, 2, nb, nullptr, 0, q));
            else {
                PIB_CHK(finalize(s, 3, 2, nb, q));
                hipLaunchKernelGGL(k_b_s_omega, dim3(1), dim3(1), 0, q, s->d_s);
            }
            // x += alpha ph + omega sh ; r = s - omega t ; |r|^2, r.rp
            OpBUpdateX op{PH, SH, S, T, RP, x, R, 0.0, 0.0, 0};
            PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, true, q));
            if (one_rank)
                PIB_CHK(finalize_post<3>(s, 0, 2, nb, s->d_hist, conv_is_its, q));
            else {
                PIB_CHK(finalize(s, 0, 2, nb, q));
                hipLaunchKernelGGL(k_b_s_end, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, conv_is_its);
            }
            PIB_HIP(hipGetLastError());
            return 0;
        };
        PIB_CHK(run_iterations(s, todo, enq, graph_key(2, x, b), q, body));
        const bool first = enq == 0;
        enq += todo;
        if (first) {  // (as in solve_cg: closing kernels behind the first batch, one synchronisation if that was the solve)
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
