// krylov_ops.hpp -- the Krylov solvers' device side (krylov.hip): the vector-kernel launcher, CG's fused vector operations and the
// one-thread / one-workgroup scalar kernels of the recurrences (standard and single-reduction CG).
// Included by krylov.hip only (one translation unit).
#pragma once
#include "pib_internal.hpp"

namespace pib {

int spmv_rows(pib_solver *s, const double *x_owned, double *y, int64_t r_begin, int64_t r_end, double *dot_part,
              bool guarded, hipStream_t st);
int spmv_launch_blocks();

#ifndef PIB_VGRID_MAX
#define PIB_VGRID_MAX 2048
#endif
constexpr int VGRID_MAX = PIB_VGRID_MAX;

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <int W>
struct Pack {
    double v[W];
};
template <int W>
__device__ __forceinline__ Pack<W> ld(const double *p, int64_t i)
{
    Pack<W> r;
    if constexpr (W == 2) {
        const double2 t = *reinterpret_cast<const double2 *>(p + 2 * i);
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else {
        r.v[0] = p[i];
    }
    return r;
}
template <int W>
__device__ __forceinline__ void st(double *p, int64_t i, const Pack<W> &r)
{
    if constexpr (W == 2) {
        *reinterpret_cast<double2 *>(p + 2 * i) = make_double2(r.v[0], r.v[1]);
    } else {
        p[i] = r.v[0];
    }
}

// Generic fused vector kernel.  Op::NRED partial sums go to
// part[(k)*PIB_MAXPART + blockIdx.x].
// [e_begin, e_end): element range (multiples of W; the odd tail element belongs to the range that ends at n)
template <int W, class Op>
__global__ __launch_bounds__(256) void k_vec(const Scalars *__restrict__ S, int64_t n, Op op, double *__restrict__ part,
                                             int64_t e_begin, int64_t e_end, int64_t per = 0)
{
    if (S != nullptr && S->done) return;
    constexpr int NR = Op::NRED > 0 ? Op::NRED : 1;
    double acc[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) acc[k] = 0.0;
    op.prepare(S);
    const int64_t ng = (e_end == n) ? n / W : e_end / W;
    if (per > 0) {
        // large vectors: a contiguous range per workgroup (a 24 B/cell stream runs 0.56 instead of 0.60 ms per 512^3 pass
        // this way: tools/vcycle_lab.hip S) -- the partial sums are then grouped by range instead of by stride
        const int64_t lo = e_begin / W + (int64_t)blockIdx.x * per, hi = min(lo + per, ng);
#pragma unroll 2
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) op.template apply<W>(i, acc);
    } else {
        const int64_t stride = (int64_t)gridDim.x * 256;
#pragma unroll 2
        for (int64_t i = e_begin / W + (int64_t)blockIdx.x * 256 + threadIdx.x; i < ng; i += stride) op.template apply<W>(i, acc);
    }
    if (W == 2 && (n & 1) && e_end == n && blockIdx.x == 0 && threadIdx.x == 0) op.template apply<1>(n - 1, acc);
    if (Op::NRED > 0) {
        __shared__ double sh[NR][4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const double v = wsum(acc[k]);
            if (lane == 0) sh[k][w] = v;
        }
        __syncthreads();
        if (threadIdx.x < NR) {
            const int k = threadIdx.x;
            part[(int64_t)k * PIB_MAXPART + blockIdx.x] = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
        }
    }
}

// The same for operations without sums, one contiguous chunk of 4 x 256 packs per workgroup instead of a grid-stride
// loop: the dispatcher then sweeps a moving address window (tools/vcycle_lab.hip: 5.5 -> 5.8 TB/s for a 2-read-1-write
// stream; the SpMV gained 10-15 % from the same change).  Kernels with sums keep the bounded grid (their partials).
template <int W, class Op>
__global__ __launch_bounds__(256) void k_vec_chunk(const Scalars *__restrict__ S, int64_t n, Op op, int64_t e_begin, int64_t e_end)
{
    if (S != nullptr && S->done) return;
    double acc[1] = {0.0};
    op.prepare(S);
    const int64_t ng = (e_end == n) ? n / W : e_end / W;
    const int64_t base = e_begin / W + (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + 256 * u;
        if (i < ng) op.template apply<W>(i, acc);
    }
    if (W == 2 && (n & 1) && e_end == n && blockIdx.x == 0 && threadIdx.x == 0) op.template apply<1>(n - 1, acc);
}

// sum `count` partials of each of `nslots` slots (one block per slot), fixed order.
__global__ __launch_bounds__(256) void k_finalize(Scalars *__restrict__ S, const double *__restrict__ part, int slot0,
                                                  int count)
{
    if (S->done) return;
    const int slot = slot0 + blockIdx.x;
    const double *p = part + (int64_t)slot * PIB_MAXPART;
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) v += p[i];
    __shared__ double sh[4];
    v = wsum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) S->red[slot] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <class Op>
static int launch_vec(pib_solver *s, int64_t n, const Op &op, bool vec2, int slot0, int *nblocks_out, bool guarded,
                      hipStream_t stq, int64_t e_begin = 0, int64_t e_end = -1)
{
    if (e_end < 0) e_end = n;
    if (e_end <= e_begin) return 0;
    int64_t ng = vec2 ? (e_end - e_begin + 1) / 2 : (e_end - e_begin);
    int nb = (int)std::min<int64_t>(VGRID_MAX, std::max<int64_t>(1, (ng + 255) / 256));
    double *part = s->d_part + (int64_t)slot0 * PIB_MAXPART;
    const Scalars *S = guarded ? s->d_s : nullptr;
    if constexpr (Op::NRED == 0) {
        if (ng >= ((int64_t)1 << 20)) {  // large streaming update: chunked
            const unsigned nc = (unsigned)((ng + 1023) / 1024);
            if (vec2)
                hipLaunchKernelGGL((k_vec_chunk<2, Op>), dim3(nc), dim3(256), 0, stq, S, n, op, e_begin, e_end);
            else
                hipLaunchKernelGGL((k_vec_chunk<1, Op>), dim3(nc), dim3(256), 0, stq, S, n, op, e_begin, e_end);
            PIB_HIP(hipGetLastError());
            if (nblocks_out) *nblocks_out = (int)nc;
            return 0;
        }
    }
    int64_t per = 0;
    if (ng >= ((int64_t)1 << 22)) per = (((ng + nb - 1) / nb + 255) / 256) * 256;
    if (vec2)
        hipLaunchKernelGGL((k_vec<2, Op>), dim3(nb), dim3(256), 0, stq, S, n, op, part, e_begin, e_end, per);
    else
        hipLaunchKernelGGL((k_vec<1, Op>), dim3(nb), dim3(256), 0, stq, S, n, op, part, e_begin, e_end, per);
    PIB_HIP(hipGetLastError());
    if (nblocks_out) *nblocks_out = nb;
    return 0;
}

int allreduce_slots(pib_solver *s, int first, int count, hipStream_t stq)
{
    return comm_allreduce_sum(s, &s->d_s->red[first], count, stq);
}

static int finalize(pib_solver *s, int slot0, int nslots, int count, hipStream_t stq)
{
    hipLaunchKernelGGL(k_finalize, dim3(nslots), dim3(256), 0, stq, s->d_s, s->d_part, slot0, count);
    PIB_HIP(hipGetLastError());
    return allreduce_slots(s, slot0, nslots, stq);
}

// ------------------------------------------------------------------ ops
// reduction slot map (CG): 0 z.r  1 z.z  2 sum z  3 z[0] (pinned GMG)  4 r.r  5 sum r  6 p.w
constexpr int SLOT_PW = 6;
enum { PCM_NONE = 0, PCM_JACOBI = 1, PCM_EXTERNAL = 2 };

// r = b - w (guess) or r = b ; z = M^-1 r ; partials 0..5
template <int PCM>
struct OpInit {
    static constexpr int NRED = 6;
    const double *b, *w, *dinv;
    double *r, *z;
    double omega;
    int guess;
    int pin0;  // this rank owns the pinned row 0: its residual is exactly 0 (x[0] = b[0] was set)
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[6]) const
    {
        Pack<W> vb = ld<W>(b, i), vr, vz;
        if (guess) {
            Pack<W> vw = ld<W>(w, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vr.v[k] = vb.v[k] - vw.v[k];
        } else {
            vr = vb;
        }
        if (pin0 && i == 0) vr.v[0] = 0.0;
        if (PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vz.v[k] = omega * (vd.v[k] * vr.v[k]);
            st<W>(z, i, vz);
        } else {
            vz = vr;
        }
        st<W>(r, i, vr);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[0] += vz.v[k] * vr.v[k];
            acc[1] += vz.v[k] * vz.v[k];
            acc[2] += vz.v[k];
            acc[4] += vr.v[k] * vr.v[k];
            acc[5] += vr.v[k];
        }
    }
};

// r -= a w ; z = M^-1 r ; partials 0..4.  (x += a p rides on the next p-update, which reads p anyway: OpUpdateP.)
template <int PCM>
struct OpUpdateXR {
    static constexpr int NRED = 6;
    const double *w, *dinv;
    double *r, *z;
    double omega;
    double a;
    __device__ void prepare(const Scalars *S) { a = S->a; }
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[6]) const
    {
        Pack<W> vw = ld<W>(w, i), vr = ld<W>(r, i), vz;
#pragma unroll
        for (int k = 0; k < W; ++k) vr.v[k] = vr.v[k] - a * vw.v[k];
        st<W>(r, i, vr);
        if (PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vz.v[k] = omega * (vd.v[k] * vr.v[k]);
            st<W>(z, i, vz);
        } else {
            vz = vr;
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[0] += vz.v[k] * vr.v[k];
            acc[1] += vz.v[k] * vz.v[k];
            acc[2] += vz.v[k];
            acc[4] += vr.v[k] * vr.v[k];
            acc[5] += vr.v[k];
        }
    }
};

// partials 0 z.r, 1 z.z, 2 sum z -- after an external PC apply
struct OpDotZR {
    static constexpr int NRED = 3;
    const double *z, *r;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[3]) const
    {
        Pack<W> vz = ld<W>(z, i), vr = ld<W>(r, i);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[0] += vz.v[k] * vr.v[k];
            acc[1] += vz.v[k] * vz.v[k];
            acc[2] += vz.v[k];
        }
    }
};

// Pinned pressure row + multigrid where the shift cannot be lazy (single-reduction CG: the matrix is applied to z itself;
// BiCGStab: the cycle's output goes straight into a product): z <- z - z[0], z[0] = r[0] in a pass of its own -- what the
// oracle's PCAPPLY / pcapply do for nullspace 2 (oracle/csrc/gmg.c, oracle.c) -- with the sums z.r, z.z, sum z of the result
// (DOTS).  red[slot] holds z[0] of the raw cycle output (k_fetch_z0, all-reduced); `owner`: this rank holds row 0.
template <int DOTS>
struct OpPinShift {
    static constexpr int NRED = DOTS ? 3 : 0;
    double *z;
    const double *r;
    int owner, slot;
    double m;
    __device__ void prepare(const Scalars *S) { m = S->red[slot]; }
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[DOTS ? 3 : 1]) const
    {
        Pack<W> vz = ld<W>(z, i), vr = ld<W>(r, i);
#pragma unroll
        for (int k = 0; k < W; ++k) vz.v[k] = vz.v[k] - m;
        if (owner && i == 0) vz.v[0] = vr.v[0];
        st<W>(z, i, vz);
        if (DOTS) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                acc[0] += vz.v[k] * vr.v[k];
                acc[1] += vz.v[k] * vz.v[k];
                acc[DOTS ? 2 : 0] += vz.v[k];
            }
        }
    }
};

// x += a p (the update the previous iteration owes, elements [xlo, xhi) of the index space: x has no ghost entries) ;
// p = (z - mean) + b p      (first iteration: p = z - mean)
struct OpUpdateP {
    static constexpr int NRED = 0;
    const double *z;
    double *p;
    double *x;          // indexed like p; may be null (no x update)
    int64_t xlo, xhi;   // both even
    double bcoef, mean, a;
    int first, pend;
    __device__ void prepare(const Scalars *S)
    {
        bcoef = S->b;
        mean = S->mean;
        first = (S->its == 0);
        a = S->a;
        pend = (x != nullptr && S->xa_it != S->xapplied);
    }
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        Pack<W> vz = ld<W>(z, i), vp;
        if (first) {
#pragma unroll
            for (int k = 0; k < W; ++k) vp.v[k] = vz.v[k] - mean;
        } else {
            vp = ld<W>(p, i);
            if (pend && i * W >= xlo && i * W < xhi) {
                Pack<W> vx = ld<W>(x, i);
#pragma unroll
                for (int k = 0; k < W; ++k) vx.v[k] = vx.v[k] + a * vp.v[k];
                st<W>(x, i, vx);
            }
#pragma unroll
            for (int k = 0; k < W; ++k) vp.v[k] = (vz.v[k] - mean) + bcoef * vp.v[k];
        }
        st<W>(p, i, vp);
    }
};

// single-reduction CG: the whole vector part of an iteration in one pass --
//   p = (z - mean) + b p ;  w = s + b w  (= A p) ;  x += a p ;  r -= a w ;  z = M^-1 r (Jacobi / none) ; partials 0, 1, 2, 4, 5
// (the first iteration: p = z - mean, w = s).  PCM_EXTERNAL (multigrid): z is left alone, only r.r and sum r are summed.
template <int PCM>
struct OpSRUpdate {
    static constexpr int NRED = 6;
    const double *sv, *dinv;
    double *z, *p, *w, *x, *r;
    double omega;
    double bcoef, a, mean;
    int first;
    __device__ void prepare(const Scalars *S)
    {
        bcoef = S->b;
        a = S->a;
        mean = S->mean;
        first = (S->its == 0);
    }
    template <int W>
    __device__ void apply(int64_t i, double (&acc)[6]) const
    {
        Pack<W> vz = ld<W>(z, i), vs = ld<W>(sv, i), vp, vw;
        if (first) {
#pragma unroll
            for (int k = 0; k < W; ++k) {
                vp.v[k] = vz.v[k] - mean;
                vw.v[k] = vs.v[k];
            }
        } else {
            vp = ld<W>(p, i);
            vw = ld<W>(w, i);
#pragma unroll
            for (int k = 0; k < W; ++k) {
                vp.v[k] = (vz.v[k] - mean) + bcoef * vp.v[k];
                vw.v[k] = vs.v[k] + bcoef * vw.v[k];
            }
        }
        st<W>(p, i, vp);
        st<W>(w, i, vw);
        Pack<W> vx = ld<W>(x, i), vr = ld<W>(r, i);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            vx.v[k] = vx.v[k] + a * vp.v[k];
            vr.v[k] = vr.v[k] - a * vw.v[k];
        }
        st<W>(x, i, vx);
        st<W>(r, i, vr);
        if (PCM == PCM_JACOBI) {
            Pack<W> vd = ld<W>(dinv, i);
#pragma unroll
            for (int k = 0; k < W; ++k) vz.v[k] = omega * (vd.v[k] * vr.v[k]);
            st<W>(z, i, vz);
        } else if (PCM == PCM_NONE) {
            vz = vr;  // z aliases r
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (PCM != PCM_EXTERNAL) {
                acc[0] += vz.v[k] * vr.v[k];
                acc[1] += vz.v[k] * vz.v[k];
                acc[2] += vz.v[k];
            }
            acc[4] += vr.v[k] * vr.v[k];
            acc[5] += vr.v[k];
        }
    }
};

// the x update still owed when the iteration stops: x += a p
// (if_done: launched speculatively after the first batch of iterations -- acts only if the solve has stopped, see solve_cg)
__global__ __launch_bounds__(256) void k_flush_x(const Scalars *__restrict__ S, int64_t n, const double *__restrict__ p,
                                                 double *__restrict__ x, int if_done)
{
    if (if_done && !S->done) return;
    if (S->xa_it == S->xapplied) return;
    const double a = S->a;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = x[i] + a * p[i];
}
__global__ void k_flush_done(Scalars *S, int if_done)
{
    if (if_done && !S->done) return;
    S->xapplied = S->xa_it;
}

struct OpCopy {
    static constexpr int NRED = 0;
    const double *src;
    double *dst;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        st<W>(dst, i, ld<W>(src, i));
    }
};

struct OpFill {
    static constexpr int NRED = 0;
    double *dst;
    double value;
    __device__ void prepare(const Scalars *) {}
    template <int W>
    __device__ void apply(int64_t i, double (&)[1]) const
    {
        Pack<W> v;
#pragma unroll
        for (int k = 0; k < W; ++k) v.v[k] = value;
        st<W>(dst, i, v);
    }
};

// ------------------------------------------------------------ scalar kernels
__device__ __forceinline__ void converged_default(Scalars *S, double dp)
{
    if (dp != dp) {
        S->reason = PIB_DIVERGED_NANORINF;
        S->done = 1;
    } else if (dp <= S->ttol) {
        S->reason = (dp < S->atol) ? PIB_CONVERGED_ATOL : PIB_CONVERGED_RTOL;
        S->done = 1;
    } else if (S->dtol > 0.0 && dp >= S->dtol * S->rnorm0) {
        S->reason = PIB_DIVERGED_DTOL;
        S->done = 1;
    }
}

// lazy: 0 no shift; 1 z <- z - mean(z) (constant null space); 2 z <- z - z[0] (pinned pressure + multigrid).
// The shift m is applied lazily: p = (z - m) + b p in OpUpdateP, and the dot products are corrected here:
//   (z-m).r = z.r - m sum(r) ;  |z-m|^2 = z.z - 2 m sum(z) + n m^2
__device__ __forceinline__ void lazy_shift(Scalars *S, double n_global, int lazy, double &zr, double &zz)
{
    double m = 0.0;
    if (lazy == 1) m = S->red[2] / n_global;
    if (lazy == 2) m = S->red[3];
    if (lazy) {
        zr = zr - m * S->red[5];
        zz = (zz - 2.0 * m * S->red[2]) + n_global * m * m;
        if (zz < 0.0) zz = 0.0;
    }
    S->mean = m;
}

__global__ void k_fetch_z0(Scalars *S, const double *z, int owner, int slot = 3)
{
    if (S->done) return;
    S->red[slot] = owner ? z[0] : 0.0;
}

__global__ void k_pin_x0(double *x, const double *b) { x[0] = b[0]; }

__global__ void k_cg_s_init(Scalars *S, double *hist, double n_global, int lazy_mean, int monitor)
{
    double zr = S->red[0], zz = S->red[1], rr = S->red[4];
    lazy_shift(S, n_global, lazy_mean, zr, zz);
    const double dp = (S->normtype == 0) ? sqrt(zz) : sqrt(rr);
    S->dp = dp;
    S->rnorm0 = dp;
    S->ttol = monitor ? fmax(S->rtol * dp, S->atol) : -1.0;
    S->its = 0;
    S->reason = 0;
    S->done = 0;
    S->dpi = 0.0;
    S->dpiold = 0.0;
    S->b = 0.0;
    hist[0] = dp;
    converged_default(S, dp);
    S->beta = zr;
    S->betaold = zr;
    if (!S->done && S->maxit <= 0) {
        S->reason = PIB_DIVERGED_ITS;
        S->done = 1;
    }
    if (!S->done && zr == 0.0) {
        S->reason = PIB_CONVERGED_ATOL;
        S->its = 1;
        hist[1] = dp;
        S->done = 1;
    }
}

// pr.n > 0 (a pinned pressure row, this rank owns cell 0, the residual's sum is wanted ahead of the pass that forms the
// residual): sum r_new = sum r - a sum w, and sum w = -sum_f coef[f] p[off[f]] -- the columns of the singular operator sum to
// zero and p[0] = 0 (PinRow, pib_internal.hpp).  red[5] is the sum the pass that formed r delivered: re-based every iteration,
// the recurrence is one step long and carries no drift.
__device__ __forceinline__ void cg_pin_sigma(Scalars *S, const PinRowDev &pr)
{
    if (pr.n <= 0) return;
    double t = 0.0;
    for (int f = 0; f < pr.n; ++f) t = fma(pr.coef[f], pr.p[pr.off[f]], t);
    S->pin_sigma = S->red[5] + S->a * t;
}
__device__ __forceinline__ void cg_s1(Scalars *S)
{
    S->xapplied = S->xa_it;  // this iteration's p-update has applied what the previous one owed
    S->dpiold = S->dpi;
    const double dpi = S->red[6];
    S->dpi = dpi;
    if (dpi == 0.0 || dpi != dpi || (S->its > 0 && ((dpi > 0.0) != (S->dpiold > 0.0)))) {
        S->reason = (dpi != dpi) ? PIB_DIVERGED_NANORINF : PIB_DIVERGED_INDEFINITE_MAT;
        S->its += 1;
        S->done = 1;
        return;
    }
    S->a = S->beta / dpi;
    S->betaold = S->beta;
    S->xa_it += 1;  // x += a p is owed
}
__global__ void k_cg_s1(Scalars *S, PinRowDev pr)
{
    if (S->done) return;
    cg_s1(S);
    if (!S->done) cg_pin_sigma(S, pr);
}

// do_norm: evaluate the monitored norm + convergence; do_beta: new beta, b.
__device__ __forceinline__ void cg_s2(Scalars *S, double *hist, double n_global, int lazy_mean, int do_norm, int do_beta,
                                      int conv_is_its)
{
    double zr = S->red[0], zz = S->red[1], rr = S->red[4];
    if (do_beta) lazy_shift(S, n_global, lazy_mean, zr, zz);
    if (do_norm) {
        const double dp = (S->normtype == 0) ? sqrt(zz) : sqrt(rr);
        S->dp = dp;
        S->its += 1;
        hist[S->its] = dp;
        converged_default(S, dp);
        if (!S->done && S->its >= S->maxit) {
            S->reason = conv_is_its ? PIB_CONVERGED_ITS : PIB_DIVERGED_ITS;
            S->done = 1;
        }
    }
    if (do_beta && !S->done) {
        S->beta = zr;
        if (zr == 0.0) {
            S->reason = PIB_CONVERGED_ATOL;
            S->its += 1;
            hist[S->its] = S->dp;
            S->done = 1;
        } else if ((zr > 0.0) != (S->betaold > 0.0)) {
            S->reason = PIB_DIVERGED_INDEFINITE_PC;
            S->its += 1;
            hist[S->its] = S->dp;
            S->done = 1;
        } else {
            S->b = zr / S->betaold;
        }
    }
}
__global__ void k_cg_s2(Scalars *S, double *hist, double n_global, int lazy_mean, int do_norm, int do_beta,
                        int conv_is_its)
{
    if (S->done) return;
    cg_s2(S, hist, n_global, lazy_mean, do_norm, do_beta, conv_is_its);
}

// The closing kernel of a preconditioner application whose Krylov sums the V-cycle left as per-workgroup partials (gmg.hip
// reduce_dots: at most DEFER_DOTS_MAX of them per sum): z.r, z.z, sum z reduced in a fixed order by ONE workgroup, z[0] fetched
// (pinned null space), and -- POST, one rank: nothing sits between the sums and their consumer -- the iteration's scalar step.
// One launch where k_reduce_big, k_finalize_big, k_fetch_z0 and k_cg_s2 were four (round 5).
template <int POST>
__global__ __launch_bounds__(1024) void k_dots_tail(Scalars *__restrict__ S, const double *__restrict__ part, int stride, int count,
                                                    const double *__restrict__ z, int owner, double *hist, double n_global, int lazy_mean,
                                                    int do_norm, int do_beta, int conv_is_its)
{
    if (S->done) return;
    __shared__ double sh[16];
    for (int slot = 0; slot < 3; ++slot) {
        const double *p = part + (int64_t)slot * stride;
        double v = 0.0;
        for (int i = threadIdx.x; i < count; i += 1024) v += p[i];
        v = wsum(v);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 16; ++w) t += sh[w];
            S->red[slot] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        S->red[3] = owner ? z[0] : 0.0;
        if (POST) cg_s2(S, hist, n_global, lazy_mean, do_norm, do_beta, conv_is_its);
    }
}

// ---- single-reduction CG (KSPCGUseSingleReduction; oracle/csrc/oracle.c:orc_cg_single_reduction): the matrix is applied to z,
// s = A z, and the top of an iteration needs no sum of its own --
//     dpi = p'w = delta - beta^2 dpiold / betaold^2   (delta = z's; the first iteration: p = z, dpi = delta),   a = beta / dpi
// -- so ALL sums of an iteration (z.r, z.z, sum z, z[0], r.r, sum r, z.s) go through ONE all-reduce, behind the product.
// (z - m).s = z.s - m sum(s), and sum(s) = 1'A z = 0 for the symmetric operator with A 1 = 0 the lazy shift is used with: the
// term is dropped.
__device__ __forceinline__ void cg_sr_top(Scalars *S)
{
    S->dpiold = S->dpi;
    const double delta = S->red[6], beta = S->beta, bo = S->betaold;
    const double dpi = (S->its == 0) ? delta : delta - beta * beta * S->dpiold / (bo * bo);
    S->dpi = dpi;
    if (dpi == 0.0 || dpi != dpi || (S->its > 0 && ((dpi > 0.0) != (S->dpiold > 0.0)))) {
        S->reason = (dpi != dpi) ? PIB_DIVERGED_NANORINF : PIB_DIVERGED_INDEFINITE_MAT;
        S->its += 1;
        S->done = 1;
        return;
    }
    S->a = beta / dpi;
    S->betaold = beta;
}
// after the set-up (k_cg_s_init has beta, b = 0) and the first product s = A z
__global__ void k_cg_sr_first(Scalars *S)
{
    if (S->done) return;
    cg_sr_top(S);
}
// end of an iteration (norm, convergence, the new beta and b) and the top of the next one (dpi, a)
__global__ void k_cg_sr_step(Scalars *S, double *hist, double n_global, int lazy_mean, int conv_is_its)
{
    if (S->done) return;
    cg_s2(S, hist, n_global, lazy_mean, 1, 1, conv_is_its);
    if (S->done) return;
    cg_sr_top(S);
}
}  // namespace pib
