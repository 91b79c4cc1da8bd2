// dense.hip -- direct solve for the small systems of the path: `-ksp_type preonly -pc_type lu` (every
// forces_solver.info of the reference's decoupled-IBPM examples, e.g. examples/decoupledibpm/cylinder2dRe40_GPU/
// config/forces_solver.info: preonly + LU through superlu_dist) and AmgX `solver=DENSE_LU_SOLVER`.
//
// The force system E BN H (applications/decoupledibpm/decoupledibpm.cpp:186-196) is symmetric positive definite
// with a few hundred to a few thousand unknowns and is solved once per time step, so the set-up cost is paid once
// (static bodies) and the solve must be a single short launch: setMatrix forms the explicit inverse in HBM by
// Gauss-Jordan elimination (no pivoting: SPD / diagonally dominant matrices; a vanishing pivot is an error), one
// pair of launches per column, each a rank-1 update streamed at HBM rate (n^3 * 16 B in total: 0.13 s at n = 4000);
// solve is one dense mat-vec, one wave per row, fixed summation order.
#include <cmath>

#include "pib_internal.hpp"

namespace pib {

constexpr int64_t DENSE_MAX_ROWS = 32768;

template <typename RP>
__global__ __launch_bounds__(256) void k_dense_fill(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, double *__restrict__ M)
{
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x)
        for (int64_t q = rp[r] + threadIdx.x; q < rp[r + 1]; q += 256) M[r * n + col[q]] += val[q];
}

__global__ __launch_bounds__(256) void k_dense_identity(int64_t n, double *__restrict__ I)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) I[r * n + r] = 1.0;
}

// step k, part 1: scaled pivot row of [M | Inv] into rowM / rowI, elimination factors (column k of M) into fac
__global__ __launch_bounds__(256) void k_gj_pivot(int64_t n, int64_t k, double *__restrict__ M, double *__restrict__ Inv,
                                                  double *__restrict__ rowM, double *__restrict__ rowI,
                                                  double *__restrict__ fac, int *__restrict__ bad)
{
    const double p = M[k * n + k];
    if (!(fabs(p) > 1e-300)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *bad = (int)k + 1;
        return;
    }
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
        rowM[c] = M[k * n + c] / p;
        rowI[c] = Inv[k * n + c] / p;
        fac[c] = (c == k) ? 0.0 : M[c * n + k];
    }
}

// step k, part 2: rows i != k:  M[i, k:] -= f_i rowM[k:],  Inv[i, :k+1] -= f_i rowI[:k+1]; row k := the scaled row.
// (columns < k of M are already unit vectors, columns > k of Inv still are.)
__global__ __launch_bounds__(256) void k_gj_update(int64_t n, int64_t k, double *__restrict__ M, double *__restrict__ Inv,
                                                   const double *__restrict__ rowM, const double *__restrict__ rowI,
                                                   const double *__restrict__ fac, const int *__restrict__ bad)
{
    if (*bad) return;
    const int64_t i = blockIdx.y;
    const double f = fac[i];
    const bool piv = (i == k);
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
        if (c >= k) M[i * n + c] = piv ? rowM[c] : M[i * n + c] - f * rowM[c];
        if (c <= k) Inv[i * n + c] = piv ? rowI[c] : Inv[i * n + c] - f * rowI[c];
    }
}

// y = Inv b, one wave per row
__global__ __launch_bounds__(256) void k_dense_apply(int64_t n, const double *__restrict__ Inv, const double *__restrict__ b,
                                                     double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    double s = 0.0;
    for (int64_t c = lane; c < n; c += 64) s += Inv[r * n + c] * b[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) y[r] = s;
}

void dense_release(pib_solver *s)
{
    if (s->dense_inv) (void)hipFree(s->dense_inv);
    s->dense_inv = nullptr;
    s->dense_n = 0;
}

int dense_setup(pib_solver *s)
{
    dense_release(s);
    if (s->comm.nranks > 1) return fail(PIB_ERR_SUP, "solver %s: the direct solver is single-rank", s->name.c_str());
    const DeviceCsr &A = s->A;
    const int64_t n = A.n;
    if (n > DENSE_MAX_ROWS)
        return fail(PIB_ERR_SUP, "solver %s: direct solve asked for %lld rows (limit %lld); use a Krylov method", s->name.c_str(),
                    (long long)n, (long long)DENSE_MAX_ROWS);
    if (n == 0) return 0;
    hipStream_t q = s->stream;
    double *M = nullptr, *rows = nullptr;
    int *bad = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n * (size_t)n;
    PIB_HIP(hipMalloc(&s->dense_inv, bytes));
    PIB_HIP(hipMalloc(&M, bytes));
    PIB_HIP(hipMalloc(&rows, sizeof(double) * 3 * (size_t)n));
    PIB_HIP(hipMalloc(&bad, sizeof(int)));
    PIB_HIP(hipMemsetAsync(M, 0, bytes, q));
    PIB_HIP(hipMemsetAsync(s->dense_inv, 0, bytes, q));
    PIB_HIP(hipMemsetAsync(bad, 0, sizeof(int), q));
    const int gb = (int)std::min<int64_t>(4096, n);
    if (A.rp64)
        hipLaunchKernelGGL(k_dense_fill<int64_t>, dim3(gb), dim3(256), 0, q, n, (const int64_t *)A.rowptr, A.col, A.val, M);
    else
        hipLaunchKernelGGL(k_dense_fill<int32_t>, dim3(gb), dim3(256), 0, q, n, (const int32_t *)A.rowptr, A.col, A.val, M);
    const int g1 = (int)((n + 255) / 256);
    hipLaunchKernelGGL(k_dense_identity, dim3(g1), dim3(256), 0, q, n, s->dense_inv);
    double *rowM = rows, *rowI = rows + n, *fac = rows + 2 * n;
    for (int64_t k = 0; k < n; ++k) {
        hipLaunchKernelGGL(k_gj_pivot, dim3(g1), dim3(256), 0, q, n, k, M, s->dense_inv, rowM, rowI, fac, bad);
        hipLaunchKernelGGL(k_gj_update, dim3(g1, (unsigned)n), dim3(256), 0, q, n, k, M, s->dense_inv, rowM, rowI, fac, bad);
    }
    PIB_HIP(hipGetLastError());
    int hbad = 0;
    PIB_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    PIB_HIP(hipFree(M));
    PIB_HIP(hipFree(rows));
    PIB_HIP(hipFree(bad));
    if (hbad) {
        dense_release(s);
        return fail(PIB_ERR_MAT_LU_ZRPVT, "solver %s: zero pivot in row %d of the direct factorisation", s->name.c_str(), hbad - 1);
    }
    s->dense_n = n;
    return 0;
}

// KSPPREONLY: one application of the (exact) preconditioner; its = 1, no residual norm is computed
int solve_direct(pib_solver *s, double *x, const double *b)
{
    if (s->dense_inv == nullptr && s->A.n > 0)
        return fail(PIB_ERR_ORDER, "solver %s: direct solve before the factorisation", s->name.c_str());
    const int64_t n = s->A.n;
    hipStream_t q = s->stream;
    const double *bb = b;
    if (x == b) {  // in-place call: the mat-vec needs the whole right-hand side
        PIB_CHK(ensure_work(s, 1));
        PIB_HIP(hipMemcpyAsync(s->vec(0), b, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, q));
        bb = s->vec(0);
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_dense_apply, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, q, n, s->dense_inv, bb, x);
        PIB_HIP(hipGetLastError());
    }
    PIB_HIP(hipStreamSynchronize(q));
    s->iters = 1;
    s->reason = 4;  // KSP_CONVERGED_ITS
    s->residual = 0.0;
    s->history.assign(1, 0.0);
    return 0;
}

}  // namespace pib
