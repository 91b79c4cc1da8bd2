// dense.hip -- direct solve for the small systems of the path: `-ksp_type preonly -pc_type lu` (every
// forces_solver.info of the reference's decoupled-IBPM examples, e.g. examples/decoupledibpm/cylinder2dRe40_GPU/
// config/forces_solver.info: preonly + LU through superlu_dist) and AmgX `solver=DENSE_LU_SOLVER`.
//
// The force system E BN H (applications/decoupledibpm/decoupledibpm.cpp:186-196) is symmetric positive definite
// with a few hundred to a few thousand unknowns and is solved once per time step, so the set-up cost is paid once
// (static bodies) and the solve must be a single short launch: setMatrix forms the explicit inverse in HBM by
// Gauss-Jordan elimination (no pivoting: SPD / diagonally dominant matrices; a vanishing pivot is an error), one
// launch per column (workgroup i owns row i), all of them captured once per matrix order into a hipGraph -- a moving
// body re-factorises every time step; each launch is a rank-1 update streamed at HBM rate (n^3 * 16 B in total).  From
// 128 unknowns on the elimination runs 64 columns at a time with its products on the matrix cores (k_bgj_*, below).
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

// Gauss-Jordan step k, ONE launch: workgroup i owns row i of [M | Inv] (nobody else writes it), reads the pivot row k
// (which no workgroup modifies in step k) and its own factor f_i = M[i,k] / M[k,k] before touching anything:
//   row_i -= f_i * row_k   (i != k);   columns < k of M are already zero outside the diagonal, columns > k of Inv still are.
// The pivots stay unscaled until k_gj_scale divides every row by its diagonal entry.
// The elimination does not pivot (EBNH is symmetric positive definite for distinct Lagrangian points: E BN H with
// H = E^T up to the diagonal scalings), so a pivot that has collapsed RELATIVE to the matrix -- coincident or duplicated
// points, bodies closer than the kernel support -- is reported as PETSC_ERR_MAT_LU_ZRPVT like the LU of the reference
// would, instead of producing a garbage inverse: |pivot| <= 1e-13 * max |diagonal of the original matrix|.
__global__ __launch_bounds__(256) void k_dense_maxdiag(int64_t n, const double *__restrict__ M, double *__restrict__ out)
{
    double v = 0.0;
    for (int64_t r = threadIdx.x; r < n; r += 256) v = fmax(v, fabs(M[r * n + r]));
    __shared__ double sh[256];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}

__global__ __launch_bounds__(256) void k_gj_step(int64_t n, int64_t k, double *__restrict__ M, double *__restrict__ Inv,
                                                 int *__restrict__ bad, const double *__restrict__ maxdiag)
{
    if (*bad) return;
    const int64_t i = blockIdx.x;
    const double p = M[k * n + k];
    if (!(fabs(p) > 1e-300) || !(fabs(p) > 1e-13 * *maxdiag)) {
        if (i == 0 && threadIdx.x == 0) *bad = (int)k + 1;
        return;
    }
    if (i == k) return;
    const double f = M[i * n + k] / p;
    __syncthreads();  // every thread has its factor before the row (including M[i,k]) changes
    if (f == 0.0) return;
    for (int64_t c = k + threadIdx.x; c < n; c += 256) M[i * n + c] = M[i * n + c] - f * M[k * n + c];
    for (int64_t c = threadIdx.x; c <= k; c += 256) Inv[i * n + c] = Inv[i * n + c] - f * Inv[k * n + c];
}

__global__ __launch_bounds__(256) void k_gj_scale(int64_t n, const double *__restrict__ M, double *__restrict__ Inv,
                                                  const int *__restrict__ bad)
{
    if (*bad) return;
    const int64_t i = blockIdx.x;
    const double d = M[i * n + i];
    for (int64_t c = threadIdx.x; c < n; c += 256) Inv[i * n + c] = Inv[i * n + c] / d;
}

// ---- blocked in-place Gauss-Jordan (orders >= 2 blocks): the same elimination 64 columns at a time, so that nearly all
// of its 2 n^3 flops are 64 x 64 x 64 products out of LDS instead of n rank-1 updates streamed through the caches
// (n = 3456, the force system of a 1152-point plate: 3456 launches and 27 ms per factorisation before, a moving body pays
// that every time step).  W is the matrix padded with an identity to a multiple of 64 (leading dimension np); for the
// pivot block K:   P = inv(W_KK)   W_KJ = P W_KJ   W_IJ -= W_IK W_KJ   W_IK = -W_IK P   W_KK = P      (I, J != K)
// and after the last block W is the inverse.  No pivoting (see above); the scalar pivots inside a block are the ones the
// unblocked elimination meets, with the same relative test.
constexpr int GB = 64;

// P = inv(W_KK), one workgroup: scalar in-place Gauss-Jordan on the 64 x 64 block, a 4 x 4 patch per thread in registers;
// a step's pivot row and column go through LDS (two buffers in turn: one barrier per step)
__global__ __launch_bounds__(256) void k_bgj_diag(int64_t np, int64_t kb, double *__restrict__ W, double *__restrict__ P,
                                                  int *__restrict__ bad, const double *__restrict__ maxdiag)
{
    if (*bad) return;
    __shared__ double rowp[2][4][GB], colp[2][4][GB];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double *blk = W + (kb + 4 * ty) * np + kb + 4 * tx;
    double a[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[r][c] = blk[(int64_t)r * np + c];
    const double tiny = 1e-13 * *maxdiag;
    for (int p = 0; p < GB; ++p) {
        const int buf = p & 1, pr = p >> 2, pq = p & 3;
        // the threads holding the pivot row (column) publish their whole patch rows (columns): the readers pick line pq
        // (a selection among the registers here would turn the patch into an indexed array, i.e. scratch memory)
        if (ty == pr) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) rowp[buf][r][4 * tx + c] = a[r][c];
        }
        if (tx == pr) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) colp[buf][c][4 * ty + r] = a[r][c];
        }
        __syncthreads();
        const double piv = rowp[buf][pq][p];
        if (!(fabs(piv) > 1e-300) || !(fabs(piv) > tiny)) {
            if (tid == 0) *bad = (int)(kb + p) + 1;
            return;  // uniform: every thread reads the same pivot
        }
        const double d = 1.0 / piv;
        double rs[4], f[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rs[c] = rowp[buf][pq][4 * tx + c] * d;  // the scaled pivot row
#pragma unroll
        for (int r = 0; r < 4; ++r) f[r] = colp[buf][pq][4 * ty + r];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = 4 * ty + r, j = 4 * tx + c;
                if (i == p) a[r][c] = (j == p) ? d : rs[c];
                else a[r][c] = (j == p) ? -f[r] * d : __builtin_fma(-f[r], rs[c], a[r][c]);
            }
    }
    double *Pp = P + (4 * ty) * GB + 4 * tx;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            blk[(int64_t)r * np + c] = a[r][c];
            Pp[r * GB + c] = a[r][c];
        }
}

// The 64 x 64 x 64 block products on the matrix cores: v_mfma_f64_16x16x4_f64, wave w of the workgroup owns rows
// 16 w .. 16 w + 15 of the tile and its four 16 x 16 column tiles.  Operand layout (cdna_hip_programming.md): lane l feeds
// A[row = l & 15][k' = l >> 4] and B[k' = l >> 4][col = l & 15]; of its four results, number v is C[row = (l >> 4) + 4 v][col = l & 15].
// Which k of the product a lane group feeds in which of the 16 instructions is free as long as A and B agree: group g takes
// k = 16 g + step, so a lane's sixteen A operands are 128 contiguous bytes of its row -- they come straight from memory
// into registers, only B goes through LDS (an odd leading dimension spreads a wave's operand reads over the banks), and the
// C tile of the update is fetched ahead of the products.  37 KB of LDS and ~150 registers: three workgroups per CU.
// MODE 0: W_KJ = P W_KJ (J = blockIdx.x, J != K) ; MODE 1: W_IK = -W_IK P (I = blockIdx.x, I != K) ;
// MODE 2: W_IJ -= W_IK W_KJ (I = blockIdx.y, J = blockIdx.x, both != K)
constexpr int GLB = GB + 1;
template <int MODE>
__global__ __launch_bounds__(256) void k_bgj_block(int64_t np, int64_t kb, double *__restrict__ W, const double *__restrict__ P,
                                                   const int *__restrict__ bad, int64_t only = -1, int64_t skip = -1)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));
    if (*bad) return;
    const int64_t K = kb / GB;
    // MODE 2 with the look-ahead of dense_setup: `only` >= 0 -- the one block (only, only), a 1 x 1 grid; `skip` >= 0 -- every block
    // but (skip, skip), which the look-ahead has updated (and inverted) already
    const int64_t J = (MODE == 1) ? K : ((MODE == 2 && only >= 0) ? only : (int64_t)blockIdx.x),
                  I = (MODE == 0) ? K : (MODE == 1 ? (int64_t)blockIdx.x : ((only >= 0) ? only : (int64_t)blockIdx.y));
    if ((MODE != 1 && J == K) || (MODE != 0 && I == K)) return;
    if (MODE == 2 && I == skip && J == skip) return;
    __shared__ double Bs[GB][GLB];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, g = l >> 4;
    const double *Ap = (MODE == 0) ? P : W + I * GB * np + kb;        // left factor and its leading dimension
    const int64_t lda = (MODE == 0) ? GB : np;
    const double *Bp = (MODE == 1) ? P : W + kb * np + J * GB;        // right factor
    const int64_t ldb = (MODE == 1) ? GB : np;
    double av[16];
    const v2d *arow = reinterpret_cast<const v2d *>(Ap + (int64_t)(16 * w + lr) * lda + 16 * g);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const v2d t = arow[q];
        av[2 * q] = t[0];
        av[2 * q + 1] = t[1];
    }
    for (int e = tid; e < GB * GB / 2; e += 256) {
        const int r = e / (GB / 2), c = 2 * (e % (GB / 2));
        const v2d t = *reinterpret_cast<const v2d *>(Bp + (int64_t)r * ldb + c);
        Bs[r][c] = t[0];
        Bs[r][c + 1] = t[1];
    }
    double *Cp = W + (I * GB + 16 * w + g) * np + J * GB + lr;
    v4d acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (MODE == 2) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[t][v] = -Cp[(int64_t)(4 * v) * np + 16 * t];  // acc = -(C - A B) = -C + A B
        } else {
            acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
        }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], Bs[16 * g + kk][16 * t + lr], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double *o = Cp + (int64_t)(4 * v) * np + 16 * t;
            *o = (MODE == 0) ? acc[t][v] : -acc[t][v];
        }
}

// W = [M 0 ; 0 I] (leading dimension np) and back
__global__ __launch_bounds__(256) void k_bgj_pack(int64_t n, int64_t np, const double *__restrict__ M, double *__restrict__ W)
{
    const int64_t r = blockIdx.x;
    for (int64_t c = threadIdx.x; c < np; c += 256) W[r * np + c] = (r < n && c < n) ? M[r * n + c] : (r == c ? 1.0 : 0.0);
}
__global__ __launch_bounds__(256) void k_bgj_unpack(int64_t n, int64_t np, const double *__restrict__ W, double *__restrict__ Inv,
                                                    const int *__restrict__ bad)
{
    if (*bad) return;
    const int64_t r = blockIdx.x;
    for (int64_t c = threadIdx.x; c < n; c += 256) Inv[r * n + c] = W[r * np + c];
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

// y = Inv b with the explicit inverse of a solver set up for a direct solve, on the caller's stream and without a host
// synchronisation (the coupled immersed-boundary operator applies EBNH^-1 inside every Krylov product)
int dense_apply_raw(const pib_solver *s, const double *b, double *y, hipStream_t q)
{
    const int64_t n = s->dense_n;
    if (s->dense_inv == nullptr || n <= 0) return fail(PIB_ERR_ORDER, "solver %s: no explicit inverse", s->name.c_str());
    hipLaunchKernelGGL(k_dense_apply, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, q, n, s->dense_inv, b, y);
    PIB_HIP(hipGetLastError());
    return 0;
}

void dense_release(pib_solver *s)
{
    drop_iteration_graph(s);
    if (s->dense_graph) {  // (the factorisation's graph: before its buffers, its second stream and its events go)
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        (void)hipGraphExecDestroy(s->dense_graph);
    }
    if (s->dense_inv) (void)hipFree(s->dense_inv);
    if (s->dense_work) (void)hipFree(s->dense_work);
    if (s->dense_bad) (void)hipFree(s->dense_bad);
    if (s->dense_pad) (void)hipFree(s->dense_pad);
    if (s->dense_stream2) {
        (void)hipStreamDestroy(s->dense_stream2);
        for (int e = 0; e < 3; ++e)
            if (s->dense_ev[e]) (void)hipEventDestroy(s->dense_ev[e]);
        s->dense_stream2 = nullptr;
        s->dense_ev[0] = s->dense_ev[1] = s->dense_ev[2] = nullptr;
    }
    s->dense_pad = nullptr;
    s->dense_graph = nullptr;
    s->dense_inv = s->dense_work = nullptr;
    s->dense_bad = nullptr;
    s->dense_n = 0;
}

int dense_setup(pib_solver *s)
{
    if (s->comm.nranks > 1) return fail(PIB_ERR_SUP, "solver %s: the direct solver is single-rank (replicated)", s->name.c_str());
    const DeviceCsr &A = s->A;
    const int64_t n = A.n;
    if (n > DENSE_MAX_ROWS)
        return fail(PIB_ERR_SUP, "solver %s: direct solve asked for %lld rows (limit %lld); use a Krylov method", s->name.c_str(),
                    (long long)n, (long long)DENSE_MAX_ROWS);
    if (n == 0) {
        dense_release(s);
        return 0;
    }
    hipStream_t q = s->stream;
    const size_t bytes = sizeof(double) * (size_t)n * (size_t)n;
    const int64_t np = (n + GB - 1) / GB * GB;
    const bool blocked = n >= 2 * GB;
    // buffers (and the captured elimination graph) are kept while the size stays the same: a moving body re-factorises
    // a matrix of the same order every time step (rigidkinematics.cpp:135-139)
    if (s->dense_n != n || s->dense_inv == nullptr) {
        dense_release(s);
        PIB_HIP(hipMalloc(&s->dense_inv, bytes));
        PIB_HIP(hipMalloc(&s->dense_work, bytes));
        PIB_HIP(hipMalloc(&s->dense_bad, sizeof(int) + sizeof(double) * 2));  // flag + (8-byte aligned) max |diagonal|
        if (blocked) PIB_HIP(hipMalloc(&s->dense_pad, sizeof(double) * ((size_t)np * (size_t)np + 2 * GB * GB)));
        s->dense_n = n;
    }
    if (blocked && s->dense_stream2 == nullptr) {
        int prio_lo = 0, prio_hi = 0;  // (the look-ahead's one workgroup should not queue behind the trailing update's thousands)
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        PIB_HIP(hipStreamCreateWithPriority(&s->dense_stream2, hipStreamNonBlocking, prio_hi));
        for (int e = 0; e < 3; ++e) PIB_HIP(hipEventCreateWithFlags(&s->dense_ev[e], hipEventDisableTiming));
    }
    double *M = s->dense_work;
    PIB_HIP(hipMemsetAsync(M, 0, bytes, q));
    PIB_HIP(hipMemsetAsync(s->dense_inv, 0, bytes, q));
    PIB_HIP(hipMemsetAsync(s->dense_bad, 0, sizeof(int), q));
    const int gb = (int)std::min<int64_t>(4096, n);
    if (A.rp64)
        hipLaunchKernelGGL(k_dense_fill<int64_t>, dim3(gb), dim3(256), 0, q, n, (const int64_t *)A.rowptr, A.col, A.val, M);
    else
        hipLaunchKernelGGL(k_dense_fill<int32_t>, dim3(gb), dim3(256), 0, q, n, (const int32_t *)A.rowptr, A.col, A.val, M);
    // every rank holds partial sums of the entries: the full matrix on all of them (the inverse is replicated)
    if (s->reduce_via != nullptr) PIB_CHK(comm_allreduce_big(s->reduce_via, M, n * n, q));
    hipLaunchKernelGGL(k_dense_identity, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, n, s->dense_inv);
    double *maxdiag = reinterpret_cast<double *>(s->dense_bad) + 1;
    hipLaunchKernelGGL(k_dense_maxdiag, dim3(1), dim3(256), 0, q, n, M, maxdiag);
    PIB_HIP(hipGetLastError());
    // the n elimination launches as one hipGraph (their arguments are fixed for a given order n and buffers).  Not under
    // the test-only loopback transport: its ranks are threads of one process, and capturing / replaying graphs from
    // several threads at once proved unreliable in the HIP runtime (sporadic garbage in the eliminated matrix)
    auto eliminate = [&]() {
        if (blocked) {
            // Look-ahead (round 4): the inversion of a 64 x 64 diagonal block is one workgroup walking 64 pivots (50 us) and stood
            // between the trailing updates (47 us each on the matrix cores): 54 + 54 serial stages for the 3456 force unknowns of the
            // config-5 plate, 6 ms per factorisation, every time step of a moving body.  The next diagonal block only needs ITS
            // OWN update: it is updated alone and inverted on a second stream (into the other of two P buffers) while the trailing
            // update of the whole matrix -- less that block -- runs on the first.  The same operations on every block: same bits.
            double *W = s->dense_pad, *Pb[2] = {s->dense_pad + (size_t)np * (size_t)np, s->dense_pad + (size_t)np * (size_t)np + GB * GB};
            const unsigned nblk = (unsigned)(np / GB);
            hipStream_t q2 = s->dense_stream2;
            hipLaunchKernelGGL(k_bgj_pack, dim3((unsigned)np), dim3(256), 0, q, n, np, M, W);
            hipLaunchKernelGGL(k_bgj_diag, dim3(1), dim3(256), 0, q, np, (int64_t)0, W, Pb[0], s->dense_bad, maxdiag);
            for (int64_t kb = 0, K = 0; kb < np; kb += GB, ++K) {
                double *P = Pb[K & 1];
                const bool ahead = kb + GB < np;
                hipLaunchKernelGGL(k_bgj_block<0>, dim3(nblk), dim3(256), 0, q, np, kb, W, P, s->dense_bad, (int64_t)-1, (int64_t)-1);
                if (ahead) {
                    (void)hipEventRecord(s->dense_ev[0], q);
                    (void)hipStreamWaitEvent(q2, s->dense_ev[0], 0);
                    hipLaunchKernelGGL(k_bgj_block<2>, dim3(1, 1), dim3(256), 0, q2, np, kb, W, P, s->dense_bad, K + 1, (int64_t)-1);
                    (void)hipEventRecord(s->dense_ev[1], q2);  // column K of row block K + 1 has been read: block<1> may overwrite it
                    hipLaunchKernelGGL(k_bgj_diag, dim3(1), dim3(256), 0, q2, np, kb + GB, W, Pb[(K + 1) & 1], s->dense_bad, maxdiag);
                    (void)hipEventRecord(s->dense_ev[2], q2);
                }
                hipLaunchKernelGGL(k_bgj_block<2>, dim3(nblk, nblk), dim3(256), 0, q, np, kb, W, P, s->dense_bad, (int64_t)-1, ahead ? K + 1 : (int64_t)-1);
                if (ahead) (void)hipStreamWaitEvent(q, s->dense_ev[1], 0);
                hipLaunchKernelGGL(k_bgj_block<1>, dim3(nblk), dim3(256), 0, q, np, kb, W, P, s->dense_bad, (int64_t)-1, (int64_t)-1);
                if (ahead) (void)hipStreamWaitEvent(q, s->dense_ev[2], 0);
            }
            hipLaunchKernelGGL(k_bgj_unpack, dim3((unsigned)n), dim3(256), 0, q, n, np, W, s->dense_inv, s->dense_bad);
            return;
        }
        for (int64_t k = 0; k < n; ++k)
            hipLaunchKernelGGL(k_gj_step, dim3((unsigned)n), dim3(256), 0, q, n, k, M, s->dense_inv, s->dense_bad, maxdiag);
        hipLaunchKernelGGL(k_gj_scale, dim3((unsigned)n), dim3(256), 0, q, n, M, s->dense_inv, s->dense_bad);
    };
    if (s->reduce_via != nullptr && s->reduce_via->comm.loop != nullptr) {
        eliminate();
        PIB_HIP(hipGetLastError());
    } else {
    if (s->dense_graph == nullptr) {
        hipGraph_t g = nullptr;
        PIB_HIP(hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal));
        eliminate();
        const hipError_t e = hipStreamEndCapture(q, &g);
        if (e != hipSuccess || g == nullptr) return fail(PIB_ERR_LIB, "solver %s: capturing the factorisation failed (%s)", s->name.c_str(), hipGetErrorString(e));
        const hipError_t ei = hipGraphInstantiate(&s->dense_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) {
            s->dense_graph = nullptr;
            return fail(PIB_ERR_LIB, "solver %s: hipGraphInstantiate failed (%s)", s->name.c_str(), hipGetErrorString(ei));
        }
    }
    PIB_HIP(hipGraphLaunch(s->dense_graph, q));
    }
    int hbad = 0;
    PIB_HIP(hipMemcpyAsync(&hbad, s->dense_bad, sizeof(int), hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    if (hbad) {
        dense_release(s);
        return fail(PIB_ERR_MAT_LU_ZRPVT, "solver %s: zero (or relatively vanishing) pivot in row %d of the direct factorisation", s->name.c_str(), hbad - 1);
    }
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
