// spmv_lab.hip -- stand-alone A/B harness for the CSR SpMV kernel (K1) on gfx950.
// Not part of the product: it exists to try kernel variants against each
// other on the 512^3 7-point operator and print achieved GB/s per variant
// (algorithmic bytes = 12 nnz + 4 (n+1) + 16 n).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/spmv_lab.hip -o tools/spmv_lab
//   tools/spmv_lab [n=512] [reps=20]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__host__ __device__ inline int64_t nnz_before(int64_t g, int64_t nx, int64_t ny, int64_t nz)
{
    const int64_t pl = nx * ny;
    int64_t c = g;
    c += g - (g + nx - 1) / nx;
    c += g - g / nx;
    const int64_t kq = g / pl, rem = g % pl;
    c += g - (kq * nx + (rem < nx ? rem : nx));
    const int64_t top = rem - (ny - 1) * nx;
    c += g - (kq * nx + (top > 0 ? top : 0));
    c += g - (g < pl ? g : pl);
    const int64_t last = g - (nz - 1) * pl;
    c += g - (last > 0 ? last : 0);
    return c;
}

__global__ void k_build(int64_t nx, int64_t ny, int64_t nz, int32_t *rowptr, int32_t *col, double *val)
{
    const int64_t n = nx * ny * nz, pl = nx * ny;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= n; r += (int64_t)gridDim.x * 256) {
        int64_t p = nnz_before(r, nx, ny, nz);
        rowptr[r] = (int32_t)p;
        if (r == n) break;
        const int64_t i = r % nx, j = (r / nx) % ny, k = r / pl;
        double d = 0.0;
        const double c = 1.0 + 1e-3 * (double)(r % 7);
        if (k > 0) { col[p] = (int32_t)(r - pl); val[p] = c; d -= c; ++p; }
        if (j > 0) { col[p] = (int32_t)(r - nx); val[p] = c; d -= c; ++p; }
        if (i > 0) { col[p] = (int32_t)(r - 1); val[p] = c; d -= c; ++p; }
        const int64_t pd = p++;
        if (i < nx - 1) { col[p] = (int32_t)(r + 1); val[p] = c; d -= c; ++p; }
        if (j < ny - 1) { col[p] = (int32_t)(r + nx); val[p] = c; d -= c; ++p; }
        if (k < nz - 1) { col[p] = (int32_t)(r + pl); val[p] = c; d -= c; ++p; }
        col[pd] = (int32_t)r;
        val[pd] = d;
    }
}

__global__ void k_fill(int64_t n, double *x)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        uint64_t h = (uint64_t)p * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        x[p] = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

struct Order {
    int mode;          // 0: contiguous per XCD; 1: plane-interleaved tiles; 2: plain round-robin (no XCD awareness)
    int64_t cpp;       // chunks per plane
    int64_t cpt;       // chunks per tile
    int64_t nplanes;
};

// sequence position -> chunk id
__device__ __forceinline__ int64_t chunk_of(const Order &o, int64_t seq, int64_t nchunks)
{
    if (o.mode != 1 && o.mode != 4) return seq;
    const int64_t per_tile = o.cpt * o.nplanes;
    const int64_t tile = seq / per_tile, rem = seq % per_tile;
    const int64_t k = rem / o.cpt, w = rem % o.cpt;
    const int64_t c = k * o.cpp + tile * o.cpt + w;
    return c < nchunks ? c : nchunks;  // callers skip >= nchunks
}

template <int ROWS, int NPT, bool WIDE, bool NT>
__global__ __launch_bounds__(256) void k_spmv(int64_t n, const int32_t *__restrict__ rowptr,
                                              const int32_t *__restrict__ col, const double *__restrict__ val,
                                              const double *__restrict__ x, double *__restrict__ y, Order o)
{
    constexpr int TILE = 256 * NPT;
    __shared__ double prod[TILE + 2];
    __shared__ int32_t srow[ROWS + 1];
    const int tid = threadIdx.x;
    const int64_t nchunks = (n + ROWS - 1) / ROWS;
    int64_t s_lo, s_hi, s_step, s0;
    if (o.mode == 2) {
        s_lo = 0; s_hi = nchunks; s0 = blockIdx.x; s_step = gridDim.x;
    } else if (o.mode >= 3) {  // non-persistent: one chunk per block, XCD x walks its contiguous sequence range
        const int xcd = blockIdx.x & 7;
        const int64_t cpx = (nchunks + 7) >> 3;
        s_lo = (int64_t)xcd * cpx;
        s_hi = (s_lo + cpx < nchunks) ? s_lo + cpx : nchunks;
        s0 = s_lo + (blockIdx.x >> 3);
        s_step = (int64_t)1 << 60;
    } else {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, bpx = gridDim.x >> 3;
        const int64_t cpx = (nchunks + 7) >> 3;
        s_lo = (int64_t)xcd * cpx;
        s_hi = (s_lo + cpx < nchunks) ? s_lo + cpx : nchunks;
        s0 = s_lo + j;
        s_step = bpx;
    }
    for (int64_t sq = s0; sq < s_hi; sq += s_step) {
        const int64_t c = chunk_of(o, sq, nchunks);
        if (c >= nchunks) continue;
        const int64_t r0 = c * ROWS;
        const int nr = (int)((n - r0 < ROWS) ? (n - r0) : ROWS);
        for (int t = tid; t <= nr; t += 256) srow[t] = rowptr[r0 + t];
        __syncthreads();
        const int32_t p0 = srow[0], p1 = srow[nr];
        double sum[ROWS / 256];
        int32_t rs[ROWS / 256], re[ROWS / 256];
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) {
            const int t = tid + q * 256;
            rs[q] = (t < nr) ? srow[t] : 0;
            re[q] = (t < nr) ? srow[t + 1] : 0;
            sum[q] = 0.0;
        }
        const int32_t a0 = WIDE ? (p0 & ~1) : p0;
        for (int32_t t0 = a0; t0 < p1; t0 += TILE) {
            if (WIDE) {
#pragma unroll
                for (int u = 0; u < NPT / 2; ++u) {
                    const int32_t q = t0 + 2 * (tid + u * 256);
                    if (q < p1) {
                        double2 v;
                        int2 cc;
                        if (NT) {
                            v.x = __builtin_nontemporal_load(val + q);
                            v.y = __builtin_nontemporal_load(val + q + 1);
                            cc.x = __builtin_nontemporal_load(col + q);
                            cc.y = __builtin_nontemporal_load(col + q + 1);
                        } else {
                            v = *reinterpret_cast<const double2 *>(val + q);
                            cc = *reinterpret_cast<const int2 *>(col + q);
                        }
                        const double x0 = (q >= p0) ? x[cc.x] : 0.0;
                        const double x1 = (q + 1 < p1) ? x[cc.y] : 0.0;
                        prod[q - t0] = v.x * x0;
                        prod[q - t0 + 1] = v.y * x1;
                    }
                }
            } else {
                int32_t cc[NPT];
                double vv[NPT];
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int32_t q = t0 + tid + u * 256;
                    if (NT) {
                        cc[u] = (q < p1) ? __builtin_nontemporal_load(col + q) : 0;
                        vv[u] = (q < p1) ? __builtin_nontemporal_load(val + q) : 0.0;
                    } else {
                        cc[u] = (q < p1) ? col[q] : 0;
                        vv[u] = (q < p1) ? val[q] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const int32_t q = t0 + tid + u * 256;
                    if (q < p1) prod[q - t0] = vv[u] * x[cc[u]];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < ROWS / 256; ++q) {
                const int32_t lo = (rs[q] > t0) ? rs[q] : t0;
                const int32_t hi = (re[q] < t0 + TILE) ? re[q] : (t0 + TILE);
                for (int32_t p = lo; p < hi; ++p) sum[q] = sum[q] + prod[p - t0];
            }
            __syncthreads();
        }
        if (p1 <= a0) __syncthreads();
#pragma unroll
        for (int q = 0; q < ROWS / 256; ++q) {
            const int t = tid + q * 256;
            if (t < nr) y[r0 + t] = sum[q];
        }
    }
}


// ---- wave-private CSR-stream: every wave owns 64 rows, stages its products in its own 4 KiB LDS slice,
// no block barrier anywhere (LDS ops of one wave are executed in order).
// PERSIST: grid-stride over chunk sequence; otherwise one 256-row group per block (4 waves x 64 rows).
template <bool WIDE, bool PERSIST>
__global__ __launch_bounds__(256) void k_spmv_wave(int64_t n, const int32_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ col, const double *__restrict__ val,
                                                   const double *__restrict__ x, double *__restrict__ y, Order o)
{
    constexpr int WT = 512;  // products per wave tile
    __shared__ double prod_all[4][WT + 2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *prod = prod_all[w];
    const int64_t nchunks = (n + 255) / 256;  // 256-row groups, as in the block kernel
    int64_t s_hi, s_step, s0;
    if (PERSIST) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, bpx = gridDim.x >> 3;
        const int64_t cpx = (nchunks + 7) >> 3;
        const int64_t s_lo = (int64_t)xcd * cpx;
        s_hi = (s_lo + cpx < nchunks) ? s_lo + cpx : nchunks;
        s0 = s_lo + j;
        s_step = bpx;
    } else {
        // one group per block; XCD-aware: block b runs on XCD b%8, which walks its own contiguous sequence range
        const int xcd = blockIdx.x & 7;
        const int64_t cpx = (nchunks + 7) >> 3;
        s0 = (int64_t)xcd * cpx + (blockIdx.x >> 3);
        s_hi = ((int64_t)xcd * cpx + cpx < nchunks) ? (int64_t)xcd * cpx + cpx : nchunks;
        s_step = (int64_t)1 << 60;
    }
    for (int64_t sq = s0; sq < s_hi; sq += s_step) {
        const int64_t c = chunk_of(o, sq, nchunks);
        if (c >= nchunks) continue;
        const int64_t r = c * 256 + w * 64 + lane;
        int32_t rs = 0, re = 0;
        if (r < n) {
            rs = rowptr[r];
            re = rowptr[r + 1];
        }
        const int32_t p0 = __shfl(rs, 0, 64);
        // last valid lane's end
        int32_t p1 = re;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int32_t t = __shfl_down(p1, off, 64);
            p1 = (t > p1) ? t : p1;
        }
        p1 = __shfl(p1, 0, 64);
        if (c * 256 + w * 64 >= n) continue;
        double sum = 0.0;
        const int32_t a0 = WIDE ? (p0 & ~1) : p0;
        for (int32_t t0 = a0; t0 < p1; t0 += WT) {
            if (WIDE) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t q = t0 + 2 * (lane + u * 64);
                    if (q < p1) {
                        const double2 v = *reinterpret_cast<const double2 *>(val + q);
                        const int2 cc = *reinterpret_cast<const int2 *>(col + q);
                        const double x0 = (q >= p0) ? x[cc.x] : 0.0;
                        const double x1 = (q + 1 < p1) ? x[cc.y] : 0.0;
                        prod[q - t0] = v.x * x0;
                        prod[q - t0 + 1] = v.y * x1;
                    }
                }
            } else {
                int32_t cc[8];
                double vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t q = t0 + lane + u * 64;
                    cc[u] = (q < p1) ? col[q] : 0;
                    vv[u] = (q < p1) ? val[q] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t q = t0 + lane + u * 64;
                    if (q < p1) prod[q - t0] = vv[u] * x[cc[u]];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int32_t lo = (rs > t0) ? rs : t0;
            const int32_t hi = (re < t0 + WT) ? re : (t0 + WT);
            for (int32_t p = lo; p < hi; ++p) sum = sum + prod[p - t0];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (r < n) y[r] = sum;
    }
}


// ---- variant T: LDS transpose.  Phase 1 copies the chunk's val/col span into LDS with wide coalesced loads
// (no gather).  Phase 2: thread t owns row t, reads its (val,col) pairs back from LDS and gathers x with the
// row-per-lane mapping, so every gather instruction reads ~64 consecutive x values (4-5 cache lines) instead of
// ~9 rows x 7 diagonals (10+ lines).  Row sum sequential in CSR order, products rounded: bit-identical.
template <int ROWS, int NPT, bool PERSIST>
__global__ __launch_bounds__(256) void k_spmv_t(int64_t n, const int32_t *__restrict__ rowptr,
                                                const int32_t *__restrict__ col, const double *__restrict__ val,
                                                const double *__restrict__ x, double *__restrict__ y, Order o)
{
    constexpr int TILE = 256 * NPT;
    __shared__ __attribute__((aligned(16))) double vals[TILE + 2];
    __shared__ __attribute__((aligned(16))) int32_t cols[TILE + 2];
    __shared__ int32_t srow[ROWS + 1];
    const int tid = threadIdx.x;
    const int64_t nchunks = (n + ROWS - 1) / ROWS;
    const int xcd = blockIdx.x & 7;
    const int64_t cpx = (nchunks + 7) >> 3;
    const int64_t s_lo = (int64_t)xcd * cpx;
    const int64_t s_hi = (s_lo + cpx < nchunks) ? s_lo + cpx : nchunks;
    const int64_t s0 = s_lo + (blockIdx.x >> 3);
    const int64_t s_step = PERSIST ? (gridDim.x >> 3) : ((int64_t)1 << 60);
    for (int64_t sq = s0; sq < s_hi; sq += s_step) {
        const int64_t c = chunk_of(o, sq, nchunks);
        if (c >= nchunks) continue;
        const int64_t r0 = c * ROWS;
        const int nr = (int)((n - r0 < ROWS) ? (n - r0) : ROWS);
        for (int t = tid; t <= nr; t += 256) srow[t] = rowptr[r0 + t];
        __syncthreads();
        const int32_t p0 = srow[0], p1 = srow[nr];
        int32_t rs = 0, re = 0;
        if (tid < nr) {
            rs = srow[tid];
            re = srow[tid + 1];
        }
        double sum = 0.0;
        const int32_t a0 = p0 & ~1;
        for (int32_t t0 = a0; t0 < p1; t0 += TILE) {
#pragma unroll
            for (int u = 0; u < NPT / 2; ++u) {
                const int32_t q = t0 + 2 * (tid + u * 256);
                if (q < p1) {
                    *reinterpret_cast<double2 *>(&vals[q - t0]) = *reinterpret_cast<const double2 *>(val + q);
                    *reinterpret_cast<int2 *>(&cols[q - t0]) = *reinterpret_cast<const int2 *>(col + q);
                }
            }
            __syncthreads();
            const int32_t lo = (rs > t0) ? rs : t0;
            const int32_t hi = (re < t0 + TILE) ? re : (t0 + TILE);
            for (int32_t p = lo; p < hi; p += 8) {
                double vv[8], xx[8];
                int32_t cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = p + u < hi;
                    cc[u] = ok ? cols[p + u - t0] : 0;
                    vv[u] = ok ? vals[p + u - t0] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xx[u] = (p + u < hi) ? x[cc[u]] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p + u < hi) sum = sum + vv[u] * xx[u];
            }
            __syncthreads();
        }
        if (p1 <= a0) __syncthreads();
        if (tid < nr) y[r0 + tid] = sum;
    }
}


// ---- variant T2: as T (non-persistent), but (i) p0/p1 come from two uniform loads so phase 1 starts at once,
// per-row offsets go straight to registers (no srow in LDS, one barrier less); (ii) LDS sized TILE=1792+pad for
// 7-entry rows (7 instead of 6 workgroups per CU); (iii) optional fused x.y dot reusing the gathered diagonal.
template <int TILE, bool DOT>
__global__ __launch_bounds__(256) void k_spmv_t2(int64_t n, const int32_t *__restrict__ rowptr,
                                                 const int32_t *__restrict__ col, const double *__restrict__ val,
                                                 const double *__restrict__ x, double *__restrict__ y, Order o,
                                                 double *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) double vals[TILE + 2];
    __shared__ __attribute__((aligned(16))) int32_t cols[TILE + 2];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t nchunks = (n + 255) / 256;
    const int64_t cpx = (nchunks + 7) >> 3;
    const int64_t sq = (int64_t)(blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    double dacc = 0.0;
    if ((int64_t)(blockIdx.x >> 3) < cpx && sq < nchunks) {
        const int64_t c = chunk_of(o, sq, nchunks);
        const int64_t r0 = c * 256;
        const int nr = (int)((n - r0 < 256) ? (n - r0) : 256);
        const int32_t p0 = rowptr[r0], p1 = rowptr[r0 + nr];  // uniform: scalar loads
        int32_t rs = 0, re = 0;
        if (tid < nr) {
            rs = rowptr[r0 + tid];
            re = rowptr[r0 + tid + 1];
        }
        double sum = 0.0;
        const int32_t a0 = p0 & ~1;
        for (int32_t t0 = a0; t0 < p1; t0 += TILE) {
#pragma unroll
            for (int u = 0; u < (TILE + 511) / 512; ++u) {
                const int32_t q = t0 + 2 * (tid + u * 256);
                if (q < p1 && q - t0 < TILE) {
                    *reinterpret_cast<double2 *>(&vals[q - t0]) = *reinterpret_cast<const double2 *>(val + q);
                    *reinterpret_cast<int2 *>(&cols[q - t0]) = *reinterpret_cast<const int2 *>(col + q);
                }
            }
            __syncthreads();
            const int32_t lo = (rs > t0) ? rs : t0;
            const int32_t hi = (re < t0 + TILE) ? re : (t0 + TILE);
            for (int32_t p = lo; p < hi; p += 8) {
                double vv[8], xx[8];
                int32_t cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = p + u < hi;
                    cc[u] = ok ? cols[p + u - t0] : 0;
                    vv[u] = ok ? vals[p + u - t0] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xx[u] = (p + u < hi) ? x[cc[u]] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p + u < hi) {
                        sum = sum + vv[u] * xx[u];
                        if (DOT && cc[u] == (int32_t)(r0 + tid)) dacc = xx[u];
                    }
            }
            if (t0 + TILE < p1) __syncthreads();
        }
        if (tid < nr) {
            y[r0 + tid] = sum;
            if (DOT) dacc = dacc * sum;
        }
    }
    if (DOT) {
        // wave sums -> LDS -> one value per workgroup
        double v = dacc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// streaming floor: read val/col/rowptr, write y, no gather (x[0] only)
__global__ __launch_bounds__(256) void k_stream_floor(int64_t n, int64_t nnz, const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ x, double *__restrict__ y)
{
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t q = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); q + 1 < nnz; q += 2 * stride) {
        const double2 v = *reinterpret_cast<const double2 *>(val + q);
        const int2 c = *reinterpret_cast<const int2 *>(col + q);
        acc += v.x * (double)c.x + v.y * (double)c.y;
    }
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) y[r] = acc + (double)rowptr[r] + x[r];
}

int main(int argc, char **argv)
{
    const int64_t nn = argc > 1 ? atoll(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int64_t nx = nn, ny = nn, nz = nn, n = nx * ny * nz;
    const int64_t nnz = nnz_before(n, nx, ny, nz);
    printf("grid %lld^3 n=%lld nnz=%lld\n", (long long)nn, (long long)n, (long long)nnz);
    int32_t *rowptr, *col;
    double *val, *x, *y, *yref;
    CK(hipMalloc(&rowptr, 4 * (n + 1)));
    CK(hipMalloc(&col, 4 * (nnz + 4)));
    CK(hipMalloc(&val, 8 * (nnz + 4)));
    CK(hipMalloc(&x, 8 * n));
    CK(hipMalloc(&y, 8 * n));
    CK(hipMalloc(&yref, 8 * n));
    CK(hipMemset(col, 0, 4 * (nnz + 4)));
    CK(hipMemset(val, 0, 8 * (nnz + 4)));
    hipLaunchKernelGGL(k_build, dim3(8192), dim3(256), 0, 0, nx, ny, nz, rowptr, col, val);
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, n, x);
    CK(hipDeviceSynchronize());
    const double alg = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> h0(4096), h1(4096);

    auto bench = [&](const char *name, auto launch, bool check) {
        launch();
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        int bad = -1;
        if (check) {
            // compare a few windows against the reference result
            bad = 0;
            for (int64_t off : {int64_t(0), n / 3, n / 2 + 12345, n - 4096}) {
                CK(hipMemcpy(h0.data(), yref + off, 8 * 4096, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h1.data(), y + off, 8 * 4096, hipMemcpyDeviceToHost));
                for (int i = 0; i < 4096; ++i) bad += (h0[i] != h1[i]);
            }
        }
        printf("%-44s %8.3f ms  %7.1f GB/s  %5.1f%% of 8 TB/s  mismatches=%d\n", name, ms, alg / ms / 1e6,
               alg / ms / 1e6 / 80.0, bad);
        fflush(stdout);
    };

    Order nat{0, 0, 0, 0}, rr{2, 0, 0, 0};
    // reference result
    hipLaunchKernelGGL((k_spmv<256, 8, false, false>), dim3(2048), dim3(256), 0, 0, n, rowptr, col, val, x, yref, nat);
    CK(hipDeviceSynchronize());

    bench("floor: stream val/col/rowptr + y (no gather)", [&] {
        hipLaunchKernelGGL(k_stream_floor, dim3(2048), dim3(256), 0, 0, n, nnz, rowptr, col, val, x, y); }, false);
#define RUN(NAME, ROWS, NPT, WIDE, NT, GRID, ORD)                                                              \
    bench(NAME, [&] {                                                                                          \
        hipLaunchKernelGGL((k_spmv<ROWS, NPT, WIDE, NT>), dim3(GRID), dim3(256), 0, 0, n, rowptr, col, val, x, y, ORD); \
    }, true)
    RUN("A  rows256 scalar xcd-contig g2048", 256, 8, false, false, 2048, nat);
    RUN("A2 rows256 scalar round-robin g2048", 256, 8, false, false, 2048, rr);
    RUN("A3 rows256 scalar xcd-contig g1024", 256, 8, false, false, 1024, nat);
    RUN("A4 rows256 scalar xcd-contig g4096", 256, 8, false, false, 4096, nat);
    RUN("B  rows256 wide   xcd-contig g2048", 256, 8, true, false, 2048, nat);
    RUN("C  rows256 wide+nt xcd-contig g2048", 256, 8, true, true, 2048, nat);
    RUN("C2 rows256 scalar+nt xcd-contig g2048", 256, 8, false, true, 2048, nat);
    RUN("D  rows512 wide   xcd-contig g2048", 512, 16, true, false, 2048, nat);
    RUN("D2 rows512 scalar xcd-contig g2048", 512, 16, false, false, 2048, nat);
    const int NPG = (int)(((n + 255) / 256 + 7) / 8 * 8);
    Order np3{3, 0, 0, 0};
    RUN("NP rows256 scalar one-chunk-per-block", 256, 8, false, false, NPG, np3);
    RUN("NP rows256 wide   one-chunk-per-block", 256, 8, true, false, NPG, np3);
#define RUNW(NAME, WIDE, PERSIST, GRID, ORD)                                                                   \
    bench(NAME, [&] {                                                                                          \
        hipLaunchKernelGGL((k_spmv_wave<WIDE, PERSIST>), dim3(GRID), dim3(256), 0, 0, n, rowptr, col, val, x, y, ORD); \
    }, true)
    RUNW("W  wave-private scalar persist g2048 nat", false, true, 2048, nat);
    RUNW("W  wave-private wide   persist g2048 nat", true, true, 2048, nat);
    RUNW("W  wave-private scalar non-persist nat", false, false, NPG, nat);
    RUNW("W  wave-private wide   non-persist nat", true, false, NPG, nat);
#define RUNT(NAME, PERSIST, GRID, ORD)                                                                         \
    bench(NAME, [&] {                                                                                          \
        hipLaunchKernelGGL((k_spmv_t<256, 8, PERSIST>), dim3(GRID), dim3(256), 0, 0, n, rowptr, col, val, x, y, ORD); \
    }, true)
    RUNT("T  lds-transpose persist g2048 nat", true, 2048, nat);
    RUNT("T  lds-transpose persist g1536 nat", true, 1536, nat);
    RUNT("T  lds-transpose non-persist nat", false, NPG, nat);
    for (int tj : {16, 32, 64, 128}) {
        const int64_t cpp = nx * ny / 256, cpt = (int64_t)tj * nx / 256;
        if ((nx * ny) % 256 || (tj * nx) % 256 || cpp % cpt) continue;
        Order til{1, cpp, cpt, nz};
        char nm[96];
        snprintf(nm, sizeof nm, "T  lds-transpose persist g2048 tiles tj=%d", tj);
        RUNT(nm, true, 2048, til);
        snprintf(nm, sizeof nm, "T  lds-transpose non-persist tiles tj=%d", tj);
        RUNT(nm, false, NPG, til);
    }
    {
        double *part;
        CK(hipMalloc(&part, 8 * (size_t)(NPG + 8)));
        const int64_t cpp = nx * ny / 256, cpt = (int64_t)64 * nx / 256;
        Order til{1, cpp, cpt, nz};
        bench("T2 tile2048 nodot tiles tj=64", [&] {
            hipLaunchKernelGGL((k_spmv_t2<2048, false>), dim3(NPG), dim3(256), 0, 0, n, rowptr, col, val, x, y, til, part); }, true);
        bench("T2 tile1792 nodot tiles tj=64", [&] {
            hipLaunchKernelGGL((k_spmv_t2<1792, false>), dim3(NPG), dim3(256), 0, 0, n, rowptr, col, val, x, y, til, part); }, true);
        bench("T2 tile1792 DOT   tiles tj=64", [&] {
            hipLaunchKernelGGL((k_spmv_t2<1792, true>), dim3(NPG), dim3(256), 0, 0, n, rowptr, col, val, x, y, til, part); }, true);
        bench("T2 tile2048 DOT   tiles tj=64", [&] {
            hipLaunchKernelGGL((k_spmv_t2<2048, true>), dim3(NPG), dim3(256), 0, 0, n, rowptr, col, val, x, y, til, part); }, true);
        bench("T  (product kernel shape) tiles tj=64", [&] {
            hipLaunchKernelGGL((k_spmv_t<256, 8, false>), dim3(NPG), dim3(256), 0, 0, n, rowptr, col, val, x, y, til); }, true);
    }
    for (int tj : {16, 64, 128}) {
        const int64_t cpp = nx * ny / 256, cpt = (int64_t)tj * nx / 256;
        if ((nx * ny) % 256 || (tj * nx) % 256 || cpp % cpt) continue;
        Order til{1, cpp, cpt, nz}, til4{4, cpp, cpt, nz};
        char nm[96];
        snprintf(nm, sizeof nm, "W  wave-private wide persist tiles tj=%d", tj);
        RUNW(nm, true, true, 2048, til);
        snprintf(nm, sizeof nm, "W  wave-private wide non-persist tiles tj=%d", tj);
        RUNW(nm, true, false, NPG, til);
        snprintf(nm, sizeof nm, "W  wave-private scalar non-persist tiles tj=%d", tj);
        RUNW(nm, false, false, NPG, til);
        snprintf(nm, sizeof nm, "NP rows256 wide one-chunk-per-block tiles tj=%d", tj);
        RUN(nm, 256, 8, true, false, NPG, til4);
    }
    for (int tj : {64}) {
        const int64_t cpp = nx * ny / 256, cpt = (int64_t)tj * nx / 256;
        if ((nx * ny) % 256 || (tj * nx) % 256 || cpp % cpt) continue;
        Order til{1, cpp, cpt, nz};
        char nm[96];
        snprintf(nm, sizeof nm, "E  rows256 scalar tiles tj=%d g2048", tj);
        RUN(nm, 256, 8, false, false, 2048, til);
        snprintf(nm, sizeof nm, "E' rows256 wide   tiles tj=%d g2048", tj);
        RUN(nm, 256, 8, true, false, 2048, til);
        snprintf(nm, sizeof nm, "E\" rows256 wide+nt tiles tj=%d g2048", tj);
        RUN(nm, 256, 8, true, true, 2048, til);
    }
    return 0;
}
