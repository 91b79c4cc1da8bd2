// vcycle_lab.hip -- A/B harness for ONE damped-Jacobi step of the multigrid's level operator on a 512^3 level
// (x, b read; x' written: 24 B per cell), z-marching LDS-tiled form of gmg.hip:k_level_march, in variants that differ in
// how much arithmetic a cell costs.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/vcycle_lab.hip -o tools/vcycle_lab
//
//   V0  the product's expressions: face coefficients (w_a w_b) g per cell and plane, diagonal = -(sum of the six)
//   V1  volume-scaled rows: the row of cell (i,j,k) divided by its volume has the face coefficients g_d[s] / w_d[s] --
//       functions of ONE index each -- and the diagonal -(ax_i + ay_j + az_k); D^-1 (b - A x) is unchanged by a row
//       scaling, so this is the same smoother with 1-D coefficient tables (two multiplications for b / volume)
//   V2  V1 + the reciprocal of the diagonal hoisted when the z direction is uniform (interior planes share it)
//   V3  V1 on 128 x 16 tiles (512 threads): the halo rows are 1/8 instead of 1/4 of the tile
//   S   stream: out = a + 0.9 * b, same 24 B per cell, no stencil (what the memory system gives this access pattern)
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

typedef double v4 __attribute__((ext_vector_type(4)));

struct L {
    int nx, ny, nz;
    const double *wx, *wy, *wz, *gx, *gy, *gz;
    // volume-scaled tables: cm[s] = g[s-1] / w[s] (0 at the wall), cp[s] = g[s] / w[s] (0 at the wall), rw[s] = 1 / w[s]
    const double *cmx, *cpx, *rwx, *cmy, *cpy, *rwy, *cmz, *cpz, *rwz;
};

constexpr int TX = 128;

struct Cell0 {
    double wx, wy, gxm, gxp, gym, gyp;
};
__device__ __forceinline__ Cell0 cell0(const L &l, int i, int j)
{
    Cell0 c;
    c.wx = l.wx[i];
    c.wy = l.wy[j];
    c.gxm = (i > 0) ? l.gx[i - 1] : 0.0;
    c.gxp = (i < l.nx - 1) ? l.gx[i] : 0.0;
    c.gym = (j > 0) ? l.gy[j - 1] : 0.0;
    c.gyp = (j < l.ny - 1) ? l.gy[j] : 0.0;
    return c;
}
struct Cell1 {
    double cxm, cxp, cym, cyp, s4, rxy;
};
__device__ __forceinline__ Cell1 cell1(const L &l, int i, int j)
{
    Cell1 c;
    c.cxm = l.cmx[i];
    c.cxp = l.cpx[i];
    c.cym = l.cmy[j];
    c.cyp = l.cpy[j];
    c.s4 = ((c.cxm + c.cxp) + c.cym) + c.cyp;
    c.rxy = l.rwx[i] * l.rwy[j];
    return c;
}

// VAR 0: product expressions; 1: volume-scaled; 2: volume-scaled + hoisted reciprocal (uniform z)
template <int VAR, int TY, int KZ, int PF = 0, int DOTS = 0>
__global__ __launch_bounds__(32 * TY) void k_step(L l, double omega, const double *__restrict__ b, const double *__restrict__ xi,
                                                  double *__restrict__ xo, double *__restrict__ part = nullptr)
{
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    constexpr int SX = TX + 2, SY = TY + 2;
    __shared__ double sp[2][SY][SX];
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int i0 = blockIdx.x * TX, j0 = blockIdx.y * TY, k0 = blockIdx.z * KZ;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    constexpr int nth = 32 * TY;
    const int hx_col = (tid & 1) ? TX : -1, hx_y = (tid >> 1);
    const bool hx_use = tid < 2 * TY;
    const int hxj = j0 + hx_y, hxi = i0 + hx_col;
    const bool hx_ok = hx_use && hxi >= 0 && hxi < l.nx;
    const int64_t off_c = (int64_t)j * l.nx + ic, off_hx = (int64_t)hxj * l.nx + hxi;
    Cell0 q0[4];
    Cell1 q1[4];
    double rinv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (VAR == 0) q0[c] = cell0(l, ic + c, j);
        else q1[c] = cell1(l, ic + c, j);
        if (VAR == 2) rinv[c] = 1.0 / (-((q1[c].s4 + l.cmz[1]) + l.cpz[1]));  // interior planes of a uniform z direction
    }
    const int kend = (k0 + KZ < l.nz) ? k0 + KZ : l.nz;
    v4 zm = {0, 0, 0, 0}, xc, zp = {0, 0, 0, 0}, zq = {0, 0, 0, 0}, bv = {0, 0, 0, 0}, bn = {0, 0, 0, 0};
    // PF (TY == 8): plane two ahead, next plane's halo cells and right-hand side requested an iteration early
    const int hy_row = (tid < 128) ? -1 : TY, hy_x = tid & 127, hyj = j0 + hy_row;
    const bool hy_ok = hyj >= 0 && hyj < l.ny;
    const int64_t off_hy = (int64_t)hyj * l.nx + i0 + hy_x;
    double hyv = 0.0, hxv = 0.0, hyn = 0.0, hxn = 0.0;
    if (k0 > 0) zm = *reinterpret_cast<const v4 *>(xi + (int64_t)(k0 - 1) * plane + off_c);
    xc = *reinterpret_cast<const v4 *>(xi + (int64_t)k0 * plane + off_c);
    if (PF) {
        if (k0 + 1 < l.nz) zp = *reinterpret_cast<const v4 *>(xi + (int64_t)(k0 + 1) * plane + off_c);
        hyv = hy_ok ? xi[(int64_t)k0 * plane + off_hy] : 0.0;
        hxv = hx_ok ? xi[(int64_t)k0 * plane + off_hx] : 0.0;
        bv = *reinterpret_cast<const v4 *>(b + (int64_t)k0 * plane + off_c);
    }
    for (int kk = k0; kk < kend; ++kk) {
        const int slot = kk & 1;
        const double *px = xi + (int64_t)kk * plane;
        if (PF) {
            if (kk + 2 < l.nz && kk + 1 < kend) zq = *reinterpret_cast<const v4 *>(px + 2 * plane + off_c);
            if (kk + 1 < kend) {
                hyn = hy_ok ? px[plane + off_hy] : 0.0;
                hxn = hx_ok ? px[plane + off_hx] : 0.0;
                bn = *reinterpret_cast<const v4 *>(b + (int64_t)(kk + 1) * plane + off_c);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) sp[slot][ty + 1][4 * tx + 1 + c] = xc[c];
            sp[slot][hy_row + 1][hy_x + 1] = hyv;
            if (hx_use) sp[slot][hx_y + 1][hx_col + 1] = hxv;
        } else {
        if (kk + 1 < l.nz) zp = *reinterpret_cast<const v4 *>(px + plane + off_c);
        bv = *reinterpret_cast<const v4 *>(b + (int64_t)kk * plane + off_c);
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[slot][ty + 1][4 * tx + 1 + c] = xc[c];
#pragma unroll
        for (int h = tid; h < 2 * TX; h += nth) {  // the two y-halo rows
            const int row = (h < TX) ? -1 : TY, hx = h & (TX - 1), gj = j0 + row;
            sp[slot][row + 1][hx + 1] = (gj >= 0 && gj < l.ny) ? px[(int64_t)gj * l.nx + i0 + hx] : 0.0;
        }
        if (hx_use) sp[slot][hx_y + 1][hx_col + 1] = hx_ok ? px[off_hx] : 0.0;
        }
        __syncthreads();
        v4 out;
        if (VAR == 0) {
            const double wzk = l.wz[kk];
            const double gzm = (kk > 0) ? l.gz[kk - 1] : 0.0, gzp = (kk < l.nz - 1) ? l.gz[kk] : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = ic + c, lx = 4 * tx + 1 + c;
                const Cell0 &q = q0[c];
                const double ax = q.wy * wzk, ay = q.wx * wzk, az = q.wx * q.wy;
                const double c0 = ax * q.gxm, c1 = ax * q.gxp, c2 = ay * q.gym, c3 = ay * q.gyp, c4 = az * gzm, c5 = az * gzp;
                const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
                const double xcc = xc[c];
                double s = 0.0;
                if (i > 0) s += c0 * (sp[slot][ty + 1][lx - 1] - xcc);
                if (i < l.nx - 1) s += c1 * (sp[slot][ty + 1][lx + 1] - xcc);
                if (j > 0) s += c2 * (sp[slot][ty][lx] - xcc);
                if (j < l.ny - 1) s += c3 * (sp[slot][ty + 2][lx] - xcc);
                if (kk > 0) s += c4 * (zm[c] - xcc);
                if (kk < l.nz - 1) s += c5 * (zp[c] - xcc);
                out[c] = xcc + omega * ((bv[c] - s) / d);
            }
        } else {
            const double czm = l.cmz[kk], czp = l.cpz[kk], rwz = l.rwz[kk];
            const bool zin = kk > 0 && kk < l.nz - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int lx = 4 * tx + 1 + c;
                const Cell1 &q = q1[c];
                const double xcc = xc[c];
                // wall faces carry a zero coefficient and the tile's halo cells outside the domain hold 0: no branches
                double s = 0.0;
                s += q.cxm * (sp[slot][ty + 1][lx - 1] - xcc);
                s += q.cxp * (sp[slot][ty + 1][lx + 1] - xcc);
                s += q.cym * (sp[slot][ty][lx] - xcc);
                s += q.cyp * (sp[slot][ty + 2][lx] - xcc);
                s += czm * (zm[c] - xcc);
                s += czp * (zp[c] - xcc);
                const double bs = (bv[c] * q.rxy) * rwz;
                if (VAR == 2 && zin)
                    out[c] = xcc + omega * ((bs - s) * rinv[c]);
                else {
                    const double d = -((q.s4 + czm) + czp);
                    out[c] = xcc + omega * ((bs - s) / d);
                }
                if (DOTS) {  // the sums k_level_march<8> leaves for CG: z.r, z.z, sum z
                    acc0 += out[c] * bv[c];
                    acc1 += out[c] * out[c];
                    acc2 += out[c];
                }
            }
        }
        *reinterpret_cast<v4 *>(xo + (int64_t)kk * plane + off_c) = out;
        zm = xc;
        xc = zp;
        if (PF) {
            zp = zq;
            hyv = hyn;
            hxv = hxn;
            bv = bn;
        }
    }
    if (DOTS) {
        __shared__ double sh[3][TY / 2];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        double v[3] = {acc0, acc1, acc2};
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v[k2] += __shfl_down(v[k2], o, 64);
            if (lane == 0) sh[k2][w] = v[k2];
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double t = 0.0;
            for (int w2 = 0; w2 < TY / 2; ++w2) t += sh[threadIdx.x][w2];
            const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            part[(int64_t)threadIdx.x * 4096 + blk] = t;
        }
    }
}

// M: y = A x (the stencil twin's product, 16 B per cell: one input stream).  PF: the plane two ahead and the next plane's
// halo cells are requested an iteration early; NT: nontemporal stores
template <int TY, int KZ, int PF, int NT>
__global__ __launch_bounds__(32 * TY) void k_mv(L l, const double *__restrict__ xi, double *__restrict__ xo)
{
    constexpr int SX = TX + 2, SY = TY + 2;
    __shared__ double sp[2][SY][SX];
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int i0 = blockIdx.x * TX, j0 = blockIdx.y * TY, k0 = blockIdx.z * KZ;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    static_assert(TY == 8, "halo mapping below is for 256 threads");
    const int hy_row = (tid < 128) ? -1 : TY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? TX : -1, hx_y = (tid >> 1) & 7;
    const int hyj = j0 + hy_row, hyi = i0 + hy_x, hxj = j0 + hx_y, hxi = i0 + hx_col;
    const bool hy_ok = hyj >= 0 && hyj < l.ny, hx_ok = tid < 16 && hxi >= 0 && hxi < l.nx;
    const int64_t off_c = (int64_t)j * l.nx + ic, off_hy = (int64_t)hyj * l.nx + hyi, off_hx = (int64_t)hxj * l.nx + hxi;
    Cell1 q1[4];
    double vxy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        q1[c] = cell1(l, ic + c, j);
        vxy[c] = l.wx[ic + c] * l.wy[j];
    }
    const int kend = (k0 + KZ < l.nz) ? k0 + KZ : l.nz;
    v4 zm = {0, 0, 0, 0}, xc, zp = {0, 0, 0, 0}, zq = {0, 0, 0, 0};
    double hyv = 0.0, hxv = 0.0, hyn = 0.0, hxn = 0.0;
    if (k0 > 0) zm = *reinterpret_cast<const v4 *>(xi + (int64_t)(k0 - 1) * plane + off_c);
    xc = *reinterpret_cast<const v4 *>(xi + (int64_t)k0 * plane + off_c);
    if (PF) {
        if (k0 + 1 < l.nz) zp = *reinterpret_cast<const v4 *>(xi + (int64_t)(k0 + 1) * plane + off_c);
        hyv = hy_ok ? xi[(int64_t)k0 * plane + off_hy] : 0.0;
        hxv = hx_ok ? xi[(int64_t)k0 * plane + off_hx] : 0.0;
    }
    for (int kk = k0; kk < kend; ++kk) {
        const int slot = kk & 1;
        const double *px = xi + (int64_t)kk * plane;
        if (PF) {
            if (kk + 2 < l.nz && kk + 1 < kend) zq = *reinterpret_cast<const v4 *>(px + 2 * plane + off_c);
            if (kk + 1 < kend) {
                hyn = hy_ok ? px[plane + off_hy] : 0.0;
                hxn = hx_ok ? px[plane + off_hx] : 0.0;
            }
        } else {
            if (kk + 1 < l.nz) zp = *reinterpret_cast<const v4 *>(px + plane + off_c);
            hyv = hy_ok ? px[off_hy] : 0.0;
            hxv = hx_ok ? px[off_hx] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[slot][ty + 1][4 * tx + 1 + c] = xc[c];
        sp[slot][hy_row + 1][hy_x + 1] = hyv;
        if (tid < 16) sp[slot][hx_y + 1][hx_col + 1] = hxv;
        __syncthreads();
        const double czm = l.cmz[kk], czp = l.cpz[kk], wzk = l.wz[kk];
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int lx = 4 * tx + 1 + c;
            const Cell1 &q = q1[c];
            const double xcc = xc[c];
            double s = 0.0;
            s += q.cxm * (sp[slot][ty + 1][lx - 1] - xcc);
            s += q.cxp * (sp[slot][ty + 1][lx + 1] - xcc);
            s += q.cym * (sp[slot][ty][lx] - xcc);
            s += q.cyp * (sp[slot][ty + 2][lx] - xcc);
            s += czm * (zm[c] - xcc);
            s += czp * (zp[c] - xcc);
            out[c] = (s * vxy[c]) * wzk;
        }
        if (NT) __builtin_nontemporal_store(out, reinterpret_cast<v4 *>(xo + (int64_t)kk * plane + off_c));
        else *reinterpret_cast<v4 *>(xo + (int64_t)kk * plane + off_c) = out;
        zm = xc;
        xc = zp;
        if (PF) {
            zp = zq;
            hyv = hyn;
            hxv = hxn;
        }
    }
}
// MD: y = A x with the planes brought in by LDS-DMA (global_load_lds_dwordx4: 16 B per lane straight into LDS, no staging
// registers) D planes ahead into a ring of D + 1 slots.  Every wave issues exactly three DMA instructions per plane (two
// body rows; the third is a y-halo row, the x-halo cells as dwords, or a dummy), so s_waitcnt vmcnt(3 D) means "the oldest
// plane has landed" (loads return in order; outstanding stores only make the wait longer).
constexpr int DSX = TX + 4;  // row stride 132 doubles: the body starts at column 2 (16-byte aligned), halo cells at 1 and 130
template <int KZ, int D>
__global__ __launch_bounds__(256) void k_mv_dma(L l, const double *__restrict__ xi, double *__restrict__ xo)
{
    // ring of D + 2 slots: planes kk - 1 (other waves may still read it when this wave issues the next DMA) .. kk + D
    constexpr int TY = 8, SY = TY + 2, R = D + 2;
    __shared__ __attribute__((aligned(16))) double sp[R][SY][DSX];
    __shared__ __attribute__((aligned(16))) double hx[R][32];  // x-halo cells of a plane: [0..7] left column, [8..15] right; 64 dwords per DMA
    __shared__ __attribute__((aligned(16))) double dummy[32];
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31, wave = tid >> 6, lane = tid & 63;
    const int i0 = blockIdx.x * TX, j0 = blockIdx.y * TY, k0 = blockIdx.z * KZ;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    const int64_t off_c = (int64_t)j * l.nx + ic;
    const bool top_ok = j0 > 0, bot_ok = j0 + TY < l.ny, left_ok = i0 > 0, right_ok = i0 + TX < l.nx;
    Cell1 q1[4];
    double vxy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        q1[c] = cell1(l, ic + c, j);
        vxy[c] = l.wx[ic + c] * l.wy[j];
    }
    // rows beyond the domain hold zero in every slot, once and for all (the DMA never writes them)
    for (int e = tid; e < R * SY * DSX; e += 256) (&sp[0][0][0])[e] = 0.0;
    __syncthreads();
    typedef __attribute__((address_space(3))) void *lds_ptr;
    // exactly three DMA instructions per wave and plane
    auto dma_plane = [&](int kk) {
        const int slot = kk % R;
        const double *px = xi + (int64_t)kk * plane;
        __builtin_amdgcn_global_load_lds((const void *)(px + (int64_t)(j0 + wave) * l.nx + i0 + 2 * lane), (lds_ptr)&sp[slot][wave + 1][2 + 2 * lane], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(px + (int64_t)(j0 + wave + 4) * l.nx + i0 + 2 * lane), (lds_ptr)&sp[slot][wave + 5][2 + 2 * lane], 16, 0, 0);
        if (wave == 0 && top_ok)
            __builtin_amdgcn_global_load_lds((const void *)(px + (int64_t)(j0 - 1) * l.nx + i0 + 2 * lane), (lds_ptr)&sp[slot][0][2 + 2 * lane], 16, 0, 0);
        else if (wave == 1 && bot_ok)
            __builtin_amdgcn_global_load_lds((const void *)(px + (int64_t)(j0 + TY) * l.nx + i0 + 2 * lane), (lds_ptr)&sp[slot][SY - 1][2 + 2 * lane], 16, 0, 0);
        else if (wave == 2) {
            // dword `lane` of hx[slot]: double lane / 2 = row (lane / 2) % 8 of the left (lane < 16) / right column; lanes beyond
            // the 32 dwords and columns beyond the domain fetch something harmless (their values are never used)
            const int dbl = (lane >> 1) & 15, row = dbl & 7, right = dbl >> 3;
            const bool ok = lane < 32 && (right ? right_ok : left_ok);
            const int gi = ok ? (right ? i0 + TX : i0 - 1) : i0;
            const int *src = reinterpret_cast<const int *>(px + (int64_t)(j0 + row) * l.nx + gi) + (lane & 1);
            __builtin_amdgcn_global_load_lds((const void *)src, (lds_ptr)(reinterpret_cast<int *>(&hx[slot][0]) + lane), 4, 0, 0);
        } else
            __builtin_amdgcn_global_load_lds((const void *)(reinterpret_cast<const int *>(px + off_c)), (lds_ptr)(reinterpret_cast<int *>(&dummy[0]) + lane), 4, 0, 0);
    };
    const int kend = (k0 + KZ < l.nz) ? k0 + KZ : l.nz;
    const int kfirst = k0 > 0 ? k0 - 1 : k0, klast = kend < l.nz ? kend : kend - 1;  // planes this workgroup reads: [kfirst, klast]
    int issued = kfirst;  // planes [kfirst, issued) are on their way or here
    for (; issued <= klast && issued < k0 + D; ++issued) dma_plane(issued);
    auto land = [&](int kk) {  // plane kk has landed for every wave: at most the planes issued after it are outstanding
        const int behind = issued - 1 - kk;
        if (behind >= 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto own = [&](int kk) -> v4 {
        const double *row = &sp[kk % R][ty + 1][2 + 4 * tx];
        const double2 a = *reinterpret_cast<const double2 *>(row), b = *reinterpret_cast<const double2 *>(row + 2);
        v4 r = {a.x, a.y, b.x, b.y};
        return r;
    };
    v4 zm = {0, 0, 0, 0}, xc, zp = {0, 0, 0, 0};
    if (k0 > 0) {
        land(k0 - 1);
        zm = own(k0 - 1);
    }
    land(k0);
    xc = own(k0);
    for (int kk = k0; kk < kend; ++kk) {
        // slot (kk + D) % R held plane kk - 2: every wave has passed the barrier that followed its use
        if (kk + D <= klast) {
            dma_plane(kk + D);
            issued = kk + D + 1;
        }
        if (kk + 1 < l.nz) {
            land(kk + 1);
            zp = own(kk + 1);
        } else
            __syncthreads();
        const int slot = kk % R;
        const double czm = l.cmz[kk], czp = l.cpz[kk], wzk = l.wz[kk];
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int lx = 4 * tx + 2 + c;
            const Cell1 &q = q1[c];
            const double xcc = xc[c];
            const double left = (c == 0 && tx == 0) ? (left_ok ? hx[slot][ty] : 0.0) : sp[slot][ty + 1][lx - 1];
            const double right = (c == 3 && tx == 31) ? (right_ok ? hx[slot][TY + ty] : 0.0) : sp[slot][ty + 1][lx + 1];
            double s = 0.0;
            s += q.cxm * (left - xcc);
            s += q.cxp * (right - xcc);
            s += q.cym * (sp[slot][ty][lx] - xcc);
            s += q.cyp * (sp[slot][ty + 2][lx] - xcc);
            s += czm * (zm[c] - xcc);
            s += czp * (zp[c] - xcc);
            out[c] = (s * vxy[c]) * wzk;
        }
        *reinterpret_cast<v4 *>(xo + (int64_t)kk * plane + off_c) = out;
        zm = xc;
        xc = zp;
    }
}

__global__ __launch_bounds__(256) void k_copy1(const v4 *__restrict__ a, v4 *__restrict__ o)
{
    const int64_t base = (int64_t)blockIdx.x * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[base + u * 256 + threadIdx.x] = a[base + u * 256 + threadIdx.x];
}

__global__ __launch_bounds__(256) void k_stream(int64_t n4, const v4 *__restrict__ a, const v4 *__restrict__ b, v4 *__restrict__ o)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const v4 x = a[i], y = b[i];
        v4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = x[c] + 0.9 * y[c];
        o[i] = r;
    }
}
// one workgroup per chunk (no grid stride): the dispatcher sweeps a moving address window
__global__ __launch_bounds__(256) void k_stream1(const v4 *__restrict__ a, const v4 *__restrict__ b, v4 *__restrict__ o)
{
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + 256 * u;
        const v4 x = a[i], y = b[i];
        v4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = x[c] + 0.9 * y[c];
        o[i] = r;
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int uniform = argc > 3 ? atoi(argv[3]) : 1;
    const int64_t N = (int64_t)n * n * n;
    std::vector<double> hw(n), hg(n), cm(n), cp(n), rw(n);
    for (int i = 0; i < n; ++i) hw[i] = uniform ? 1.0 / n : (1.0 + 0.3 * ((i * 37) % 11) / 11.0) / n;
    for (int i = 0; i + 1 < n; ++i) hg[i] = 5e-4 * (1.0 / (0.5 * (hw[i] + hw[i + 1])));
    hg[n - 1] = 0.0;
    for (int i = 0; i < n; ++i) {
        cm[i] = (i > 0) ? hg[i - 1] / hw[i] : 0.0;
        cp[i] = (i < n - 1) ? hg[i] / hw[i] : 0.0;
        rw[i] = 1.0 / hw[i];
    }
    auto up = [&](const std::vector<double> &h) {
        double *d;
        CK(hipMalloc(&d, 8 * (h.size() + 8)));
        CK(hipMemset(d, 0, 8 * (h.size() + 8)));
        CK(hipMemcpy(d, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
        return d;
    };
    double *w = up(hw), *g = up(hg), *dcm = up(cm), *dcp = up(cp), *drw = up(rw);
    double *b, *x0, *y0, *y1;
    CK(hipMalloc(&b, 8 * N));
    CK(hipMalloc(&x0, 8 * N));
    CK(hipMalloc(&y0, 8 * N));
    CK(hipMalloc(&y1, 8 * N));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int64_t o = 0; o < N; o += (int64_t)h.size()) CK(hipMemcpy(b + o, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 40503u + 17) % 977) / 977.0 - 0.5;
    for (int64_t o = 0; o < N; o += (int64_t)h.size()) CK(hipMemcpy(x0 + o, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    L l{n, n, n, w, w, w, g, g, g, dcm, dcp, drw, dcm, dcp, drw, dcm, dcp, drw};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.3f ms  %6.2f TB/s (24 B/cell)\n", name, ms / reps, 24.0 * N / (ms / reps) / 1e9);
        fflush(stdout);
    };
    printf("n = %d, %s mesh\n", n, uniform ? "uniform" : "stretched");
    timeit("S  stream out = a + 0.9 b, grid-stride 4096 wg", [&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, N / 4, (const v4 *)x0, (const v4 *)b, (v4 *)y0); });
    timeit("S  stream out = a + 0.9 b, one chunk per workgroup", [&] { hipLaunchKernelGGL(k_stream1, dim3((unsigned)(N / 4 / 1024)), dim3(256), 0, 0, (const v4 *)x0, (const v4 *)b, (v4 *)y0); });
    timeit("V0 product expressions, 128x8 tile, KZ 64", [&] { hipLaunchKernelGGL((k_step<0, 8, 64>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y0); });
    timeit("V1 volume-scaled rows, 128x8 tile, KZ 64", [&] { hipLaunchKernelGGL((k_step<1, 8, 64>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    {
        std::vector<double> a0(1 << 22), a1(1 << 22);
        double worst = 0.0;
        for (int64_t off : {int64_t(0), N / 2 - (int64_t)a0.size() / 2, N - (int64_t)a0.size()}) {
            CK(hipMemcpy(a0.data(), y0 + off, 8 * a0.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(a1.data(), y1 + off, 8 * a0.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < a0.size(); ++i) {
                const double e = a0[i] - a1[i], s = a0[i] < 0 ? -a0[i] : a0[i];
                const double r = (e < 0 ? -e : e) / (s > 1e-300 ? s : 1.0);
                if (r > worst) worst = r;
            }
        }
        printf("   V1 vs V0: largest relative difference %.3e\n", worst);
    }
    if (uniform) timeit("V2 volume-scaled + hoisted reciprocal, 128x8 tile, KZ 64", [&] { hipLaunchKernelGGL((k_step<2, 8, 64>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V1 volume-scaled rows, 128x8 tile, KZ 32", [&] { hipLaunchKernelGGL((k_step<1, 8, 32>), dim3(n / TX, n / 8, n / 32), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V3 volume-scaled rows, 128x16 tile (512 threads), KZ 64", [&] { hipLaunchKernelGGL((k_step<1, 16, 64>), dim3(n / TX, n / 16, n / 64), dim3(512), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V3 volume-scaled rows, 128x16 tile (512 threads), KZ 32", [&] { hipLaunchKernelGGL((k_step<1, 16, 32>), dim3(n / TX, n / 16, n / 32), dim3(512), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V0 product expressions, 128x16 tile (512 threads), KZ 64", [&] { hipLaunchKernelGGL((k_step<0, 16, 64>), dim3(n / TX, n / 16, n / 64), dim3(512), 0, 0, l, 0.9, b, x0, y0); });
    timeit("V1 volume-scaled rows, 128x8 tile, KZ 64, prefetch", [&] { hipLaunchKernelGGL((k_step<1, 8, 64, 1>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V1 volume-scaled rows, 128x8 tile, KZ 32, prefetch", [&] { hipLaunchKernelGGL((k_step<1, 8, 32, 1>), dim3(n / TX, n / 8, n / 32), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V1 volume-scaled rows, 128x8 tile, KZ 16, prefetch", [&] { hipLaunchKernelGGL((k_step<1, 8, 16, 1>), dim3(n / TX, n / 8, n / 16), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    timeit("V1 volume-scaled rows, 128x8 tile, KZ 16", [&] { hipLaunchKernelGGL((k_step<1, 8, 16, 0>), dim3(n / TX, n / 8, n / 16), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    {
        double *part;
        CK(hipMalloc(&part, 8 * 3 * 4096));
        timeit("V1 + the three CG sums, 128x8 tile, KZ 64", [&] { hipLaunchKernelGGL((k_step<1, 8, 64, 0, 1>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y1, part); });
        timeit("V1 + the three CG sums, 128x8 tile, KZ 64, prefetch", [&] { hipLaunchKernelGGL((k_step<1, 8, 64, 1, 1>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y1, part); });
        timeit("V1 volume-scaled rows, 128x8 tile, KZ 64 (again)", [&] { hipLaunchKernelGGL((k_step<1, 8, 64>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, y1); });
    }
    printf("one input stream (16 B/cell; the TB/s column counts 24: multiply by 2/3)\n");
    timeit("C  copy, one chunk per workgroup", [&] { hipLaunchKernelGGL(k_copy1, dim3((unsigned)(N / 4 / 1024)), dim3(256), 0, 0, (const v4 *)x0, (v4 *)y0); });
    timeit("M  y = A x, 128x8 tile, KZ 64", [&] { hipLaunchKernelGGL((k_mv<8, 64, 0, 0>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y1); });
    timeit("M  y = A x, 128x8 tile, KZ 32", [&] { hipLaunchKernelGGL((k_mv<8, 32, 0, 0>), dim3(n / TX, n / 8, n / 32), dim3(256), 0, 0, l, x0, y1); });
    timeit("M  y = A x, 128x8 tile, KZ 16", [&] { hipLaunchKernelGGL((k_mv<8, 16, 0, 0>), dim3(n / TX, n / 8, n / 16), dim3(256), 0, 0, l, x0, y1); });
    timeit("M  y = A x, 128x8 tile, KZ 64, prefetch", [&] { hipLaunchKernelGGL((k_mv<8, 64, 1, 0>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y1); });
    timeit("M  y = A x, 128x8 tile, KZ 32, prefetch", [&] { hipLaunchKernelGGL((k_mv<8, 32, 1, 0>), dim3(n / TX, n / 8, n / 32), dim3(256), 0, 0, l, x0, y1); });
    timeit("M  y = A x, 128x8 tile, KZ 64, nontemporal stores", [&] { hipLaunchKernelGGL((k_mv<8, 64, 0, 1>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y1); });
    timeit("M  y = A x, 128x8 tile, KZ 64, prefetch + nontemporal", [&] { hipLaunchKernelGGL((k_mv<8, 64, 1, 1>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y1); });
    timeit("MD y = A x, LDS-DMA ring, 2 planes ahead, KZ 64", [&] { hipLaunchKernelGGL((k_mv_dma<64, 2>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y0); });
    {
        std::vector<double> a0(1 << 22), a1(1 << 22);
        double worst = 0.0;
        for (int64_t off : {int64_t(0), N / 2 - (int64_t)a0.size() / 2, N - (int64_t)a0.size()}) {
            CK(hipMemcpy(a0.data(), y0 + off, 8 * a0.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(a1.data(), y1 + off, 8 * a0.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < a0.size(); ++i) {
                const double e = a0[i] - a1[i];
                if ((e < 0 ? -e : e) > worst) worst = e < 0 ? -e : e;
            }
        }
        printf("   MD vs M: largest difference %.3e\n", worst);
    }
    timeit("MD y = A x, LDS-DMA ring, 3 planes ahead, KZ 64", [&] { hipLaunchKernelGGL((k_mv_dma<64, 3>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y0); });
    timeit("MD y = A x, LDS-DMA ring, 3 planes ahead, KZ 32", [&] { hipLaunchKernelGGL((k_mv_dma<32, 3>), dim3(n / TX, n / 8, n / 32), dim3(256), 0, 0, l, x0, y0); });
    timeit("MD y = A x, LDS-DMA ring, 1 plane ahead, KZ 64", [&] { hipLaunchKernelGGL((k_mv_dma<64, 1>), dim3(n / TX, n / 8, n / 64), dim3(256), 0, 0, l, x0, y0); });
    timeit("V1 volume-scaled rows, 128x4 tile (128 threads), KZ 64", [&] { hipLaunchKernelGGL((k_step<1, 4, 64>), dim3(n / TX, n / 4, n / 64), dim3(128), 0, 0, l, 0.9, b, x0, y1); });
    return 0;
}
