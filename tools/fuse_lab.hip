// fuse_lab.hip -- A/B harness: the two pre-smoothing steps of the V-cycle from a zero guess,
//   x1 = omega b / d ;  x2 = x1 + omega (b - A x1) / d,
// as two streaming kernels (the product's MODE 1 + MODE 2: 5 vector passes over HBM) against ONE kernel that marches
// through z with the x1 planes in an LDS ring (2 passes).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/fuse_lab.hip -o tools/fuse_lab
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

struct L {
    int nx, ny, nz;
    const double *wx, *wy, *wz, *gx, *gy, *gz;
};

__device__ __forceinline__ double diag_of(const L &l, int i, int j, int k, double c[6])
{
    const double wxi = l.wx[i], wyj = l.wy[j], wzk = l.wz[k];
    const double ax = wyj * wzk, ay = wxi * wzk, az = wxi * wyj;
    c[0] = (i > 0) ? ax * l.gx[i - 1] : 0.0;
    c[1] = (i < l.nx - 1) ? ax * l.gx[i] : 0.0;
    c[2] = (j > 0) ? ay * l.gy[j - 1] : 0.0;
    c[3] = (j < l.ny - 1) ? ay * l.gy[j] : 0.0;
    c[4] = (k > 0) ? az * l.gz[k - 1] : 0.0;
    c[5] = (k < l.nz - 1) ? az * l.gz[k] : 0.0;
    return -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
}

// reference pair, one cell per lane
__global__ __launch_bounds__(256) void k_m1(L l, double omega, const double *__restrict__ b, double *__restrict__ xo)
{
    const unsigned plane = (unsigned)l.nx * l.ny;
    const int k = blockIdx.y;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < plane; q += gridDim.x * 256u) {
        const int j = q / (unsigned)l.nx, i = q - j * l.nx;
        double c[6];
        const double d = diag_of(l, i, j, k, c);
        const int64_t p = (int64_t)k * plane + q;
        xo[p] = omega * (b[p] / d);
    }
}
template <int C>
__global__ __launch_bounds__(256) void k_m2(L l, double omega, const double *__restrict__ b, const double *__restrict__ xi,
                                            double *__restrict__ xo)
{
    typedef double vt __attribute__((ext_vector_type(C)));
    const unsigned nxc = (unsigned)l.nx / C, planec = nxc * l.ny;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int k = blockIdx.y;
    const double wzk = l.wz[k];
    const double gzm = (k > 0) ? l.gz[k - 1] : 0.0, gzp = (k < l.nz - 1) ? l.gz[k] : 0.0;
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < planec; q += gridDim.x * 256u) {
        const int j = q / nxc, i0 = (q - j * nxc) * C;
        const int64_t p = (int64_t)k * plane + (int64_t)j * l.nx + i0;
        const double wyj = l.wy[j];
        const double gym = (j > 0) ? l.gy[j - 1] : 0.0, gyp = (j < l.ny - 1) ? l.gy[j] : 0.0;
        const double ax = wyj * wzk;
        vt xc = *reinterpret_cast<const vt *>(xi + p), ym = xc, yp = xc, zm = xc, zp = xc, out;
        const vt bv = *reinterpret_cast<const vt *>(b + p);
        double xl = 0.0, xr = 0.0;
        if (i0 > 0) xl = xi[p - 1];
        if (i0 + C < l.nx) xr = xi[p + C];
        if (j > 0) ym = *reinterpret_cast<const vt *>(xi + p - l.nx);
        if (j < l.ny - 1) yp = *reinterpret_cast<const vt *>(xi + p + l.nx);
        if (k > 0) zm = *reinterpret_cast<const vt *>(xi + p - plane);
        if (k < l.nz - 1) zp = *reinterpret_cast<const vt *>(xi + p + plane);
        double gxm = (i0 > 0) ? l.gx[i0 - 1] : 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int i = i0 + c;
            const double wxi = l.wx[i];
            const double gxp = (i < l.nx - 1) ? l.gx[i] : 0.0;
            const double ay = wxi * wzk, az = wxi * wyj;
            const double c0 = ax * gxm, c1 = ax * gxp, c2 = ay * gym, c3 = ay * gyp, c4 = az * gzm, c5 = az * gzp;
            gxm = gxp;
            const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
            const double left = (c == 0) ? xl : xc[c > 0 ? c - 1 : 0], right = (c == C - 1) ? xr : xc[c < C - 1 ? c + 1 : 0];
            const double xcc = xc[c];
            double s = 0.0;
            if (i > 0) s += c0 * (left - xcc);
            if (i < l.nx - 1) s += c1 * (right - xcc);
            if (j > 0) s += c2 * (ym[c] - xcc);
            if (j < l.ny - 1) s += c3 * (yp[c] - xcc);
            if (k > 0) s += c4 * (zm[c] - xcc);
            if (k < l.nz - 1) s += c5 * (zp[c] - xcc);
            out[c] = xcc + omega * ((bv[c] - s) / d);
        }
        *reinterpret_cast<vt *>(xo + p) = out;
    }
}

// ---- fused: tile TX x TY cells of a plane per workgroup, marching over KZ planes; x1 planes in a ring of 4 LDS slots.
// Every k-independent mesh coefficient of the thread's cells (4 tile cells, one y-halo cell, one x-halo cell for 16 of the
// threads) is loaded once before the march: the first version issued ~55 vector-memory instructions per thread and plane
// for the 1-D arrays alone.
constexpr int TX = 128, TY = 8, SX = TX + 2, SY = TY + 2;
struct Cell1 {  // k-independent part of one cell's coefficients
    double wx, wy, gxm, gxp, gym, gyp;
};
__device__ __forceinline__ Cell1 cell1(const L &l, int i, int j)
{
    Cell1 c;
    c.wx = l.wx[i];
    c.wy = l.wy[j];
    c.gxm = (i > 0) ? l.gx[i - 1] : 0.0;
    c.gxp = (i < l.nx - 1) ? l.gx[i] : 0.0;
    c.gym = (j > 0) ? l.gy[j - 1] : 0.0;
    c.gyp = (j < l.ny - 1) ? l.gy[j] : 0.0;
    return c;
}
__device__ __forceinline__ double diag1(const Cell1 &q, double wzk, double gzm, double gzp)
{
    const double ax = q.wy * wzk, ay = q.wx * wzk, az = q.wx * q.wy;
    const double c0 = ax * q.gxm, c1 = ax * q.gxp, c2 = ay * q.gym, c3 = ay * q.gyp, c4 = az * gzm, c5 = az * gzp;
    return -(((((c0 + c1) + c2) + c3) + c4) + c5);
}
template <int KZ, int STRIDED, int PF>
__global__ __launch_bounds__(256) void k_fused(L l, double omega, const double *__restrict__ b, double *__restrict__ xo)
{
    __shared__ double x1[4][SY][SX];
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int i0 = blockIdx.x * TX, j0 = blockIdx.y * TY, k0 = blockIdx.z * KZ;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;  // this thread's 4 cells: (ic..ic+3, j); STRIDED: (i0 + tx + 32 c, j)
    const int xs = STRIDED ? 32 : 1, xb = STRIDED ? tx : 4 * tx;  // cell c sits at tile column xb + xs * c
    const int hy_row = (tid < 128) ? -1 : TY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? TX : -1, hx_y = (tid >> 1) & 7;
    const int hyj = j0 + hy_row, hyi = i0 + hy_x, hxj = j0 + hx_y, hxi = i0 + hx_col;
    const bool hy_ok = hyj >= 0 && hyj < l.ny, hx_ok = tid < 16 && hxi >= 0 && hxi < l.nx;
    const int64_t off_c = (int64_t)j * l.nx + ic, off_hy = (int64_t)hyj * l.nx + hyi, off_hx = (int64_t)hxj * l.nx + hxi;
    Cell1 q4[4], qhy = {}, qhx = {};
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = cell1(l, i0 + xb + xs * c, j);
    if (hy_ok) qhy = cell1(l, hyi, hyj);
    if (hx_ok) qhx = cell1(l, hxi, hxj);
    v4 bprev = {0, 0, 0, 0}, bcur = {0, 0, 0, 0}, bn = {0, 0, 0, 0};
    double hyn = 0.0, hxn = 0.0;
    auto load_plane = [&](int kq, v4 &bv, double &hy, double &hx) {
        if (kq < 0 || kq >= l.nz) return;
        const double *pb = b + (int64_t)kq * plane;
        if (STRIDED) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = pb[(int64_t)j * l.nx + i0 + xb + xs * c];
        } else
            bv = *reinterpret_cast<const v4 *>(pb + off_c);
        hy = hy_ok ? pb[off_hy] : 0.0;
        hx = hx_ok ? pb[off_hx] : 0.0;
    };
    if (PF) load_plane(k0 - 1, bn, hyn, hxn);
    for (int kk = k0 - 1; kk <= k0 + KZ; ++kk) {
        const int slot = (kk + 4) & 3;
        bprev = bcur;
        v4 bv = bn;
        double hyv = hyn, hxv = hxn;
        if (PF) {
            if (kk < k0 + KZ) load_plane(kk + 1, bn, hyn, hxn);
        } else
            load_plane(kk, bv, hyv, hxv);
        if (kk >= 0 && kk < l.nz) {
            const double wzk = l.wz[kk];
            const double gzm = (kk > 0) ? l.gz[kk - 1] : 0.0, gzp = (kk < l.nz - 1) ? l.gz[kk] : 0.0;
            bcur = bv;
#pragma unroll
            for (int c = 0; c < 4; ++c) x1[slot][ty + 1][xb + xs * c + 1] = omega * (bv[c] / diag1(q4[c], wzk, gzm, gzp));
            x1[slot][hy_row + 1][hy_x + 1] = hy_ok ? omega * (hyv / diag1(qhy, wzk, gzm, gzp)) : 0.0;
            if (tid < 16) x1[slot][hx_y + 1][hx_col + 1] = hx_ok ? omega * (hxv / diag1(qhx, wzk, gzm, gzp)) : 0.0;
        }
        __syncthreads();
        const int kc = kk - 1;  // the plane whose x2 is complete now
        if (kc < k0 || kc >= l.nz) continue;
        const int sc = (kc + 4) & 3, sm = (kc + 3) & 3, sp = (kc + 5) & 3;
        const double wzk = l.wz[kc];
        const double gzm = (kc > 0) ? l.gz[kc - 1] : 0.0, gzp = (kc < l.nz - 1) ? l.gz[kc] : 0.0;
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + xb + xs * c, lx = xb + xs * c + 1;
            const Cell1 &q = q4[c];
            const double ax = q.wy * wzk, ay = q.wx * wzk, az = q.wx * q.wy;
            const double c0 = ax * q.gxm, c1 = ax * q.gxp, c2 = ay * q.gym, c3 = ay * q.gyp, c4 = az * gzm, c5 = az * gzp;
            const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
            const double xcc = x1[sc][ty + 1][lx];
            double s = 0.0;
            if (i > 0) s += c0 * (x1[sc][ty + 1][lx - 1] - xcc);
            if (i < l.nx - 1) s += c1 * (x1[sc][ty + 1][lx + 1] - xcc);
            if (j > 0) s += c2 * (x1[sc][ty][lx] - xcc);
            if (j < l.ny - 1) s += c3 * (x1[sc][ty + 2][lx] - xcc);
            if (kc > 0) s += c4 * (x1[sm][ty + 1][lx] - xcc);
            if (kc < l.nz - 1) s += c5 * (x1[sp][ty + 1][lx] - xcc);
            out[c] = xcc + omega * ((bprev[c] - s) / d);
        }
        if (STRIDED) {
#pragma unroll
            for (int c = 0; c < 4; ++c) xo[(int64_t)kc * plane + (int64_t)j * l.nx + i0 + xb + xs * c] = out[c];
        } else
            *reinterpret_cast<v4 *>(xo + (int64_t)kc * plane + (int64_t)j * l.nx + ic) = out;
    }
}

// ---- two general Jacobi steps fused: x1 = S(x0) on tile + 1, x2 = S(x1) on the tile; x0 planes (tile + 2) in a ring of 3
// LDS slots, x1 planes (tile + 1) in a ring of 4
template <int KZ>
__global__ __launch_bounds__(256) void k_fused_pair(L l, double omega, const double *__restrict__ b, const double *__restrict__ x0,
                                                    double *__restrict__ xo)
{
    constexpr int HX = TX + 4, HY = TY + 4;
    __shared__ double s0[3][HY][HX];
    __shared__ double s1[4][SY][SX];
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int i0 = blockIdx.x * TX, j0 = blockIdx.y * TY, k0 = blockIdx.z * KZ;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    const int hy_row = (tid < 128) ? -1 : TY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? TX : -1, hx_y = (tid >> 1) & 7;
    const int hyj = j0 + hy_row, hyi = i0 + hy_x, hxj = j0 + hx_y, hxi = i0 + hx_col;
    const bool hy_ok = hyj >= 0 && hyj < l.ny, hx_ok = tid < 16 && hxi >= 0 && hxi < l.nx;
    const int64_t off_c = (int64_t)j * l.nx + ic, off_hy = (int64_t)hyj * l.nx + hyi, off_hx = (int64_t)hxj * l.nx + hxi;
    Cell1 q4[4], qhy = {}, qhx = {};
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = cell1(l, ic + c, j);
    if (hy_ok) qhy = cell1(l, hyi, hyj);
    if (hx_ok) qhx = cell1(l, hxi, hxj);
    v4 b1 = {0, 0, 0, 0}, b2 = {0, 0, 0, 0};  // b of planes kk-1 and kk-2 at the thread's cells
    // one Jacobi step at a cell of plane kq from the x0 ring; (lx, ly) = position in the tile + 2 frame
    auto step_a = [&](const Cell1 &q, int i, int jj, int kq, int lx, int ly, double bq) -> double {
        const double wzk = l.wz[kq];
        const double gzm = (kq > 0) ? l.gz[kq - 1] : 0.0, gzp = (kq < l.nz - 1) ? l.gz[kq] : 0.0;
        const double ax = q.wy * wzk, ay = q.wx * wzk, az = q.wx * q.wy;
        const double c0 = ax * q.gxm, c1 = ax * q.gxp, c2 = ay * q.gym, c3 = ay * q.gyp, c4 = az * gzm, c5 = az * gzp;
        const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
        const int sc = (kq + 3) % 3, sm = (kq + 2) % 3, sp = (kq + 4) % 3;
        const double xcc = s0[sc][ly][lx];
        double s = 0.0;
        if (i > 0) s += c0 * (s0[sc][ly][lx - 1] - xcc);
        if (i < l.nx - 1) s += c1 * (s0[sc][ly][lx + 1] - xcc);
        if (jj > 0) s += c2 * (s0[sc][ly - 1][lx] - xcc);
        if (jj < l.ny - 1) s += c3 * (s0[sc][ly + 1][lx] - xcc);
        if (kq > 0) s += c4 * (s0[sm][ly][lx] - xcc);
        if (kq < l.nz - 1) s += c5 * (s0[sp][ly][lx] - xcc);
        return xcc + omega * ((bq - s) / d);
    };
    for (int kk = k0 - 2; kk <= k0 + KZ + 1; ++kk) {
        // (a) x0 plane kk -> ring (tile + 2; zeros outside the domain)
        {
            const int slot = (kk + 3) % 3;
            const bool in = kk >= 0 && kk < l.nz;
            const double *px = x0 + (int64_t)kk * plane;
            v4 v = {0, 0, 0, 0};
            if (in) v = *reinterpret_cast<const v4 *>(px + off_c);
#pragma unroll
            for (int c = 0; c < 4; ++c) s0[slot][ty + 2][4 * tx + 2 + c] = v[c];
            for (int h = tid; h < 4 * HX + 4 * TY; h += 256) {
                int ly, lx;
                if (h < 4 * HX) {
                    const int r = h / HX;
                    ly = (r < 2) ? r : TY + r;  // rows 0, 1, TY+2, TY+3
                    lx = h - r * HX;
                } else {
                    const int q = h - 4 * HX, r = q >> 2, cidx = q & 3;
                    ly = r + 2;
                    lx = (cidx < 2) ? cidx : TX + cidx;  // columns 0, 1, TX+2, TX+3
                }
                const int gi = i0 + lx - 2, gj = j0 + ly - 2;
                double val = 0.0;
                if (in && gi >= 0 && gi < l.nx && gj >= 0 && gj < l.ny) val = px[(int64_t)gj * l.nx + gi];
                s0[slot][ly][lx] = val;
            }
        }
        __syncthreads();
        // (b) x1 of plane kk-1 on tile + 1
        const int ka = kk - 1;
        if (ka >= 0 && ka < l.nz && ka >= k0 - 1 && ka <= k0 + KZ) {
            const double *pb = b + (int64_t)ka * plane;
            const v4 bv = *reinterpret_cast<const v4 *>(pb + off_c);
            b2 = b1;
            b1 = bv;
            const int slot = (ka + 4) & 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) s1[slot][ty + 1][4 * tx + 1 + c] = step_a(q4[c], ic + c, j, ka, 4 * tx + 2 + c, ty + 2, bv[c]);
            s1[slot][hy_row + 1][hy_x + 1] = hy_ok ? step_a(qhy, hyi, hyj, ka, hy_x + 2, hy_row + 2, pb[off_hy]) : 0.0;
            if (tid < 16) s1[slot][hx_y + 1][hx_col + 1] = hx_ok ? step_a(qhx, hxi, hxj, ka, hx_col + 2, hx_y + 2, pb[off_hx]) : 0.0;
        } else {
            b2 = b1;
        }
        __syncthreads();
        // (c) x2 of plane kk-2 on the tile
        const int kc = kk - 2;
        if (kc < k0 || kc >= l.nz || kc >= k0 + KZ) continue;
        const int sc = (kc + 4) & 3, sm = (kc + 3) & 3, sp = (kc + 5) & 3;
        const double wzk = l.wz[kc];
        const double gzm = (kc > 0) ? l.gz[kc - 1] : 0.0, gzp = (kc < l.nz - 1) ? l.gz[kc] : 0.0;
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ic + c, lx = 4 * tx + 1 + c;
            const Cell1 &q = q4[c];
            const double ax = q.wy * wzk, ay = q.wx * wzk, az = q.wx * q.wy;
            const double c0 = ax * q.gxm, c1 = ax * q.gxp, c2 = ay * q.gym, c3 = ay * q.gyp, c4 = az * gzm, c5 = az * gzp;
            const double d = -(((((c0 + c1) + c2) + c3) + c4) + c5);
            const double xcc = s1[sc][ty + 1][lx];
            double s = 0.0;
            if (i > 0) s += c0 * (s1[sc][ty + 1][lx - 1] - xcc);
            if (i < l.nx - 1) s += c1 * (s1[sc][ty + 1][lx + 1] - xcc);
            if (j > 0) s += c2 * (s1[sc][ty][lx] - xcc);
            if (j < l.ny - 1) s += c3 * (s1[sc][ty + 2][lx] - xcc);
            if (kc > 0) s += c4 * (s1[sm][ty + 1][lx] - xcc);
            if (kc < l.nz - 1) s += c5 * (s1[sp][ty + 1][lx] - xcc);
            out[c] = xcc + omega * ((b2[c] - s) / d);
        }
        *reinterpret_cast<v4 *>(xo + (int64_t)kc * plane + off_c) = out;
    }
}

// ---- ONE general Jacobi step, 2.5-D blocked: a workgroup owns a TX x TY tile and marches through KZ planes; a thread keeps
// its cells' z neighbours in registers (the plane it loads ahead becomes the centre, then the lower neighbour) and only the
// CURRENT plane (tile + one halo cell in x and y) sits in LDS, double-buffered: 2 vector loads + ~1 scalar + 1 store per
// thread and plane instead of 6 vector + 2 scalar loads.
template <int KZ, int XCD>
__global__ __launch_bounds__(256) void k_sweep_march(L l, double omega, const double *__restrict__ b, const double *__restrict__ xi,
                                                     double *__restrict__ xo)
{
    __shared__ double sp[2][SY][SX];
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (XCD) {
        // workgroup b runs on XCD b % 8: give every XCD a contiguous range of tiles (y fastest, then x, then z chunks) so
        // that the y-halo rows of a tile are fetched by the L2 that also serves its neighbours
        const unsigned nb = gridDim.x, per = nb >> 3;
        const unsigned t = (nb & 7u) ? blockIdx.x : (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        const unsigned ntx = l.nx / TX, nty = l.ny / TY;
        by = t % nty;
        bx = (t / nty) % ntx;
        bz = t / (nty * ntx);
    }
    const int i0 = bx * TX, j0 = by * TY, k0 = bz * KZ;
    const int64_t plane = (int64_t)l.nx * l.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    const int hy_row = (tid < 128) ? -1 : TY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? TX : -1, hx_y = (tid >> 1) & 7;
    const int hyj = j0 + hy_row, hyi = i0 + hy_x, hxj = j0 + hx_y, hxi = i0 + hx_col;
    const bool hy_ok = hyj >= 0 && hyj < l.ny, hx_ok = tid < 16 && hxi >= 0 && hxi < l.nx;
    const int64_t off_c = (int64_t)j * l.nx + ic, off_hy = (int64_t)hyj * l.nx + hyi, off_hx = (int64_t)hxj * l.nx + hxi;
    Cell1 q4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = cell1(l, ic + c, j);
    const int kend = (k0 + KZ < l.nz) ? k0 + KZ : l.nz;
    v4 zm = {0, 0, 0, 0}, xc, zp = {0, 0, 0, 0};
    if (k0 > 0) zm = *reinterpret_cast<const v4 *>(xi + (int64_t)(k0 - 1) * plane + off_c);
    xc = *reinterpret_cast<const v4 *>(xi + (int64_t)k0 * plane + off_c);
    for (int kk = k0; kk < kend; ++kk) {
        const int slot = kk & 1;
        const double *px = xi + (int64_t)kk * plane;
        if (kk + 1 < l.nz) zp = *reinterpret_cast<const v4 *>(px + plane + off_c);
        const v4 bv = *reinterpret_cast<const v4 *>(b + (int64_t)kk * plane + off_c);
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[slot][ty + 1][4 * tx + 1 + c] = xc[c];
        sp[slot][hy_row + 1][hy_x + 1] = hy_ok ? px[off_hy] : 0.0;
        if (tid < 16) sp[slot][hx_y + 1][hx_col + 1] = hx_ok ? px[off_hx] : 0.0;
        __syncthreads();
        const double wzk = l.wz[kk];
        const double gzm = (kk > 0) ? l.gz[kk - 1] : 0.0, gzp = (kk < l.nz - 1) ? l.gz[kk] : 0.0;
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ic + c, lx = 4 * tx + 1 + c;
            const Cell1 &q = q4[c];
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
        *reinterpret_cast<v4 *>(xo + (int64_t)kk * plane + off_c) = out;
        zm = xc;
        xc = zp;
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int64_t N = (int64_t)n * n * n;
    std::vector<double> hw(n), hg(n);
    for (int i = 0; i < n; ++i) hw[i] = (1.0 + 0.3 * ((i * 37) % 11) / 11.0) / n;
    for (int i = 0; i + 1 < n; ++i) hg[i] = 5e-4 * (1.0 / (0.5 * (hw[i] + hw[i + 1])));
    double *w, *g, *b, *x1, *y, *yref;
    CK(hipMalloc(&w, 8 * n));
    CK(hipMalloc(&g, 8 * n));
    CK(hipMemcpy(w, hw.data(), 8 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), 8 * n, hipMemcpyHostToDevice));
    CK(hipMalloc(&b, 8 * N));
    CK(hipMalloc(&x1, 8 * N));
    CK(hipMalloc(&y, 8 * N));
    CK(hipMalloc(&yref, 8 * N));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int64_t o = 0; o < N; o += (int64_t)h.size()) CK(hipMemcpy(b + o, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    L l{n, n, n, w, w, w, g, g, g};
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
        printf("%-52s %8.3f ms\n", name, ms / reps);
        fflush(stdout);
    };
    const unsigned plane = (unsigned)n * n;
    timeit("two kernels: x1 = w b/d ; x2 = x1 + w (b - A x1)/d", [&] {
        hipLaunchKernelGGL(k_m1, dim3((plane + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x1);
        hipLaunchKernelGGL(k_m2<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x1, yref);
    });
    timeit("fused, z-marching, KZ 64", [&] {
        hipLaunchKernelGGL((k_fused<64, 0, 0>), dim3(n / TX, n / TY, n / 64), dim3(256), 0, 0, l, 0.9, b, y); });
    timeit("fused, z-marching, KZ 64, prefetch", [&] {
        hipLaunchKernelGGL((k_fused<64, 0, 1>), dim3(n / TX, n / TY, n / 64), dim3(256), 0, 0, l, 0.9, b, y); });
    timeit("fused, z-marching, KZ 64, strided, prefetch", [&] {
        hipLaunchKernelGGL((k_fused<64, 1, 1>), dim3(n / TX, n / TY, n / 64), dim3(256), 0, 0, l, 0.9, b, y); });
    timeit("fused, z-marching, KZ 128, strided, prefetch", [&] {
        hipLaunchKernelGGL((k_fused<128, 1, 1>), dim3(n / TX, n / TY, n / 128), dim3(256), 0, 0, l, 0.9, b, y); });
    {
        // general pair: x0 = the fused result above (any field), reference = two mode-2 steps
        double *x0 = y, *t1 = x1, *r2 = yref, *o2 = nullptr;
        CK(hipMalloc(&o2, 8 * N));
        CK(hipMemcpy(x0, b, 8 * N, hipMemcpyDeviceToDevice));
        timeit("two general steps, two kernels", [&] {
            hipLaunchKernelGGL(k_m2<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x0, t1);
            hipLaunchKernelGGL(k_m2<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, t1, r2);
        });
        timeit("ONE general step, streaming kernel (product's mode 2)", [&] {
            hipLaunchKernelGGL(k_m2<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x0, t1); });
        CK(hipMemcpy(r2, t1, 8 * N, hipMemcpyDeviceToDevice));
        for (int rep = 0; rep < 1; ++rep) {
            timeit("ONE general step, 2.5-D blocked, KZ 32", [&] {
                hipLaunchKernelGGL((k_sweep_march<32, 0>), dim3(n / TX, n / TY, n / 32), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
            timeit("ONE general step, 2.5-D blocked, KZ 64", [&] {
                hipLaunchKernelGGL((k_sweep_march<64, 0>), dim3(n / TX, n / TY, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
            timeit("ONE general step, 2.5-D blocked, KZ 32, XCD ranges", [&] {
                hipLaunchKernelGGL((k_sweep_march<32, 1>), dim3((n / TX) * (n / TY) * (n / 32)), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
            timeit("ONE general step, 2.5-D blocked, KZ 64, XCD ranges", [&] {
                hipLaunchKernelGGL((k_sweep_march<64, 1>), dim3((n / TX) * (n / TY) * (n / 64)), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
            timeit("ONE general step, 2.5-D blocked, KZ 16, XCD ranges", [&] {
                hipLaunchKernelGGL((k_sweep_march<16, 1>), dim3((n / TX) * (n / TY) * (n / 16)), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
        }
        {
            std::vector<double> a0(1 << 22), a1(1 << 22);
            int64_t bad1 = 0;
            for (int64_t off : {int64_t(0), N / 2 - (int64_t)a0.size() / 2, N - (int64_t)a0.size()}) {
                CK(hipMemcpy(a0.data(), r2 + off, 8 * a0.size(), hipMemcpyDeviceToHost));
                CK(hipMemcpy(a1.data(), o2 + off, 8 * a0.size(), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < a0.size(); ++i) bad1 += (a0[i] != a1[i]);
            }
            printf("  2.5-D step vs streaming step: mismatching values %lld\n", (long long)bad1);
        }
        timeit("two general steps, two kernels (again)", [&] {
            hipLaunchKernelGGL(k_m2<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, x0, t1);
            hipLaunchKernelGGL(k_m2<4>, dim3((plane / 4 + 255) / 256, n), dim3(256), 0, 0, l, 0.9, b, t1, r2);
        });
        timeit("two general steps fused, KZ 64", [&] {
            hipLaunchKernelGGL((k_fused_pair<64>), dim3(n / TX, n / TY, n / 64), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
        timeit("two general steps fused, KZ 128", [&] {
            hipLaunchKernelGGL((k_fused_pair<128>), dim3(n / TX, n / TY, n / 128), dim3(256), 0, 0, l, 0.9, b, x0, o2); });
        CK(hipMemcpy(y, o2, 8 * N, hipMemcpyDeviceToDevice));
    }
    std::vector<double> h0(N > (1 << 24) ? (1 << 24) : N), h1(h0.size());
    int64_t bad = 0;
    for (int64_t off : {int64_t(0), N / 2 - (int64_t)h0.size() / 2, N - (int64_t)h0.size()}) {
        CK(hipMemcpy(h0.data(), yref + off, 8 * h0.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), y + off, 8 * h0.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h0.size(); ++i) bad += (h0[i] != h1[i]);
    }
    printf("mismatching values: %lld\n", (long long)bad);
    return 0;
}
