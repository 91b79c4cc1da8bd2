// gmg_level_kernels.hpp -- geometric multigrid (gmg.hip), device side: the level operator, the smoothers (flat and 2.5-D blocked, the fused first pair from a zero guess) and the cycle's reductions.
// Included by gmg.hip only (one translation unit: the launches there instantiate these templates).
#pragma once
#include "pib_internal.hpp"

namespace pib {

// 1-D transfer table of one direction towards the next coarser level (see grid_register)
struct Tr1 {
    const int *par, *oth, *fst;  // parent / other coarse cell of fine cell s; first child of coarse cell I
    const double *wpar, *woth;   // their interpolation weights
};
// x direction, packed per COARSE cell I for the row kernels (one lane <-> one coarse cell and its children)
struct TrX {
    const int2 *fc;     // first child, number of children (1 or 2)
    const double4 *pw;  // prolongation: child0 <- (I, I-1) weights, child1 <- (I, I+1) weights (zeros: none)
    const double4 *rw;  // restriction: weights with which fine cells f0-1, f0, f0+1 (0 if lone), f0+cnt feed I
};
struct LevelDev {
    int nx, ny, nzg;  // global cells (each < 2^31; the local cell count fits int32 like the CSR columns)
    int k0, nk;       // owned planes [k0, k0+nk)
    int per;          // bit 0/1/2: x/y/z periodic (the operator wraps: g[n-1] couples cell n-1 and cell 0)
    int tper;         // ... and the transfers towards the next coarser level reach across the seam
    int zring;        // distributed level of a periodic slab axis: the z wrap goes through the halo planes (+-plane)
    const double *wx, *wy, *wz, *gx, *gy, *gz;
    // volume-scaled rows (see "level operator" below): coefficient towards -d / +d of cell s, and 1 / width
    const double *cmx, *cpx, *rwx, *cmy, *cpy, *rwy, *cmz, *cpz, *rwz;
    Tr1 t[3];         // x, y, z tables (null on the coarsest level)
    TrX tx;
};

// ---- level operator --------------------------------------------------------------------------------------------
// Row (i,j,k) of the level's finite-volume operator has the face coefficient (w_a w_b) g_d[s] towards +d.  Divided by
// the cell volume w_x w_y w_z it becomes g_d[s] / w_d[s]: a function of ONE index, tabulated per level as
//     cm_d[s] (towards -d), cp_d[s] (towards +d)            -- zero at a wall, the wrap face on a periodic direction --
// and the diagonal is -(sum of the six).  Jacobi only ever uses D^-1 (b - A x), which a row scaling leaves unchanged, so
// every smoothing kernel works with the scaled row
//     t = sum_faces c (x_nb - x_c),   d = -(((((cxm + cxp) + cym) + cyp) + czm) + czp),   bs = (b (1/wx 1/wy)) 1/wz,
//     x' = x + omega (bs - t) / d
// -- no per-cell coefficient products, no boundary branches (a missing neighbour is a zero coefficient times a value
// that is 0 or the centre's own) -- and only the residual and the operator itself multiply the volume back in:
//     r = b - (t (wx wy)) wz.
// The expressions and their order are the same in every kernel below and in the oracle (oracle/csrc/gmg.c): fused and
// unfused, tiled and streaming forms give the same bits.  tools/vcycle_lab.hip: 0.76 -> 0.61 ms per 512^3 Jacobi step.
//
// Fused multiply-adds, spelled out (round 4).  The library is built -ffp-contract=off so that nothing contracts by accident;
// these three helpers are the places where a product is NOT rounded before it is added -- v_fma_f64 here, fma() of <math.h>
// (vfmadd under -march=x86-64-v3) in oracle/csrc/gmg.c, the same call in the same order on both sides, so the bits still
// agree -- and a face term costs two fp64 instructions instead of three (the marching kernels are bound by VALU issue):
//     facc : s + c (x_nb - x_c)        one face of the scaled row sum
//     resid: b - t w                   the residual's last factor (t = row sum times two widths, w the third)
__device__ __forceinline__ double facc(double s, double c, double nb, double xc) { return fma(c, nb - xc, s); }
__device__ __forceinline__ double resid(double b, double t, double w) { return fma(-t, w, b); }
//     tacc : s + w v                   one term of an interpolation / restriction sum (w = the product of the 1-D weights)
__device__ __forceinline__ double tacc(double s, double w, double v) { return fma(w, v, s); }
// The damped-Jacobi step in its weighted-average form (round 4, second half).  With sum_faces c = -d,
//     x + omega (bs - sum c (x_nb - x)) / d  =  (1 - omega) x + (omega / d) (bs - sum c x_nb)
// -- the same step in exact arithmetic; in this form a face costs ONE fp64 instruction (six instead of twelve per cell), and
// the only division, wr = omega / d, depends on the cell column's in-plane coefficients and on the PLANE's two z coefficients:
// the marching kernels keep wr of their cells in registers and divide again only when a plane's (czm, czp) differ from the
// previous plane's (a workgroup-uniform comparison; on a mesh with uniform spacing along z: at the two walls only).  The step
// cost 33 fp64 instructions per cell in the difference form (10 of them the division), 11 here.  Every kernel below and
// oracle/csrc/gmg.c use these three calls in this order:
//     jweight: wr = omega / d                       (d = -(((((cxm + cxp) + cym) + cyp) + czm) + czp), one IEEE division)
//     nacc   : t - c x_nb                           one face, starting from t = bs, in the order -x +x -y +y -z +z
//     jrelax : (1 - omega) x + wr t                 as fma(wr, t, omc * x), omc = 1.0 - omega
// and a step from a zero guess is wr * bs (what jrelax gives for x = 0 and zero neighbours).  The residual and the operator
// itself keep the difference form (facc): they need d x_c, and cancellation there would cost them digits.
__device__ __forceinline__ double jweight(double omega, double d) { return omega / d; }
// a workgroup-uniform double the compiler loaded through the vector path (a table entry of the plane a march is on, read inside
// a loop that also stores: no scalar load) moved to scalar registers
__device__ __forceinline__ double uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double nacc(double t, double c, double nb) { return fma(-c, nb, t); }
__device__ __forceinline__ double jrelax(double x, double omc, double wr, double t) { return fma(wr, t, omc * x); }
__device__ __forceinline__ void face_coefs(const LevelDev &L, int i, int j, int k, double c[6])
{
    c[0] = L.cmx[i];
    c[1] = L.cpx[i];
    c[2] = L.cmy[j];
    c[3] = L.cpy[j];
    c[4] = L.cmz[k];
    c[5] = L.cpz[k];
}
// b / volume in the association every kernel uses: (b (1/wx 1/wy)) 1/wz
__device__ __forceinline__ double scale_b(const LevelDev &L, int i, int j, int k, double b) { return (b * (L.rwx[i] * L.rwy[j])) * L.rwz[k]; }
__device__ __forceinline__ double unscale(const LevelDev &L, int i, int j, int k, double t) { return (t * (L.wx[i] * L.wy[j])) * L.wz[k]; }

// the scaled row sum t at local cell p (x points at the first OWNED plane; halo planes sit at -plane and +nk*plane)
__device__ __forceinline__ double apply_cell(const LevelDev &L, const double *__restrict__ x, int64_t p, int i, int j,
                                             int k, double *diag)
{
    double c[6];
    face_coefs(L, i, j, k, c);
    const int64_t sy = L.nx, sz = (int64_t)L.nx * L.ny;
    const double xc = x[p];
    double s = 0.0;
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    if (i > 0) s = facc(s, c[0], x[p - 1], xc);
    else if (px) s = facc(s, c[0], x[p + (L.nx - 1)], xc);
    if (i < L.nx - 1) s = facc(s, c[1], x[p + 1], xc);
    else if (px) s = facc(s, c[1], x[p - (L.nx - 1)], xc);
    if (j > 0) s = facc(s, c[2], x[p - sy], xc);
    else if (py) s = facc(s, c[2], x[p + (L.ny - 1) * sy], xc);
    if (j < L.ny - 1) s = facc(s, c[3], x[p + sy], xc);
    else if (py) s = facc(s, c[3], x[p - (L.ny - 1) * sy], xc);
    if (k > 0) s = facc(s, c[4], x[p - sz], xc);
    else if (pz) s = facc(s, c[4], x[L.zring ? p - sz : p + (L.nzg - 1) * sz], xc);
    if (k < L.nzg - 1) s = facc(s, c[5], x[p + sz], xc);
    else if (pz) s = facc(s, c[5], x[L.zring ? p + sz : p - (L.nzg - 1) * sz], xc);
    *diag = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
    return s;
}

// bs - sum_faces c x_nb at local cell p (what jrelax takes), and the scaled diagonal: apply_cell's walk in the weighted-average form
__device__ __forceinline__ double relax_cell(const LevelDev &L, const double *__restrict__ x, int64_t p, int i, int j, int k, double bs,
                                             double *diag)
{
    double c[6];
    face_coefs(L, i, j, k, c);
    const int64_t sy = L.nx, sz = (int64_t)L.nx * L.ny;
    double t = bs;
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    if (i > 0) t = nacc(t, c[0], x[p - 1]);
    else if (px) t = nacc(t, c[0], x[p + (L.nx - 1)]);
    if (i < L.nx - 1) t = nacc(t, c[1], x[p + 1]);
    else if (px) t = nacc(t, c[1], x[p - (L.nx - 1)]);
    if (j > 0) t = nacc(t, c[2], x[p - sy]);
    else if (py) t = nacc(t, c[2], x[p + (L.ny - 1) * sy]);
    if (j < L.ny - 1) t = nacc(t, c[3], x[p + sy]);
    else if (py) t = nacc(t, c[3], x[p - (L.ny - 1) * sy]);
    if (k > 0) t = nacc(t, c[4], x[p - sz]);
    else if (pz) t = nacc(t, c[4], x[L.zring ? p - sz : p + (L.nzg - 1) * sz]);
    if (k < L.nzg - 1) t = nacc(t, c[5], x[p + sz]);
    else if (pz) t = nacc(t, c[5], x[L.zring ? p + sz : p - (L.nzg - 1) * sz]);
    *diag = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
    return t;
}

// Launch geometry of every level kernel: grid (ceil(plane/256) capped, nk); blockIdx.y is the local plane, so
// k is workgroup-uniform (its coefficients come through the scalar path) and only ONE 32-bit division per
// cell is left (j = q / nx).  64-bit div/mod per cell made the first version of these kernels ALU-bound.
#define PIB_PLANE_LOOP(L)                                                                  \
    const unsigned plane_ = (unsigned)(L).nx * (unsigned)(L).ny;                           \
    const int kk_ = blockIdx.y;                                                            \
    const int k = (L).k0 + kk_;                                                            \
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < plane_; q += gridDim.x * 256u)
#define PIB_PLANE_IJ(L)                                  \
    const int j = (int)(q / (unsigned)(L).nx);           \
    const int i = (int)(q - (unsigned)j * (unsigned)(L).nx); \
    const int64_t p = (int64_t)kk_ * plane_ + q;

// mode 0: y = A x                       (stencil twin K2)
// mode 1: xo = omega * b / diag          (Jacobi from a zero guess)
// mode 2: xo = xi + omega (b - A xi)/diag
// mode 3: r  = b - A xi                  (written to xo)
// pin_sum != nullptr: effective b at global cell 0 is b[0] - *pin_sum (PINNED null space)
//
// C cells per lane along i (C = 4, 2 or 1 by divisibility of nx): the centre, +-y and +-z neighbours are
// read as one 16/32-byte access each, index arithmetic and the j/k coefficients are amortised over C cells.
// One cell per lane ran at 1.9 TB/s (24 B/cell) on the 512^3 level, four cells per lane at 3.7 TB/s
// (tools/gmg_lab.hip); the arithmetic per cell is unchanged, so results are bit-identical.
// mode 5: Chebyshev-Jacobi step   d = a_d d + a_z (b - A xi)/diag ; xo = xi + d      (omega carries a_z)
// mode 6: first Chebyshev step from a zero guess:  d = a_z b/diag ; xo = d
// mode 8: mode 2 + the sums the Krylov solver wants of the result (the LAST post-smoothing step of level 0 writes
//         z = M^-1 r): per-workgroup partials of z.b, z.z, sum z go to part[k * part_stride + block] -- saves the
//         separate pass over z and r (0.39 ms per 512^3 iteration).  b here is the unmodified residual.
template <int MODE, int C>
__global__ __launch_bounds__(256) void k_level(const Scalars *__restrict__ S, LevelDev L, double omega,
                                               const double *__restrict__ b, const double *__restrict__ xi,
                                               double *__restrict__ xo, const double *__restrict__ pin_sum,
                                               double *__restrict__ dvec, double a_d, double *__restrict__ part,
                                               int part_stride, int dlo, int dhi)
{
    if (S != nullptr && S->done) return;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    const bool dots = MODE == 8 && (int)blockIdx.y >= dlo && (int)blockIdx.y < dhi;  // the sums cover the OWNED planes only
    typedef double vt __attribute__((ext_vector_type(C), aligned(C == 1 ? 8 : 16)));
    const unsigned nxc = (unsigned)L.nx / C;  // lane groups per grid line
    const unsigned planec = nxc * (unsigned)L.ny;
    const int64_t plane = (int64_t)L.nx * L.ny;
    const int kk = blockIdx.y;
    const int k = L.k0 + kk;
    const double wzk = L.wz[k], rwz = L.rwz[k], czm = L.cmz[k], czp = L.cpz[k];
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    // Workgroup b runs on XCD b % 8.  With the plain order the two grid lines of a workgroup have their +-y neighbours
    // in the workgroups of OTHER XCDs, so every L2 fetched x twice (PMC: 3.22 GB read per 512^3 sweep for 2.15 GB
    // of b and x).  Dealing each XCD a contiguous band of the plane leaves 8 band edges per plane instead.
    const unsigned bx = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // rocprof SQ counters: these kernels stall on vector-memory ISSUE (SQ_WAIT_INST_ANY 0.6 of the wave cycles, 26 VMEM
    // reads per wave), not on data.  When a wave stays inside one grid line (nxc a multiple of 64) j is wave-uniform:
    // its coefficients then come through the scalar path; the x coefficients of the lane's C cells are one vector load.
    const bool j_uniform = (nxc & 63u) == 0u;
    for (unsigned q = bx * 256u + threadIdx.x; q < planec; q += gridDim.x * 256u) {
        int j = (int)(q / nxc);
        const int i0 = (int)(q - (unsigned)j * nxc) * C;
        if (j_uniform) j = __builtin_amdgcn_readfirstlane(j);
        const int64_t p = (int64_t)kk * plane + (int64_t)j * L.nx + i0;
        const double wyj = L.wy[j], rwy = L.rwy[j], cym = L.cmy[j], cyp = L.cpy[j];
        vt xc, bv, ym, yp, zm, zp, out;
        double xl = 0.0, xr = 0.0;
        vt dv;
        if (MODE == 5 && a_d != 0.0) dv = *reinterpret_cast<const vt *>(dvec + p);
        if (MODE != 1 && MODE != 6) {
            xc = *reinterpret_cast<const vt *>(xi + p);
            ym = yp = zm = zp = xc;
            if (i0 > 0) xl = xi[p - 1];
            else if (px) xl = xi[p + (L.nx - 1)];
            if (i0 + C < L.nx) xr = xi[p + C];
            else if (px) xr = xi[p + C - L.nx];
            if (j > 0) ym = *reinterpret_cast<const vt *>(xi + p - L.nx);
            else if (py) ym = *reinterpret_cast<const vt *>(xi + p + (int64_t)(L.ny - 1) * L.nx);
            if (j < L.ny - 1) yp = *reinterpret_cast<const vt *>(xi + p + L.nx);
            else if (py) yp = *reinterpret_cast<const vt *>(xi + p - (int64_t)(L.ny - 1) * L.nx);
            if (k > 0) zm = *reinterpret_cast<const vt *>(xi + p - plane);
            else if (pz) zm = *reinterpret_cast<const vt *>(xi + p + (L.zring ? -plane : (int64_t)(L.nzg - 1) * plane));
            if (k < L.nzg - 1) zp = *reinterpret_cast<const vt *>(xi + p + plane);
            else if (pz) zp = *reinterpret_cast<const vt *>(xi + p + (L.zring ? plane : -(int64_t)(L.nzg - 1) * plane));
        }
        vt braw;
        if (MODE != 0) {
            bv = *reinterpret_cast<const vt *>(b + p);
            if (MODE == 8) braw = bv;
            if (pin_sum != nullptr && p == 0 && L.k0 == 0) bv[0] = bv[0] - *pin_sum;
        }
        // the 1-D tables are padded: aligned vectors of C entries may be read at any i0
        const vt cxmv = *reinterpret_cast<const vt *>(L.cmx + i0), cxpv = *reinterpret_cast<const vt *>(L.cpx + i0);
        const vt rwxv = *reinterpret_cast<const vt *>(L.rwx + i0);
        vt wxv;
        if (MODE == 0 || MODE == 3) wxv = *reinterpret_cast<const vt *>(L.wx + i0);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double cxm = cxmv[c], cxp = cxpv[c];
            const double d = -(((((cxm + cxp) + cym) + cyp) + czm) + czp);
            double bs = 0.0;
            if (MODE != 0) bs = (bv[c] * (rwxv[c] * rwy)) * rwz;
            if (MODE == 1) {
                out[c] = jweight(omega, d) * bs;
                continue;
            }
            if (MODE == 6) {
                out[c] = omega * (bs / d);
                dv[c] = out[c];
                continue;
            }
            const double left = (c == 0) ? xl : xc[c > 0 ? c - 1 : 0];
            const double right = (c == C - 1) ? xr : xc[c < C - 1 ? c + 1 : 0];
            const double xcc = xc[c];
            if (MODE == 2 || MODE == 8) {
                double t = bs;
                t = nacc(t, cxm, left);
                t = nacc(t, cxp, right);
                t = nacc(t, cym, ym[c]);
                t = nacc(t, cyp, yp[c]);
                t = nacc(t, czm, zm[c]);
                t = nacc(t, czp, zp[c]);
                out[c] = jrelax(xcc, 1.0 - omega, jweight(omega, d), t);
                if (MODE == 8 && dots) {
                    acc0 += out[c] * braw[c];
                    acc1 += out[c] * out[c];
                    acc2 += out[c];
                }
                continue;
            }
            // a missing neighbour: zero coefficient, and the value is 0 (xl, xr) or the centre's own (ym .. zp)
            double s = 0.0;
            s = facc(s, cxm, left, xcc);
            s = facc(s, cxp, right, xcc);
            s = facc(s, cym, ym[c], xcc);
            s = facc(s, cyp, yp[c], xcc);
            s = facc(s, czm, zm[c], xcc);
            s = facc(s, czp, zp[c], xcc);
            if (MODE == 0)
                out[c] = (s * (wxv[c] * wyj)) * wzk;
            else if (MODE == 5) {
                const double z = (bs - s) / d;
                const double dn = (a_d != 0.0) ? a_d * dv[c] + omega * z : omega * z;
                dv[c] = dn;
                out[c] = xcc + dn;
            } else
                out[c] = resid(bv[c], s * (wxv[c] * wyj), wzk);
        }
        if (MODE == 5 || MODE == 6) *reinterpret_cast<vt *>(dvec + p) = dv;
        *reinterpret_cast<vt *>(xo + p) = out;
    }
    if (MODE == 0 && part != nullptr) {
        // one partial per workgroup; the slots up to part_stride that no workgroup owns are zeroed (the consumer sums a
        // fixed number of them)
        __shared__ double sh0[4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc0 += __shfl_down(acc0, o, 64);
        if (lane == 0) sh0[w] = acc0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int64_t nwg = (int64_t)gridDim.x * gridDim.y * gridDim.z;
            const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            part[blk] = (sh0[0] + sh0[1]) + (sh0[2] + sh0[3]);
            for (int64_t e = blk + nwg; e < part_stride; e += nwg) part[e] = 0.0;
        }
    }
    if (MODE == 8) {
        __shared__ double sh[3][4];
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
            const int k2 = threadIdx.x;
            part[(int64_t)k2 * part_stride + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (sh[k2][0] + sh[k2][1]) + (sh[k2][2] + sh[k2][3]);
        }
    }
}

// ---- the first two pre-smoothing steps from a zero guess in ONE kernel:
//   x1 = omega b / d            (mode 1)
//   x2 = x1 + omega (b - A x1) / d   (mode 2)
// as two streaming kernels these are 5 vector passes over HBM (b, x1 | x1, b, x2); here a workgroup owns a 128 x 8 tile
// of the plane and marches through FZ planes with the x1 planes (tile + one halo cell in x and y) in a ring of three LDS
// slots (the thread's own x1 values of three consecutive planes stay in registers): b is read once (1.27 x with the halo), x2 written once.  Every k-independent mesh coefficient of the thread's
// cells is loaded before the march (the 1-D arrays alone cost ~55 vector-memory instructions per thread and plane
// otherwise).  Same expressions in the same order as modes 1 and 2: bit-identical (tools/fuse_lab.hip: 1.34 -> 0.78 ms
// per 512^3 pair).  Levels that are whole on this rank, not periodic, 3-D, nx % 128 == 0, ny % 8 == 0.
constexpr int FX = 128, FY = 8, FSY = FY + 2;
// LDS rows of the marching kernels that hand 4-cell pieces to a lane (round 4, second half).  In the natural order a lane's
// piece is 32 bytes and a ds_read_b128 / ds_write_b128 of half a piece across the lanes has a 32-byte stride: its 16-lane
// groups use every other 16-byte slot of the 256-byte bank row -- a two-way conflict on every access (PMC: SQ_LDS_BANK_CONFLICT
// half of SQ_LDS_IDX_ACTIVE in k_prolong_smooth2 and k_resid_restrict_march, the LDS busy half of their time).  Swizzled row:
// the FIRST halves (cells 0, 1) of all pieces side by side, the SECOND halves (cells 2, 3) SWH doubles further on -- both
// 16-byte strides; SWH = 40 slots = 8 (mod 16), so that an access whose lanes alternate between the halves (the restriction's
// reads of the cells 2 l + 4, 2 l + 5) spreads over all sixteen slots too.  Cell X of a row sits at swz(X).
constexpr int SWR = 160, SWH = 80;  // doubles per swizzled row (>= 2 SWH, rows 136 cells wide), offset of the second halves
__device__ __forceinline__ int swz(int X) { return ((X >> 2) << 1) + (X & 1) + ((X >> 1) & 1) * SWH; }
typedef double swv2 __attribute__((ext_vector_type(2)));
typedef double swv4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void swz_put4(double *row, int X, const swv4 &v)  // X a multiple of 4
{
    const swv2 a = {v[0], v[1]}, b = {v[2], v[3]};
    *reinterpret_cast<swv2 *>(row + (X >> 1)) = a;
    *reinterpret_cast<swv2 *>(row + (X >> 1) + SWH) = b;
}
__device__ __forceinline__ swv4 swz_get4(const double *row, int X)  // X a multiple of 4
{
    const swv2 a = *reinterpret_cast<const swv2 *>(row + (X >> 1)), b = *reinterpret_cast<const swv2 *>(row + (X >> 1) + SWH);
    const swv4 v = {a[0], a[1], b[0], b[1]};
    return v;
}
// A workgroup barrier that orders LDS only: __syncthreads() is a release / acquire fence over ALL memory, i.e. s_waitcnt vmcnt(0)
// in front of every s_barrier -- which ends the flight of the global loads a marching kernel has requested for its NEXT plane
// at the first barrier of the current one.  The LDS hand-over between the stages of a plane needs lgkmcnt(0) only.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Register budgets of the LDS-tiled kernels.  A 256-thread workgroup is four waves, one per SIMD, and the compiler sizes
// its register use for whatever occupancy it happens to reach: k_level_march<8> took 144 VGPRs (three waves per SIMD),
// k_presmooth2 142 (three), k_prolong_smooth 212 (two).  amdgpu_waves_per_eu(n) asks for n: the march fits 126 without
// a spill (kept), the pre-smoothing pair 128 with five spilled dwords, the fused prolongation 168 with 43.  Measured on
// the 512^3 solve (two runs each): none 89.7 / 88.1 ms, march at four waves 86.7 / 88.0, + pre-smoothing at four 87.4 /
// 88.1, + prolongation at three 112.6 / 113.2 -- occupancy is not what holds these kernels back, spills are poison.
// (The `vgpr` column of rocprofv3's kernel trace counts in units of two on gfx950 -- 72 there is 144 here; the numbers
// above are the code object's .vgpr_count.)  With the two-plane prefetch of k_level_march the step with the Krylov sums
// (mode 8) no longer fits four waves without 17 spilled dwords -- 823 instead of 620 us per 512^3 launch inside the solve
// (tools/ab_trace.sh) -- so it asks for three (162 VGPRs, no spill); the other modes keep four (128, three dwords).
#ifndef PIB_WAVES_MARCH
#define PIB_WAVES_MARCH 4
#endif
#ifndef PIB_WAVES_PRESMOOTH
#define PIB_WAVES_PRESMOOTH 0
#endif
#ifndef PIB_WAVES_PROLONG
#define PIB_WAVES_PROLONG 0
#endif
#define PIB_WAVES_ATTR_0
#define PIB_WAVES_ATTR_2 __attribute__((amdgpu_waves_per_eu(2)))
#define PIB_WAVES_ATTR_3 __attribute__((amdgpu_waves_per_eu(3)))
#define PIB_WAVES_ATTR_4 __attribute__((amdgpu_waves_per_eu(4)))
#define PIB_WAVES_ATTR_5 __attribute__((amdgpu_waves_per_eu(5)))
#define PIB_WAVES_CAT(a, b) a##b
#define PIB_WAVES_ATTR(n) PIB_WAVES_CAT(PIB_WAVES_ATTR_, n)

// tile of this workgroup.  Workgroup b (in dispatch order: x fastest) runs on XCD b % 8; every XCD is dealt
// a contiguous band of y-tiles (all x-tiles of it, z-chunk after z-chunk), so that the halo rows and columns two neighbouring
// tiles both read are fetched by ONE L2 (profiles: k_presmooth2 reads 1.46 x its algorithmic bytes in the plain order)
struct Tile3 {
    int x, y, z;
};
template <bool BANDS = true>
__device__ __forceinline__ Tile3 tile_of_block()
{
#ifndef PIB_NO_XCD_BANDS
    const unsigned nbx = gridDim.x, nby = gridDim.y;
    if (BANDS && (nby & 7u) == 0u) {
        const unsigned id = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);
        const unsigned xcd = id & 7u, m = id >> 3, band = nby >> 3;
        const unsigned r = m / nbx;
        return {(int)(m - r * nbx), (int)(xcd * band + r % band), (int)(r / band)};
    }
#endif
    return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
}

// what a thread keeps of a cell column (i, j) across the planes: the scaled in-plane coefficients, their part of the
// diagonal sum, 1 / (wx wy) and wx wy
struct FCell {
    double cxm, cxp, cym, cyp, s4, rxy, vxy;
};
__device__ __forceinline__ FCell fcell(const LevelDev &L, int i, int j)
{
    FCell c;
    c.cxm = L.cmx[i];
    c.cxp = L.cpx[i];
    c.cym = L.cmy[j];
    c.cyp = L.cpy[j];
    c.s4 = ((c.cxm + c.cxp) + c.cym) + c.cyp;
    c.rxy = L.rwx[i] * L.rwy[j];
    c.vxy = L.wx[i] * L.wy[j];
    return c;
}
__device__ __forceinline__ double fdiag(const FCell &q, double czm, double czp) { return -((q.s4 + czm) + czp); }
// RES = 1 (a V(1,.) cycle: ONE pre-smoothing step): the second stage is the residual r = b - A x1 instead of the second
// Jacobi step; x1 goes to xo, r to ro -- b read once, two vectors written, instead of mode 1 + mode 3 (2 + 3 passes).
// UPD = 1 (level 0 of the preconditioner inside PCG, one rank): the right-hand side is the Krylov residual, and its update
// r = r_old - alpha w is done HERE as the planes are read -- b is r_old, `uw` is w = A p, alpha is S->a -- instead of in a pass
// of its own (24 B/row): every loaded cell (halo cells included) is updated on the fly, the tile's own cells of its own
// planes are written to `unew` (a second buffer: a neighbouring tile still reads the old values of these cells) and their
// r.r and sum r go to upart[0 / 1][workgroup] for the solver's finalize kernel.  Same expression as OpUpdateXR: same r.
template <int RES, int UPD = 0>
__global__ __launch_bounds__(256) PIB_WAVES_ATTR(PIB_WAVES_PRESMOOTH) void k_presmooth2(const Scalars *__restrict__ S, LevelDev L, double omega,
                                                    const double *__restrict__ b, double *__restrict__ xo,
                                                    const double *__restrict__ pin_sum, int FZ, double *__restrict__ ro,
                                                    const double *__restrict__ uw = nullptr, double *__restrict__ unew = nullptr,
                                                    double *__restrict__ upart = nullptr, int upart_stride = 0, int wext = 0,
                                                    int sum_lo = -(1 << 30), int sum_hi = 1 << 30, int blk_base = 0)
{
    // UPD on z-slabs (round 4): the run covers ghost planes too; the new residual is also written on the plane just below the
    // run (wext bit 0, first z-chunk) / just above it (bit 1, last z-chunk) -- planes the march loads and updates anyway -- so
    // that the neighbours' planes of r are kept by recurrence (w is exchanged, r never again); the sums cover the owned
    // planes [sum_lo, sum_hi) only; the partials of the launches of one cycle sit side by side (blk_base).
    if (S != nullptr && S->done) return;
    // rows of the tile's plane in LDS: the cells i0 - 4 .. i0 + 131 in the swizzled order (swz: the thread's four cells at X = 4 + 4 tx as
    // two aligned 16-byte halves with 16-byte lane strides; the x halo cells are X = 3 and X = 132) -- the natural order with one
    // halo cell put a thread's cells at an odd offset: 8-byte accesses with a 32-byte stride, four-way bank conflicts
    __shared__ __attribute__((aligned(32))) double x1[3][FSY][SWR];
    const double ua = UPD ? S->a : 0.0;
    double ur0 = 0.0, ur1 = 0.0;
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    // processed planes: [L.k0, L.k0 + L.nk) (global); b / xo / ro point at the first of them.  A whole level, or a run of
    // planes of a z-slab whose right-hand side is valid one plane beyond the run on every side that has a neighbour.
    const Tile3 tb = tile_of_block();
    const int i0 = tb.x * FX, j0 = tb.y * FY, k0 = L.k0 + tb.z * FZ;
    const int kend = min(k0 + FZ, L.k0 + L.nk);
    const int64_t plane = (int64_t)L.nx * L.ny;
    b -= (int64_t)L.k0 * plane;  // index by global plane below
    xo -= (int64_t)L.k0 * plane;
    if (RES) ro -= (int64_t)L.k0 * plane;
    if (UPD) {
        uw -= (int64_t)L.k0 * plane;
        unew -= (int64_t)L.k0 * plane;
    }
    const int j = j0 + ty, ic = i0 + 4 * tx;  // this thread's 4 cells: (ic .. ic+3, j)
    // halo duty: every thread one cell of the two y-halo rows, 16 threads one cell of the two x-halo columns
    const int hy_row = (tid < 128) ? -1 : FY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? FX : -1, hx_y = (tid >> 1) & 7;
    // periodic directions (whole levels only): the halo cells are the ones across the seam, plane -1 is plane nz - 1
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    int hyj = j0 + hy_row, hxi = i0 + hx_col;
    const int hyi = i0 + hy_x, hxj = j0 + hx_y;
    if (py) hyj = hyj < 0 ? L.ny - 1 : (hyj >= L.ny ? 0 : hyj);
    if (px) hxi = hxi < 0 ? L.nx - 1 : (hxi >= L.nx ? 0 : hxi);
    const bool hy_ok = hyj >= 0 && hyj < L.ny, hx_ok = tid < 16 && hxi >= 0 && hxi < L.nx;
    const int64_t off_c = (int64_t)j * L.nx + ic, off_hy = (int64_t)hyj * L.nx + hyi, off_hx = (int64_t)hxj * L.nx + hxi;
    FCell q4[4], qhy = {}, qhx = {};
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = fcell(L, ic + c, j);
    if (hy_ok) qhy = fcell(L, hyi, hyj);
    if (hx_ok) qhx = fcell(L, hxi, hxj);
    // x1 of the thread's own cells on the planes kk-2, kk-1, kk stays in registers (the z neighbours of step 2); LDS holds
    // the planes for the x / y neighbours only: plane kk is written while plane kk-1 is read, three slots, one barrier
    v4 bprev = {0, 0, 0, 0}, bcur = {0, 0, 0, 0};
    v4 x1m = {0, 0, 0, 0}, x1c = {0, 0, 0, 0}, x1p = {0, 0, 0, 0};
    // wr = omega / d of the thread's cells on the plane the first step works on (wn) and on the plane before it (wc: what
    // the second step needs); divided again only when a plane's z coefficients differ from the previous plane's
    const double omc = 1.0 - omega;
    v4 wn = {0, 0, 0, 0}, wc = {0, 0, 0, 0};
    double wn_hy = 0.0, wn_hx = 0.0, key_zm = __builtin_nan(""), key_zp = __builtin_nan("");
    for (int kk = k0 - 1; kk <= kend; ++kk) {
        const int slot = (kk + 3) % 3;
        bprev = bcur;
        x1m = x1c;
        x1c = x1p;
        wc = wn;
        const int kw = pz ? (kk < 0 ? L.nzg - 1 : (kk >= L.nzg ? 0 : kk)) : kk;
        if (kw >= 0 && kw < L.nzg) {
            const double *pb = b + (int64_t)kw * plane;
            v4 bv = *reinterpret_cast<const v4 *>(pb + off_c);
            double hyv = hy_ok ? pb[off_hy] : 0.0, hxv = hx_ok ? pb[off_hx] : 0.0;
            if (UPD) {
                const double *pw = uw + (int64_t)kw * plane;
                const v4 wv = *reinterpret_cast<const v4 *>(pw + off_c);
#pragma unroll
                for (int c = 0; c < 4; ++c) bv[c] = bv[c] - ua * wv[c];
                if (hy_ok) hyv = hyv - ua * pw[off_hy];
                if (hx_ok) hxv = hxv - ua * pw[off_hx];
                const bool own = kk >= k0 && kk < kend;
                if (own || (kk == k0 - 1 && tb.z == 0 && (wext & 1)) || (kk == kend && kend == L.k0 + L.nk && (wext & 2)))
                    *reinterpret_cast<v4 *>(unew + (int64_t)kw * plane + off_c) = bv;  // this workgroup's own cells: the new residual
                if (own && kw >= sum_lo && kw < sum_hi) {                                // ... and its sums
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ur0 += bv[c] * bv[c];
                        ur1 += bv[c];
                    }
                }
            }
            if (pin_sum != nullptr && kw == 0) {  // PINNED: effective b at cell 0 (as a tile's own cell, or as the halo cell of
                                                  // the tiles across a periodic seam)
                if (off_c == 0) bv[0] = bv[0] - *pin_sum;
                if (hy_ok && off_hy == 0) hyv = hyv - *pin_sum;
                if (hx_ok && off_hx == 0) hxv = hxv - *pin_sum;
            }
            const double rwz = L.rwz[kw], czm = L.cmz[kw], czp = L.cpz[kw];
            bcur = bv;
            if (czm != key_zm || czp != key_zp) {  // (workgroup-uniform)
                key_zm = czm, key_zp = czp;
#pragma unroll
                for (int c = 0; c < 4; ++c) wn[c] = jweight(omega, fdiag(q4[c], czm, czp));
                wn_hy = hy_ok ? jweight(omega, fdiag(qhy, czm, czp)) : 0.0;
                wn_hx = hx_ok ? jweight(omega, fdiag(qhx, czm, czp)) : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) x1p[c] = wn[c] * ((bv[c] * q4[c].rxy) * rwz);
            swz_put4(x1[slot][ty + 1], 4 + 4 * tx, x1p);
            x1[slot][hy_row + 1][swz(4 + hy_x)] = hy_ok ? wn_hy * ((hyv * qhy.rxy) * rwz) : 0.0;
            if (tid < 16) x1[slot][hx_y + 1][swz(4 + hx_col)] = hx_ok ? wn_hx * ((hxv * qhx.rxy) * rwz) : 0.0;
        }
        __syncthreads();
        const int kc = kk - 1;  // the plane whose x1 neighbours are complete now
        if (kc < k0 || kc >= kend) continue;
        const int sc = (kc + 3) % 3;
        const double wzk = L.wz[kc], rwz = L.rwz[kc], czm = L.cmz[kc], czp = L.cpz[kc];
        v4 out;
        // in-plane neighbours: the rows above and below as the thread's aligned pieces, the cells left and right of its four
        // (its own values from the registers: the same numbers the LDS holds)
        const v4 ylo = swz_get4(x1[sc][ty], 4 + 4 * tx), yhi = swz_get4(x1[sc][ty + 2], 4 + 4 * tx);
        const double xleft = x1[sc][ty + 1][swz(3 + 4 * tx)], xright = x1[sc][ty + 1][swz(8 + 4 * tx)];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const FCell &q = q4[c];
            const double xcc = x1c[c];
            const double left = (c == 0) ? xleft : x1c[c > 0 ? c - 1 : 0], right = (c == 3) ? xright : x1c[c < 3 ? c + 1 : 0];
            // missing neighbours: zero coefficients; the LDS halo cells and x1m / x1p outside the domain hold 0
            if (RES) {
                double sum = 0.0;
                sum = facc(sum, q.cxm, left, xcc);
                sum = facc(sum, q.cxp, right, xcc);
                sum = facc(sum, q.cym, ylo[c], xcc);
                sum = facc(sum, q.cyp, yhi[c], xcc);
                sum = facc(sum, czm, x1m[c], xcc);
                sum = facc(sum, czp, x1p[c], xcc);
                out[c] = resid(bprev[c], sum * q.vxy, wzk);
            } else {
                double t = (bprev[c] * q.rxy) * rwz;
                t = nacc(t, q.cxm, left);
                t = nacc(t, q.cxp, right);
                t = nacc(t, q.cym, ylo[c]);
                t = nacc(t, q.cyp, yhi[c]);
                t = nacc(t, czm, x1m[c]);
                t = nacc(t, czp, x1p[c]);
                out[c] = jrelax(xcc, omc, wc[c], t);
            }
        }
        if (RES) {
            *reinterpret_cast<v4 *>(ro + (int64_t)kc * plane + off_c) = out;
            *reinterpret_cast<v4 *>(xo + (int64_t)kc * plane + off_c) = x1c;
        } else
            *reinterpret_cast<v4 *>(xo + (int64_t)kc * plane + off_c) = out;
    }
    if (UPD) {
        __shared__ double ush[2][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ur0 += __shfl_down(ur0, o, 64);
            ur1 += __shfl_down(ur1, o, 64);
        }
        __syncthreads();
        if ((tid & 63) == 0) {
            ush[0][tid >> 6] = ur0;
            ush[1][tid >> 6] = ur1;
        }
        __syncthreads();
        if (tid < 2) {
            const int64_t blk = blk_base + ((int64_t)tb.z * gridDim.y + tb.y) * gridDim.x + tb.x;
            upart[(int64_t)tid * upart_stride + blk] = (ush[tid][0] + ush[tid][1]) + (ush[tid][2] + ush[tid][3]);
        }
    }
}

// ---- one Jacobi step / residual, 2.5-D blocked (modes 2, 3, 8 of k_level on the levels k_presmooth2 serves): a workgroup
// owns a 128 x 8 tile and marches through FZ planes; a thread keeps its cells' z neighbours in registers (the plane it
// loads ahead becomes the centre, then the lower neighbour) and only the CURRENT plane (tile + one halo cell in x and y)
// sits in LDS, double-buffered -- two vector loads, about one scalar load and one store per thread and plane instead of
// six vector and two scalar loads: 0.76 instead of 0.88 ms per 512^3 sweep (tools/fuse_lab.hip), same expressions in the
// same order (modes 2 and 3 bit-identical; mode 8's sums are grouped by tile instead of by line segment, i.e. equal to
// rounding).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE != 0 ? 3 : PIB_WAVES_MARCH))) void k_level_march(const Scalars *__restrict__ S, LevelDev L, double omega,
                                                     const double *__restrict__ b, const double *__restrict__ xi,
                                                     double *__restrict__ xo, const double *__restrict__ pin_sum,
                                                     double *__restrict__ part, int part_stride, int FZ, int dlo, int dhi)
{
    if (S != nullptr && S->done) return;
    __shared__ __attribute__((aligned(32))) double sp[2][FSY][SWR];  // (swizzled rows, see k_presmooth2: cells i0 - 4 .. i0 + 131)
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    // owned planes [L.k0, L.k0 + L.nk) of the level (a z-slab or a part of one); the vectors point at the first of them,
    // the planes below / above hold the neighbours' values (halo planes) where they exist
    const Tile3 tb = tile_of_block<MODE == 8>();
    const int i0 = tb.x * FX, j0 = tb.y * FY, l0 = tb.z * FZ;
    const int64_t plane = (int64_t)L.nx * L.ny;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    const int hy_row = (tid < 128) ? -1 : FY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? FX : -1, hx_y = (tid >> 1) & 7;
    // periodic directions: the halo cells are the ones across the seam; a periodic z needs the whole level here
    // (plane -1 is plane nz - 1), slabs are served with per == 0 only
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    int hyj = j0 + hy_row, hxi = i0 + hx_col;
    const int hyi = i0 + hy_x, hxj = j0 + hx_y;
    if (py) hyj = hyj < 0 ? L.ny - 1 : (hyj >= L.ny ? 0 : hyj);
    if (px) hxi = hxi < 0 ? L.nx - 1 : (hxi >= L.nx ? 0 : hxi);
    const bool hy_ok = hyj >= 0 && hyj < L.ny, hx_ok = tid < 16 && hxi >= 0 && hxi < L.nx;
    const int64_t off_c = (int64_t)j * L.nx + ic, off_hy = (int64_t)hyj * L.nx + hyi, off_hx = (int64_t)hxj * L.nx + hxi;
    FCell q4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = fcell(L, ic + c, j);
    const int lend = (l0 + FZ < L.nk) ? l0 + FZ : L.nk;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    // software pipeline: the own cells of the plane TWO ahead, and the halo cells and right-hand side of the NEXT plane,
    // are requested an iteration before their first use (tools/vcycle_lab.hip: -4 % for the Jacobi step, -13 % for the
    // one-input product at 512^3).  plane_of: a plane's place in memory (local index; across the periodic seam)
    auto plane_of = [&](int lk) -> int64_t {
        const int kk = L.k0 + lk;
        if (pz) return kk < 0 ? L.nzg - 1 : (kk >= L.nzg ? 0 : lk);
        return lk;
    };
    auto have = [&](int lk) { return pz || (L.k0 + lk >= 0 && L.k0 + lk < L.nzg); };
    v4 zm = {0, 0, 0, 0}, xc, zp = {0, 0, 0, 0}, zq = {0, 0, 0, 0}, bv = {0, 0, 0, 0}, bn = {0, 0, 0, 0};
    double hyv, hxv, hyn = 0.0, hxn = 0.0;
    if (have(l0 - 1)) zm = *reinterpret_cast<const v4 *>(xi + plane_of(l0 - 1) * plane + off_c);
    xc = *reinterpret_cast<const v4 *>(xi + (int64_t)l0 * plane + off_c);
    if (have(l0 + 1)) zp = *reinterpret_cast<const v4 *>(xi + plane_of(l0 + 1) * plane + off_c);
    hyv = hy_ok ? xi[(int64_t)l0 * plane + off_hy] : 0.0;
    hxv = hx_ok ? xi[(int64_t)l0 * plane + off_hx] : 0.0;
    if (MODE != 0) bv = *reinterpret_cast<const v4 *>(b + (int64_t)l0 * plane + off_c);
    // wr = omega / d of the thread's cells, divided again only when a plane's z coefficients differ from the previous plane's
    const double omc = 1.0 - omega;
    v4 wr = {0, 0, 0, 0};
    double key_zm = __builtin_nan(""), key_zp = __builtin_nan("");
    for (int lk = l0; lk < lend; ++lk) {
        const int kk = L.k0 + lk;  // global plane
        const int slot = lk & 1;
        if (lk + 1 < lend) {
            const double *pn = xi + (int64_t)(lk + 1) * plane;
            if (have(lk + 2)) zq = *reinterpret_cast<const v4 *>(xi + plane_of(lk + 2) * plane + off_c);
            hyn = hy_ok ? pn[off_hy] : 0.0;
            hxn = hx_ok ? pn[off_hx] : 0.0;
            if (MODE != 0) bn = *reinterpret_cast<const v4 *>(b + (int64_t)(lk + 1) * plane + off_c);
        }
        const v4 braw = bv;
        if (MODE != 0 && pin_sum != nullptr && kk == 0 && off_c == 0) bv[0] = bv[0] - *pin_sum;
        swz_put4(sp[slot][ty + 1], 4 + 4 * tx, xc);
        sp[slot][hy_row + 1][swz(4 + hy_x)] = hyv;
        if (tid < 16) sp[slot][hx_y + 1][swz(4 + hx_col)] = hxv;
        __syncthreads();
        const double wzk = L.wz[kk], rwz = L.rwz[kk], czm = L.cmz[kk], czp = L.cpz[kk];
        v4 out;
        if ((MODE == 2 || MODE == 8) && (czm != key_zm || czp != key_zp)) {  // (workgroup-uniform)
            key_zm = czm, key_zp = czp;
#pragma unroll
            for (int c = 0; c < 4; ++c) wr[c] = jweight(omega, fdiag(q4[c], czm, czp));
        }
        // in-plane neighbours: the rows above and below as aligned pieces, the cells left and right of the thread's four (its own
        // values from the registers)
        const v4 ylo = swz_get4(sp[slot][ty], 4 + 4 * tx), yhi = swz_get4(sp[slot][ty + 2], 4 + 4 * tx);
        const double xleft = sp[slot][ty + 1][swz(3 + 4 * tx)], xright = sp[slot][ty + 1][swz(8 + 4 * tx)];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const FCell &q = q4[c];
            const double xcc = xc[c];
            const double nb_l = (c == 0) ? xleft : xc[c > 0 ? c - 1 : 0], nb_r = (c == 3) ? xright : xc[c < 3 ? c + 1 : 0];
            if (MODE == 2 || MODE == 8) {
                double t = (bv[c] * q.rxy) * rwz;
                t = nacc(t, q.cxm, nb_l);
                t = nacc(t, q.cxp, nb_r);
                t = nacc(t, q.cym, ylo[c]);
                t = nacc(t, q.cyp, yhi[c]);
                t = nacc(t, czm, zm[c]);
                t = nacc(t, czp, zp[c]);
                out[c] = jrelax(xcc, omc, wr[c], t);
                if (MODE == 8 && lk >= dlo && lk < dhi) {  // the sums cover the OWNED planes only
                    acc0 += out[c] * braw[c];
                    acc1 += out[c] * out[c];
                    acc2 += out[c];
                }
                continue;
            }
            double sum = 0.0;
            sum = facc(sum, q.cxm, nb_l, xcc);
            sum = facc(sum, q.cxp, nb_r, xcc);
            sum = facc(sum, q.cym, ylo[c], xcc);
            sum = facc(sum, q.cyp, yhi[c], xcc);
            sum = facc(sum, czm, zm[c], xcc);
            sum = facc(sum, czp, zp[c], xcc);
            if (MODE == 0) {  // y = A x (the Krylov product of the stencil twin), x.y over the owned planes on request
                out[c] = (sum * q.vxy) * wzk;
                if (part != nullptr && lk >= dlo && lk < dhi) acc0 += out[c] * xcc;
            } else
                out[c] = resid(bv[c], sum * q.vxy, wzk);
        }
        *reinterpret_cast<v4 *>(xo + (int64_t)lk * plane + off_c) = out;
        zm = xc;
        xc = zp;
        zp = zq;
        hyv = hyn;
        hxv = hxn;
        bv = bn;
    }
    if (MODE == 0 && part != nullptr) {
        // one partial per workgroup; the slots up to part_stride that no workgroup owns are zeroed (the consumer sums a
        // fixed number of them)
        __shared__ double sh0[4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc0 += __shfl_down(acc0, o, 64);
        if (lane == 0) sh0[w] = acc0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int64_t nwg = (int64_t)gridDim.x * gridDim.y * gridDim.z;
            const int64_t blk = ((int64_t)tb.z * gridDim.y + tb.y) * gridDim.x + tb.x;
            part[blk] = (sh0[0] + sh0[1]) + (sh0[2] + sh0[3]);
            for (int64_t e = blk + nwg; e < part_stride; e += nwg) part[e] = 0.0;
        }
    }
    if (MODE == 8) {
        __shared__ double sh[3][4];
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
            const int k2 = threadIdx.x;
            const int64_t blk = ((int64_t)tb.z * gridDim.y + tb.y) * gridDim.x + tb.x;
            part[(int64_t)k2 * part_stride + blk] = (sh[k2][0] + sh[k2][1]) + (sh[k2][2] + sh[k2][3]);
        }
    }
}

// the per-workgroup partials of slot k = blockIdx.y (up to 5 * 10^5 of them) in two fixed-order stages: 64 workgroups
// per slot sum a contiguous chunk each, one workgroup per slot sums the 64 results into S->red[k]
constexpr int BIG_STAGE = 64;
__global__ __launch_bounds__(256) void k_reduce_big(const Scalars *__restrict__ S, const double *__restrict__ part, int stride,
                                                    int count, double *__restrict__ out /* [3][BIG_STAGE] */)
{
    if (S->done) return;
    const double *p = part + (int64_t)blockIdx.y * stride;
    const int chunk = (count + BIG_STAGE - 1) / BIG_STAGE;
    const int lo = blockIdx.x * chunk, hi = min(lo + chunk, count);
    double v = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) v += p[i];
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.y * BIG_STAGE + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void k_finalize_big(Scalars *__restrict__ S, const double *__restrict__ in)
{
    if (S->done) return;
    double v = in[blockIdx.x * BIG_STAGE + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (threadIdx.x == 0) S->red[blockIdx.x] = v;
}

// 1-D transfer stencil of fine cell s: its parent aggregate (weight 1 - t) and the coarse cell on the child's
// side (weight t = sibling width / (W_parent + W_neighbour); 3/4, 1/4 on a uniform mesh); a lone child, a child
// at a wall and a direction that is not coarsened have oth == par with weights (1, 0).
__device__ __forceinline__ void tr1d(const Tr1 &t, int s, int I[2], double wt[2])
{
    I[0] = t.par[s];
    I[1] = t.oth[s];
    wt[0] = t.wpar[s];
    wt[1] = t.woth[s];
}
}  // namespace pib
