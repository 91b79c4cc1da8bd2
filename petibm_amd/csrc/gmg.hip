// gmg.hip -- K2/K5/K6/K7: matrix-free stencil twin and geometric multigrid for
// the 5/7-point Poisson operator on the stretched Cartesian mesh (gfx950).
//
// Stands in for the algebraic multigrid the reference gets from third parties
// (AmgX CLASSICAL/AGGREGATION AMG: examples/navierstokes/
// liddrivencavity2dRe1000_GPU/config/poisson_solver.info:20-42; PCGAMG / hypre:
// examples/navierstokes/liddrivencavity2dRe100/config/poisson_solver.info:6-9).
// The algorithm is restated operation-for-operation on the CPU in
// oracle/csrc/gmg.c (the parity oracle of these kernels).
//
//   * grid stored (nx, ny, nz) in natural order; 2-D grids are (nx, 1, ny) so
//     the slab (decomposition) axis is always the last one;
//   * level operator = rediscretised FV operator from three 1-D width arrays
//     (16-24 B/row of HBM traffic instead of the CSR's 104 B/row);
//   * selective coarsening per direction (neighbours merge only while their
//     combined width is <= 1.5 hmin 2^(level+1): the stretched far field of a
//     PetIBM mesh waits for the refined region; plain pairing when uniform);
//   * tri-linear cell-centred prolongation with width-based weights from 1-D
//     tables (3/4, 1/4 on a uniform mesh), restriction = P^T,
//     damped-Jacobi V(nu1,nu2), pre-smoothing from a zero guess;
//   * multi-GPU: levels are z-slab distributed (one halo plane, RCCL
//     send/recv; aggregates never straddle a slab boundary) while every rank
//     keeps >= 2 planes and the level is large; below that the
//     level's right-hand side is all-gathered and the remaining levels are
//     solved redundantly on every GPU (no further communication);
//   * null space: CONSTANT leaves z un-projected (the Krylov kernels subtract
//     the mean lazily), PINNED feeds r'[0] = r[0] - sum(r) to the singular
//     operator; both keep the preconditioner symmetric.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

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

// ---- transfer kernels, row form -------------------------------------------------------------------------
// One wave <-> one grid row (fixed j, k: the y / z stencils are wave-uniform, i.e. scalar loads), one lane <-> one
// COARSE cell I of that row and its one or two fine children.  The x neighbours I-1 / I+1 come from the
// neighbouring lanes (__shfl).  Prolongation: consecutive waves overlap by two lanes (62 producing lanes, lanes 0 and
// 63 only feed their neighbours -- single-lane edge loads were most of its vector-memory instructions: 737 -> 566 us);
// restriction: aligned 64-lane chunks whose two edge lanes load their outer neighbour (the overlap measured slower
// there: 740 -> 854 us).  A coarse value is loaded once per row
// instead of three times (prolongation) and a fine value once instead of twice (restriction) -- these kernels
// are bound by the vector-memory issue rate, not by HBM (rocprof r01: 1.2 ms / 0.83 ms per 512^3 launch with
// per-lane table gathers, 0.47 / 0.25 ms of HBM time).  Row groups are dealt to the XCDs in contiguous ranges
// (workgroup b runs on XCD b % 8), so a coarse row is fetched by one L2 only.
// Summation order = the oracle's: z, then y, then x ascending, weights ((wz*wy)*wx); zero weights add exactly 0.
constexpr int ROW_LANES = 62;  // producing lanes per wave; lanes 0 and 63 are the overlap with the neighbouring waves
__device__ __forceinline__ bool row_of_wave(int ngroups, int per_xcd, int nrows, int *row)
{
    const int b = blockIdx.x;
    const int rg = (b & 7) * per_xcd + (b >> 3);
    *row = __builtin_amdgcn_readfirstlane(rg * 4 + (int)threadIdx.y);
    return rg < ngroups && *row < nrows;
}

// xf += P xc.   xc points at the coarse level's first owned plane (coarse k0c); coarse halo planes must be valid
// when the level is distributed.  One wave does RP consecutive fine rows and issues every load of all of them
// (4 coarse rows + the old fine values each) before the first use: a single row per wave left ~3 dependent memory
// round trips of latency per 1 KB written.
template <int RP>
__global__ __launch_bounds__(256) void k_prolong_rows(const Scalars *__restrict__ S, LevelDev F, LevelDev C,
                                                      const double *__restrict__ xc, double *__restrict__ xf,
                                                      int ngroups, int per_xcd, int vec_ok)
{
    if (S != nullptr && S->done) return;
    int row0;
    const int nrows = F.ny * F.nk;
    if (!row_of_wave(ngroups, per_xcd, (nrows + RP - 1) / RP, &row0)) return;
    row0 *= RP;
    const int lane = threadIdx.x;
    const int Iraw = blockIdx.y * ROW_LANES + lane - 1;
    const bool valid = lane >= 1 && lane <= ROW_LANES && Iraw < C.nx;
    const int I = min(max(Iraw, 0), C.nx - 1);
    // the coarse cell whose value this lane holds: across the periodic seam for the two lanes next to the row's ends
    const int Iload = (F.tper & 1) ? (Iraw < 0 ? C.nx - 1 : (Iraw >= C.nx ? min(Iraw - C.nx, C.nx - 1) : Iraw)) : I;
    const int2 fc = F.tx.fc[I];
    const double4 pw = F.tx.pw[I];
    const int64_t cplane = (int64_t)C.nx * C.ny, fplane = (int64_t)F.nx * F.ny;
    const bool vec = vec_ok && __all(!valid || (fc.y == 2 && !(fc.x & 1)));
    double vP[RP][4], w4[RP][4], d0[RP], d1[RP];
    int64_t off[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        const int row = min(row0 + r, nrows - 1);  // a clamped duplicate row is loaded but never stored
        const int kk = row / F.ny, j = row - kk * F.ny, k = F.k0 + kk;
        int J[2], K[2];
        double wj[2], wk[2];
        tr1d(F.t[1], j, J, wj);
        tr1d(F.t[2], k, K, wk);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const double *rowp = xc + (int64_t)C.nx * J[b2] + cplane * (K[c2] - C.k0);
                w4[r][c2 * 2 + b2] = wk[c2] * wj[b2];
                vP[r][c2 * 2 + b2] = rowp[Iload];
            }
        off[r] = (int64_t)kk * fplane + (int64_t)j * F.nx + fc.x;
        d0[r] = d1[r] = 0.0;
        if (valid) {
            if (vec) {
                const double2 v = *reinterpret_cast<const double2 *>(xf + off[r]);
                d0[r] = v.x;
                d1[r] = v.y;
            } else {
                d0[r] = xf[off[r]];
                if (fc.y == 2) d1[r] = xf[off[r] + 1];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        double sl = 0.0, sr = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double wkj = w4[r][q];
            const double v = vP[r][q];
            const double vL = __shfl_up(v, 1, 64), vR = __shfl_down(v, 1, 64);
            if (wkj == 0.0) continue;  // wave-uniform: the oracle skips zero weights too
            sl = tacc(sl, (wkj * pw.x), v);
            sl = tacc(sl, (wkj * pw.y), vL);
            sr = tacc(sr, (wkj * pw.z), v);
            sr = tacc(sr, (wkj * pw.w), vR);
        }
        if (!valid || row0 + r >= nrows) continue;
        double *dst = xf + off[r];
        if (vec)
            *reinterpret_cast<double2 *>(dst) = make_double2(d0[r] + sl, d1[r] + sr);
        else {
            dst[0] = d0[r] + sl;
            if (fc.y == 2) dst[1] = d1[r] + sr;
        }
    }
}

// ---- prolongation + the first post-smoothing step in one march (fully paired levels that k_level_march serves).  The
// corrected iterate x + P e exists only on chip: a workgroup keeps the coarse planes a fine plane interpolates from in a
// three-slot LDS ring (66 x 6 values each), corrects its 128 x 8 tile of the plane ahead (and that plane's x / y halo
// cells) and relaxes the current plane exactly as k_level_march<2> does -- the iterate is read once and written once
// instead of twice each.  Per-cell sums in the order of k_prolong_rows (z slot, y slot; own coarse cell, then the x
// neighbour) followed by k_level's expression: bit-identical to the two kernels it replaces.
constexpr int PCX = FX / 2 + 2, PCY = FY / 2 + 2;
struct PHalo {
    int r0, r1, lx, lxo;
    double wj0, wj1, wa, wb;
};
// row of coarse row J in a tile that starts at coarse row J0 - 1: a row across the periodic seam sits at the tile's edge
__device__ __forceinline__ int ptile_row(int J, int J0, int ncy)
{
    const int r = J - J0 + 1;
    return r < 0 ? r + ncy : (r >= PCY ? r - ncy : r);
}
// i: the cell's position (-1 and nx are the cells across a periodic seam), iw / j: its indices in the tables
__device__ __forceinline__ PHalo phalo(const LevelDev &F, int ncx, int ncy, int i, int iw, int j, int I0, int J0)
{
    PHalo h;
    int J[2];
    double wj[2];
    tr1d(F.t[1], j, J, wj);
    h.r0 = ptile_row(J[0], J0, ncy);
    h.r1 = ptile_row(J[1], J0, ncy);
    h.wj0 = wj[0];
    h.wj1 = wj[1];
    const int I = i >> 1;
    const double4 pw = F.tx.pw[iw >> 1];
    const bool right = i & 1;
    h.lx = I - I0 + 1;
    h.lxo = right ? h.lx + 1 : h.lx - 1;
    h.wa = right ? pw.z : pw.x;
    h.wb = right ? pw.w : pw.y;
    return h;
}
// DOTS (the only post-smoothing step of level 0 writes z = M^-1 r): the partial sums of k_level_march<8>, same grouping.
template <int DOTS>
__global__ __launch_bounds__(256) PIB_WAVES_ATTR(PIB_WAVES_PROLONG) void k_prolong_smooth(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega,
                                                        const double *__restrict__ b, const double *__restrict__ xc,
                                                        const double *__restrict__ xi, double *__restrict__ xo,
                                                        const double *__restrict__ pin_sum, int FZ, double *__restrict__ part,
                                                        int part_stride, int dlo, int dhi)
{
    if (S != nullptr && S->done) return;
    __shared__ __attribute__((aligned(32))) double sp[2][FSY][SWR];  // (swizzled rows, see k_presmooth2: cells i0 - 4 .. i0 + 131)
    __shared__ __attribute__((aligned(16))) double cs[3][PCY][PCX];
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    // relaxed planes: [F.k0, F.k0 + F.nk) (global) -- a whole level or a run of planes of a z-slab; b / xi / xo point at
    // the first of them, xc at coarse plane C.k0.  The planes one below / above the run are corrected too (they are the
    // z neighbours of the relaxation): the old iterate and the coarse planes they interpolate from must be valid there.
    const Tile3 tb = tile_of_block();
    const int i0 = tb.x * FX, j0 = tb.y * FY, l0 = F.k0 + tb.z * FZ;
    const int I0 = i0 >> 1, J0 = j0 >> 1;
    const int64_t plane = (int64_t)F.nx * F.ny, cplane = (int64_t)C.nx * C.ny;
    b -= (int64_t)F.k0 * plane;  // index by global plane below
    xi -= (int64_t)F.k0 * plane;
    xo -= (int64_t)F.k0 * plane;
    xc -= (int64_t)C.k0 * cplane;
    dlo += F.k0;
    dhi += F.k0;
    const int j = j0 + ty, ic = i0 + 4 * tx;
    const int hy_row = (tid < 128) ? -1 : FY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? FX : -1, hx_y = (tid >> 1) & 7;
    // periodic directions (operator and transfers alike: the caller checks per == tper): the cells beyond the domain are
    // the ones across the seam; a periodic z has the whole level here, plane -1 is plane nz - 1 and coarse plane -1 is
    // coarse plane nzc - 1 (planes are counted through the seam below, `zw` / `Kw` give their place in memory)
    const bool px = F.per & 1, py = F.per & 2, pz = F.per & 4;
    const int hyj = j0 + hy_row, hyi = i0 + hy_x, hxj = j0 + hx_y, hxi = i0 + hx_col;
    const int hyjw = py ? (hyj < 0 ? F.ny - 1 : (hyj >= F.ny ? 0 : hyj)) : hyj;
    const int hxiw = px ? (hxi < 0 ? F.nx - 1 : (hxi >= F.nx ? 0 : hxi)) : hxi;
    const bool hy_ok = hyjw >= 0 && hyjw < F.ny, hx_ok = tid < 16 && hxiw >= 0 && hxiw < F.nx;
    const int64_t off_c = (int64_t)j * F.nx + ic, off_hy = (int64_t)hyjw * F.nx + hyi, off_hx = (int64_t)hxj * F.nx + hxiw;
    FCell q4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = fcell(F, ic + c, j);
    // interpolation data of the own cells (coarse columns I0 + 2 tx, + 1) and of the halo cells
    const double4 pwA = F.tx.pw[I0 + 2 * tx], pwB = F.tx.pw[I0 + 2 * tx + 1];
    int rr[2];
    double wjv[2];
    {
        int J[2];
        tr1d(F.t[1], j, J, wjv);
        rr[0] = ptile_row(J[0], J0, C.ny);
        rr[1] = ptile_row(J[1], J0, C.ny);
    }
    PHalo hy = {}, hx = {};
    if (hy_ok) hy = phalo(F, C.nx, C.ny, hyi, hyi, hyjw, I0, J0);
    if (hx_ok) hx = phalo(F, C.nx, C.ny, hxi, hxiw, hxj, I0, J0);
    auto zw = [&](int k) { return pz ? (k < 0 ? k + F.nzg : (k >= F.nzg ? k - F.nzg : k)) : k; };
    // coarse plane K (tile + one cell around it, zero outside the domain) into its ring slot
    auto stage = [&](int K) {
        const int Kw = pz ? (K < 0 ? K + C.nzg : (K >= C.nzg ? K - C.nzg : K)) : K;
        const double *pc = xc + (int64_t)Kw * cplane;
        double *dst = &cs[(K + 3) % 3][0][0];
        for (int e = tid; e < PCX * PCY; e += 256) {
            const int row = e / PCX, cx = e - row * PCX;
            int I = I0 - 1 + cx, J = J0 - 1 + row;
            if (px) I = I < 0 ? I + C.nx : (I >= C.nx ? I - C.nx : I);
            if (py) J = J < 0 ? J + C.ny : (J >= C.ny ? J - C.ny : J);
            dst[e] = (I >= 0 && I < C.nx && J >= 0 && J < C.ny) ? pc[(int64_t)J * C.nx + I] : 0.0;
        }
    };
    // the old iterate on plane k (own cells, halo cells): loaded one plane ahead of its use, across the barrier
    struct Old {
        v4 c;
        double hy, hx;
    };
    auto fetch = [&](int k, bool halo) -> Old {
        Old o;
        const double *pl = xi + (int64_t)zw(k) * plane;
        o.c = *reinterpret_cast<const v4 *>(pl + off_c);
        o.hy = (halo && hy_ok) ? pl[off_hy] : 0.0;
        o.hx = (halo && hx_ok) ? pl[off_hx] : 0.0;
        return o;
    };
    // x + P e on plane k: the own cells (returned) and, with `halo`, the tile's halo cells -> LDS slot
    auto correct = [&](int k, bool halo, const Old &o) -> v4 {
        int K[2];
        double wk[2];
        tr1d(F.t[2], zw(k), K, wk);
        if (pz) {  // the tables hold the planes' places in memory: count them through the seam like k
            const int Kc = k >> 1;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) K[c2] = K[c2] > Kc + 1 ? K[c2] - C.nzg : (K[c2] < Kc - 1 ? K[c2] + C.nzg : K[c2]);
        }
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, sy = 0.0, sx = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const double(*cp)[PCX] = cs[(K[c2] + 3) % 3];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const double w = wk[c2] * wjv[b2];
                const double *row = &cp[rr[b2]][2 * tx];
                const double2 v01 = *reinterpret_cast<const double2 *>(row), v23 = *reinterpret_cast<const double2 *>(row + 2);
                s0 = tacc(s0, (w * pwA.x), v01.y);
                s0 = tacc(s0, (w * pwA.y), v01.x);
                s1 = tacc(s1, (w * pwA.z), v01.y);
                s1 = tacc(s1, (w * pwA.w), v23.x);
                s2 = tacc(s2, (w * pwB.x), v23.x);
                s2 = tacc(s2, (w * pwB.y), v01.y);
                s3 = tacc(s3, (w * pwB.z), v23.x);
                s3 = tacc(s3, (w * pwB.w), v23.y);
                if (halo) {
                    const double wy = wk[c2] * (b2 ? hy.wj1 : hy.wj0);
                    const double *rowy = cp[b2 ? hy.r1 : hy.r0];
                    sy = tacc(sy, (wy * hy.wa), rowy[hy.lx]);
                    sy = tacc(sy, (wy * hy.wb), rowy[hy.lxo]);
                    if (tid < 16) {
                        const double wx = wk[c2] * (b2 ? hx.wj1 : hx.wj0);
                        const double *rowx = cp[b2 ? hx.r1 : hx.r0];
                        sx = tacc(sx, (wx * hx.wa), rowx[hx.lx]);
                        sx = tacc(sx, (wx * hx.wb), rowx[hx.lxo]);
                    }
                }
            }
        }
        v4 out;
        out[0] = o.c[0] + s0;
        out[1] = o.c[1] + s1;
        out[2] = o.c[2] + s2;
        out[3] = o.c[3] + s3;
        if (halo) {
            const int slot = k & 1;
            swz_put4(sp[slot][ty + 1], 4 + 4 * tx, out);
            sp[slot][hy_row + 1][swz(4 + hy_x)] = hy_ok ? o.hy + sy : 0.0;
            if (tid < 16) sp[slot][hx_y + 1][swz(4 + hx_col)] = hx_ok ? o.hx + sx : 0.0;
        }
        return out;
    };
    const int lend = min(l0 + FZ, F.k0 + F.nk);
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    // prologue: the coarse planes under l0 - 1 and l0 (K0 - 1 and K0 for an even l0 = 2 K0, K0 - 1 .. K0 + 1 for an odd
    // l0 = 2 K0 + 1: three distinct ring slots), then those two corrected planes
    for (int K = pz ? (l0 - 2) >> 1 : max((l0 - 2) >> 1, 0); K <= (pz ? (l0 + 1) >> 1 : min((l0 + 1) >> 1, C.nzg - 1)); ++K) stage(K);
    v4 zm = {0, 0, 0, 0}, xcur, zp = {0, 0, 0, 0};
    Old om = {}, o0 = fetch(l0, true), on = {};
    if (l0 > 0 || pz) om = fetch(l0 - 1, false);
    if (l0 + 1 < F.nzg || pz) on = fetch(l0 + 1, l0 + 1 < lend);
    v4 bc = *reinterpret_cast<const v4 *>(b + (int64_t)l0 * plane + off_c), bn = {0, 0, 0, 0};
    __syncthreads();
    if (l0 > 0 || pz) zm = correct(l0 - 1, false, om);
    xcur = correct(l0, true, o0);
    // wr = omega / d of the thread's cells, divided again only when a plane's z coefficients differ from the previous plane's
    const double omc = 1.0 - omega;
    v4 wr = {0, 0, 0, 0};
    double key_zm = __builtin_nan(""), key_zp = __builtin_nan("");
    for (int lk = l0; lk < lend; ++lk) {
        const int slot = lk & 1;
        const int kn = lk + 1;
        // loads for the next step go out before the barrier: the old iterate two planes ahead, b one plane ahead
        Old o2 = {};
        if ((kn + 1 < F.nzg || pz) && kn < lend) o2 = fetch(kn + 1, kn + 1 < lend);
        if (kn < lend) bn = *reinterpret_cast<const v4 *>(b + (int64_t)kn * plane + off_c);
        if ((kn & 1) && kn < F.nzg && ((kn + 1) / 2 < C.nzg || pz)) stage((kn + 1) / 2);  // an odd plane reaches up to the next coarse plane
        __syncthreads();
        if (kn < F.nzg || pz) zp = correct(kn, kn < lend, on);
        v4 bv = bc;
        const v4 braw = bc;
        if (pin_sum != nullptr && lk == 0 && off_c == 0) bv[0] = bv[0] - *pin_sum;
        const double rwz = F.rwz[lk], czm = F.cmz[lk], czp = F.cpz[lk];
        v4 out;
        if (czm != key_zm || czp != key_zp) {  // (workgroup-uniform)
            key_zm = czm, key_zp = czp;
#pragma unroll
            for (int c = 0; c < 4; ++c) wr[c] = jweight(omega, fdiag(q4[c], czm, czp));
        }
        const v4 ylo = swz_get4(sp[slot][ty], 4 + 4 * tx), yhi = swz_get4(sp[slot][ty + 2], 4 + 4 * tx);
        const double xleft = sp[slot][ty + 1][swz(3 + 4 * tx)], xright = sp[slot][ty + 1][swz(8 + 4 * tx)];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const FCell &q = q4[c];
            const double xcc = xcur[c];
            const double nb_l = (c == 0) ? xleft : xcur[c > 0 ? c - 1 : 0], nb_r = (c == 3) ? xright : xcur[c < 3 ? c + 1 : 0];
            double t = (bv[c] * q.rxy) * rwz;
            t = nacc(t, q.cxm, nb_l);
            t = nacc(t, q.cxp, nb_r);
            t = nacc(t, q.cym, ylo[c]);
            t = nacc(t, q.cyp, yhi[c]);
            t = nacc(t, czm, zm[c]);
            t = nacc(t, czp, zp[c]);
            out[c] = jrelax(xcc, omc, wr[c], t);
            if (DOTS && lk >= dlo && lk < dhi) {  // the sums cover the OWNED planes only
                acc0 += out[c] * braw[c];
                acc1 += out[c] * out[c];
                acc2 += out[c];
            }
        }
        *reinterpret_cast<v4 *>(xo + (int64_t)lk * plane + off_c) = out;
        zm = xcur;
        xcur = zp;
        on = o2;
        bc = bn;
    }
    if (DOTS) {
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

// ---- prolongation + BOTH post-smoothing steps in one march (V(.,2) on the levels k_prolong_smooth serves, whole on this
// rank).  As two kernels (k_prolong_smooth, k_level_march<2 / 8>) the once-smoothed iterate goes to HBM and comes back and b
// is read twice: 49 B per fine cell; here 25 (b, the old iterate and the coarse values read once, the result written once).
// The scheme of k_resid_restrict_march, one stage deeper: a workgroup's region is its 128 x 8 tile and two cells around it, in
// aligned 4-cell pieces -- every thread owns its tile piece (the cells k_level_march gives it: the Krylov sums keep their
// grouping and their bits) and, 152 of the threads, one piece of the margin; a piece's z neighbours are the thread's
// registers.  Per fine plane k: the corrected iterate x + P e of plane k + 2 on the whole region, the first step of plane
// k + 1 on the region less its outer ring (in-plane neighbours from the LDS copy of the corrected plane), the second step of
// plane k on the tile (neighbours from the LDS copy of the first step's plane).  Same expressions in the same order as the
// two kernels: the same bits.
// The tile is 128 x 16 -- two of k_level_march's tiles, 512 threads: the margins are 1.33 x / 1.2 x the tile's work instead of
// 1.6 x / 1.33 x with eight rows, and the kernel is bound by instruction issue -- and each half of the workgroup forms the
// Krylov sums of its own 128 x 8 tile.
constexpr int UTY = 2 * FY, UNT = 32 * UTY;                 // tile rows, threads
constexpr int UY = UTY + 4, UX = FX + 8, UPR = UX / 4;      // region rows, columns, pieces per row
constexpr int UCX = FX / 2 + 8, UCY = UTY / 2 + 4;          // the coarse planes' tile: columns I0 - 3 .. I0 + 68, rows J0 - 2 .. J0 + UTY / 2 + 1
constexpr int UMARGIN = 4 * UPR + 2 * UTY;                  // pieces of the margin
template <int DOTS>
__global__ __launch_bounds__(UNT) void k_prolong_smooth2(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega,
                                                         const double *__restrict__ b, const double *__restrict__ xc,
                                                         const double *__restrict__ xi, double *__restrict__ xo, int FZ,
                                                         double *__restrict__ part, int part_stride, int dlo, int dhi,
                                                         const double *__restrict__ pin_sum = nullptr)
{
    if (S != nullptr && S->done) return;
    // x + P e on the plane the first step works on, the first step's result on the plane the second works on: two copies each
    // (one read, one written per iteration: a single barrier)
    __shared__ __attribute__((aligned(32))) double XP[2][UY][SWR];  // (swizzled rows: cell X of a row at swz(X))
    __shared__ __attribute__((aligned(32))) double S1[2][UY][SWR];
    __shared__ __attribute__((aligned(16))) double cs[3][UCY][UCX];
    __shared__ __attribute__((aligned(16))) double tcx[3][SWR];  // cm, cp, 1/w of the region's columns (swizzled like the rows) ...
    __shared__ double tcy[3][UY];                                // ... and of its rows
    // x interpolation weights of a piece's two coarse columns (I0 - 3 + q0, q0 + 1 with q0 = 2 p + 1 for piece p of a row), as four
    // 16-byte chunks per piece, chunk by chunk: a lane's reads have a 16-byte stride (a double4 per coarse column had 64: 4-way conflicts)
    __shared__ __attribute__((aligned(16))) double pwc[4][UPR][2];
    __shared__ double tyw[2][UY];                                // y interpolation weights of the region's rows
    __shared__ int tyr[2][UY];                                   // ... and the coarse tile's rows they apply to
    // The per-plane entries of the z tables (interpolation weights, 1 / w, the two face coefficients) of the planes this
    // workgroup touches, staged once: inside the march the compiler reads such a (workgroup-uniform) entry through the vector
    // path -- the kernel stores to global memory, so no scalar load -- and waits for it on the spot: three memory round trips at
    // the head of the three stages of EVERY plane.  From LDS it is a broadcast read.  Entry e <-> plane l0 - 4 + e.
    constexpr int ZT = 144;  // >= planes per workgroup + 6
    __shared__ double tz[5][ZT];
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const Tile3 tb = tile_of_block();
    // relaxed planes: [F.k0, F.k0 + F.nk) (global) -- a whole level or a run of planes of a z-slab (its own planes and the
    // ghost planes the caller wants the result on); b / xi / xo point at the first of them, xc at coarse plane C.k0.  The two
    // planes below / above a workgroup's planes are corrected and the one next to them relaxed once as well: the old iterate
    // and the coarse planes they interpolate from, and b one plane out, must be valid there.
    const int i0 = tb.x * FX, j0 = tb.y * UTY, l0 = F.k0 + tb.z * FZ, lend = min(l0 + FZ, F.k0 + F.nk);
    const int I0 = i0 >> 1, J0 = j0 >> 1;
    const int64_t plane = (int64_t)F.nx * F.ny, cplane = (int64_t)C.nx * C.ny;
    b -= (int64_t)F.k0 * plane;  // index by global plane below
    xi -= (int64_t)F.k0 * plane;
    xo -= (int64_t)F.k0 * plane;
    xc -= (int64_t)C.k0 * cplane;
    const bool px = F.per & 1, py = F.per & 2, pz = F.per & 4;  // (operator and transfers wrap alike: the caller checks)
    for (int e = tid; e < ZT; e += UNT) {
        const int kp = l0 - 4 + e;
        const bool in = pz || (kp >= 0 && kp < F.nzg);
        const int kw = pz ? (kp < 0 ? kp + F.nzg : (kp >= F.nzg ? kp - F.nzg : kp)) : kp;
        const bool use = in && e < FZ + 6;
        tz[0][e] = use ? F.t[2].wpar[kw] : 0.0;
        tz[1][e] = use ? F.t[2].woth[kw] : 0.0;
        tz[2][e] = use ? F.rwz[kw] : 0.0;
        tz[3][e] = use ? F.cmz[kw] : 0.0;
        tz[4][e] = use ? F.cpz[kw] : 0.0;
    }
    const int zt0 = l0 - 4;
    // ---- tables of the region
    for (int e = tid; e < UX; e += UNT) {
        int gi = i0 - 4 + e;
        if (px) gi = gi < 0 ? gi + F.nx : (gi >= F.nx ? gi - F.nx : gi);
        const bool in = gi >= 0 && gi < F.nx;
        tcx[0][swz(e)] = in ? F.cmx[gi] : 0.0;
        tcx[1][swz(e)] = in ? F.cpx[gi] : 0.0;
        tcx[2][swz(e)] = in ? F.rwx[gi] : 0.0;
    }
    if (tid < UY) {
        const int gu = j0 - 2 + tid;
        int gj = gu;
        if (py) gj = gj < 0 ? gj + F.ny : (gj >= F.ny ? gj - F.ny : gj);
        const bool in = gj >= 0 && gj < F.ny;
        tcy[0][tid] = in ? F.cmy[gj] : 0.0;
        tcy[1][tid] = in ? F.cpy[gj] : 0.0;
        tcy[2][tid] = in ? F.rwy[gj] : 0.0;
        // the two coarse rows the row interpolates from: its parent and the coarse row on the child's side (pairs: the parent
        // of fine row g is g >> 1), as rows of the coarse tile; weights from the level's table
        const int par = gu >> 1, oth = (gu & 1) ? par + 1 : par - 1;
        tyr[0][tid] = min(max(par - (J0 - 2), 0), UCY - 1);
        tyr[1][tid] = min(max(oth - (J0 - 2), 0), UCY - 1);
        tyw[0][tid] = in ? F.t[1].wpar[gj] : 0.0;
        tyw[1][tid] = in ? F.t[1].woth[gj] : 0.0;
    }
    for (int e = tid; e < 2 * UPR; e += UNT) {
        const int pp = e >> 1, q = 2 * pp + 1 + (e & 1);  // piece, coarse column of the tile
        int I = I0 - 3 + q;
        if (px) I = I < 0 ? I + C.nx : (I >= C.nx ? I - C.nx : I);
        const double4 w4 = (I >= 0 && I < C.nx) ? F.tx.pw[I] : make_double4(0.0, 0.0, 0.0, 0.0);
        pwc[2 * (e & 1)][pp][0] = w4.x;
        pwc[2 * (e & 1)][pp][1] = w4.y;
        pwc[2 * (e & 1) + 1][pp][0] = w4.z;
        pwc[2 * (e & 1) + 1][pp][1] = w4.w;
    }
    // ---- the thread's pieces: 0 the tile piece, 1 a piece of the margin (threads 0 .. UMARGIN - 1)
    int prow[2], pcol[2];
    int64_t goff[2];
    bool ok[2], has[2], first[2];
    prow[0] = 2 + ty;
    pcol[0] = 4 + 4 * tx;
    has[0] = true;
    has[1] = tid < UMARGIN;
    {
        const int h = tid;
        if (h < 4 * UPR) {
            const int r4 = h / UPR;
            prow[1] = r4 < 2 ? r4 : UTY + r4;  // rows 0, 1, UTY + 2, UTY + 3
            pcol[1] = 4 * (h - r4 * UPR);
        } else {
            const int q2 = h - 4 * UPR;
            prow[1] = 2 + (q2 >> 1);
            pcol[1] = (q2 & 1) ? UX - 4 : 0;
        }
        if (!has[1]) prow[1] = 0, pcol[1] = 0;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int gi = i0 - 4 + pcol[e], gj = j0 - 2 + prow[e];
        if (px) gi = gi < 0 ? gi + F.nx : (gi >= F.nx ? gi - F.nx : gi);
        if (py) gj = gj < 0 ? gj + F.ny : (gj >= F.ny ? gj - F.ny : gj);
        ok[e] = has[e] && gi >= 0 && gi < F.nx && gj >= 0 && gj < F.ny;
        first[e] = ok[e] && prow[e] >= 1 && prow[e] <= UTY + 2;  // carries the first step (the region less its outer rows)
        goff[e] = (int64_t)gj * F.nx + gi;
    }
    const v4 zero = {0, 0, 0, 0};
    auto zw = [&](int k) { return pz ? (k < 0 ? k + F.nzg : (k >= F.nzg ? k - F.nzg : k)) : k; };
    auto inz = [&](int k) { return pz || (k >= 0 && k < F.nzg); };
    // coarse plane K (its tile, zero outside the domain) into its ring slot
    auto stage = [&](int K) {
        const bool kin = pz || (K >= 0 && K < C.nzg);
        const int Kw = pz ? (K < 0 ? K + C.nzg : (K >= C.nzg ? K - C.nzg : K)) : K;
        const double *pc = xc + (int64_t)(kin ? Kw : 0) * cplane;
        double *dst = &cs[((K % 3) + 3) % 3][0][0];
        for (int e = tid; e < UCX * UCY; e += UNT) {
            const int row = e / UCX, cx = e - row * UCX;
            int I = I0 - 3 + cx, J = J0 - 2 + row;
            if (px) I = I < 0 ? I + C.nx : (I >= C.nx ? I - C.nx : I);
            if (py) J = J < 0 ? J + C.ny : (J >= C.ny ? J - C.ny : J);
            dst[e] = (kin && I >= 0 && I < C.nx && J >= 0 && J < C.ny) ? pc[(int64_t)J * C.nx + I] : 0.0;
        }
    };
    // the same in two halves, for the march: the values are requested at the top of an iteration and go to the ring slot at its end
    // (as one piece the loads were waited for on the spot -- a memory round trip at the head of every other plane)
    auto stage_load = [&](int K, double cv[2]) {
        const bool kin = pz || (K >= 0 && K < C.nzg);
        const int Kw = pz ? (K < 0 ? K + C.nzg : (K >= C.nzg ? K - C.nzg : K)) : K;
        const double *pc = xc + (int64_t)(kin ? Kw : 0) * cplane;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * UNT;
            const int row = e / UCX, cx = e - row * UCX;
            int I = I0 - 3 + cx, J = J0 - 2 + row;
            if (px) I = I < 0 ? I + C.nx : (I >= C.nx ? I - C.nx : I);
            if (py) J = J < 0 ? J + C.ny : (J >= C.ny ? J - C.ny : J);
            cv[u] = (e < UCX * UCY && kin && I >= 0 && I < C.nx && J >= 0 && J < C.ny) ? pc[(int64_t)J * C.nx + I] : 0.0;
        }
    };
    auto stage_store = [&](int K, const double cv[2]) {
        double *dst = &cs[((K % 3) + 3) % 3][0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + u * UNT < UCX * UCY) dst[tid + u * UNT] = cv[u];
    };
    static_assert(UCX * UCY <= 2 * UNT, "two coarse values per thread");
    auto fetch = [&](const double *v, int k, const bool *which, v4 out[2]) {
        const bool in = inz(k);
        const double *pl = v + (int64_t)zw(k) * plane;
#pragma unroll
        for (int e = 0; e < 2; ++e) out[e] = (in && which[e]) ? *reinterpret_cast<const v4 *>(pl + goff[e]) : zero;
    };
    // PINNED (level 0): the effective right-hand side at global cell 0 is b[0] - *pin_sum; cell 0 is the first cell of an aligned
    // piece -- the tile piece of workgroup (0, 0), a margin piece of its neighbours (across the seam on a periodic level)
    auto pin_b = [&](int k, v4 bv[2]) {
        if (pin_sum == nullptr || !inz(k) || zw(k) != 0) return;
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (first[e] && goff[e] == 0) bv[e][0] = bv[e][0] - *pin_sum;
    };
    // x + P e of the thread's pieces on plane k (old: the old iterate there): own coarse cell, then the x neighbour, per
    // z slot and y slot -- the order of k_prolong_rows / k_prolong_smooth
    auto correct = [&](int k, const v4 old[2], v4 out[2]) {
        const bool in = inz(k);
        const int Kp = k >> 1, Ko = (k & 1) ? Kp + 1 : Kp - 1;
        const double wk[2] = {uniform(tz[0][k - zt0]), uniform(tz[1][k - zt0])};  // (zero outside the domain)
        const int Ks[2] = {((Kp % 3) + 3) % 3, ((Ko % 3) + 3) % 3};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!(in && ok[e])) {
                out[e] = zero;
                continue;
            }
            const int R = prow[e], q0 = (pcol[e] >> 1) + 1;  // the piece's first coarse column in the tile (I0 - 3 + q0)
            const int pp = pcol[e] >> 2;
            const swv2 wA0 = *reinterpret_cast<const swv2 *>(pwc[0][pp]), wA1 = *reinterpret_cast<const swv2 *>(pwc[1][pp]),
                       wB0 = *reinterpret_cast<const swv2 *>(pwc[2][pp]), wB1 = *reinterpret_cast<const swv2 *>(pwc[3][pp]);
            const double4 pwA = make_double4(wA0[0], wA0[1], wA1[0], wA1[1]), pwB = make_double4(wB0[0], wB0[1], wB1[0], wB1[1]);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const double w = wk[c2] * tyw[b2][R];
                    const double *row = &cs[Ks[c2]][tyr[b2][R]][q0 - 1];
                    const double v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3];
                    s0 = tacc(s0, (w * pwA.x), v1);
                    s0 = tacc(s0, (w * pwA.y), v0);
                    s1 = tacc(s1, (w * pwA.z), v1);
                    s1 = tacc(s1, (w * pwA.w), v2);
                    s2 = tacc(s2, (w * pwB.x), v2);
                    s2 = tacc(s2, (w * pwB.y), v1);
                    s3 = tacc(s3, (w * pwB.z), v2);
                    s3 = tacc(s3, (w * pwB.w), v3);
                }
            }
            out[e][0] = old[e][0] + s0;
            out[e][1] = old[e][1] + s1;
            out[e][2] = old[e][2] + s2;
            out[e][3] = old[e][3] + s3;
        }
    };
    // one damped-Jacobi step of piece e on a plane: centre values cc, z neighbours zm / zp, in-plane neighbours from `pl`
    const double omc = 1.0 - omega;
    // wr = omega / d of piece e's cells on a plane with the z coefficients czm, czp
    auto weights = [&](int e, double czm, double czp) -> v4 {
        const int R = prow[e], X = pcol[e];
        const double cym = tcy[0][R], cyp = tcy[1][R];
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double cxm = tcx[0][swz(X + c)], cxp = tcx[1][swz(X + c)];
            const double s4 = ((cxm + cxp) + cym) + cyp;
            out[c] = jweight(omega, -((s4 + czm) + czp));
        }
        return out;
    };
    auto step = [&](int e, const double (*pl)[SWR], const v4 &cc, const v4 &zm, const v4 &zp, const v4 &bv, double rwz, double czm,
                    double czp, const v4 &wr) -> v4 {
        const int R = prow[e], X = pcol[e];
        const double cym = tcy[0][R], cyp = tcy[1][R], rwy = tcy[2][R];
        const v4 ylo = swz_get4(pl[R - 1], X), yhi = swz_get4(pl[R + 1], X);
        const v4 cxm4 = swz_get4(tcx[0], X), cxp4 = swz_get4(tcx[1], X), rwx4 = swz_get4(tcx[2], X);
        const double xleft = X > 0 ? pl[R][swz(X - 1)] : 0.0, xright = X + 4 < UX ? pl[R][swz(X + 4)] : 0.0;
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double xcc = cc[c];
            const double left = (c == 0) ? xleft : cc[c > 0 ? c - 1 : 0];
            const double right = (c == 3) ? xright : cc[c < 3 ? c + 1 : 0];
            const double cxm = cxm4[c], cxp = cxp4[c];
            double t = (bv[c] * (rwx4[c] * rwy)) * rwz;
            t = nacc(t, cxm, left);
            t = nacc(t, cxp, right);
            t = nacc(t, cym, ylo[c]);
            t = nacc(t, cyp, yhi[c]);
            t = nacc(t, czm, zm[c]);
            t = nacc(t, czp, zp[c]);
            out[c] = jrelax(xcc, omc, wr[c], t);
        }
        return out;
    };
    // ---- the march.  Iteration k: x + P e of plane k + 2, first step of plane k + 1, second step of plane k; four
    // iterations ahead of the first owned plane fill the pipeline.
    v4 xpm[2] = {zero, zero}, xpc[2] = {zero, zero}, xpn[2];       // x + P e on the planes k, k + 1 (k + 2: xpn)
    v4 s1m[2] = {zero, zero}, s1c[2] = {zero, zero}, s1n[2];       // first step on the planes k - 1, k (k + 1: s1n)
    v4 bcur[2] = {zero, zero}, bnext[2], anext[2], a2[2];
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    // wr of the pieces on the plane the first step works on (w1) and of the tile piece on the plane the second works on (w2):
    // divided again only when a plane's z coefficients differ from those of the plane the weights were formed for
    v4 w1[2] = {zero, zero}, w2 = zero;
    double key1_zm = __builtin_nan(""), key1_zp = __builtin_nan(""), key2_zm = __builtin_nan(""), key2_zp = __builtin_nan("");
    int staged = ((l0 - 2) >> 1) - 2;  // highest coarse plane in the ring
    auto need_stage = [&](int k) {  // the coarse planes plane k interpolates from: k >> 1 and the one above (odd k) or below
        const int hi = (k & 1) ? (k >> 1) + 1 : (k >> 1);
        while (staged < hi) stage(++staged);
    };
    need_stage(l0 - 2);
    fetch(xi, l0 - 2, ok, anext);
    fetch(b, l0 - 3, first, bnext);
    pin_b(l0 - 3, bnext);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending on entry either (see the wait inside the march)
    for (int k = l0 - 4; k < lend; ++k) {
        const bool do1 = k + 1 >= l0 - 1 && k + 1 <= lend && inz(k + 1), do2 = k >= l0;
        // loads of the next iteration go out first: the coarse plane the plane after next reaches (into the ring slot no plane
        // of this iteration reads), the old iterate three planes ahead, b two
#pragma unroll
        for (int e = 0; e < 2; ++e) a2[e] = anext[e];
        const v4 b1[2] = {bnext[0], bnext[1]};  // b of plane k + 1
        const int want = ((k + 3) & 1) ? ((k + 3) >> 1) + 1 : ((k + 3) >> 1);  // (need_stage(k + 3): at most one plane per iteration)
        const bool staging = staged < want;
        double cv[2] = {0.0, 0.0};
        if (staging) stage_load(++staged, cv);
        if (k + 1 < lend) {
            fetch(xi, k + 3, ok, anext);
            fetch(b, k + 2, first, bnext);
            pin_b(k + 2, bnext);
        }
        correct(k + 2, a2, xpn);
        const int cur = k & 1, nxt = cur ^ 1;
        if (do1) {
            const double rwz = uniform(tz[2][k + 1 - zt0]), czm = uniform(tz[3][k + 1 - zt0]), czp = uniform(tz[4][k + 1 - zt0]);
            if (czm != key1_zm || czp != key1_zp) {  // (workgroup-uniform)
                key1_zm = czm, key1_zp = czp;
#pragma unroll
                for (int e = 0; e < 2; ++e) w1[e] = first[e] ? weights(e, czm, czp) : zero;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) s1n[e] = first[e] ? step(e, XP[cur], xpc[e], xpm[e], xpn[e], b1[e], rwz, czm, czp, w1[e]) : zero;
        } else {
            s1n[0] = s1n[1] = zero;
        }
        // Loads and stores share one counter (vmcnt) and may complete out of order with each other: the wait for the planes
        // requested at the top of this iteration, which the compiler would place at the top of the NEXT one, would also wait for
        // the store below -- issued a few instructions earlier, a full write latency on every plane.  Waiting HERE (on every
        // path: the builtin, which the compiler's own wait insertion takes into account), where those loads are long done, lets the
        // store fly during the whole next iteration.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (do2) {
            const double rwz = uniform(tz[2][k - zt0]), czm = uniform(tz[3][k - zt0]), czp = uniform(tz[4][k - zt0]);
            if (czm != key2_zm || czp != key2_zp) {
                key2_zm = czm, key2_zp = czp;
                w2 = weights(0, czm, czp);
            }
            const v4 out = step(0, S1[cur], s1c[0], s1m[0], s1n[0], bcur[0], rwz, czm, czp, w2);
            *reinterpret_cast<v4 *>(xo + (int64_t)k * plane + goff[0]) = out;
            if (DOTS && k >= dlo && k < dhi) {
                // (z.r takes the UNMODIFIED residual, as k_level<8>'s braw: at the pinned cell that is 0 = (0 - sum) + sum exactly)
                v4 br = bcur[0];
                if (pin_sum != nullptr && k == 0 && goff[0] == 0) br[0] = br[0] + *pin_sum;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc0 += out[c] * br[c];
                    acc1 += out[c] * out[c];
                    acc2 += out[c];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (has[e]) {
                swz_put4(XP[nxt][prow[e]], pcol[e], xpn[e]);
                swz_put4(S1[nxt][prow[e]], pcol[e], s1n[e]);
            }
            xpm[e] = xpc[e];
            xpc[e] = xpn[e];
            s1m[e] = s1c[e];
            s1c[e] = s1n[e];
            bcur[e] = b1[e];
        }
        if (staging) stage_store(staged, cv);
        __syncthreads();
    }
    if (DOTS) {
        // one partial per 128 x 8 tile of k_level_march<8> (waves 0-3: the upper, 4-7: the lower one), summed as there
        __shared__ double sh[2][3][4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        double v[3] = {acc0, acc1, acc2};
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v[k2] += __shfl_down(v[k2], o, 64);
            if (lane == 0) sh[w >> 2][k2][w & 3] = v[k2];
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int half = threadIdx.x / 3, k2 = threadIdx.x - 3 * half;
            const int64_t blk = ((int64_t)tb.z * (2 * gridDim.y) + 2 * tb.y + half) * gridDim.x + tb.x;
            part[(int64_t)k2 * part_stride + blk] = (sh[half][k2][0] + sh[half][k2][1]) + (sh[half][k2][2] + sh[half][k2][3]);
        }
    }
}

// The restriction is summed direction by direction (round 4, second half): x within a fine row, y over the rows of a plane, z over
// the planes --
//     t = ((wx0 r0 + wx1 r1) + wx2 r2) + wx3 r3 ;  u = sum_y wy t ;  s = sum_z wz u      (every + an fma: tacc)
// -- 84 fused multiply-adds per coarse cell instead of 64 and 80 weight products, and in the marching kernels the row sums t are
// shared by the two coarse planes and the two coarse rows a fine row feeds (the restriction was more than half of
// k_resid_restrict_march's arithmetic).  Same order in every kernel and in oracle/csrc/gmg.c:restrict_t.
__device__ __forceinline__ double rsum_x(const double4 &rw, double vl, double c0, double c1, double vr)
{
    return tacc(tacc(tacc(tacc(0.0, rw.x, vl), rw.y, c0), rw.z, c1), rw.w, vr);
}
// one fine plane's share of the coarse cells (I, J), (I, J + 1) of a marching kernel: the six fine rows' x sums, the two coarse rows'
// y sums, then the plane's weight towards the lower (slots 2 / 3) and the upper (slots 0 / 1) coarse plane
__device__ __forceinline__ void restrict_plane(const double4 &rw, const double (&wj)[2][4], const double (&vl)[6], const double (&c0)[6],
                                               const double (&c1)[6], const double (&vr)[6], bool dolo, double wklo, bool dohi, double wkhi,
                                               double (&lo)[2], double (&hi)[2])
{
    double t[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) t[r] = rsum_x(rw, vl[r], c0[r], c1[r], vr[r]);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        double u = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) u = tacc(u, wj[a][b2], t[2 * a + b2]);
        if (dolo) lo[a] = tacc(lo[a], wklo, u);
        if (dohi) hi[a] = tacc(hi[a], wkhi, u);
    }
}

// 1-D restriction stencil of coarse cell I in fixed 4-slot form: slot o <-> fine cell fst[I] - 1 + o (the left
// neighbour, the one or two children, the right neighbour) with the weight that cell gives to I (0 where there
// is no such fine cell or it does not feed I).  Indices are clamped so the loads are always legal; a zero
// weight adds exactly 0.
__device__ __forceinline__ void rs1d4(const Tr1 &t, int I, int nf, bool wrap, double w[4], int f[4])
{
    const int f0 = t.fst[I] - 1;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        int ff = f0 + o;
        if (wrap) ff = ff < 0 ? ff + nf : (ff >= nf ? ff - nf : ff);  // across the periodic seam
        double wt = 0.0;
        if (ff >= 0 && ff < nf) {
            if (t.par[ff] == I)
                wt = t.wpar[ff];
            else if (t.oth[ff] == I)
                wt = t.woth[ff];
        }
        w[o] = wt;
        f[o] = ff < 0 ? 0 : (ff >= nf ? nf - 1 : ff);
    }
}

// bc = P^T rf over the owned coarse rows; fine halo planes valid.
__global__ __launch_bounds__(256) void k_restrict_rows(const Scalars *__restrict__ S, LevelDev F, LevelDev C,
                                                       const double *__restrict__ rf, double *__restrict__ bc,
                                                       int ngroups, int per_xcd, int vec_ok)
{
    if (S != nullptr && S->done) return;
    int row;
    if (!row_of_wave(ngroups, per_xcd, C.ny * C.nk, &row)) return;
    const int KK = row / C.ny, J = row - KK * C.ny, K = C.k0 + KK;
    double wk[4], wj[4];
    int sk[4], sj[4];
    rs1d4(F.t[2], K, F.nzg, F.tper & 4, wk, sk);
    rs1d4(F.t[1], J, F.ny, F.tper & 2, wj, sj);
    const int lane = threadIdx.x;
    const int Iraw = blockIdx.y * 64 + lane;
    const bool valid = Iraw < C.nx;
    const int I = valid ? Iraw : C.nx - 1;
    const int2 fc = F.tx.fc[I];
    const double4 rw = F.tx.rw[I];
    const bool pair = (fc.y == 2);
    const int f0 = fc.x, f1 = pair ? f0 + 1 : f0;
    const bool wrapx = F.tper & 1;
    const bool edgeL = (lane == 0 && (I > 0 || wrapx)), edgeR = (lane == 63 && I + 1 < C.nx) || (wrapx && Iraw == C.nx - 1);
    const int fL = (f0 > 0) ? f0 - 1 : F.nx - 1, fR = (f1 + 1 < F.nx) ? f1 + 1 : 0;  // wrapped only when wrapx (else unused)
    const int64_t fplane = (int64_t)F.nx * F.ny;
    double s = 0.0;
    // all sixteen row loads are issued before the first use (no branch on the wave-uniform zero weights: a zero
    // weight adds exactly 0 and its clamped row index is legal) -- 32 loads in flight per wave
    double c0[4][4], c1[4][4];
    // plain pairing along x (every lane's children are the aligned pair 2I, 2I+1): one 16-byte load per lane and row
    // instead of two 8-byte loads with a stride of two
    const bool vec = vec_ok && __all(pair && !(f0 & 1));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double *pk = rf + fplane * (sk[c] - F.k0);
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            const double *pj = pk + (int64_t)F.nx * sj[b2];
            if (vec) {
                const double2 v = *reinterpret_cast<const double2 *>(pj + f0);
                c0[c][b2] = v.x;
                c1[c][b2] = v.y;
            } else {
                c0[c][b2] = pj[f0];
                c1[c][b2] = pj[f1];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double *pk = rf + fplane * (sk[c] - F.k0);
        double u = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            const double *pj = pk + (int64_t)F.nx * sj[b2];
            double vl = __shfl_up(c1[c][b2], 1, 64), vr = __shfl_down(c0[c][b2], 1, 64);
            if (edgeL) vl = pj[fL];
            if (edgeR) vr = pj[fR];
            u = tacc(u, wj[b2], rsum_x(rw, vl, c0[c][b2], c1[c][b2], vr));
        }
        s = tacc(s, wk[c], u);
    }
    if (valid) bc[(int64_t)KK * C.nx * C.ny + (int64_t)J * C.nx + I] = s;
}

// ---- restriction, z-marching form for ANY aggregation (selective coarsening on a stretched mesh: lone cells among the pairs).
// The row kernel above loads sixteen fine rows per coarse cell -- every fine row by up to four waves (two coarse rows, two coarse
// planes) -- and is bound by the vector-memory issue rate (0.38 ms for the 25 M-cell level of the config-5 plate: 0.55 TB/s; it was
// 30 % of that case's V-cycle).  Here a wave owns ONE coarse row (64 coarse cells per lane group) and walks up through the fine
// planes: a plane's four fine rows are loaded once and summed in x and y (rsum_x, then the row weights) into u, and u goes with
// the plane's weight into the one or two coarse planes it feeds -- three running sums per lane, a coarse plane stored when the walk
// has left it behind.  The order of the sums is the oracle's (z ascending outermost, then y, then x): the bits of k_restrict_rows.
// Levels that are whole on this rank and have no periodic z seam; a fine plane's rows are requested a plane ahead.
__global__ __launch_bounds__(256) void k_restrict_zmarch(const Scalars *__restrict__ S, LevelDev F, LevelDev C,
                                                         const double *__restrict__ rf, double *__restrict__ bc, int CZ, int vec_ok)
{
    if (S != nullptr && S->done) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int J = (int)blockIdx.x * 4 + w;
    if (J >= C.ny) return;  // (no barrier in this kernel: a wave may leave)
    const int KA = (int)blockIdx.z * CZ, KB = min(KA + CZ, C.nzg);
    double wj[4];
    int sj[4];
    rs1d4(F.t[1], J, F.ny, F.tper & 2, wj, sj);
    const int Iraw = (int)blockIdx.y * 64 + lane;
    const bool valid = Iraw < C.nx;
    const int I = valid ? Iraw : C.nx - 1;
    const int2 fc = F.tx.fc[I];
    const double4 rw = F.tx.rw[I];
    const bool pair = (fc.y == 2);
    const int f0 = fc.x, f1 = pair ? f0 + 1 : f0;
    const bool wrapx = F.tper & 1;
    const bool edgeL = (lane == 0 && (I > 0 || wrapx)), edgeR = (lane == 63 && I + 1 < C.nx) || (wrapx && Iraw == C.nx - 1);
    const int fL = (f0 > 0) ? f0 - 1 : F.nx - 1, fR = (f1 + 1 < F.nx) ? f1 + 1 : 0;  // wrapped only when wrapx (else unused)
    const int64_t fplane = (int64_t)F.nx * F.ny, cplane = (int64_t)C.nx * C.ny;
    const bool vec = vec_ok && __all(pair && !(f0 & 1));
    const Tr1 tz = F.t[2];
    // fine planes that feed [KA, KB): from the lower neighbour of KA's first child to the first child of KB (its weight towards
    // KB - 1), clipped to the level
    const int k_lo = max(tz.fst[KA] - 1, 0), k_hi = min(KB < C.nzg ? tz.fst[KB] : F.nzg - 1, F.nzg - 1);
    double c0[4], c1[4], el[4], er[4], n0[4], n1[4], nl[4], nr[4];
    int Kp = 0, Ko = 0, Kpn = 0, Kon = 0;       // the plane's parent and other coarse plane, its weights towards them: they travel with
    double wp = 0.0, wo = 0.0, wpn = 0.0, won = 0.0;  // the rows (read behind them they would be waited for at once, and the rows with them)
    auto fetch = [&](int k, double (&a0)[4], double (&a1)[4], double (&al)[4], double (&ar)[4], int &kp, int &ko, double &vp, double &vo) {
        kp = tz.par[k];
        ko = tz.oth[k];
        vp = tz.wpar[k];
        vo = tz.woth[k];
        const double *pk = rf + fplane * k;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            const double *pj = pk + (int64_t)F.nx * sj[b2];
            if (vec) {
                const double2 v = *reinterpret_cast<const double2 *>(pj + f0);
                a0[b2] = v.x;
                a1[b2] = v.y;
            } else {
                a0[b2] = pj[f0];
                a1[b2] = pj[f1];
            }
            al[b2] = edgeL ? pj[fL] : 0.0;
            ar[b2] = edgeR ? pj[fR] : 0.0;
        }
    };
    // three running sums: the coarse planes Kb, Kb + 1, Kb + 2
    int Kb = KA - 1;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    auto store = [&](int K, double v) {
        if (valid && K >= KA && K < KB) bc[(int64_t)K * cplane + (int64_t)J * C.nx + I] = v;
    };
    if (k_lo <= k_hi) fetch(k_lo, c0, c1, el, er, Kp, Ko, wp, wo);
    for (int k = k_lo; k <= k_hi; ++k) {
        if (k + 1 <= k_hi) fetch(k + 1, n0, n1, nl, nr, Kpn, Kon, wpn, won);
        double u = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            double vl = __shfl_up(c1[b2], 1, 64), vr = __shfl_down(c0[b2], 1, 64);
            if (edgeL) vl = el[b2];
            if (edgeR) vr = er[b2];
            u = tacc(u, wj[b2], rsum_x(rw, vl, c0[b2], c1[b2], vr));
        }
        // the walk leaves plane Kb behind when this plane's parent is two above it (a plane touches its parent and one neighbour)
        while (Kp > Kb + 1) {
            store(Kb, a0);
            a0 = a1;
            a1 = a2;
            a2 = 0.0;
            ++Kb;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {  // the parent first or the other first: ascending coarse plane is not an order of a sum --
            const int K = t ? Ko : Kp;  // each coarse plane gets this fine plane's ONE term
            const double wt = t ? wo : wp;
            if (K == Kb) a0 = tacc(a0, wt, u);
            else if (K == Kb + 1) a1 = tacc(a1, wt, u);
            else if (K == Kb + 2) a2 = tacc(a2, wt, u);
        }
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) {
            c0[b2] = n0[b2];
            c1[b2] = n1[b2];
            el[b2] = nl[b2];
            er[b2] = nr[b2];
        }
        Kp = Kpn;
        Ko = Kon;
        wp = wpn;
        wo = won;
    }
    store(Kb, a0);
    store(Kb + 1, a1);
    store(Kb + 2, a2);
}

// ---- restriction, z-marching form (fully paired 3-D levels: every coarse cell has the children 2I, 2I+1 in all three
// directions).  The row kernel above is bound by the vector-memory issue rate (sixteen row loads per coarse cell, every
// fine row fetched by four waves): 0.68 ms per 512^3 launch against 0.15 ms of HBM time.  Here a workgroup owns 64 x 8
// coarse columns and walks up through the fine planes; a plane's 128 x 16 tile (+ the halo the 4-point stencils reach)
// goes through LDS once, double-buffered, and feeds the two coarse planes it belongs to (slots 0/1 of the upper, 2/3 of
// the lower one).  A coarse value is still the sum over z slot, y slot, x slot in that order with the same weight
// products, i.e. the bits of k_restrict_rows (out-of-range slots carry the weight 0 there and are skipped here).
constexpr int RX = 128, RY = 16, RSX = RX + 8, RSY = RY + 2, RV4 = (RSX / 4) * RSY;

__device__ __forceinline__ double rz_weight(const Tr1 &t, int kf, int K)
{
    return t.par[kf] == K ? t.wpar[kf] : (t.oth[kf] == K ? t.woth[kf] : 0.0);
}
__global__ __launch_bounds__(256) void k_restrict_march(const Scalars *__restrict__ S, LevelDev F, LevelDev C,
                                                        const double *__restrict__ rf, double *__restrict__ bc, int CZ)
{
    if (S != nullptr && S->done) return;
    __shared__ __attribute__((aligned(32))) double sp[2][RSY][SWR];  // (swizzled rows: swz)
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, tw = __builtin_amdgcn_readfirstlane(tid >> 6);  // (the wave's index: scalar, and so are J and the y weights)
    const Tile3 tb = tile_of_block();
    const int i0 = tb.x * RX, j0 = tb.y * RY;
    const int I = tb.x * (RX / 2) + lane, J = tb.y * (RY / 2) + 2 * tw;  // coarse cells (I, J) and (I, J + 1)
    const int KA = C.k0 + tb.z * CZ, KB = min(KA + CZ, C.k0 + C.nk);    // coarse planes [KA, KB) (global)
    const double4 rw = F.tx.rw[I];
    // transfers that reach across a periodic seam (F.tper): the tile's cells beyond the domain are the ones at the other
    // end (whole aligned pieces: nx % 128 == 0), plane -1 is plane nz - 1 (the whole level is here then)
    const bool wx = F.tper & 1, wy = F.tper & 2, wz = F.tper & 4;
    double wj[2][4];
    {
        int sj[4];
        rs1d4(F.t[1], J, F.ny, wy, wj[0], sj);
        rs1d4(F.t[1], J + 1, F.ny, wy, wj[1], sj);
    }
    const int64_t fplane = (int64_t)F.nx * F.ny, cplane = (int64_t)C.nx * C.ny;
    // this thread's share of a plane's tile: up to three aligned 4-cell pieces (zero outside the domain)
    int64_t goff[3];
    int loff[3];
    bool ok[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int idx = tid + 256 * e, row = idx / (RSX / 4), cx = idx - row * (RSX / 4);
        int gi = i0 - 4 + 4 * cx, gj = j0 - 1 + row;
        if (wx) gi = gi < 0 ? gi + F.nx : (gi >= F.nx ? gi - F.nx : gi);
        if (wy) gj = gj < 0 ? gj + F.ny : (gj >= F.ny ? gj - F.ny : gj);
        ok[e] = idx < RV4 && gi >= 0 && gi < F.nx && gj >= 0 && gj < F.ny;
        goff[e] = (int64_t)gj * F.nx + gi;
        loff[e] = idx < RV4 ? row * SWR + 2 * cx : -1;  // the piece's first half in its (swizzled) row
    }
    const int kf0 = 2 * KA - 1, kf1 = 2 * (KB - 1) + 2;  // fine planes that feed [KA, KB) (global, both ends inclusive)
    const v4 zero = {0, 0, 0, 0};
    v4 pre[3] = {zero, zero, zero};
    auto fetch = [&](int kf) {
        if (wz) kf = kf < 0 ? kf + F.nzg : (kf >= F.nzg ? kf - F.nzg : kf);
        const double *pf = rf + (int64_t)(kf - F.k0) * fplane;
#pragma unroll
        for (int e = 0; e < 3; ++e) pre[e] = ok[e] ? *reinterpret_cast<const v4 *>(pf + goff[e]) : zero;
    };
    if (kf0 >= 0 || wz) fetch(kf0);
    double lo[2] = {0.0, 0.0}, hi[2] = {0.0, 0.0};
    for (int kf = kf0; kf <= kf1; ++kf) {
        const bool inz = wz || (kf >= 0 && kf < F.nzg);
        const int kfw = wz ? (kf < 0 ? kf + F.nzg : (kf >= F.nzg ? kf - F.nzg : kf)) : kf;  // the plane's index in the tables
        const int slot = kf & 1;
        if (inz) {
            double *dst = &sp[slot][0][0];
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (loff[e] >= 0) swz_put4(dst + loff[e], 0, pre[e]);
        }
        __syncthreads();
        if (kf + 1 <= kf1 && (kf + 1 < F.nzg || wz)) fetch(kf + 1);
        const bool odd = kf & 1;
        const int Khi = odd ? (kf + 1) / 2 : kf / 2, Klo = Khi - 1;  // kf is slot 0 / 1 of Khi and slot 2 / 3 of Klo
        if (inz) {
            const bool dohi = Khi >= KA && Khi < KB, dolo = Klo >= KA && Klo < KB;
            const double wkhi = dohi ? rz_weight(F.t[2], kfw, Khi) : 0.0, wklo = dolo ? rz_weight(F.t[2], kfw, Klo) : 0.0;
            // the six fine rows of the two coarse rows
            double vl[6], c0[6], c1[6], vr[6];
            const int qc = swz(2 * lane + 4), ql = swz(2 * lane + 3), qr = swz(2 * lane + 6);  // the children, their left / right neighbours
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double *rowp = sp[slot][4 * tw + r];
                const double2 cc = *reinterpret_cast<const double2 *>(rowp + qc);
                vl[r] = rowp[ql];
                c0[r] = cc.x;
                c1[r] = cc.y;
                vr[r] = rowp[qr];
            }
            restrict_plane(rw, wj, vl, c0, c1, vr, dolo, wklo, dohi, wkhi, lo, hi);
        }
        if (!odd) {  // slot 3 of Klo is behind us: store it, the upper plane moves down
            if (Klo >= KA && Klo < KB) {
                double *dst = bc + (int64_t)(Klo - C.k0) * cplane + (int64_t)J * C.nx + I;
                dst[0] = lo[0];
                dst[C.nx] = lo[1];
            }
            lo[0] = hi[0];
            lo[1] = hi[1];
            hi[0] = hi[1] = 0.0;
        }
    }
}

// ---- residual + restriction in one march (fully paired 3-D levels that are whole on this rank): bc = P^T (b - A x).
// As two kernels the residual goes to HBM and comes back (k_level_march<3>: 24 B per fine cell, k_restrict_march: 9); here
// a workgroup walks up through the fine planes of its 128 x 16 tile like k_restrict_march, but what it stages in LDS is the
// ITERATE's plane (tile + two cells around it), from which every thread computes the residual of the cells it loaded --
// their z neighbours are its own registers, the plane below / the plane / the plane above -- into the LDS tile the
// restriction part reads: b and x are read once (~17 B per fine cell with the halos), nothing but the coarse right-hand
// side is written.  The residual's expression and the restriction's order of summation are those of k_level_march<3> /
// k_restrict_march: the same bits.
constexpr int QSY = RSY + 2;                 // rows of the iterate's tile: the residual's rows and one more on either side
constexpr int QV4 = (RSX / 4) * QSY;         // its aligned 4-cell pieces (680: up to three per thread)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_resid_restrict_march(const Scalars *__restrict__ S, LevelDev F, LevelDev C,
                                                              const double *__restrict__ b, const double *__restrict__ x,
                                                              double *__restrict__ bc, int CZ, const double *__restrict__ pin_sum = nullptr)
{
    if (S != nullptr && S->done) return;
    __shared__ __attribute__((aligned(32))) double xs[QSY][SWR];   // the iterate on the current plane: cols i0-4 .. i0+131, rows j0-2 .. j0+17 (swizzled rows: swz)
    __shared__ __attribute__((aligned(32))) double rs[RSY][SWR];   // its residual: rows j0-1 .. j0+16
    __shared__ __attribute__((aligned(16))) double tcx[3][SWR];    // cm, cp, w of the tile's columns (swizzled like the rows) ...
    __shared__ double tcy[3][QSY];                                 // ... and of its rows
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, tw = __builtin_amdgcn_readfirstlane(tid >> 6);  // (the wave's index: scalar, and so are J and the y weights)
    const Tile3 tb = tile_of_block();
    const int i0 = tb.x * RX, j0 = tb.y * RY;
    const int I = tb.x * (RX / 2) + lane, J = tb.y * (RY / 2) + 2 * tw;  // coarse cells (I, J) and (I, J + 1)
    const int KA = C.k0 + tb.z * CZ, KB = min(KA + CZ, C.k0 + C.nk);    // coarse planes [KA, KB)
    const double4 rw = F.tx.rw[I];
    const bool wx = F.tper & 1, wy = F.tper & 2, wz = F.tper & 4;       // (the caller checks per == tper)
    double wj[2][4];
    {
        int sj[4];
        rs1d4(F.t[1], J, F.ny, wy, wj[0], sj);
        rs1d4(F.t[1], J + 1, F.ny, wy, wj[1], sj);
    }
    const int64_t fplane = (int64_t)F.nx * F.ny, cplane = (int64_t)C.nx * C.ny;
    // the tile's 1-D coefficients (zero beyond the domain: such cells carry no residual)
    for (int e = tid; e < RSX; e += 256) {
        int gi = i0 - 4 + e;
        if (wx) gi = gi < 0 ? gi + F.nx : (gi >= F.nx ? gi - F.nx : gi);
        const bool in = gi >= 0 && gi < F.nx;
        tcx[0][swz(e)] = in ? F.cmx[gi] : 0.0;
        tcx[1][swz(e)] = in ? F.cpx[gi] : 0.0;
        tcx[2][swz(e)] = in ? F.wx[gi] : 0.0;
    }
    if (tid < QSY) {
        int gj = j0 - 2 + tid;
        if (wy) gj = gj < 0 ? gj + F.ny : (gj >= F.ny ? gj - F.ny : gj);
        const bool in = gj >= 0 && gj < F.ny;
        tcy[0][tid] = in ? F.cmy[gj] : 0.0;
        tcy[1][tid] = in ? F.cpy[gj] : 0.0;
        tcy[2][tid] = in ? F.wy[gj] : 0.0;
    }
    // this thread's share of a plane: up to three aligned 4-cell pieces of the iterate's tile (zero outside the domain);
    // a piece in the rows 1 .. RSY of that tile also carries the residual of its cells (and reads b there)
    int64_t goff[3];
    int prow[3], pcol[3];
    bool ok[3], mine[3], res[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int idx = tid + 256 * e, row = idx / (RSX / 4), cx = idx - row * (RSX / 4);
        int gi = i0 - 4 + 4 * cx, gj = j0 - 2 + row;
        if (wx) gi = gi < 0 ? gi + F.nx : (gi >= F.nx ? gi - F.nx : gi);
        if (wy) gj = gj < 0 ? gj + F.ny : (gj >= F.ny ? gj - F.ny : gj);
        mine[e] = idx < QV4;
        ok[e] = mine[e] && gi >= 0 && gi < F.nx && gj >= 0 && gj < F.ny;
        res[e] = ok[e] && row >= 1 && row <= RSY;
        goff[e] = (int64_t)gj * F.nx + gi;
        prow[e] = row;
        pcol[e] = 4 * cx;
    }
    const int kf0 = 2 * KA - 1, kf1 = 2 * (KB - 1) + 2;  // fine planes that feed [KA, KB) (both ends inclusive)
    // the per-plane table entries of the planes this workgroup walks, staged once (k_prolong_smooth2 says why): w, the two face
    // coefficients, the plane's restriction weights towards its upper and its lower coarse plane.  Entry e <-> fine plane kf0 + e.
    constexpr int ZT = 80;  // >= 2 CZ + 2 planes (CZ <= 32)
    __shared__ double tz[5][ZT];
    for (int e = tid; e < ZT; e += 256) {
        const int kf = kf0 + e;
        const bool in = (wz || (kf >= 0 && kf < F.nzg)) && kf <= kf1;
        const int kw = wz ? (kf < 0 ? kf + F.nzg : (kf >= F.nzg ? kf - F.nzg : kf)) : kf;
        const int Kh = (kf & 1) ? (kf + 1) / 2 : kf / 2;
        tz[0][e] = in ? F.wz[kw] : 0.0;
        tz[1][e] = in ? F.cmz[kw] : 0.0;
        tz[2][e] = in ? F.cpz[kw] : 0.0;
        tz[3][e] = in ? rz_weight(F.t[2], kw, Kh) : 0.0;
        tz[4][e] = in ? rz_weight(F.t[2], kw, Kh - 1) : 0.0;
    }
    const v4 zero = {0, 0, 0, 0};
    auto zwrap = [&](int kf) { return wz ? (kf < 0 ? kf + F.nzg : (kf >= F.nzg ? kf - F.nzg : kf)) : kf; };
    auto inz = [&](int kf) { return wz || (kf >= 0 && kf < F.nzg); };
    auto fetch_x = [&](int kf, v4 out[3]) {
        const bool in = inz(kf);
        const double *pf = x + (int64_t)(zwrap(kf) - F.k0) * fplane;
#pragma unroll
        for (int e = 0; e < 3; ++e) out[e] = (in && ok[e]) ? *reinterpret_cast<const v4 *>(pf + goff[e]) : zero;
    };
    auto fetch_b = [&](int kf, v4 out[3]) {
        const bool in = inz(kf);
        const double *pf = b + (int64_t)(zwrap(kf) - F.k0) * fplane;
#pragma unroll
        for (int e = 0; e < 3; ++e) out[e] = (in && res[e]) ? *reinterpret_cast<const v4 *>(pf + goff[e]) : zero;
        if (pin_sum != nullptr && in && zwrap(kf) == 0) {  // PINNED (level 0): effective b at global cell 0, the first cell of an aligned piece
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (res[e] && goff[e] == 0) out[e][0] = out[e][0] - *pin_sum;
        }
    };
    auto put_x = [&](const v4 v[3]) {
#pragma unroll
        for (int e = 0; e < 3; ++e)
            if (mine[e]) swz_put4(xs[prow[e]], pcol[e], v[e]);
    };
    // the iterate of the thread's pieces on the planes kf - 1, kf, kf + 1, the plane kf + 2 and the right-hand side of plane
    // kf + 1 on their way
    v4 xm[3], xc[3], xp[3], xn[3], bcur[3], bnext[3];
    fetch_x(kf0 - 1, xm);
    fetch_x(kf0, xc);
    fetch_x(kf0 + 1, xp);
    fetch_b(kf0, bcur);
    __syncthreads();  // the coefficient tables
    put_x(xc);
    __syncthreads();
    double lo[2] = {0.0, 0.0}, hi[2] = {0.0, 0.0};
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending on entry either
    for (int kf = kf0; kf <= kf1; ++kf) {
        const bool in = inz(kf);
        if (kf + 1 <= kf1) {
            fetch_x(kf + 2, xn);
            fetch_b(kf + 1, bnext);
        }
        // ---- the residual of plane kf (xs holds the iterate of plane kf)
        {
            const double wzk = tz[0][kf - kf0], czm = tz[1][kf - kf0], czp = tz[2][kf - kf0];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (!mine[e] || prow[e] < 1 || prow[e] > RSY) continue;
                v4 out = zero;
                if (res[e] && in) {
                    const int R = prow[e], X = pcol[e];
                    const double cym = tcy[0][R], cyp = tcy[1][R], wyj = tcy[2][R];
                    const v4 ylo = swz_get4(xs[R - 1], X), yhi = swz_get4(xs[R + 1], X);
                    const v4 cxm4 = swz_get4(tcx[0], X), cxp4 = swz_get4(tcx[1], X), wx4 = swz_get4(tcx[2], X);
                    const double xleft = X > 0 ? xs[R][swz(X - 1)] : 0.0, xright = X + 4 < RSX ? xs[R][swz(X + 4)] : 0.0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double xcc = xc[e][c];
                        const double left = (c == 0) ? xleft : xc[e][c > 0 ? c - 1 : 0];
                        const double right = (c == 3) ? xright : xc[e][c < 3 ? c + 1 : 0];
                        double sum = 0.0;
                        sum = facc(sum, cxm4[c], left, xcc);
                        sum = facc(sum, cxp4[c], right, xcc);
                        sum = facc(sum, cym, ylo[c], xcc);
                        sum = facc(sum, cyp, yhi[c], xcc);
                        sum = facc(sum, czm, xm[e][c], xcc);
                        sum = facc(sum, czp, xp[e][c], xcc);
                        out[c] = resid(bcur[e][c], sum * (wx4[c] * wyj), wzk);
                    }
                }
                swz_put4(rs[prow[e] - 1], pcol[e], out);
            }
        }
        __syncthreads();
        // ---- the iterate of the next plane takes the tile's place; the restriction's share of plane kf
        put_x(xp);
        const bool odd = kf & 1;
        const int Khi = odd ? (kf + 1) / 2 : kf / 2, Klo = Khi - 1;  // kf is slot 0 / 1 of Khi and slot 2 / 3 of Klo
        if (in) {
            const bool dohi = Khi >= KA && Khi < KB, dolo = Klo >= KA && Klo < KB;
            const double wkhi = dohi ? tz[3][kf - kf0] : 0.0, wklo = dolo ? tz[4][kf - kf0] : 0.0;
            double vl[6], c0[6], c1[6], vr[6];
            const int qc = swz(2 * lane + 4), ql = swz(2 * lane + 3), qr = swz(2 * lane + 6);  // the children, their left / right neighbours
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double *rowp = rs[4 * tw + r];
                const double2 cc = *reinterpret_cast<const double2 *>(rowp + qc);
                vl[r] = rowp[ql];
                c0[r] = cc.x;
                c1[r] = cc.y;
                vr[r] = rowp[qr];
            }
            restrict_plane(rw, wj, vl, c0, c1, vr, dolo, wklo, dohi, wkhi, lo, hi);
        }
        // (the planes requested at the top of this iteration are waited for HERE, on every path and through the builtin, so that the
        // compiler does not place that wait behind the stores below, which would then be waited for too: see k_prolong_smooth2)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (!odd) {  // slot 3 of Klo is behind us: store it, the upper plane moves down
            if (Klo >= KA && Klo < KB) {
                double *dst = bc + (int64_t)(Klo - C.k0) * cplane + (int64_t)J * C.nx + I;
                dst[0] = lo[0];
                dst[C.nx] = lo[1];
            }
            lo[0] = hi[0];
            lo[1] = hi[1];
            hi[0] = hi[1] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            xm[e] = xc[e];
            xc[e] = xp[e];
            xp[e] = xn[e];
            bcur[e] = bnext[e];
        }
    }
}

// ---- the way down of a V(2, .) cycle on a large level in ONE march (end of round 5: docs/design/down_march.md).
// k_presmooth2 (two Jacobi steps from zero; with UPD the Krylov residual's update on the fly) and k_resid_restrict_march
// (residual of the smoothed iterate, restricted) as one kernel: x1 = omega D^-1 b is pointwise, x2 needs x1 one cell around,
// the residual x2 one cell around, the restriction the residual one cell around -- all of it a function of the right-hand
// side within three cells.  A workgroup walks up through the fine planes of its 128 x TY tile; every one of its active threads
// owns ONE aligned 4-cell piece of the tile + 4 columns / 3 rows around it (34 x (TY + 6) pieces) and keeps that piece's
// right-hand side, x1 and x2 on three consecutive planes each in registers (the z neighbours); the x / y neighbours come
// from LDS rows (x1 on TY + 6 rows, x2 on TY + 4, the residual on TY + 2; two slots each for x1 and x2, so that the plane a
// stage reads was completed an iteration earlier: two barriers per plane).  The tile's own cells of the chunk's own planes
// are written (the new residual with UPD, x2), nothing else but the coarse right-hand side: with TY = 8 the right-hand side
// (and w) is read 1.86 x, 17 B per cell written -- 47 instead of 57 B per cell (TY = 16: 40, but one piece per thread then
// needs 768 threads and their 168 registers do not hold a piece's thirteen plane values without spilling).  Every value by the expression of the kernel it replaces:
// the same bits (the Krylov sums of UPD in k_presmooth2's grouping: its 128 x 8 tiles, its FZ planes = this kernel's 2 CZ).
constexpr int down_threads(int TX, int TY) { return ((((TX + 8) / 4) * (TY + 6) + 63) / 64) * 64; }  // 128 x 8: 512, 64 x 16: 448
template <int UPD, int TX, int TY>
__global__ __launch_bounds__(down_threads(TX, TY)) void k_down_march(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega,
                                                    const double *__restrict__ b, double *__restrict__ xo, double *__restrict__ bc, int CZ,
                                                    const double *__restrict__ pin_sum, const double *__restrict__ uw, double *__restrict__ unew,
                                                    double *__restrict__ upart, int upart_stride)
{
    if (S != nullptr && S->done) return;
    typedef double v4 __attribute__((ext_vector_type(4)));
    constexpr int DSX = TX + 8, DPR = DSX / 4;                                          // columns i0 - 4 .. i0 + TX + 3 in 4-cell pieces
    constexpr int DNT = down_threads(TX, TY), DRY = TY + 6, DXY = TY + 4, DSY = TY + 2;  // threads; rows of the right-hand side / x1 (j0 - 3 ..), of x2, of the residual
    constexpr int NRT = (TX / 2) * (TY / 2), CL = TX / 2;                                // threads of the restriction: one per coarse cell of the tile, CL a row
    __shared__ __attribute__((aligned(32))) double x1s[2][DRY][SWR];
    __shared__ __attribute__((aligned(32))) double xs[2][DXY][SWR];
    __shared__ __attribute__((aligned(32))) double rs[DSY][SWR];
    __shared__ __attribute__((aligned(16))) double tcx[4][SWR];  // cm, cp, w, 1 / w of the tile's columns i0 - 4 .. i0 + 131 (swizzled)
    __shared__ double tcy[4][DRY];                               // ... and of its rows j0 - 3 .. j0 + 18
    constexpr int ZT = 80;                                       // >= 2 CZ + 7 planes (CZ <= 32)
    __shared__ double tz[6][ZT];                                 // w, cm, cp, 1 / w, restriction weight up / down of the planes kfs + e
    const double ua = UPD ? S->a : 0.0;
    const int tid = threadIdx.x;
    const Tile3 tb = tile_of_block();
    const int i0 = tb.x * TX, j0 = tb.y * TY;
    const int ci = tid % CL, cj = (tid / CL) % (TY / 2);               // the coarse cell of a thread of the restriction, within the tile
    const int I = tb.x * CL + ci, J = tb.y * (TY / 2) + cj;  // the coarse cell of a thread of the first eight waves
    const int KA = C.k0 + tb.z * CZ, KB = min(KA + CZ, C.k0 + C.nk);     // coarse planes [KA, KB): fine planes [2 KA, 2 KB) are this chunk's own
    const bool rthread = tid < NRT;
    const double4 rw = F.tx.rw[rthread ? I : 0];
    double wj[4] = {0.0, 0.0, 0.0, 0.0};
    if (rthread) {
        int sj[4];
        rs1d4(F.t[1], J, F.ny, false, wj, sj);
    }
    const int64_t fplane = (int64_t)F.nx * F.ny, cplane = (int64_t)C.nx * C.ny;
    for (int e = tid; e < DSX; e += DNT) {
        const int gi = i0 - 4 + e;
        const bool in = gi >= 0 && gi < F.nx;
        tcx[0][swz(e)] = in ? F.cmx[gi] : 0.0;
        tcx[1][swz(e)] = in ? F.cpx[gi] : 0.0;
        tcx[2][swz(e)] = in ? F.wx[gi] : 0.0;
        tcx[3][swz(e)] = in ? F.rwx[gi] : 0.0;
    }
    if (tid < DRY) {
        const int gj = j0 - 3 + tid;
        const bool in = gj >= 0 && gj < F.ny;
        tcy[0][tid] = in ? F.cmy[gj] : 0.0;
        tcy[1][tid] = in ? F.cpy[gj] : 0.0;
        tcy[2][tid] = in ? F.wy[gj] : 0.0;
        tcy[3][tid] = in ? F.rwy[gj] : 0.0;
    }
    const int kf0 = 2 * KA - 1, kf1 = 2 * (KB - 1) + 2;  // fine planes whose residual feeds [KA, KB)
    const int kfs = kf0 - 2;                              // the march starts two planes earlier: x2 of kf0 - 1 and kf0 first
    for (int e = tid; e < ZT; e += DNT) {
        const int kf = kfs + e;
        const bool in = kf >= 0 && kf < F.nzg && kf <= kf1 + 2;
        const int Kh = (kf & 1) ? (kf + 1) / 2 : kf / 2;
        tz[0][e] = in ? F.wz[kf] : 0.0;
        tz[1][e] = in ? F.cmz[kf] : 0.0;
        tz[2][e] = in ? F.cpz[kf] : 0.0;
        tz[3][e] = in ? F.rwz[kf] : 0.0;
        tz[4][e] = in ? rz_weight(F.t[2], kf, Kh) : 0.0;
        tz[5][e] = in ? rz_weight(F.t[2], kf, Kh - 1) : 0.0;
    }
    // this thread's piece: row R of the 22, columns X .. X + 3 of the 136
    const int R = tid / DPR, X = 4 * (tid - R * DPR);
    const int gi = i0 - 4 + X, gj = j0 - 3 + R;
    const bool mine = tid < DPR * DRY;
    const bool ok = mine && gi >= 0 && gi < F.nx && gj >= 0 && gj < F.ny;
    const bool has2 = mine && R >= 1 && R <= DXY, hasr = mine && R >= 2 && R <= DSY + 1;   // carries x2 / the residual
    const bool own = ok && R >= 3 && R < 3 + TY && X >= 4 && X < 4 + TX;                    // a piece of the tile itself
    const int64_t goff = (int64_t)(ok ? gj : 0) * F.nx + (ok ? gi : 0);
    __syncthreads();  // the tables
    // the piece's in-plane coefficients
    v4 rxy4 = {0, 0, 0, 0}, cxm4 = {0, 0, 0, 0}, cxp4 = {0, 0, 0, 0}, vxy4 = {0, 0, 0, 0};
    double cym = 0.0, cyp = 0.0;
    if (mine) {
        const v4 rwx4 = swz_get4(tcx[3], X), wx4 = swz_get4(tcx[2], X);
        cxm4 = swz_get4(tcx[0], X);
        cxp4 = swz_get4(tcx[1], X);
        cym = tcy[0][R];
        cyp = tcy[1][R];
        const double rwyj = tcy[3][R], wyj = tcy[2][R];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rxy4[c] = rwx4[c] * rwyj;
            vxy4[c] = wx4[c] * wyj;
        }
    }
    const v4 zero = {0, 0, 0, 0};
    const double omc = 1.0 - omega;
    double ur0 = 0.0, ur1 = 0.0;
    auto inz = [&](int kf) { return kf >= 0 && kf < F.nzg; };
    // the right-hand side of the piece on plane kf: requested ...
    // (a vector is zeroed on the path that needs the zeros, not ahead of the branch: the kernel is bound by instruction issue)
    auto request = [&](int kf, v4 &vb, v4 &vw) {
        if (ok && inz(kf)) {
            vb = *reinterpret_cast<const v4 *>(b + (int64_t)(kf - F.k0) * fplane + goff);
            if (UPD) vw = *reinterpret_cast<const v4 *>(uw + (int64_t)(kf - F.k0) * fplane + goff);
        } else {
            vb = zero;
            vw = zero;
        }
    };
    // ... and taken in: the Krylov update, the tile's share of the new residual and of its sums, the pinned cell
    auto take = [&](int kf, v4 vb, const v4 &vw) -> v4 {
        if (!(ok && inz(kf))) return zero;
        if (UPD) {
#pragma unroll
            for (int c = 0; c < 4; ++c) vb[c] = vb[c] - ua * vw[c];
            if (own && kf >= 2 * KA && kf < 2 * KB) {
                *reinterpret_cast<v4 *>(unew + (int64_t)(kf - F.k0) * fplane + goff) = vb;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ur0 += vb[c] * vb[c];
                    ur1 += vb[c];
                }
            }
        }
        if (pin_sum != nullptr && kf == 0 && goff == 0) vb[0] = vb[0] - *pin_sum;
        return vb;
    };
    // omega / d of the piece's cells on a plane
    // (divided again only when a plane's z coefficients differ from the previous plane's: workgroup-uniform)
    double key_zm = __builtin_nan(""), key_zp = __builtin_nan("");
    v4 wlast = zero;
    auto weights = [&](int e) -> v4 {
        const double czm = tz[1][e], czp = tz[2][e];
        if (czm != key_zm || czp != key_zp) {
            key_zm = czm, key_zp = czp;
            wlast = zero;
            if (ok) {
#pragma unroll
                for (int c = 0; c < 4; ++c) wlast[c] = jweight(omega, -(((((cxm4[c] + cxp4[c]) + cym) + cyp) + czm) + czp));
            }
        }
        return wlast;
    };
    // (a piece outside the domain has w = 0 and a zero right-hand side: its x1 is 0 * 0 without a branch)
    auto first_step = [&](const v4 &vb, const v4 &w, int e) -> v4 {
        v4 o;
        const double rwz = tz[3][e];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = w[c] * ((vb[c] * rxy4[c]) * rwz);
        return o;
    };
    // registers: the right-hand side on the planes kf, kf + 1, kf + 2; x1 on kf, kf + 1 (kf + 2 is formed in the iteration);
    // x2 on kf - 1, kf (kf + 1 is formed in the iteration); omega / d of plane kf + 1
    v4 r0, r1, r2, x1a, x1b, xm = zero, xc = zero, wb, nb, nw;
    {
        v4 tb0, tw0, tb1, tw1, tb2, tw2;
        request(kfs, tb0, tw0);
        request(kfs + 1, tb1, tw1);
        request(kfs + 2, tb2, tw2);
        r0 = take(kfs, tb0, tw0);
        r1 = take(kfs + 1, tb1, tw1);
        r2 = take(kfs + 2, tb2, tw2);
        const v4 w0 = weights(0);
        wb = weights(1);
        x1a = first_step(r0, w0, 0);
        x1b = first_step(r1, wb, 1);
        if (mine) swz_put4(x1s[(kfs + 1) & 1][R], X, x1b);
    }
    __syncthreads();
    double lo = 0.0, hi = 0.0;
    // One plane of the march.  The planes a piece keeps rotate through NAMES, not through registers: three calls with the names
    // rotated make one pass of the loop below (the copies r0 = r1, r1 = r2 ... at the end of a plane were 56 of its ~250 vector
    // instructions, and the kernel is bound by instruction issue: 2.85e8 of them per 512^3 launch against 1.63e8 in the pair it
    // replaces).  A0, A1, A2: the right-hand side on kf, kf + 1, kf + 2 (A0 takes plane kf + 3 at the end); X0, X1: x1 on kf, kf + 1
    // (X0 takes plane kf + 2); M, C: x2 on kf - 1, kf (M takes plane kf + 1).
    auto plane = [&](int kf, v4 &A0, v4 &A1, v4 &A2, v4 &X0, v4 &X1, v4 &Q0, v4 &Q1) {
        const int e = kf - kfs;
        if (kf + 3 <= kf1 + 2) request(kf + 3, nb, nw);
        // ---- x1 of plane kf + 2
        const v4 wa = weights(e + 2);
        const v4 x1c = first_step(A2, wa, e + 2);
        if (mine) swz_put4(x1s[kf & 1][R], X, x1c);  // (slot of plane kf + 2)
        // ---- x2 of plane kf + 1: the second step, from x1 of the planes kf .. kf + 2 and its own plane's x / y neighbours in LDS
        v4 xp;
        if (has2 && ok && inz(kf + 1)) {
            const double rwz = tz[3][e + 1], czm = tz[1][e + 1], czp = tz[2][e + 1];
            const double(*pl)[SWR] = x1s[(kf + 1) & 1];
            const v4 ylo = swz_get4(pl[R - 1], X), yhi = swz_get4(pl[R + 1], X);
            const double xleft = X > 0 ? pl[R][swz(X - 1)] : 0.0, xright = X + 4 < DSX ? pl[R][swz(X + 4)] : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double xcc = X1[c];
                const double left = (c == 0) ? xleft : X1[c > 0 ? c - 1 : 0], right = (c == 3) ? xright : X1[c < 3 ? c + 1 : 0];
                double t = (A1[c] * rxy4[c]) * rwz;
                t = nacc(t, cxm4[c], left);
                t = nacc(t, cxp4[c], right);
                t = nacc(t, cym, ylo[c]);
                t = nacc(t, cyp, yhi[c]);
                t = nacc(t, czm, X0[c]);
                t = nacc(t, czp, x1c[c]);
                xp[c] = jrelax(xcc, omc, wb[c], t);
            }
            if (own && kf + 1 >= 2 * KA && kf + 1 < 2 * KB) *reinterpret_cast<v4 *>(xo + (int64_t)(kf + 1 - F.k0) * fplane + goff) = xp;
        } else
            xp = zero;
        if (has2) swz_put4(xs[(kf + 1) & 1][R - 1], X, xp);
        // ---- the residual of plane kf (x2 of the planes kf - 1, kf, kf + 1; plane kf's x / y neighbours in LDS)
        if (kf >= kf0 && hasr) {
            v4 out;
            if (ok && inz(kf)) {
                const double wzk = tz[0][e], czm = tz[1][e], czp = tz[2][e];
                const double(*pl)[SWR] = xs[kf & 1];
                const int Q = R - 1;  // the piece's row among x2's
                const v4 ylo = swz_get4(pl[Q - 1], X), yhi = swz_get4(pl[Q + 1], X);
                const double xleft = X > 0 ? pl[Q][swz(X - 1)] : 0.0, xright = X + 4 < DSX ? pl[Q][swz(X + 4)] : 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double xcc = Q1[c];
                    const double left = (c == 0) ? xleft : Q1[c > 0 ? c - 1 : 0], right = (c == 3) ? xright : Q1[c < 3 ? c + 1 : 0];
                    double sum = 0.0;
                    sum = facc(sum, cxm4[c], left, xcc);
                    sum = facc(sum, cxp4[c], right, xcc);
                    sum = facc(sum, cym, ylo[c], xcc);
                    sum = facc(sum, cyp, yhi[c], xcc);
                    sum = facc(sum, czm, Q0[c], xcc);
                    sum = facc(sum, czp, xp[c], xcc);
                    out[c] = resid(A0[c], sum * vxy4[c], wzk);
                }
            } else
                out = zero;
            swz_put4(rs[R - 2], X, out);
        }
        lds_barrier();
        // ---- the restriction's share of plane kf
        const bool odd = kf & 1;
        const int Khi = odd ? (kf + 1) / 2 : kf / 2, Klo = Khi - 1;
        if (kf >= kf0 && rthread && inz(kf)) {
            const bool dohi = Khi >= KA && Khi < KB, dolo = Klo >= KA && Klo < KB;
            const double wkhi = dohi ? tz[4][e] : 0.0, wklo = dolo ? tz[5][e] : 0.0;
            const int qc = swz(2 * ci + 4), ql = swz(2 * ci + 3), qr = swz(2 * ci + 6);
            double t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *rowp = rs[2 * cj + r];
                const double2 cc = *reinterpret_cast<const double2 *>(rowp + qc);
                t[r] = rsum_x(rw, rowp[ql], cc.x, cc.y, rowp[qr]);
            }
            double u = 0.0;
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) u = tacc(u, wj[b2], t[b2]);
            if (dolo) lo = tacc(lo, wklo, u);
            if (dohi) hi = tacc(hi, wkhi, u);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the plane requested at the top (placed here: see k_resid_restrict_march)
        A0 = (kf + 3 <= kf1 + 2) ? take(kf + 3, nb, nw) : zero;
        if (kf >= kf0 && !odd) {
            if (rthread && Klo >= KA && Klo < KB) bc[(int64_t)(Klo - C.k0) * cplane + (int64_t)J * C.nx + I] = lo;
            lo = hi;
            hi = 0.0;
        }
        lds_barrier();
        X0 = x1c;
        Q0 = xp;
        wb = wa;
    };
    // (the right-hand side's names come round after three planes, x1's and x2's after two: six planes a pass)
    for (int kf = kfs; kf <= kf1; kf += 6) {
        plane(kf, r0, r1, r2, x1a, x1b, xm, xc);
        if (kf + 1 <= kf1) plane(kf + 1, r1, r2, r0, x1b, x1a, xc, xm);
        if (kf + 2 <= kf1) plane(kf + 2, r2, r0, r1, x1a, x1b, xm, xc);
        if (kf + 3 <= kf1) plane(kf + 3, r0, r1, r2, x1b, x1a, xc, xm);
        if (kf + 4 <= kf1) plane(kf + 4, r1, r2, r0, x1a, x1b, xm, xc);
        if (kf + 5 <= kf1) plane(kf + 5, r2, r0, r1, x1b, x1a, xc, xm);
    }
    if (UPD) {
        // TX = 128, TY = 8: the sums in k_presmooth2's grouping -- its workgroup summed thread (ty, tx) by thread over the lanes of its four
        // waves, then (w0 + w1) + (w2 + w3): the same bits.  Other tiles: the tile's pieces in rows of TX / 4, summed the same way over
        // the lanes of up to four waves (equal to rounding: the residual NORMS the solver prints move in their last digits).
        constexpr int NP = TY * (TX / 4);  // pieces of the tile: 256
        static_assert(NP == 256, "the sums are formed by four waves");
        double(*ush)[NP] = reinterpret_cast<double(*)[NP]>(&x1s[0][0][0]);  // (the planes are done with)
        __shared__ double uw4[2][4];
        if (R >= 3 && R < 3 + TY && X >= 4 && X < 4 + TX && mine) {
            ush[0][(R - 3) * (TX / 4) + (X - 4) / 4] = ur0;
            ush[1][(R - 3) * (TX / 4) + (X - 4) / 4] = ur1;
        }
        __syncthreads();
        if (tid < NP) {
            double v0 = ush[0][tid], v1 = ush[1][tid];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                v0 += __shfl_down(v0, o, 64);
                v1 += __shfl_down(v1, o, 64);
            }
            if ((tid & 63) == 0) {
                uw4[0][tid >> 6] = v0;
                uw4[1][tid >> 6] = v1;
            }
        }
        __syncthreads();
        if (tid < 2) {
            const int64_t blk = ((int64_t)tb.z * gridDim.y + tb.y) * gridDim.x + tb.x;
            upart[(int64_t)tid * upart_stride + blk] = (uw4[tid][0] + uw4[tid][1]) + (uw4[tid][2] + uw4[tid][3]);
        }
    }
}

// coarsest level in ONE workgroup: `sweeps` damped-Jacobi sweeps from zero,
// ping-pong between xa / xb (global, L2-resident), block barrier between sweeps.
__global__ __launch_bounds__(256) void k_coarsest(const Scalars *__restrict__ S, LevelDev L, double omega, int sweeps,
                                                  const double *__restrict__ b, double *__restrict__ xa,
                                                  double *__restrict__ xb, double *__restrict__ xout)
{
    if (S != nullptr && S->done) return;
    const int plane = L.nx * L.ny, n = plane * L.nk;
    double *cur = xa, *nxt = xb;
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int p = threadIdx.x; p < n; p += blockDim.x) {
            const int i = p % L.nx, j = (p / L.nx) % L.ny, k = L.k0 + p / plane;
            double d;
            if (sw == 0) {
                double c[6];
                face_coefs(L, i, j, k, c);
                d = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
                nxt[p] = jweight(omega, d) * scale_b(L, i, j, k, b[p]);
            } else {
                const double t = relax_cell(L, cur, p, i, j, k, scale_b(L, i, j, k, b[p]), &d);
                nxt[p] = jrelax(cur[p], 1.0 - omega, jweight(omega, d), t);
            }
        }
        __threadfence_block();
        __syncthreads();
        double *t = cur;
        cur = nxt;
        nxt = t;
    }
    if (cur != xout) {
        for (int p = threadIdx.x; p < n; p += blockDim.x) xout[p] = cur[p];
    }
}


// ---- coarse tail: every level with <= TAIL_MAX_CELLS cells (replicated / single rank) in ONE workgroup ----
// The levels below ~32^3 are launch-latency bound: five 4.4-us launches per level and V-cycle (rocprof: 40 of
// the 53 kernels of a 512^3 iteration are such launches).  One 1024-thread workgroup walks the whole remaining
// V-cycle (pre-smooth, residual, restriction, ..., coarsest sweeps, ..., prolongation, post-smooth) with block
// barriers between the phases; the per-cell arithmetic is the same as in k_level / k_restrict_rows / k_prolong_rows,
// so results are bit-identical to the per-level launches.  Damped Jacobi only.
//
// What a phase costs decides whether the tail pays (tools/tail_probe.py with the kernel's phase stamps,
// profiles/r03_coarse_tail_phases.txt).  With vectors and tables in HBM a phase is a chain of L2 round trips (83 us per V-cycle
// on a 448^2 mesh: five levels); as first written for LDS it was no better -- table-by-table staging, per-level pointer
// arrays in scratch memory, and every access through a pointer that may be LDS or HBM, i.e. a FLAT instruction at the latency
// of a vector-cache hit.  Now: one coalesced copy of the packed tables, the level descriptors in LDS, the kernel compiled
// twice so that the LDS instance addresses pool and tables with LDS instructions, a level of at most one cell per thread
// keeps the cell's row in registers for its whole visit, a level of at most 64 cells is walked by one wave without block
// barriers, and the visit / restriction / prolongation code exists once, in a loop over the V-cycle's legs (the kernel runs
// once per V-cycle from a cold instruction cache): 46 us on that mesh.
constexpr int TAIL_MAX_CELLS = 32768;
constexpr int TAIL_MAX_LEVELS = 12;
struct TailLevel {
    LevelDev L;
    double *xa, *xb, *b, *r;
};
constexpr int TAIL_POOL = 16384;  // doubles of LDS for the tail's vectors (128 KB of the CU's 160)
constexpr int TAIL_GUARD = 512;   // ... and of margin around them
constexpr int TAIL_TAB = 2048;    // doubles of LDS for the 1-D tables of the tail's levels
struct TailArgs {
    int nlev;
    TailLevel lv[TAIL_MAX_LEVELS];
    double omega;
    int pre, post, sweeps;
    // LDS instance: place of level l's three vectors (iterate, spare, right-hand side; the residual takes the spare) in the pool
    int lds_off[TAIL_MAX_LEVELS];
    // ... and the levels' 1-D coefficient and transfer tables, which the set-up packs into ONE block of HBM in the layout they
    // have in LDS (tab_src, tab_used doubles; tt[l]: where level l's tables start inside it)
    int tab_used;
    const double *tab_src;
    struct Tabs { int w[3], rw[3], cm[3], cp[3], wpar[3], woth[3], par[3], oth[3], fst[3]; } tt[TAIL_MAX_LEVELS];
    double *out0;  // where the tail's first level leaves its result (global memory)
};
// (4 KB: more than a kernel's argument segment takes beside the hidden arguments -- the kernel reads it from HBM)

// Reads of the tail's 1-D tables.  LDS: the descriptors' table pointers were redirected into the kernel's LDS block, but a
// pointer loaded from a descriptor is a generic one (a FLAT load); the reader turns it back into an index of the block,
// which the compiler addresses with LDS instructions.
template <bool LDS>
struct TailTab {
    const double *g;  // the block's generic address
    double *s;        // the block
    __device__ __forceinline__ double operator()(const double *p, int i) const
    {
        if constexpr (LDS) return s[(p - g) + i];
        else return p[i];
    }
    __device__ __forceinline__ int operator()(const int *p, int i) const
    {
        if constexpr (LDS) return reinterpret_cast<const int *>(s)[(p - reinterpret_cast<const int *>(g)) + i];
        else return p[i];
    }
};

// rs1d4 for the tail's lanes (every lane its own coarse cell, nothing wave-uniform): the four slots' table entries are loaded
// unconditionally at clamped indices and selected afterwards -- sixteen independent loads instead of twelve dependent
// little chains; the same weights
template <class RD>
__device__ __forceinline__ void rs1d4_lane(const RD &rd, const Tr1 &t, int I, int nf, bool wrap, double w[4], int f[4])
{
    const int f0 = rd(t.fst, I) - 1;
    bool in[4];
    int par[4], oth[4];
    double wp[4], wo[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        int ff = f0 + o;
        if (wrap) ff = ff < 0 ? ff + nf : (ff >= nf ? ff - nf : ff);
        in[o] = ff >= 0 && ff < nf;
        f[o] = ff < 0 ? 0 : (ff >= nf ? nf - 1 : ff);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        par[o] = rd(t.par, f[o]);
        oth[o] = rd(t.oth, f[o]);
        wp[o] = rd(t.wpar, f[o]);
        wo[o] = rd(t.woth, f[o]);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) w[o] = in[o] ? (par[o] == I ? wp[o] : (oth[o] == I ? wo[o] : 0.0)) : 0.0;
}

// One cell's row of the level operator -- face coefficients, diagonal, volume factors, where its six neighbours sit -- and
// its scaled right-hand side.  Same expressions in the same order as face_coefs / scale_b / unscale / apply_cell.
struct TailCell {
    double c[6], d, rxy, rz, wxy, wz, bs;
    int off[6];
    int has;  // bit q: neighbour q exists (or is reached across a periodic seam); bit 8: there is a cell at all
};
template <class RD>
__device__ __forceinline__ void tail_cell(const RD &rd, const LevelDev &L, int p, int n, const double *b, TailCell &t)
{
    t.has = 0;
    if (p >= n) return;
    const int plane = L.nx * L.ny;
    const int i = p % L.nx, j = (p / L.nx) % L.ny, k = L.k0 + p / plane;
    t.c[0] = rd(L.cmx, i);
    t.c[1] = rd(L.cpx, i);
    t.c[2] = rd(L.cmy, j);
    t.c[3] = rd(L.cpy, j);
    t.c[4] = rd(L.cmz, k);
    t.c[5] = rd(L.cpz, k);
    t.d = -(((((t.c[0] + t.c[1]) + t.c[2]) + t.c[3]) + t.c[4]) + t.c[5]);
    t.rxy = rd(L.rwx, i) * rd(L.rwy, j);
    t.rz = rd(L.rwz, k);
    t.wxy = rd(L.wx, i) * rd(L.wy, j);
    t.wz = rd(L.wz, k);
    t.bs = (b[p] * t.rxy) * t.rz;
    const int sy = L.nx, sz = plane;
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    int has = 256;
    t.off[0] = i > 0 ? -1 : L.nx - 1;
    if (i > 0 || px) has |= 1;
    t.off[1] = i < L.nx - 1 ? 1 : -(L.nx - 1);
    if (i < L.nx - 1 || px) has |= 2;
    t.off[2] = j > 0 ? -sy : (L.ny - 1) * sy;
    if (j > 0 || py) has |= 4;
    t.off[3] = j < L.ny - 1 ? sy : -(L.ny - 1) * sy;
    if (j < L.ny - 1 || py) has |= 8;
    t.off[4] = k > 0 ? -sz : (L.zring ? -sz : (L.nzg - 1) * sz);
    if (k > 0 || pz) has |= 16;
    t.off[5] = k < L.nzg - 1 ? sz : (L.zring ? sz : -(L.nzg - 1) * sz);
    if (k < L.nzg - 1 || pz) has |= 32;
    t.has = has;
}
__device__ __forceinline__ double tail_row(const TailCell &t, const double *x, int p)
{
    const double xc = x[p];
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if ((t.has >> q) & 1) s = facc(s, t.c[q], x[p + t.off[q]], xc);
    return s;
}
// one phase of one cell: the step from a zero guess, a damped-Jacobi step x -> out, or the residual of x
__device__ __forceinline__ void tail_cell_phase(const TailCell &t, bool zero, bool res, double omega, const double *b, const double *x,
                                                double *out, int p)
{
    if (!t.has) return;
    if (zero) {
        out[p] = jweight(omega, t.d) * t.bs;
        return;
    }
    if (res) {
        out[p] = resid(b[p], tail_row(t, x, p) * t.wxy, t.wz);
        return;
    }
    double s = t.bs;
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if ((t.has >> q) & 1) s = nacc(s, t.c[q], x[p + t.off[q]]);
    out[p] = jrelax(x[p], 1.0 - omega, jweight(omega, t.d), s);
}

// One visit of a level: `steps` smoothing steps (the first from a zero guess on the way down) and, on the way down, the
// residual; every step swaps the level's two vectors a / c, the residual goes to r.  Returns where the iterate is.
// KIND 0: at most 64 cells -- the first wave alone, a wavefront-scope fence between the steps (a wave's memory operations are
// issued and performed in order), the row in registers; 1: at most one cell per thread, block barriers, the row in
// registers; 2: several cells per thread, the rows rebuilt from the tables in every phase.
template <int KIND, class RD>
__device__ __forceinline__ double *tail_visit(const RD &rd, const LevelDev &F, int n, double omega, int steps, bool zero_first, bool resid,
                                              const double *b, double *a, double *c, double *r)
{
    const int phases = steps + (resid ? 1 : 0);
    const int p0 = threadIdx.x;
    const bool mine = KIND != 0 || p0 < 64;
    TailCell tc;
    tc.has = 0;
    if (KIND != 2 && mine) tail_cell(rd, F, p0, n, b, tc);
    for (int k = 0; k < phases; ++k) {
        const bool zero = zero_first && k == 0, res = k == steps;
        double *out = zero ? a : (res ? r : c);
        if (KIND == 2) {
            for (int p = p0; p < n; p += blockDim.x) {
                tail_cell(rd, F, p, n, b, tc);
                tail_cell_phase(tc, zero, res, omega, b, a, out, p);
            }
        } else if (mine)
            tail_cell_phase(tc, zero, res, omega, b, a, out, p0);
        if (KIND == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __threadfence_block();
            __syncthreads();
        }
        if (!zero && !res) {
            double *t = a; a = c; c = t;
        }
    }
    if (KIND == 0) {
        __threadfence_block();
        __syncthreads();
    }
    return a;
}

// LDS: the levels' vectors and tables live in LDS -- a compile-time fact, so that they are reached with LDS instructions
template <bool LDS>
__global__ __launch_bounds__(1024) void k_coarse_tail(const Scalars *__restrict__ S, const TailArgs *__restrict__ Tp)
{
    if (S != nullptr && S->done) return;
    const TailArgs &T = *Tp;
#ifdef PIB_TAIL_STAMPS
    __shared__ unsigned long long st_[256];
    __shared__ int sg_[256];
    int nst_ = 0;
#define STAMP(tag) do { if (threadIdx.x == 0 && nst_ < 256) { st_[nst_] = wall_clock64(); sg_[nst_] = (tag); } ++nst_; } while (0)
#else
#define STAMP(tag) do { } while (0)
#endif
    STAMP(0);
    // (a margin on either side of the pool: a neighbour of zero weight may be read before its weight is known; an access
    // below the LDS aperture is a fault, not a zero)
    __shared__ double pool_[LDS ? TAIL_GUARD + TAIL_POOL + TAIL_GUARD : 1];
    __shared__ double tab_[LDS ? TAIL_TAB : 1];
    __shared__ LevelDev Ls_[TAIL_MAX_LEVELS];
    double *const pool = pool_ + (LDS ? TAIL_GUARD : 0);
    const int nl = T.nlev;
    const TailTab<LDS> rd = {tab_, tab_};
    // the levels' descriptors into LDS (word by word, all threads), the tables after them; then one thread per level redirects
    // its descriptor's table pointers to the LDS copies
    {
        constexpr int words = (int)(sizeof(LevelDev) / sizeof(int));
        static_assert(sizeof(LevelDev) % sizeof(int) == 0, "LevelDev is copied in 4-byte words");
        for (int e = threadIdx.x; e < nl * words; e += blockDim.x) {
            const int l = e / words, w = e - l * words;
            reinterpret_cast<int *>(&Ls_[l])[w] = reinterpret_cast<const int *>(&T.lv[l].L)[w];
        }
        if (LDS) {
            for (int p = threadIdx.x; p < T.tab_used; p += blockDim.x) tab_[p] = T.tab_src[p];
            // (the margins finite; inside the pool every entry is written before it is read)
            for (int p = threadIdx.x; p < 2 * TAIL_GUARD; p += blockDim.x) pool_[p < TAIL_GUARD ? p : TAIL_POOL + p] = 0.0;
        }
        __syncthreads();
        if (LDS && (int)threadIdx.x < nl) {
            LevelDev &L = Ls_[threadIdx.x];
            const TailArgs::Tabs &o = T.tt[threadIdx.x];
            auto dbl = [&](int off) -> const double * { return tab_ + off; };
            auto i32 = [&](int off) -> const int * { return reinterpret_cast<const int *>(tab_ + off); };
            L.wx = dbl(o.w[0]), L.wy = dbl(o.w[1]), L.wz = dbl(o.w[2]);
            L.rwx = dbl(o.rw[0]), L.rwy = dbl(o.rw[1]), L.rwz = dbl(o.rw[2]);
            L.cmx = dbl(o.cm[0]), L.cmy = dbl(o.cm[1]), L.cmz = dbl(o.cm[2]);
            L.cpx = dbl(o.cp[0]), L.cpy = dbl(o.cp[1]), L.cpz = dbl(o.cp[2]);
            if ((int)threadIdx.x + 1 < nl) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    L.t[d].wpar = dbl(o.wpar[d]);
                    L.t[d].woth = dbl(o.woth[d]);
                    L.t[d].par = i32(o.par[d]);
                    L.t[d].oth = i32(o.oth[d]);
                    L.t[d].fst = i32(o.fst[d]);
                }
            }
        }
        if (LDS) {  // the first level's right-hand side
            const int n0 = T.lv[0].L.nx * T.lv[0].L.ny * T.lv[0].L.nk;
            double *b0 = pool + T.lds_off[0] + 2 * n0;
            for (int p = threadIdx.x; p < n0; p += blockDim.x) b0[p] = T.lv[0].b[p];
        }
        __syncthreads();
    }
    STAMP(1);
    // level l's iterate / spare / right-hand side (which = 0, 1, 2) and where its residual goes: the spare in LDS
    auto vec = [&](int l, int which) -> double * {
        const TailLevel &V = T.lv[l];
        if constexpr (LDS) return pool + T.lds_off[l] + which * (V.L.nx * V.L.ny * V.L.nk);
        else return which == 0 ? V.xa : (which == 1 ? V.xb : V.b);
    };
    // a level's two vectors swap with every step but the one from a zero guess: which of them holds the iterate after the
    // way down (no per-level pointer arrays: indexed by a runtime level they would live in scratch memory)
    const int dsteps = T.pre > 1 ? T.pre : 1;
    int l = 0;
    bool down = true;
    double *a = nullptr;  // the iterate of the level just visited
    for (;;) {
        const LevelDev &F = Ls_[l];
        const int fplane = F.nx * F.ny, nf = fplane * F.nk;
        const bool coarsest = l == nl - 1;
        const int steps = down ? (coarsest ? (T.sweeps > 1 ? T.sweeps : 1) : dsteps) : T.post;
        const bool resid = down && !coarsest;
        double *b = vec(l, 2);
        const int at = down ? 0 : ((dsteps - 1) & 1);
        double *xa = vec(l, at), *xc = vec(l, 1 - at);
        double *r = xc;
        if (resid) {
            // (the spare at the time of the residual: the vector the last step did NOT write)
            r = LDS ? vec(l, 1 - ((dsteps - 1) & 1)) : T.lv[l].r;
        }
        if (nf <= 64) a = tail_visit<0>(rd, F, nf, T.omega, steps, down, resid, b, xa, xc, r);
        else if (nf <= (int)blockDim.x) a = tail_visit<1>(rd, F, nf, T.omega, steps, down, resid, b, xa, xc, r);
        else a = tail_visit<2>(rd, F, nf, T.omega, steps, down, resid, b, xa, xc, r);
        STAMP((down ? 10 : 90) + l);
        if (resid) {
            // ---- restriction: right-hand side of level l + 1 = P^T r
            const LevelDev &C = Ls_[l + 1];
            const int cplane = C.nx * C.ny, nc = cplane * C.nk;
            double *bc = vec(l + 1, 2);
            for (int q = threadIdx.x; q < nc; q += blockDim.x) {
                const int I = q % C.nx, J = (q / C.nx) % C.ny, K = C.k0 + q / cplane;
                double wi[4], wj[4], wk[4];
                int si[4], sj[4], sk[4];
                rs1d4_lane(rd, F.t[0], I, F.nx, F.tper & 1, wi, si);
                __builtin_amdgcn_sched_barrier(0);  // (one direction's sixteen loads at a time: all three at once spill)
                rs1d4_lane(rd, F.t[1], J, F.ny, F.tper & 2, wj, sj);
                __builtin_amdgcn_sched_barrier(0);
                rs1d4_lane(rd, F.t[2], K, F.nzg, F.tper & 4, wk, sk);
                __builtin_amdgcn_sched_barrier(0);
                double sum = 0.0;
                for (int c2 = 0; c2 < 4; ++c2) {
                    if (wk[c2] == 0.0) continue;
                    const double *pk = r + fplane * (sk[c2] - F.k0);
                    double u = 0.0;
                    for (int b2 = 0; b2 < 4; ++b2) {
                        if (wj[b2] == 0.0) continue;  // (a 2-D level: one row of the four; the terms left out are +-0)
                        const double *pj = pk + F.nx * sj[b2];
                        double t = 0.0;
#pragma unroll
                        for (int a2 = 0; a2 < 4; ++a2) t = tacc(t, wi[a2], pj[si[a2]]);
                        u = tacc(u, wj[b2], t);
                    }
                    sum = tacc(sum, wk[c2], u);
                }
                bc[q] = sum;
            }
            __threadfence_block();
            __syncthreads();
            STAMP(50 + l);
            ++l;
            continue;
        }
        if (l == 0) break;
        // ---- prolongation: the iterate of level l - 1 += P a
        {
            const LevelDev &C = F;
            const LevelDev &G = Ls_[l - 1];
            const int gplane = G.nx * G.ny, ng = gplane * G.nk;
            const int cplane = C.nx * C.ny;
            double *xf = vec(l - 1, (dsteps - 1) & 1);
            for (int p = threadIdx.x; p < ng; p += blockDim.x) {
                const int i = p % G.nx, j = (p / G.nx) % G.ny, k = G.k0 + p / gplane;
                int I[2], J[2], K[2];
                double wi[2], wj[2], wk[2];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const Tr1 &t = G.t[d];
                    const int sd = d == 0 ? i : (d == 1 ? j : k);
                    int *Id = d == 0 ? I : (d == 1 ? J : K);
                    double *wd = d == 0 ? wi : (d == 1 ? wj : wk);
                    Id[0] = rd(t.par, sd);
                    Id[1] = rd(t.oth, sd);
                    wd[0] = rd(t.wpar, sd);
                    wd[1] = rd(t.woth, sd);
                }
                double sum = 0.0;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                        for (int a2 = 0; a2 < 2; ++a2) {
                            const double wgt = (wk[c2] * wj[b2]) * wi[a2];
                            if (wgt != 0.0) sum = tacc(sum, wgt, a[I[a2] + C.nx * J[b2] + cplane * (K[c2] - C.k0)]);
                        }
                xf[p] += sum;
            }
            __threadfence_block();
            __syncthreads();
            STAMP(70 + l - 1);
        }
        --l;
        down = false;
    }
    if (LDS) {
        const int n0 = T.lv[0].L.nx * T.lv[0].L.ny * T.lv[0].L.nk;
        for (int p = threadIdx.x; p < n0; p += blockDim.x) T.out0[p] = a[p];
    }
    STAMP(4);
#ifdef PIB_TAIL_STAMPS
    // tools: phase times of some launches (tag: 1 staged, 10+l way-down visit of level l -- the coarsest's sweeps for the last
    // --, 50+l restriction, 70+l prolongation onto l, 90+l post-smoothing, 4 result written); 10 ns units
    if (threadIdx.x == 0) {
        static __device__ int launches_ = 0;
        if (atomicAdd(&launches_, 1) % 40 == 20)
            for (int q = 1; q < nst_ && q < 256; ++q) printf("tail-stamp %d %llu\n", sg_[q], st_[q] - st_[q - 1]);
    }
#endif
#undef STAMP
}

// ---- small levels: a level's whole way down, and its whole way up, in ONE launch each -------------------------------
// Between the marching kernels of the large levels and the single-workgroup tail sit three or four levels (64^3 ... 16^3
// under a 512^3 grid, 224^2 ... 56^2 under a 448^2 one) that are launch-bound: seven kernels per level and V-cycle (step
// from zero, step, residual, restriction | prolongation, two steps).  Here a workgroup owns the children of a box of coarse
// cells and evaluates everything it needs on that box plus a margin in LDS -- the first step on the box grown by pre + 1
// cells, every further step and the residual one cell less, then the restriction for its own coarse cells (way down); the
// corrected iterate on the box grown by post cells, every post-smoothing step one cell less (way up).  The margins are
// recomputed by the neighbouring workgroups; no field crosses HBM between the phases.
// What such a kernel costs is the number of DEPENDENT trips to memory, ~2 us each when the lines are cold (phase stamps,
// profiles/r03_small_level_kernels.txt: a first version that fetched tables and right-hand side where it used them took 19
// us per launch, a third of it the restriction's little pointer chases).  So: ONE round of loads -- the 1-D coefficient and
// transfer tables of the region into LDS, the thread's right-hand-side values (and iterate values, way up) into registers,
// a thread keeping the same region cells through all phases -- and after it LDS only (way up: plus the gather of the coarse
// values).  Same per-cell expressions in the same order as k_level / the transfers: the same bits.
// Levels whole on this rank, Jacobi, 1-2 pre- / post-smoothing steps, operator and transfers wrapping alike.
constexpr int SM_MAXE = 40;    // largest extent of a workgroup's region per direction
constexpr int SM_MAXR = 3456;  // ... and its cells (two LDS buffers of that many doubles)
constexpr int SM_NT = 512;     // threads per workgroup
constexpr int SM_NB = (SM_MAXR + SM_NT - 1) / SM_NT;  // region cells per thread
constexpr int SM_MAXBC = 16;   // coarse cells per direction and workgroup
struct SmTabs {
    double cm[3][SM_MAXE], cp[3][SM_MAXE], rw[3][SM_MAXE], w[3][SM_MAXE];  // coefficients, by region coordinate
    double wpar[3][SM_MAXE], woth[3][SM_MAXE];                            // transfers of the region's fine cells
    int par[3][SM_MAXE], oth[3][SM_MAXE];
    int fst[3][SM_MAXBC + 1];                                              // first child of the owned coarse cells
};
struct SmGeom {
    int n[3];           // the level's cells
    int per[3];         // periodic directions
    int I0[3], I1[3];   // owned coarse cells [I0, I1)
    int F0[3], F1[3];   // their children: the owned fine cells [F0, F1)
    int lo[3], ext[3];  // the largest region, in unwrapped level coordinates (clipped to the domain where it does not wrap)
};
__device__ __forceinline__ int sm_wrap(int g, int n) { return g < 0 ? g + n : (g >= n ? g - n : g); }

// geometry of workgroup `blk`: its box of bc[] coarse cells, their children, the region grown by `grow_lo` / `grow_hi`.
// Aggregates that are all pairs (or all single cells) need no table for the children's range.
__device__ __forceinline__ void sm_geometry(const LevelDev &F, const LevelDev &C, int blk, int bcx, int bcy, int bcz, int grow_lo, int grow_hi,
                                            SmGeom &G)
{
    const int bc[3] = {bcx, bcy, bcz};
    const int nc[3] = {C.nx, C.ny, C.nzg};
    G.n[0] = F.nx, G.n[1] = F.ny, G.n[2] = F.nzg;
    int nb[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) nb[d] = (nc[d] + bc[d] - 1) / bc[d];
    const int b3[3] = {blk % nb[0], (blk / nb[0]) % nb[1], blk / (nb[0] * nb[1])};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        G.per[d] = (F.per >> d) & 1;
        G.I0[d] = b3[d] * bc[d];
        G.I1[d] = min(G.I0[d] + bc[d], nc[d]);
        if (G.n[d] == 2 * nc[d]) G.F0[d] = 2 * G.I0[d], G.F1[d] = 2 * G.I1[d];
        else if (G.n[d] == nc[d]) G.F0[d] = G.I0[d], G.F1[d] = G.I1[d];
        else {
            G.F0[d] = F.t[d].fst[G.I0[d]];
            G.F1[d] = G.I1[d] < nc[d] ? F.t[d].fst[G.I1[d]] : G.n[d];
        }
        int lo = G.F0[d] - grow_lo, hi = G.F1[d] + grow_hi;
        if (!G.per[d]) lo = max(lo, 0), hi = min(hi, G.n[d]);
        G.lo[d] = lo;
        G.ext[d] = hi - lo;
    }
}
// the one round of table loads
__device__ __forceinline__ void sm_stage_tabs(const LevelDev &F, const LevelDev &C, const SmGeom &G, SmTabs &T)
{
    const double *cm[3] = {F.cmx, F.cmy, F.cmz}, *cp[3] = {F.cpx, F.cpy, F.cpz}, *rw[3] = {F.rwx, F.rwy, F.rwz}, *w[3] = {F.wx, F.wy, F.wz};
    const int nc[3] = {C.nx, C.ny, C.nzg};
    // (thread t: direction t / 64, entry t % 64 -- all of a table's loads in one wave's single pass; the direction as a
    // compile-time constant of an unrolled loop: indexed by a runtime one the geometry would live in scratch memory)
    const int dw = threadIdx.x >> 6, r = threadIdx.x & 63;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (dw == d && r < G.ext[d]) {
            const int s = sm_wrap(G.lo[d] + r, G.n[d]);
            const Tr1 &tt = F.t[d];
            const double a0 = cm[d][s], a1 = cp[d][s], a2 = rw[d][s], a3 = w[d][s], a4 = tt.wpar[s], a5 = tt.woth[s];
            const int i0 = tt.par[s], i1 = tt.oth[s];
            T.cm[d][r] = a0, T.cp[d][r] = a1, T.rw[d][r] = a2, T.w[d][r] = a3, T.wpar[d][r] = a4, T.woth[d][r] = a5;
            T.par[d][r] = i0, T.oth[d][r] = i1;
        }
        if (dw == d + 3 && r <= G.I1[d] - G.I0[d] && G.I0[d] + r < nc[d]) T.fst[d][r] = F.t[d].fst[G.I0[d] + r];
    }
}
// the thread's cells of the largest region (the same in every phase): coordinates packed as rx | ry << 8 | rz << 16, -1: none
__device__ __forceinline__ void sm_cells(const SmGeom &G, int cell[SM_NB])
{
    const int cells = G.ext[0] * G.ext[1] * G.ext[2], e01 = G.ext[0] * G.ext[1];
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) {
        const int t = threadIdx.x + u * SM_NT;
        if (t < cells) {
            const int rz = t / e01, tr = t - rz * e01;
            const int ry = tr / G.ext[0], rx = tr - ry * G.ext[0];
            cell[u] = rx | (ry << 8) | (rz << 16);
        } else
            cell[u] = -1;
    }
}
__device__ __forceinline__ int64_t sm_global(const LevelDev &F, const SmGeom &G, int c)
{
    const int i = sm_wrap(G.lo[0] + (c & 255), G.n[0]), j = sm_wrap(G.lo[1] + ((c >> 8) & 255), G.n[1]), k = sm_wrap(G.lo[2] + (c >> 16), G.n[2]);
    return (int64_t)i + (int64_t)F.nx * (j + (int64_t)F.ny * (k - F.k0));
}
// One phase over the region's cells that lie `m` cells inside its unclipped faces (a clipped face is the domain's: no margin
// there).  MODE 1: x = omega bs / d; 2: x' = x + omega (bs - t) / d; 3: r = b - (t wx wy) wz.  src / dst: LDS fields indexed
// like the largest region; gdst (MODE 2 only): the owned cells' values go to global memory as well.
template <int MODE>
__device__ __forceinline__ void sm_phase(const LevelDev &F, const SmGeom &G, const SmTabs &T, const int cell[SM_NB], const double bq[SM_NB],
                                         int grow_lo, int grow_hi, int m, double omega, const double *src, double *dst,
                                         double *__restrict__ gdst)
{
    int a[3], e[3];  // the phase's box, relative to the region
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int lo = G.F0[d] - grow_lo + m, hi = G.F1[d] + grow_hi - m;
        if (!G.per[d]) lo = max(lo, 0), hi = min(hi, G.n[d]);
        a[d] = lo - G.lo[d];
        e[d] = hi - lo;
    }
    const int sy = G.ext[0], sz = G.ext[0] * G.ext[1];
    // Branch-free up to the store: the thread's cells are independent chains of LDS reads, ~20 dependent fp64 operations and a
    // division -- as separate basic blocks (a `continue` per cell) they ran one after the other, 0.3 us each.  A cell outside
    // the phase's box is evaluated all the same (at cell 0 if the thread has none: any finite or non-finite value will do)
    // and not stored.
    constexpr int GR = 3;  // cells evaluated together (all of them at once: 256 registers and a hundred spilled)
#pragma unroll
    for (int u0 = 0; u0 < SM_NB; u0 += GR) {
    double v[GR];
    int qv[GR];
    bool st[GR];
#pragma unroll
    for (int uu = 0; uu < GR; ++uu) {
        const int u = u0 + uu < SM_NB ? u0 + uu : SM_NB - 1;
        const int c = cell[u] < 0 ? 0 : cell[u];
        const int rx = c & 255, ry = (c >> 8) & 255, rz = c >> 16;
        st[uu] = u0 + uu < SM_NB && cell[u] >= 0 && (unsigned)(rx - a[0]) < (unsigned)e[0] && (unsigned)(ry - a[1]) < (unsigned)e[1] &&
                 (unsigned)(rz - a[2]) < (unsigned)e[2];
        const int q = rx + sy * ry + sz * rz;
        qv[uu] = q;
        const double cxm = T.cm[0][rx], cxp = T.cp[0][rx], cym = T.cm[1][ry], cyp = T.cp[1][ry], czm = T.cm[2][rz], czp = T.cp[2][rz];
        const double d = -(((((cxm + cxp) + cym) + cyp) + czm) + czp);
        const double bv = bq[u];
        const double bs = (bv * (T.rw[0][rx] * T.rw[1][ry])) * T.rw[2][rz];
        if (MODE == 1) {
            v[uu] = jweight(omega, d) * bs;
        } else if (MODE == 2) {
            // a neighbour beyond the region: only at a clipped face, i.e. a wall -- zero coefficient, the centre's own value
            const double xc = src[q];
            double t = bs;
            t = nacc(t, cxm, src[rx > 0 ? q - 1 : q]);
            t = nacc(t, cxp, src[rx < G.ext[0] - 1 ? q + 1 : q]);
            t = nacc(t, cym, src[ry > 0 ? q - sy : q]);
            t = nacc(t, cyp, src[ry < G.ext[1] - 1 ? q + sy : q]);
            t = nacc(t, czm, src[rz > 0 ? q - sz : q]);
            t = nacc(t, czp, src[rz < G.ext[2] - 1 ? q + sz : q]);
            v[uu] = jrelax(xc, 1.0 - omega, jweight(omega, d), t);
        } else {
            const double xc = src[q];
            double s = 0.0;
            s = facc(s, cxm, src[rx > 0 ? q - 1 : q], xc);
            s = facc(s, cxp, src[rx < G.ext[0] - 1 ? q + 1 : q], xc);
            s = facc(s, cym, src[ry > 0 ? q - sy : q], xc);
            s = facc(s, cyp, src[ry < G.ext[1] - 1 ? q + sy : q], xc);
            s = facc(s, czm, src[rz > 0 ? q - sz : q], xc);
            s = facc(s, czp, src[rz < G.ext[2] - 1 ? q + sz : q], xc);
            v[uu] = resid(bv, s * (T.w[0][rx] * T.w[1][ry]), T.w[2][rz]);
        }
    }
#pragma unroll
    for (int uu = 0; uu < GR; ++uu) {
        if (!st[uu]) continue;
        dst[qv[uu]] = v[uu];
        if (MODE == 2 && gdst != nullptr) {
            const int c = cell[u0 + uu < SM_NB ? u0 + uu : SM_NB - 1];
            const int gx = G.lo[0] + (c & 255), gy = G.lo[1] + ((c >> 8) & 255), gz = G.lo[2] + (c >> 16);
            if (gx >= G.F0[0] && gx < G.F1[0] && gy >= G.F0[1] && gy < G.F1[1] && gz >= G.F0[2] && gz < G.F1[2]) gdst[sm_global(F, G, c)] = v[uu];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    }
}

// way down: b -> the pre-smoothed iterate x (owned cells) and the next level's right-hand side bc = P^T (b - A x)
__global__ __launch_bounds__(SM_NT) void k_small_down(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega, int pre,
                                                      const double *__restrict__ b, double *__restrict__ x, double *__restrict__ bc, int bcx,
                                                      int bcy, int bcz, const double *__restrict__ pin_sum = nullptr)
{
    __shared__ double A_[SM_MAXR], B_[SM_MAXR];
    __shared__ SmTabs T;
#ifdef PIB_SMALL_STAMPS
    unsigned long long st_[12]; int ns_ = 0;
#define SST() do { st_[ns_++] = wall_clock64(); } while (0)
#else
#define SST() do { } while (0)
#endif
    SST();
    const int done = (S != nullptr) ? S->done : 0;
    // the residual on the owned children and one cell around them (two above: the fourth slot of the last coarse cell's
    // restriction stencil), the iterate one cell beyond that, every earlier step one more
    const int glo = 1 + pre, ghi = 2 + pre;
    SmGeom G;
    sm_geometry(F, C, blockIdx.x, bcx, bcy, bcz, glo, ghi, G);
    if (done) return;
    SST();
    int cell[SM_NB];
    double bq[SM_NB];
    sm_cells(G, cell);
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) bq[u] = cell[u] >= 0 ? b[sm_global(F, G, cell[u])] : 0.0;
    if (pin_sum != nullptr && F.k0 == 0) {  // PINNED (level 0): effective b at global cell 0, in every region that holds it
#pragma unroll
        for (int u = 0; u < SM_NB; ++u)
            if (cell[u] >= 0 && sm_global(F, G, cell[u]) == 0) bq[u] = bq[u] - *pin_sum;
    }
    sm_stage_tabs(F, C, G, T);
    __syncthreads();
    SST();
    sm_phase<1>(F, G, T, cell, bq, glo, ghi, 0, omega, nullptr, A_, nullptr);
    __syncthreads();
    SST();
    double *cur = A_, *oth = B_;
    for (int sw = 1; sw < pre; ++sw) {
        sm_phase<2>(F, G, T, cell, bq, glo, ghi, sw, omega, cur, oth, nullptr);
        __syncthreads();
        double *t = cur; cur = oth; oth = t;
    }
    SST();
    sm_phase<3>(F, G, T, cell, bq, glo, ghi, pre, omega, cur, oth, nullptr);
    // the owned cells of the iterate
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) {
        const int c = cell[u];
        if (c < 0) continue;
        const int gx = G.lo[0] + (c & 255), gy = G.lo[1] + ((c >> 8) & 255), gz = G.lo[2] + (c >> 16);
        if (gx >= G.F0[0] && gx < G.F1[0] && gy >= G.F0[1] && gy < G.F1[1] && gz >= G.F0[2] && gz < G.F1[2])
            x[sm_global(F, G, c)] = cur[(c & 255) + G.ext[0] * ((c >> 8) & 255) + G.ext[0] * G.ext[1] * (c >> 16)];
    }
    __syncthreads();
    SST();
    // restriction: one thread per owned coarse cell, the order of the sum as in k_restrict_rows / the tail
    {
        const double *r = oth;
        const int e0 = G.I1[0] - G.I0[0], e1 = G.I1[1] - G.I0[1], e2 = G.I1[2] - G.I0[2];
        const int sy = G.ext[0], sz = G.ext[0] * G.ext[1];
        for (int t = threadIdx.x; t < e0 * e1 * e2; t += blockDim.x) {
            const int tz = t / (e0 * e1), tr = t - tz * (e0 * e1);
            const int ty = tr / e0, tx = tr - ty * e0;
            const int Ic[3] = {G.I0[0] + tx, G.I0[1] + ty, G.I0[2] + tz};
            const int Tc[3] = {tx, ty, tz};
            double w[3][4];
            int pos[3][4];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int f0 = T.fst[d][Tc[d]] - 1;
                const bool wrap = (F.tper >> d) & 1;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int fu = f0 + o;  // unwrapped: the region's coordinates
                    const int ps = min(max(fu - G.lo[d], 0), G.ext[d] - 1);
                    double wt = 0.0;
                    if (wrap || (fu >= 0 && fu < G.n[d])) {
                        if (T.par[d][ps] == Ic[d]) wt = T.wpar[d][ps];
                        else if (T.oth[d][ps] == Ic[d]) wt = T.woth[d][ps];
                    }
                    w[d][o] = wt;
                    pos[d][o] = ps;
                }
            }
            // (no skipping of zero weights: a term of zero weight adds +-0 to a sum that is never -0, the residual is finite
            // on the whole region -- and sixteen loads at a time are in flight instead of four behind a branch)
            double sum = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                double rv[4][4];
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                    for (int a2 = 0; a2 < 4; ++a2) rv[b2][a2] = r[sz * pos[2][c2] + sy * pos[1][b2] + pos[0][a2]];
                double u = 0.0;
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) {
                    double t = 0.0;
#pragma unroll
                    for (int a2 = 0; a2 < 4; ++a2) t = tacc(t, w[0][a2], rv[b2][a2]);
                    u = tacc(u, w[1][b2], t);
                }
                sum = tacc(sum, w[2][c2], u);
            }
            bc[(int64_t)Ic[0] + (int64_t)C.nx * (Ic[1] + (int64_t)C.ny * (Ic[2] - C.k0))] = sum;
        }
    }
    SST();
#ifdef PIB_SMALL_STAMPS
    // tools: geometry | one round of loads | first step | further steps | residual + iterate out | restriction; 10 ns units
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        static __device__ int launches_ = 0;
        if (atomicAdd(&launches_, 1) % 60 == 30) {
            printf("small-down n=%d,%d,%d blocks=%d blk=%d:", F.nx, F.ny, F.nzg, (int)gridDim.x, (int)blockIdx.x);
            for (int q = 1; q < ns_; ++q) printf(" %llu", st_[q] - st_[q - 1]);
            printf("\n");
        }
    }
#endif
#undef SST
}

// way up: out = `post` smoothing steps on x + P xc (owned cells)
__global__ __launch_bounds__(SM_NT) void k_small_up(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega, int post,
                                                    const double *__restrict__ b, const double *__restrict__ x, const double *__restrict__ xc,
                                                    double *__restrict__ out, int bcx, int bcy, int bcz)
{
    __shared__ double A_[SM_MAXR], B_[SM_MAXR];
    __shared__ SmTabs T;
    const int done = (S != nullptr) ? S->done : 0;
    SmGeom G;
    sm_geometry(F, C, blockIdx.x, bcx, bcy, bcz, post, post, G);
    if (done) return;
    int cell[SM_NB];
    double bq[SM_NB], xq[SM_NB];
    sm_cells(G, cell);
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) {
        const int64_t p = cell[u] >= 0 ? sm_global(F, G, cell[u]) : 0;
        bq[u] = cell[u] >= 0 ? b[p] : 0.0;
        xq[u] = cell[u] >= 0 ? x[p] : 0.0;
    }
    sm_stage_tabs(F, C, G, T);
    __syncthreads();
    // the corrected iterate on the whole region
    {
        const int64_t cplane = (int64_t)C.nx * C.ny;
#pragma unroll
        for (int u = 0; u < SM_NB; ++u) {
            const int c = cell[u];
            if (c < 0) continue;
            const int r3[3] = {c & 255, (c >> 8) & 255, c >> 16};
            double sum = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        const double wgt = ((c2 ? T.woth[2][r3[2]] : T.wpar[2][r3[2]]) * (b2 ? T.woth[1][r3[1]] : T.wpar[1][r3[1]])) *
                                           (a2 ? T.woth[0][r3[0]] : T.wpar[0][r3[0]]);
                        const int I = a2 ? T.oth[0][r3[0]] : T.par[0][r3[0]], J = b2 ? T.oth[1][r3[1]] : T.par[1][r3[1]],
                                  K = c2 ? T.oth[2][r3[2]] : T.par[2][r3[2]];
                        if (wgt != 0.0) sum = tacc(sum, wgt, xc[I + (int64_t)C.nx * J + cplane * (K - C.k0)]);
                    }
            A_[r3[0] + G.ext[0] * r3[1] + G.ext[0] * G.ext[1] * r3[2]] = xq[u] + sum;
        }
    }
    __syncthreads();
    double *cur = A_, *oth = B_;
    for (int sw = 1; sw <= post; ++sw) {
        sm_phase<2>(F, G, T, cell, bq, post, post, sw, omega, cur, oth, sw == post ? out : nullptr);
        if (sw < post) __syncthreads();
        double *t = cur; cur = oth; oth = t;
    }
}

// ------------------------------------------------------------------ host side
// halo memory / deepest exchange of a distributed level (see "halos of a distributed level" below)
constexpr int HALO_PAD_PLANES = 8;   // memory per side (the fused kernels read one plane beyond the run they process)
constexpr int HALO_MAX_DEPTH = 6;    // deepest exchange: V(2,2) needs 4 planes of the residual on level 0; below it 3, or -- when the
                                     // way up is to run without exchanges of its own (pib_deep_up) -- 5 on level 1 and 6 on level 2

static LevelDev dev_of(const GridLevel &g)
{
    LevelDev L;
    L.nx = (int)g.n[0];
    L.ny = (int)g.n[1];
    L.nzg = (int)g.n[2];
    L.k0 = (int)g.k0;
    L.nk = (int)(g.k1 - g.k0);
    L.per = g.per;
    L.tper = g.tper;
    L.zring = g.zring ? 1 : 0;
    L.wx = g.w[0];
    L.wy = g.w[1];
    L.wz = g.w[2];
    L.gx = g.g[0];
    L.gy = g.g[1];
    L.gz = g.g[2];
    L.cmx = g.cm[0];
    L.cpx = g.cp[0];
    L.rwx = g.rw[0];
    L.cmy = g.cm[1];
    L.cpy = g.cp[1];
    L.rwy = g.rw[1];
    L.cmz = g.cm[2];
    L.cpz = g.cp[2];
    L.rwz = g.rw[2];
    for (int d = 0; d < 3; ++d) L.t[d] = Tr1{g.t_par[d], g.t_oth[d], g.t_fst[d], g.t_wpar[d], g.t_woth[d]};
    L.tx = TrX{g.tx_fc, g.tx_pw, g.tx_rw};
    return L;
}

static int grid_blocks(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + 255) / 256)); }
// (x, y) = (workgroups per plane, owned planes)
static dim3 level_grid(const GridLevel &g)
{
    const int64_t plane = g.n[0] * g.n[1];
    return dim3((unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, (plane + 255) / 256)), (unsigned)std::max<int64_t>(1, g.k1 - g.k0));
}

// (+8 zero entries of padding: the level kernels read the 1-D arrays in aligned vectors of up to 4)
template <class T>
static int up(const std::vector<T> &h, T **d)
{
    PIB_HIP(hipMalloc(d, sizeof(T) * (h.size() + 8)));
    PIB_MEMSET(*d, 0, sizeof(T) * (h.size() + 8));
    if (!h.empty()) PIB_HIP(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

void gmg_release(pib_solver *s)
{
    drop_iteration_graph(s);  // (krylov.hip: the captured iteration goes before the memory it points at)
    if (s->d_tail_args) (void)hipFree(s->d_tail_args);
    s->d_tail_args = nullptr;
    if (s->d_tail_tab) (void)hipFree(s->d_tail_tab);
    s->d_tail_tab = nullptr;
    s->h_tail_args.clear();
    for (auto &L : s->levels) {
        for (int d = 0; d < 3; ++d) {
            if (L.w[d]) (void)hipFree(L.w[d]);
            if (L.g[d]) (void)hipFree(L.g[d]);
            if (L.cm[d]) (void)hipFree(L.cm[d]);
            if (L.cp[d]) (void)hipFree(L.cp[d]);
            if (L.rw[d]) (void)hipFree(L.rw[d]);
            if (L.t_par[d]) (void)hipFree(L.t_par[d]);
            if (L.t_oth[d]) (void)hipFree(L.t_oth[d]);
            if (L.t_fst[d]) (void)hipFree(L.t_fst[d]);
            if (L.t_wpar[d]) (void)hipFree(L.t_wpar[d]);
            if (L.t_woth[d]) (void)hipFree(L.t_woth[d]);
        }
        if (L.tx_fc) (void)hipFree(L.tx_fc);
        if (L.tx_pw) (void)hipFree(L.tx_pw);
        if (L.tx_rw) (void)hipFree(L.tx_rw);
        if (L.x) (void)hipFree(L.x);
        if (L.b) (void)hipFree(L.b);
        if (L.r) (void)hipFree(L.r);
        if (L.x2) (void)hipFree(L.x2);
        if (L.d) (void)hipFree(L.d);
    }
    s->levels.clear();
    s->pin_row = PinRow{};
    s->has_grid = false;
}

// halo_planes of memory below and above the owned planes (HALO_PAD_PLANES on a distributed level, 1 otherwise)
static int alloc_level_vectors(GridLevel &g, bool need_b, int halo_planes)
{
    const int64_t plane = g.n[0] * g.n[1];
    g.pad = (int64_t)halo_planes * plane;
    const size_t sz = sizeof(double) * (size_t)((g.k1 - g.k0 + 2 * halo_planes) * plane);
    PIB_HIP(hipMalloc(&g.x, sz));
    PIB_HIP(hipMalloc(&g.x2, sz));
    PIB_HIP(hipMalloc(&g.r, sz));
    PIB_MEMSET(g.x, 0, sz);
    PIB_MEMSET(g.x2, 0, sz);
    PIB_MEMSET(g.r, 0, sz);
    PIB_HIP(hipMalloc(&g.d, sz));
    PIB_MEMSET(g.d, 0, sz);
    if (need_b) {
        PIB_HIP(hipMalloc(&g.b, sz));
        PIB_MEMSET(g.b, 0, sz);
    }
    g.plane = plane;
    g.nloc = (g.k1 - g.k0) * plane;
    return 0;
}

template <int MODE>
static int launch_level(pib_solver *s, const GridLevel &g, double omega, const double *b, const double *xi, double *xo,
                        const double *pin_sum, bool guarded, hipStream_t q, double *dvec = nullptr, double a_d = 0.0);

// row-kernel geometry: 4 rows per workgroup (one per wave), row groups dealt to the 8 XCDs in contiguous ranges
struct RowGrid {
    int ngroups, per_xcd;
    dim3 grid;
};
static RowGrid row_grid(int64_t nrows, int64_t ncx, int lanes = ROW_LANES)
{
    RowGrid r;
    r.ngroups = (int)((nrows + 3) / 4);
    r.per_xcd = (r.ngroups + 7) / 8;
    r.grid = dim3((unsigned)(8 * r.per_xcd), (unsigned)((ncx + lanes - 1) / lanes));
    return r;
}
static int launch_prolong(const GridLevel &f, const GridLevel &c, const double *xc, double *xf, const Scalars *S, hipStream_t q)
{
    // the planes this launch touches, checked on the host tables (a plane outside the level, or a coarse plane outside the
    // memory the coarse vector has, is a planner error -- reported, not a memory fault)
    if (f.k1 > f.k0 && (f.k0 < 0 || f.k1 > f.n[2]))
        return fail(PIB_ERR_LIB, "gmg: prolongation onto planes [%lld, %lld) of a level with %lld", (long long)f.k0, (long long)f.k1,
                    (long long)f.n[2]);
    if (!f.hz_par.empty() && c.plane > 0) {
        const int64_t padc = c.pad / c.plane;
        for (int64_t k = f.k0; k < f.k1; ++k) {
            const int64_t lo = std::min(f.hz_par[(size_t)k], f.hz_oth[(size_t)k]), hi = std::max(f.hz_par[(size_t)k], f.hz_oth[(size_t)k]);
            if (lo < c.k0 - padc || hi >= c.k1 + padc)
                return fail(PIB_ERR_LIB, "gmg: fine plane %lld interpolates from coarse planes [%lld, %lld], the coarse slab [%lld, %lld) has %lld halo planes",
                            (long long)k, (long long)lo, (long long)hi, (long long)c.k0, (long long)c.k1, (long long)padc);
        }
    }
    const int64_t nrows = f.n[1] * (f.k1 - f.k0);
    const int vec_ok = (f.n[0] % 2 == 0 && (reinterpret_cast<uintptr_t>(xf) & 15u) == 0) ? 1 : 0;
    if (nrows >= 65536) {  // enough rows to keep the chip busy with four per wave
        const RowGrid r = row_grid((nrows + 3) / 4, c.n[0]);
        hipLaunchKernelGGL(k_prolong_rows<4>, r.grid, dim3(64, 4), 0, q, S, dev_of(f), dev_of(c), xc, xf, r.ngroups, r.per_xcd, vec_ok);
    } else {
        const RowGrid r = row_grid(nrows, c.n[0]);
        hipLaunchKernelGGL(k_prolong_rows<1>, r.grid, dim3(64, 4), 0, q, S, dev_of(f), dev_of(c), xc, xf, r.ngroups, r.per_xcd, vec_ok);
    }
    PIB_HIP(hipGetLastError());
    return 0;
}
// `c` carries the coarse planes to produce in k0 / k1 (the owned ones, which may be a part of a replicated level)
// the small-level kernels' box of coarse cells per workgroup (k_small_down: steps = pre, k_small_up: steps = post); false:
// the level does not qualify
constexpr int64_t SMALL_LEVEL_CELLS = 300000, SMALL_LEVEL_CELLS_3D = 40000;  // levels of at most this many cells (3-D: the margins cost more)
static bool small_level_boxes(const pib_solver *s, const GridLevel &f, const GridLevel &c, int steps, bool down, int bc[3], unsigned *blocks)
{
    if (!s->cfg.fuse_small_levels || steps < 1 || steps > 2) return false;
    if (f.nloc > SMALL_LEVEL_CELLS || f.zring || c.zring) return false;
    if (f.k0 != 0 || f.k1 != f.n[2] || c.k0 != 0 || c.k1 != c.n[2]) return false;  // both levels whole on this rank
    if (f.per != f.tper) return false;
    int nt = 0;
    for (int d = 0; d < 3; ++d) nt += f.n[d] > 1 ? 1 : 0;
    // a 3-D level pays 5-13 x in recomputed margins, all of it instruction issue on the workgroup's one CU
    if (nt == 3 && f.nloc > SMALL_LEVEL_CELLS_3D) return false;
    // boxes of 4^3 coarse cells on a 3-D level; 8^2 on a 2-D one while that gives at most one workgroup per CU (the kernels
    // hold one workgroup per CU: a second round of workgroups doubles the launch's time), else 16^2
    for (int side = (nt == 3 ? 4 : 8);; side *= 2) {
        int64_t cells = 1, nb = 1;
        bool ok = true;
        for (int d = 0; d < 3; ++d) {
            if (f.t_fst[d] == nullptr || f.t_par[d] == nullptr) return false;
            bc[d] = f.n[d] > 1 ? side : 1;
            int e = 2 * bc[d] + (down ? 3 + 2 * steps : 2 * steps);  // aggregates of at most two cells
            if (!((f.per >> d) & 1)) e = (int)std::min<int64_t>(e, f.n[d]);
            if (e > SM_MAXE) ok = false;
            cells *= e;
            nb *= (c.n[d] + bc[d] - 1) / bc[d];
        }
        if (!ok || cells > SM_MAXR) return false;
        if (nb <= 240 || nt == 3 || side >= SM_MAXBC) {
            if (nb > 256) return false;  // (more workgroups than CUs: the per-phase launches are faster)
            *blocks = (unsigned)nb;
            return true;
        }
    }
}

static int launch_restrict(const pib_solver *s, const GridLevel &f, const GridLevel &c, const double *rf, double *bc, const Scalars *S,
                           hipStream_t q)
{
    const int64_t nkc = c.k1 - c.k0;
    // transfers across a periodic z seam: the whole fine level is on this rank (the slab axis of a distributed level never wraps)
    const bool z_ok = !(f.tper & 4) || (f.k0 == 0 && f.k1 == f.n[2]);
    if (s->cfg.march_restrict && f.plain_pair && z_ok && f.n[0] % RX == 0 && f.n[1] % RY == 0 && nkc >= 4 &&
        nkc * c.plane * 8 >= (int64_t)s->cfg.march_min_cells) {
        // coarse planes per workgroup: 32 on a 512^3 fine level (1024 workgroups), 8 below
        const int CZ = nkc * c.plane >= ((int64_t)1 << 23) ? 32 : 8;
        hipLaunchKernelGGL(k_restrict_march, dim3((unsigned)(f.n[0] / RX), (unsigned)(f.n[1] / RY), (unsigned)((nkc + CZ - 1) / CZ)),
                           dim3(256), 0, q, S, dev_of(f), dev_of(c), rf, bc, CZ);
        PIB_HIP(hipGetLastError());
        return 0;
    }
    const int vec_ok = (f.n[0] % 2 == 0 && (reinterpret_cast<uintptr_t>(rf) & 15u) == 0) ? 1 : 0;
    // any aggregation, the level whole on this rank, no periodic z seam, 3-D, large enough to fill the chip with one wave per coarse
    // row and z-chunk: the z-marching form
    if (s->cfg.march_restrict && !(f.tper & 4) && f.k0 == 0 && f.k1 == f.n[2] && c.k0 == 0 && c.k1 == c.n[2] && f.n[1] > 1 && c.n[2] >= 4 &&
        c.n[0] * c.n[1] * c.n[2] >= std::min<int64_t>((int64_t)1 << 17, s->cfg.march_min_cells)) {  // (the tests lower the bound)
        // coarse planes per workgroup: so that there are about four workgroups per CU
        const int64_t wg_plane = ((c.n[1] + 3) / 4) * ((c.n[0] + 63) / 64);
        int CZ = (int)std::max<int64_t>(2, std::min<int64_t>(32, c.n[2] * wg_plane / 1024));
        hipLaunchKernelGGL(k_restrict_zmarch, dim3((unsigned)((c.n[1] + 3) / 4), (unsigned)((c.n[0] + 63) / 64), (unsigned)((c.n[2] + CZ - 1) / CZ)),
                           dim3(256), 0, q, S, dev_of(f), dev_of(c), rf, bc, CZ, vec_ok);
        PIB_HIP(hipGetLastError());
        return 0;
    }
    const RowGrid r = row_grid(c.n[1] * (c.k1 - c.k0), c.n[0], 64);  // aligned 64-lane chunks + edge loads (62 overlapping lanes measured slower here)
    hipLaunchKernelGGL(k_restrict_rows, r.grid, dim3(64, 4), 0, q, S, dev_of(f), dev_of(c), rf, bc, r.ngroups, r.per_xcd, vec_ok);
    PIB_HIP(hipGetLastError());
    return 0;
}

// ---- hint verification: stencil twin vs CSR SpMV on a fixed pseudo-random vector
// skip0: the pinned convention (row/column 0 of the CSR replaced by the identity) is the one place where
// the CSR and the singular stencil differ by design: x[0] = 0 removes column 0, row 0 is left out.
__global__ void k_fill_hash(int64_t n, int64_t g0, double *x, int skip0)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = (uint64_t)(g0 + p) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        x[p] = (skip0 && g0 + p == 0) ? 0.0 : (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}
__global__ void k_diff_sums(int64_t n, int64_t g0, int skip0, const double *a, const double *b, double *out /* [2] */)
{
    double d2 = 0.0, b2 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        if (skip0 && g0 + p == 0) continue;
        const double d = a[p] - b[p];
        d2 += d * d;
        b2 += b[p] * b[p];
    }
    for (int o = 32; o > 0; o >>= 1) {
        d2 += __shfl_down(d2, o, 64);
        b2 += __shfl_down(b2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], d2);
        atomicAdd(&out[1], b2);
    }
}

int gmg_verify(pib_solver *s)
{
    PIB_CHK(ensure_work(s, 4));
    hipStream_t q = s->stream;
    const int64_t n = s->A.n;
    double *X = s->vec(2), *Y1 = s->vec(3), *Y2 = s->vec(0);
    const int skip0 = (s->nullspace == PIB_NULLSPACE_PINNED) ? 1 : 0;
    hipLaunchKernelGGL(k_fill_hash, dim3(grid_blocks(n)), dim3(256), 0, q, n, s->A.row0, X, skip0);
    if (s->comm.nranks > 1) PIB_CHK(halo_exchange(s, X, q));
    PIB_CHK(spmv_rows(s, X, Y1, 0, n, nullptr, false, q));
    PIB_CHK(launch_level<0>(s, s->levels[0], 0.0, nullptr, X, Y2, nullptr, false, q));
    double *d_out = nullptr;
    PIB_HIP(hipMalloc(&d_out, 2 * sizeof(double)));
    PIB_HIP(hipMemsetAsync(d_out, 0, 2 * sizeof(double), q));
    hipLaunchKernelGGL(k_diff_sums, dim3(grid_blocks(n)), dim3(256), 0, q, n, s->A.row0, skip0, Y2, Y1, d_out);
    PIB_CHK(comm_allreduce_sum(s, d_out, 2, q));  // the same verdict on every rank
    double h[2] = {0, 0};
    PIB_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    PIB_HIP(hipFree(d_out));
    const double tol = 1e-10;
    if (!(h[0] <= tol * tol * h[1]) ) {
        gmg_release(s);
        return fail(PIB_ERR_ARG_WRONG,
                    "solver %s: the grid hint does not describe the matrix (stencil vs CSR mismatch %.3e relative)",
                    s->name.c_str(), std::sqrt(h[0] / (h[1] > 0 ? h[1] : 1.0)));
    }
    return 0;
}

// Register the structure of the matrix (pib_set_grid_hint / pib_assemble_poisson)
// and build the level hierarchy.  w[d]: n[d] widths; g[d]: n[d]-1 face factors (dt included).
int grid_register(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double *const g[3],
                  int nullspace, double dt_in)
{
    gmg_release(s);
    s->nullspace = nullspace;
    s->gmg_error.clear();
    if (dim != 2 && dim != 3) return fail(PIB_ERR_ARG_OUTOFRANGE, "grid hint: dim must be 2 or 3");
    const int P = s->comm.nranks, rank = s->comm.rank;
    // internal layout (nx, ny, nz) with 2-D -> (nx, 1, ny)
    int64_t nn[3];
    std::vector<double> hw[3], hg[3];
    bool pern[3] = {false, false, false};
    const int map[3] = {0, (dim == 3) ? 1 : -1, (dim == 3) ? 2 : 1};
    for (int d = 0; d < 3; ++d) {
        if (map[d] < 0) {
            nn[d] = 1;
            hw[d] = {1.0};
            hg[d].clear();
        } else {
            nn[d] = n[map[d]];
            if (w[map[d]] == nullptr || (nn[d] > 1 && g[map[d]] == nullptr)) return fail(PIB_ERR_ARG_NULL, "grid hint: null array");
            hw[d].assign(w[map[d]], w[map[d]] + nn[d]);
            // a periodic direction has one face more: g[n-1] couples cell n-1 and cell 0
            pern[d] = s->periodic[map[d]] != 0 && nn[d] > 1;
            hg[d].assign(g[map[d]], g[map[d]] + (nn[d] - 1) + (pern[d] ? 1 : 0));
        }
    }
    if (pern[2] && P > 1 && !s->comm.ring)
        return fail(PIB_ERR_SUP, "grid hint: a periodic slab axis on several ranks needs the on-device assembly (pib_assemble_poisson)");
    if (nn[0] * nn[1] * nn[2] != s->A.n_global)
        return fail(PIB_ERR_ARG_SIZ, "grid hint: %lld x %lld x %lld cells but the matrix has %lld rows", (long long)nn[0],
                    (long long)nn[1], (long long)nn[2], (long long)s->A.n_global);
    // slab of this rank must match the matrix rows
    int64_t k0, k1;
    slab_range(nn[2], P, rank, &k0, &k1);
    const int64_t plane0 = nn[0] * nn[1];
    if (k0 * plane0 != s->A.row0 || (k1 - k0) * plane0 != s->A.n)
        return fail(PIB_ERR_ARG_WRONG, "grid hint: the matrix rows of rank %d are not the z-slab [%lld,%lld) of the grid", rank,
                    (long long)k0, (long long)k1);

    // dt is folded in g; coarse g needs dt back: g = dt/dl  ->  dt = g[0]*0.5*(w0+w1) (any direction with >= 2 cells)
    double dt = (dt_in > 0.0) ? dt_in : 0.0;
    for (int d = 0; d < 3 && dt == 0.0; ++d)
        if (nn[d] > 1) dt = hg[d][0] * (0.5 * (hw[d][1] + hw[d][0]));

    // PINNED: row 0 of the singular operator without its diagonal (what MatZeroRowsColumns took out of the matrix): the solver's
    // recurrence for the residual's sum reads the few entries of p next to cell 0 with these coefficients (krylov.hip cg_s1)
    s->pin_row = PinRow{};
    if (nullspace == PIB_NULLSPACE_PINNED) {
        PinRow &pr = s->pin_row;
        pr.ready = true;
        if (k0 == 0) {
            const double vol = (hw[0][0] * hw[1][0]) * hw[2][0];
            const int64_t stride[3] = {1, nn[0], plane0};
            for (int d = 0; d < 3; ++d) {
                const int64_t nd = nn[d];
                if (nd <= 1) continue;
                const double w0 = hw[d][0];
                pr.off[pr.n] = stride[d];  // towards +d: face 0
                pr.coef[pr.n++] = (hg[d][0] / w0) * vol;
                if (pern[d]) {  // towards -d: the wrap face, cell nd - 1 (through the low halo plane on a ring of slabs)
                    pr.off[pr.n] = (d == 2 && P > 1) ? -plane0 : stride[d] * (nd - 1);
                    pr.coef[pr.n++] = (hg[d][(size_t)nd - 1] / w0) * vol;
                }
            }
        }
    }
    std::vector<GridLevel> &lv = s->levels;
    const int max_levels = std::max(1, s->cfg.max_levels);
    bool replicated = (P == 1);
    s->gmg_own.clear();
    {
        std::vector<std::pair<int64_t, int64_t>> o0((size_t)P);
        for (int r = 0; r < P; ++r) slab_range(nn[2], P, r, &o0[(size_t)r].first, &o0[(size_t)r].second);
        s->gmg_own.push_back(o0);
    }
    // Selective coarsening (oracle/csrc/gmg.c:orc_gmg_create): walking the cells of a direction, two neighbours
    // merge only if their combined width is <= 1.5 * hmin * 2^(level+1); cells that are already larger stay
    // alone, so the stretched far field of a PetIBM mesh catches up with the refined region and the level grids
    // become uniform (plain pairing on a uniform mesh).  On a distributed level the z aggregates never straddle
    // a slab boundary, so every coarse plane has one owner and one halo plane serves both transfers.
    double hmin = 0.0;
    for (int d = 0; d < 3; ++d)
        if (nn[d] > 1)
            for (double v : hw[d])
                if (hmin == 0.0 || v < hmin) hmin = v;
    int target_shift = 0;
    for (int l = 0;; ++l) {
        GridLevel G;
        G.dim = dim;
        for (int d = 0; d < 3; ++d) G.n[d] = nn[d];
        if (replicated) {
            G.k0 = 0;
            G.k1 = nn[2];
        } else {
            G.k0 = k0;
            G.k1 = k1;
        }
        G.replicated = replicated && P > 1;
        for (int d = 0; d < 3; ++d) {
            PIB_CHK(up(hw[d], &G.w[d]));
            const bool wrap = pern[d] && nn[d] > 1;  // a direction coarsened down to one cell has no face left
            if (wrap) G.per |= 1 << d;
            if (wrap && d == 2 && !replicated && P > 1) G.zring = true;
            if (l > 0) {
                hg[d].assign((size_t)std::max<int64_t>(nn[d] - 1, 0) + (wrap ? 1 : 0), 0.0);
                for (int64_t q = 0; q + 1 < nn[d]; ++q) {
                    const double dl = 0.5 * (hw[d][(size_t)q + 1] + hw[d][(size_t)q]);
                    const double v = 1.0 / dl;
                    hg[d][(size_t)q] = dt * v;
                }
                if (wrap) {
                    const double dl = 0.5 * (hw[d][0] + hw[d][(size_t)nn[d] - 1]);
                    const double v = 1.0 / dl;
                    hg[d][(size_t)nn[d] - 1] = dt * v;
                }
            }
            PIB_CHK(up(hg[d], &G.g[d]));
            {
                // rows divided by the cell volume: g / w per face, a function of the index along d only
                const int64_t nd = nn[d];
                std::vector<double> cm((size_t)nd, 0.0), cp((size_t)nd, 0.0), rw((size_t)nd, 1.0);
                for (int64_t q = 0; q < nd; ++q) {
                    const double wq = hw[d][(size_t)q];
                    rw[(size_t)q] = 1.0 / wq;
                    if (nd > 1) {
                        if (q > 0) cm[(size_t)q] = hg[d][(size_t)q - 1] / wq;
                        else if (wrap) cm[(size_t)q] = hg[d][(size_t)nd - 1] / wq;
                        if (q < nd - 1 || wrap) cp[(size_t)q] = hg[d][(size_t)q] / wq;
                    }
                }
                PIB_CHK(up(cm, &G.cm[d]));
                PIB_CHK(up(cp, &G.cp[d]));
                PIB_CHK(up(rw, &G.rw[d]));
            }
        }
        PIB_CHK(alloc_level_vectors(G, l > 0, (P > 1 && !replicated) ? HALO_PAD_PLANES : 1));
        const bool last = (l + 1 >= max_levels) || (nn[0] <= 2 && nn[1] <= 2 && nn[2] <= 2);
        // next level: aggregates and transfer tables
        std::vector<int32_t> par[3], oth[3], fst[3];
        std::vector<double> wpar[3], woth[3], cw[3];
        int64_t nc[3] = {nn[0], nn[1], nn[2]};
        bool twrap[3] = {false, false, false};
        bool merged_any = false;
        const auto &of = s->gmg_own.back();
        for (int tries = 0; !last && tries < 64 && !merged_any; ++tries, ++target_shift) {
            const double target = 1.5 * hmin * std::ldexp(1.0, l + 1 + target_shift);
            for (int d = 0; d < 3; ++d) {
                const int64_t n = nn[d];
                // the four fine cells a coarse cell gathers from must be distinct; across ranks the seam of the slab
                // axis stays a wall for the transfers (the halo planes serve the level operator only)
                twrap[d] = pern[d] && n >= 4 && !(d == 2 && !replicated && P > 1);
                par[d].assign((size_t)n, 0);
                oth[d].assign((size_t)n, 0);
                wpar[d].assign((size_t)n, 1.0);
                woth[d].assign((size_t)n, 0.0);
                fst[d].clear();
                cw[d].clear();
                std::vector<char> slab_start((size_t)n + 1, 0);
                if (d == 2 && !replicated)
                    for (int r = 0; r < P; ++r) slab_start[(size_t)of[(size_t)r].first] = 1;
                int64_t I = 0;
                for (int64_t q = 0; q < n; ++I) {
                    fst[d].push_back((int32_t)q);
                    if (n > 2 && q + 1 < n && !slab_start[(size_t)q + 1] && hw[d][(size_t)q] + hw[d][(size_t)q + 1] <= target) {
                        par[d][(size_t)q] = par[d][(size_t)q + 1] = (int32_t)I;
                        cw[d].push_back(hw[d][(size_t)q] + hw[d][(size_t)q + 1]);
                        q += 2;
                        merged_any = true;
                    } else {
                        par[d][(size_t)q] = (int32_t)I;
                        cw[d].push_back(hw[d][(size_t)q]);
                        q += 1;
                    }
                }
                fst[d].push_back((int32_t)n);
                nc[d] = I;
                for (int64_t q = 0; q < n; ++q) {
                    const int64_t Pq = par[d][(size_t)q];
                    const int64_t f0 = fst[d][(size_t)Pq], f1 = fst[d][(size_t)Pq + 1];
                    int64_t O = Pq;
                    double t = 0.0;
                    if (f1 - f0 == 2) {
                        const bool left = (q == f0);
                        O = left ? Pq - 1 : Pq + 1;
                        if (twrap[d]) O = (O + I) % I;  // across the periodic seam
                        if (O < 0 || O >= I)
                            O = Pq;
                        else {
                            const double sib = left ? hw[d][(size_t)q + 1] : hw[d][(size_t)q - 1];
                            t = sib / (cw[d][(size_t)Pq] + cw[d][(size_t)O]);
                        }
                    }
                    oth[d][(size_t)q] = (int32_t)O;
                    wpar[d][(size_t)q] = 1.0 - t;
                    woth[d][(size_t)q] = t;
                }
            }
        }
        if (last || !merged_any) {
            lv.push_back(G);
            break;
        }
        target_shift--;  // the loop's ++ after the successful try
        for (int d = 0; d < 3; ++d)
            if (twrap[d]) G.tper |= 1 << d;
        G.plain_pair = nc[0] * 2 == nn[0] && nc[1] * 2 == nn[1] && nc[2] * 2 == nn[2];
        G.hz_par = par[2];
        G.hz_oth = oth[2];
        for (int d = 0; d < 3; ++d) {
            PIB_CHK(up(par[d], &G.t_par[d]));
            PIB_CHK(up(oth[d], &G.t_oth[d]));
            PIB_CHK(up(fst[d], &G.t_fst[d]));
            PIB_CHK(up(wpar[d], &G.t_wpar[d]));
            PIB_CHK(up(woth[d], &G.t_woth[d]));
        }
        {
            // x tables packed per coarse cell for the row kernels
            std::vector<int2> fc((size_t)nc[0]);
            std::vector<double4> pw((size_t)nc[0]), rw((size_t)nc[0]);
            for (int64_t I = 0; I < nc[0]; ++I) {
                const int f0 = fst[0][(size_t)I], cnt = fst[0][(size_t)I + 1] - f0;
                int fl = f0 - 1, fr = f0 + cnt;
                if (twrap[0]) {
                    if (fl < 0) fl = (int)nn[0] - 1;
                    if (fr >= nn[0]) fr = 0;
                }
                fc[(size_t)I] = make_int2(f0, cnt);
                pw[(size_t)I] = make_double4(wpar[0][(size_t)f0], woth[0][(size_t)f0], cnt == 2 ? wpar[0][(size_t)f0 + 1] : 0.0,
                                             cnt == 2 ? woth[0][(size_t)f0 + 1] : 0.0);
                const double wl = (fl >= 0 && par[0][(size_t)fl] != I && oth[0][(size_t)fl] == I) ? woth[0][(size_t)fl] : 0.0;
                const double wr = (fr < nn[0] && par[0][(size_t)fr] != I && oth[0][(size_t)fr] == I) ? woth[0][(size_t)fr] : 0.0;
                rw[(size_t)I] = make_double4(wl, wpar[0][(size_t)f0], cnt == 2 ? wpar[0][(size_t)f0 + 1] : 0.0, wr);
            }
            PIB_CHK(up(fc, &G.tx_fc));
            PIB_CHK(up(pw, &G.tx_pw));
            PIB_CHK(up(rw, &G.tx_rw));
        }
        lv.push_back(G);
        {
            // ownership of the coarse planes = owner of their children (recorded also for the level at which the
            // hierarchy switches to replicated: it defines who restricts what before the all-gather)
            std::vector<std::pair<int64_t, int64_t>> oc((size_t)P);
            for (int r = 0; r < P; ++r) {
                const int64_t b = of[(size_t)r].first, e = of[(size_t)r].second;
                if (replicated || e <= b)
                    oc[(size_t)r] = {0, nc[2]};
                else
                    oc[(size_t)r] = {par[2][(size_t)b], (int64_t)par[2][(size_t)e - 1] + 1};
            }
            if (!replicated) {
                // stay distributed only while every rank keeps >= 2 coarse planes and the level is big enough
                // to amortise the halo latency
                bool ok = true;
                for (int r = 0; r < P && ok; ++r)
                    if (oc[(size_t)r].second - oc[(size_t)r].first < 2) ok = false;
                if (nc[0] * nc[1] * nc[2] <= (int64_t)s->cfg.agglomerate_below) ok = false;
                if (ok) {
                    k0 = oc[(size_t)rank].first;
                    k1 = oc[(size_t)rank].second;
                } else {
                    replicated = true;
                }
            }
            s->gmg_own.push_back(oc);
        }
        for (int d = 0; d < 3; ++d) {
            hw[d].swap(cw[d]);
            nn[d] = nc[d];
        }
    }
    // every rank derives the same aggregates and ownership from the same width arrays, so neighbours agree.
    s->has_grid = true;
    // the Krylov work vectors double as level-0 vectors: the same halo memory around their owned part
    s->work_pad = (P > 1) ? HALO_PAD_PLANES * plane0 : 0;

    // verify the hint against the CSR: stencil twin vs CSR SpMV on a fixed vector (not when the structure only describes
    // the preconditioner's operator: BN order > 1, bn.hip)
    if (s->hint_pc_only) return 0;
    return gmg_verify(s);
}

// ---- halos of a distributed level ------------------------------------------------------------------------------
// A z-slab keeps up to HALO_PAD_PLANES planes of memory below and above its owned planes.  An exchange fills the first
// `depth` of them with the neighbours' owned planes (contiguous: no pack kernel); every stencil kernel that follows may
// then run on the owned planes PLUS the ghost planes whose inputs are still valid -- each application of the 7-point
// stencil costs one plane of validity per side.  The ghost values a rank computes are the bits its neighbour computes
// for the same cells (same kernels, same expressions), so a cycle on P slabs is the cycle on one rank; what it saves
// is messages: one exchange of the right-hand side per level on the way down (deep enough for the smoothing steps, the
// residual and the restriction's reach), one of the coarse correction per level on the way up, instead of one per kernel.

static int exchange_planes(pib_solver *s, const GridLevel &g, double *x_owned, int depth, hipStream_t q)
{
    const int r = s->comm.rank, P = s->comm.nranks;
    const bool ring = s->comm.ring;
    const int64_t cnt = (int64_t)depth * g.plane;
    const int64_t lo = (r > 0 || ring) ? cnt : 0, hi = (r < P - 1 || ring) ? cnt : 0;
    return halo_exchange_planes(s, x_owned, g.nloc, lo, hi, lo, hi, q);
}

// all-gather the owned coarse planes of level `lc` (ownership = the parents of the finer level's slab planes) into the
// replicated level vector
static int gather_level(pib_solver *s, int lc, int64_t coarse_plane, const double *owned, int64_t n_owned,
                        double *full_owned_base, hipStream_t q)
{
    const int P = s->comm.nranks;
    std::vector<int64_t> cnt((size_t)P), off((size_t)P);
    for (int r = 0; r < P; ++r) {
        const auto &o = s->gmg_own[(size_t)lc][(size_t)r];
        cnt[(size_t)r] = (o.second - o.first) * coarse_plane;
        off[(size_t)r] = o.first * coarse_plane;
    }
    if (cnt[(size_t)s->comm.rank] != n_owned) return fail(PIB_ERR_LIB, "gmg gather: inconsistent slab sizes");
    return comm_allgatherv(s, owned, full_owned_base, cnt, off, q);
}

// runs of planes the LDS-tiled kernels serve (k_presmooth2, k_level_march, k_prolong_smooth): 3-D, tile-divisible, enough
// cells in the run to fill the chip; a periodic z only on the whole level (plane -1 = plane nz - 1 of the same vector)
static bool tiles_ok(const GridLevel &g) { return g.n[1] > 1 && g.n[2] > 1 && g.n[0] % FX == 0 && g.n[1] % FY == 0; }
static bool run_whole(const GridLevel &g, int64_t kb, int64_t kc) { return g.k0 + kb == 0 && kc == g.n[2]; }
static bool march_run_ok(const pib_solver *s, const GridLevel &g, int64_t kb, int64_t kc)
{
    const bool z_ok = !(g.per & 4) || (!g.zring && run_whole(g, kb, kc));
    return z_ok && tiles_ok(g) && kc >= 8 && kc * g.plane >= (int64_t)s->cfg.march_min_cells;
}
// the fused kernels (two pre-smoothing steps; prolongation + first post-smoothing step)
static bool fused_run_ok(const pib_solver *s, const GridLevel &g, int64_t kb, int64_t kc)
{
    const bool z_ok = !(g.per & 4) || (!g.zring && run_whole(g, kb, kc));
    return z_ok && tiles_ok(g) && kc >= 2 && kc * g.plane >= (int64_t)s->cfg.march_min_cells;
}
// planes per workgroup: 64 on a 512^3 run (2048 workgroups), 16 on a 256^3 one (1024)
#ifndef PIB_MARCH_PLANES_BIG
#define PIB_MARCH_PLANES_BIG 64
#endif
static int march_planes(const GridLevel &g, int64_t kc)
{
    // (PIB_MARCH_PLANES_SMALL: planes per workgroup on runs below 2^23 cells -- the 2 M-cell levels under a slab, which the marches
    // only reach when pib_march_min_cells is lowered; an experiment knob, profiles/r05_slab8_iteration_timeline.md)
    static const int small_planes = std::getenv("PIB_MARCH_PLANES_SMALL") ? std::max(2, std::atoi(std::getenv("PIB_MARCH_PLANES_SMALL"))) : 16;
    if (kc * g.plane >= ((int64_t)1 << 26)) return PIB_MARCH_PLANES_BIG;
    return kc * g.plane < ((int64_t)1 << 23) ? small_planes : 16;
}

// The Krylov sums z.r, z.z, sum z a level-0 kernel left as per-workgroup partials: their fixed-order reduction into S->red[0..2].
// Round 5: when the solver asks for it (gmg_defer_dots) and the partials are few enough for ONE workgroup, the reduction is left
// to the solver's own closing kernel of the cycle (krylov.hip k_dots_tail: the three sums, z[0] and -- on one rank -- the scalar
// step of the iteration in a single launch instead of four).
constexpr int DEFER_DOTS_MAX = 32768;
static int reduce_dots(pib_solver *s, double *part, int part_stride, int count, hipStream_t q)
{
    if (s->gmg_defer_dots && count <= DEFER_DOTS_MAX) {
        s->gmg_pending_part = part;
        s->gmg_pending_stride = part_stride;
        s->gmg_pending_count = count;
        return 0;
    }
    double *stage = part + 3 * (int64_t)part_stride;
    hipLaunchKernelGGL(k_reduce_big, dim3(BIG_STAGE, 3), dim3(256), 0, q, s->d_s, part, part_stride, count, stage);
    hipLaunchKernelGGL(k_finalize_big, dim3(3), dim3(64), 0, q, s->d_s, stage);
    PIB_HIP(hipGetLastError());
    return 0;
}

// MODE on the planes [kb, kb + kc) relative to the first owned plane (kb < 0 / kb + kc > nk: ghost planes); the vectors
// point at the first OWNED plane.  dots: mode 8 sums over the owned planes only.
template <int MODE>
static int launch_level_planes(pib_solver *s, const GridLevel &g, int64_t kb, int64_t kc, double omega, const double *b,
                               const double *xi, double *xo, const double *pin_sum, bool guarded, hipStream_t q,
                               double *dvec = nullptr, double a_d = 0.0)
{
    if (kc <= 0) return 0;
    GridLevel sub = g;
    sub.k0 = g.k0 + kb;
    sub.k1 = sub.k0 + kc;
    const int64_t o = kb * g.plane;
    b = b ? b + o : b;
    xi = xi ? xi + o : xi;
    xo += o;
    dvec = dvec ? dvec + o : dvec;
    // owned planes inside the run (local indices of the run)
    const int dlo = (int)std::max<int64_t>(0, -kb), dhi = (int)std::min<int64_t>(kc, (g.k1 - g.k0) - kb);
    double *part = nullptr;
    int part_stride = 0;
    const int64_t nx = g.n[0], ny = g.n[1];
    const unsigned nk = (unsigned)kc;
    const bool march = (MODE == 2 || MODE == 3 || MODE == 8) && s->cfg.march_levels && march_run_ok(s, g, kb, kc) &&
                       ((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(xi) | reinterpret_cast<uintptr_t>(xo)) & 31u) == 0;
    const int FZ = march_planes(g, kc);
    const dim3 mg((unsigned)(nx / FX), (unsigned)(ny / FY), (unsigned)((nk + FZ - 1) / FZ));
    auto aligned = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec_ok = aligned(b) && aligned(xi) && aligned(xo) && aligned(dvec);
    const int C = (vec_ok && nx % 4 == 0) ? 4 : ((vec_ok && nx % 2 == 0) ? 2 : 1);
    const dim3 sg((unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, (nx / C * ny + 255) / 256)), nk);
    if (MODE == 8) {
        // per-workgroup partials of the fused sums
        const int64_t cap = march ? (int64_t)mg.x * mg.y * mg.z : (int64_t)sg.x * sg.y;
        if (s->gmg_part_cap < cap) {
            if (s->d_gmg_part) PIB_HIP(hipFree(s->d_gmg_part));
            s->d_gmg_part = nullptr;
            PIB_HIP(hipMalloc(&s->d_gmg_part, sizeof(double) * (3 * (size_t)cap + 3 * BIG_STAGE)));
            s->gmg_part_cap = cap;
        }
        part = s->d_gmg_part;
        part_stride = (int)s->gmg_part_cap;
    }
    const Scalars *S = guarded ? s->d_s : nullptr;
    int nparts = 0;
    if (march) {
        constexpr int M = (MODE == 3) ? 3 : (MODE == 8 ? 8 : 2);
        hipLaunchKernelGGL((k_level_march<M>), mg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, part, part_stride, FZ, dlo, dhi);
        nparts = (int)(mg.x * mg.y * mg.z);
    } else {
        if (C == 4)
            hipLaunchKernelGGL((k_level<MODE, 4>), sg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, dvec, a_d, part, part_stride, dlo, dhi);
        else if (C == 2)
            hipLaunchKernelGGL((k_level<MODE, 2>), sg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, dvec, a_d, part, part_stride, dlo, dhi);
        else
            hipLaunchKernelGGL((k_level<MODE, 1>), sg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, dvec, a_d, part, part_stride, dlo, dhi);
        nparts = (int)(sg.x * sg.y);
    }
    PIB_HIP(hipGetLastError());
    if (MODE == 8) PIB_CHK(reduce_dots(s, part, part_stride, nparts, q));
    return 0;
}

// the whole owned part of the level
template <int MODE>
static int launch_level(pib_solver *s, const GridLevel &g, double omega, const double *b, const double *xi, double *xo,
                        const double *pin_sum, bool guarded, hipStream_t q, double *dvec, double a_d)
{
    return launch_level_planes<MODE>(s, g, 0, g.k1 - g.k0, omega, b, xi, xo, pin_sum, guarded, q, dvec, a_d);
}

// coarse ghost planes (beyond a rank's owned coarse planes) that the interpolation of its boundary planes and of `e` fine
// ghost planes per side reads -- the maximum over ALL ranks, so that every rank asks for the same exchange depth
static int coarse_need(const pib_solver *s, int l, int e)
{
    const GridLevel &f = s->levels[(size_t)l];
    int need = 0;
    for (int r = 0; r < s->comm.nranks; ++r) {
        const auto &of = s->gmg_own[(size_t)l][(size_t)r];
        const auto &oc = s->gmg_own[(size_t)l + 1][(size_t)r];
        if (of.second <= of.first) continue;
        for (int g2 = 0; g2 <= e; ++g2) {
            const int64_t up = of.second - 1 + g2, dn = of.first - g2;
            if (up < f.n[2]) need = std::max<int>(need, (int)(std::max(f.hz_par[(size_t)up], f.hz_oth[(size_t)up]) - (oc.second - 1)));
            if (dn >= 0) need = std::max<int>(need, (int)(oc.first - std::min(f.hz_par[(size_t)dn], f.hz_oth[(size_t)dn])));
        }
    }
    return std::max(need, 0);
}

// Several ranks: the fused residual update changes WHICH vector travels (w instead of r) and where the reductions sit, so all
// ranks must take it or none.  Slabs differ by a plane (512 x 512 x 380 on 8 ranks: 47 or 48 planes, either side of the 12.58 M-cell
// threshold): the conditions are evaluated for EVERY rank's slab -- plane counts, the cell threshold, the partial-sum slots, the
// captured-graph limit -- from quantities all ranks hold (gmg_own); what is left to the launch site (pointer alignment, halo depths)
// is the same on every rank by construction (pads and strides are multiples of the level's plane and of 4 doubles).
static bool fused_update_slabs_all_ranks(const pib_solver *s, const GridLevel &g)
{
    for (int q2 = 0; q2 < s->comm.nranks; ++q2) {
        const int64_t nk = s->gmg_own[0][(size_t)q2].second - s->gmg_own[0][(size_t)q2].first;
        if (nk < 8 || nk * g.plane <= s->cfg.graph_max_rows) return false;
        const int FZ = march_planes(g, nk);
        if (!fused_run_ok(s, g, 0, nk) || (g.n[0] / FX) * (g.n[1] / FY) * ((nk + FZ - 1) / FZ + 2) > PIB_MAXPART) return false;
    }
    return true;
}

// One V-cycle: z = M^-1 r.   r, z: ghost-padded work vectors of the Krylov solver
// (their ghost planes double as the level-0 halo planes).
// PCG may leave its residual update to the level-0 pre-smoothing march (k_presmooth2<., 1>): one rank, the fused march serves
// the whole fine level, Jacobi smoothing, no pinned unknown, and few enough workgroups for the solver's partial-sum slots
bool gmg_fused_update_ok(const pib_solver *s)
{
    if (!s->has_grid || s->levels.empty() || !s->gmg_error.empty()) return false;
    if (s->cfg.smoother == Smoother::CHEBYSHEV || !s->cfg.fuse_presmooth) return false;
    // (a pinned pressure row: the compatible right-hand side needs the NEW residual's sum before the march that forms it --
    // it comes from the recurrence sum r - alpha sum w, krylov.hip cg_s1, when the row next to cell 0 is known: pin_row)
    if (s->nullspace == PIB_NULLSPACE_PINNED && (s->cfg.pin_sum_local == 0 || !s->pin_row.ready)) return false;
    if (s->levels.size() < 2) return false;
    const GridLevel &g = s->levels[0];
    const int64_t nk = g.k1 - g.k0;
    const int FZ = march_planes(g, nk);
    if (s->comm.nranks > 1) {
        // z-slabs (round 4): level 0 distributed with deep halos, every rank's slab thick enough for them, the two-step march
        // (the cycle decides again with its own predicate at the launch site and falls back to the separate pass if it must)
        if (!s->cfg.fuse_residual_update_slabs || !s->cfg.deep_halo || g.replicated || g.zring || (g.per & 4)) return false;
        if (std::max(1, s->cfg.presweeps) * (s->cfg.sweep_pairs ? 2 : 1) < 2) return false;
        return fused_update_slabs_all_ranks(s, g);
    }
    if (s->A.n <= s->cfg.graph_max_rows) return false;  // (a captured iteration cannot alternate the residual's two buffers)
    if (g.k0 != 0 || nk != g.n[2] || !fused_run_ok(s, g, 0, nk)) return false;
    return (g.n[0] / FX) * (g.n[1] / FY) * ((nk + FZ - 1) / FZ) <= PIB_MAXPART;
}

int gmg_apply(pib_solver *s, const double *r, double *z, hipStream_t q)
{
    if (!s->has_grid || s->levels.empty())
        return fail(PIB_ERR_ORDER, "solver %s: multigrid preconditioner without grid structure", s->name.c_str());
    if (!s->gmg_error.empty()) return fail(PIB_ERR_SUP, "solver %s: %s", s->name.c_str(), s->gmg_error.c_str());
    const bool guarded = s->gmg_guarded;
    const Scalars *S = guarded ? s->d_s : nullptr;
    s->halo_fresh = nullptr;
    s->gmg_dots_done = false;
    s->gmg_pending_count = 0;
    s->z_halo_depth = 0;
    const double omega = s->cfg.smoother_relaxation;
    const bool cheb = (s->cfg.smoother == Smoother::CHEBYSHEV);
    const int deg = cheb ? std::max(1, s->cfg.cheby_degree) : 1;  // one Chebyshev "sweep" = a degree-`deg` polynomial
    // (damped Jacobi: a sweep of the solver file is a fused pair of steps unless pib_sweep_pairs=0 -- Config::sweep_pairs)
    const int pairs = (!cheb && s->cfg.sweep_pairs) ? 2 : 1;
    const int pre = std::max(1, s->cfg.presweeps) * deg * pairs, post = std::max(0, s->cfg.postsweeps) * deg * pairs;
    const int nl = (int)s->levels.size();
    const int P = s->comm.nranks, rank = s->comm.rank;
    // PINNED: the sum of the residual that makes the right-hand side compatible -- red[5], summed by the pass that formed the
    // residual, or pin_sigma when that pass is this cycle's own first march (krylov.hip cg_s1: sum r - alpha sum w, the second sum
    // from the few entries of p next to cell 0)
    const double *pin = (s->nullspace == PIB_NULLSPACE_PINNED) ? (s->gmg_pin_local ? &s->d_s->pin_sigma : &s->d_s->red[5]) : nullptr;
    std::vector<double *> cur((size_t)nl, nullptr);  // current iterate buffer per level (owned pointer)
    const double lmax = s->cfg.cheby_lmax, lmin = lmax / s->cfg.cheby_ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;

    // ---- per-level halo facts and the validity (in ghost planes per side) of every vector this cycle touches
    struct LI {
        bool dist, lo, hi;
        int maxd;    // deepest exchange
        int cdepth;  // ghost planes a kernel may compute on (0 on a periodic slab axis: the ghost planes of the outer ranks
                     // are the planes at the other end of the axis, the kernels index the mesh arrays by plane number)
        int64_t nk;
    };
    std::vector<LI> li((size_t)nl);
    for (int l = 0; l < nl; ++l) {
        const GridLevel &g = s->levels[(size_t)l];
        LI &I = li[(size_t)l];
        I.dist = P > 1 && !g.replicated;
        I.lo = I.dist && (rank > 0 || s->comm.ring);
        I.hi = I.dist && (rank < P - 1 || s->comm.ring);
        I.nk = g.k1 - g.k0;
        I.maxd = I.cdepth = 0;
        if (I.dist) {
            int m = (s->cfg.deep_halo && !g.zring) ? HALO_MAX_DEPTH : 1;
            for (int q2 = 0; q2 < P; ++q2)  // an exchange takes planes the NEIGHBOUR owns
                m = (int)std::min<int64_t>(m, s->gmg_own[(size_t)l][(size_t)q2].second - s->gmg_own[(size_t)l][(size_t)q2].first);
            I.maxd = std::max(1, m);
            I.cdepth = g.zring ? 0 : I.maxd;
        }
    }
    std::vector<std::pair<const double *, int>> vd;
    auto valid = [&](const double *v) -> int {
        for (auto &e : vd)
            if (e.first == v) return e.second;
        return 0;
    };
    auto set_valid = [&](const double *v, int d) {
        for (auto &e : vd)
            if (e.first == v) {
                e.second = d;
                return;
            }
        vd.push_back({v, d});
    };
    // make `vec` of level l valid on d ghost planes per side
    auto need = [&](int l, const double *vec, int d) -> int {
        const LI &I = li[(size_t)l];
        if (!I.dist || d <= 0 || valid(vec) >= d) return 0;
        if (d > I.maxd) return fail(PIB_ERR_LIB, "gmg: halo depth %d not available on level %d", d, l);
        PIB_CHK(exchange_planes(s, s->levels[(size_t)l], const_cast<double *>(vec), d, q));
        set_valid(vec, d);
        return 0;
    };
    // The same exchange on the communication stream while the main stream goes on with work that needs owned planes
    // only; wait_halo() joins.  Used for the exchange of a level's right-hand side when it is large (>= 1 MiB per
    // neighbour: at 512^3 on 8 GPUs that is 8 MiB on level 0 and 1.5 MiB on level 1): the first kernel of the way down
    // runs on the interior planes meanwhile and on the planes next to the neighbours afterwards.
    bool halo_pending = false;
    auto need_async = [&](int l, const double *vec, int d) -> int {
        const LI &I = li[(size_t)l];
        if (!I.dist || d <= 0 || valid(vec) >= d) return 0;
        if (d > I.maxd) return fail(PIB_ERR_LIB, "gmg: halo depth %d not available on level %d", d, l);
        PIB_HIP(hipEventRecord(s->ev_ready, q));
        PIB_HIP(hipStreamWaitEvent(s->stream_comm, s->ev_ready, 0));
        PIB_CHK(exchange_planes(s, s->levels[(size_t)l], const_cast<double *>(vec), d, s->stream_comm));
        PIB_HIP(hipEventRecord(s->ev_halo, s->stream_comm));
        set_valid(vec, d);
        halo_pending = true;
        return 0;
    };
    auto wait_halo = [&]() -> int {
        if (halo_pending) PIB_HIP(hipStreamWaitEvent(q, s->ev_halo, 0));
        halo_pending = false;
        return 0;
    };
    // planes [a, a + c) relative to the first owned plane for a run that reaches d ghost planes into the neighbours
    auto run = [&](int l, int d, int64_t &a, int64_t &c) {
        const LI &I = li[(size_t)l];
        a = I.lo ? -d : 0;
        c = I.nk + (I.hi ? d : 0) - a;
    };
    // Depth a stencil kernel produces: what is desired, no more than its right-hand side is valid, one plane less than its
    // input; an input without any valid ghost plane is exchanged first.  The rule depends on nothing rank-specific and
    // the fused kernels below record exactly what the steps they replace would: every rank issues the same exchanges.
    auto stencil_depth = [&](int l, int desired, const double *in, const double *b, int *out) -> int {
        const LI &I = li[(size_t)l];
        *out = 0;
        if (!I.dist) return 0;
        int o = std::max(0, std::min(std::min(desired, I.maxd - 1), valid(b)));
        if (valid(in) >= 1) o = std::min(o, valid(in) - 1);
        else PIB_CHK(need(l, in, o + 1));
        *out = o;
        return 0;
    };

    // `nsteps` smoothing steps on level l: iterate in `a` (result left in `a` after the swaps), spare buffer `c`.
    // Jacobi: x <- x + omega D^-1 (b - A x).  Chebyshev-Jacobi: three-term recurrence over [lmin, lmax] of D^-1 A,
    // restarted for every segment (oracle/csrc/gmg.c:cheby).  The last step's result is valid on d_final ghost planes,
    // the one before on d_final + 1, ... (as far as the inputs allow; a missing plane is exchanged).
    auto smooth_seq = [&](int l, const double *b, const double *pin_l, double *&a, double *&c, int nsteps, bool from_zero,
                          int d_final, bool dots_in_last = false) -> int {
        GridLevel &g = s->levels[(size_t)l];
        const LI &I = li[(size_t)l];
        double rho = 1.0 / sigma;
        double *dvec = g.d + g.pad;
        for (int sw = 0; sw < nsteps; ++sw) {
            const int desired = d_final + (nsteps - 1 - sw);
            int64_t ka, kc;
            if (from_zero && sw == 0 && nsteps >= 2 && !cheb && s->cfg.fuse_presmooth) {
                // steps 0 and 1 in one kernel; the result lands where step 1 would have put it
                int o = I.dist ? std::min(std::min(desired - 1, I.maxd - 1), valid(b) - 1) : 0;
                run(l, std::max(o, 0), ka, kc);
                if (o >= 0 && fused_run_ok(s, g, ka, kc) &&
                    ((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 31u) == 0) {
                    const int FZ = march_planes(g, kc);
                    auto launch = [&](int64_t ra, int64_t rc) {
                        if (rc <= 0) return;
                        GridLevel sub = g;
                        sub.k0 = g.k0 + ra;
                        sub.k1 = sub.k0 + rc;
                        hipLaunchKernelGGL(k_presmooth2<0>, dim3((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((rc + FZ - 1) / FZ)),
                                           dim3(256), 0, q, S, dev_of(sub), omega, b + ra * g.plane, c + ra * g.plane, pin_l, FZ, nullptr);
                    };
                    if (l == 0 && s->gmg_upd.w != nullptr && I.dist) {
                        // ... on z-slabs (round 4): w came through the exchange instead of the residual, whose ghost planes follow the
                        // recurrence -- every launch updates the planes it loads, the run's planes and the one beyond it at either end
                        // (the depth the next cycle reads r on) are written to the NEW residual's buffer, the sums cover the owned planes
                        const double *ro = s->gmg_upd.r_old, *uw = s->gmg_upd.w;
                        double *rn = const_cast<double *>(b);
                        int nblk = 0;
                        auto launch_upd = [&](int64_t ra, int64_t rc, int wext) {
                            if (rc <= 0) return;
                            GridLevel sub = g;
                            sub.k0 = g.k0 + ra;
                            sub.k1 = sub.k0 + rc;
                            const dim3 grid((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((rc + FZ - 1) / FZ));
                            hipLaunchKernelGGL((k_presmooth2<0, 1>), grid, dim3(256), 0, q, S, dev_of(sub), omega, ro + ra * g.plane, c + ra * g.plane, pin_l, FZ,
                                               (double *)nullptr, uw + ra * g.plane, rn + ra * g.plane, s->d_part + 4 * (int64_t)PIB_MAXPART, (int)PIB_MAXPART,
                                               wext, (int)g.k0, (int)g.k1, nblk);
                            nblk += (int)(grid.x * grid.y * grid.z);
                        };
                        // the run reaches o ghost planes, the residual is read (and kept) one plane deeper
                        const int wlo = (I.lo && o + 1 <= valid(uw)) ? 1 : 0, whi = (I.hi && o + 1 <= valid(uw)) ? 2 : 0;
                        if (halo_pending) {
                            const int64_t ia = I.lo ? 1 : 0, ie = I.nk - (I.hi ? 1 : 0);
                            launch_upd(ia, ie - ia, 0);
                            PIB_CHK(wait_halo());
                            launch_upd(ka, ia - ka, wlo);
                            launch_upd(ie, ka + kc - ie, whi);
                        } else
                            launch_upd(ka, kc, wlo | whi);
                        PIB_HIP(hipGetLastError());
                        if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                        s->gmg_upd.used = true;
                        PIB_CHK(s->gmg_upd.after(s, nblk, q));
                    } else if (l == 0 && s->gmg_upd.w != nullptr) {
                        // PCG's residual update folded into this march (k_presmooth2<., 1>): b is the NEW residual's buffer
                        if (I.dist || halo_pending || ka != 0 || kc != I.nk) return fail(PIB_ERR_LIB, "fused residual update: level 0 is not a whole single-rank level");
                        const dim3 grid((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ));
                        const int nblk = (int)(grid.x * grid.y * grid.z);
                        if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                        hipLaunchKernelGGL((k_presmooth2<0, 1>), grid, dim3(256), 0, q, S, dev_of(g), omega, s->gmg_upd.r_old, c, pin_l, FZ,
                                           (double *)nullptr, s->gmg_upd.w, const_cast<double *>(b), s->d_part + 4 * (int64_t)PIB_MAXPART,
                                           (int)PIB_MAXPART);
                        PIB_HIP(hipGetLastError());
                        s->gmg_upd.used = true;
                        PIB_CHK(s->gmg_upd.after(s, nblk, q));
                    } else if (halo_pending) {
                        // planes whose two steps read owned planes of b only, while the exchange is in flight
                        const int64_t ia = I.lo ? 1 : 0, ie = I.nk - (I.hi ? 1 : 0);
                        launch(ia, ie - ia);
                        PIB_CHK(wait_halo());
                        launch(ka, ia - ka);
                        launch(ie, ka + kc - ie);
                    } else
                        launch(ka, kc);
                    PIB_HIP(hipGetLastError());
                    set_valid(c, o);
                    std::swap(a, c);
                    sw = 1;
                    continue;
                }
            }
            PIB_CHK(wait_halo());
            if (from_zero && sw == 0) {
                const int o = I.dist ? std::max(0, std::min(std::min(desired, I.cdepth), valid(b))) : 0;  // pointwise: as deep as b
                run(l, o, ka, kc);
                if (cheb) PIB_CHK(launch_level_planes<6>(s, g, ka, kc, 1.0 / theta, b, nullptr, a, pin_l, guarded, q, dvec, 0.0));
                else PIB_CHK(launch_level_planes<1>(s, g, ka, kc, omega, b, nullptr, a, pin_l, guarded, q));
                set_valid(a, o);
                continue;
            }
            int o;
            PIB_CHK(stencil_depth(l, desired, a, b, &o));
            run(l, o, ka, kc);
            if (cheb) {
                double a_d = 0.0, a_z = 1.0 / theta;
                if (sw > 0) {
                    const double rho_new = 1.0 / (2.0 * sigma - rho);
                    a_d = rho_new * rho;
                    a_z = 2.0 * rho_new / delta;
                    rho = rho_new;
                }
                PIB_CHK(launch_level_planes<5>(s, g, ka, kc, a_z, b, a, c, pin_l, guarded, q, dvec, a_d));
            } else {
                const bool dots = dots_in_last && sw + 1 == nsteps;
                if (dots) {
                    PIB_CHK(launch_level_planes<8>(s, g, ka, kc, omega, b, a, c, pin_l, guarded, q));
                    s->gmg_dots_done = true;
                } else
                    PIB_CHK(launch_level_planes<2>(s, g, ka, kc, omega, b, a, c, pin_l, guarded, q));
            }
            set_valid(c, o);
            std::swap(a, c);
        }
        return 0;
    };

    // first level handled by the single-workgroup coarse tail (never level 0; Jacobi only; no communication)
    int tail0 = nl;
    // the smallest levels are launch-bound (~5 us kernels, five to seven per level): one workgroup walks those of <= 1024
    // cells.  A 2-D flow case spends most of its V-cycle there; on a 256^3 / 512^3 grid it is 0.2 ms of a time step / of
    // a solve (17.4 -> 17.2 ms, 88.2 -> 88.1 ms), and taking larger levels into the one workgroup costs more than the
    // launches it saves (4096: 90.3 ms, 32768: 20.4 ms per Taylor-Green step; docs/history/measured_and_rejected.md)
    const int tail_cells = (s->cfg.coarse_tail >= 0) ? s->cfg.coarse_tail : 1024;
    if (!cheb && tail_cells) {
        for (int l = nl - 1; l >= 1; --l) {
            const GridLevel &g = s->levels[(size_t)l];
            const bool local = (s->comm.nranks == 1) || g.replicated;
            if (local && g.nloc <= std::min<int64_t>(TAIL_MAX_CELLS, tail_cells) && nl - l < TAIL_MAX_LEVELS) tail0 = l; else break;
        }
        if (nl - tail0 < 2) tail0 = nl;  // a single level is what k_coarsest already does
    }
    // depth of the level's right-hand side on the way down: the pre-smoothing steps, the residual and the restriction's
    // reach of one plane; level 0 also carries the post-smoothing steps and one plane of the result z, so that neither
    // the corrected iterate nor p = z + beta p (the next Krylov product's input) needs an exchange of its own
    // ... and a coarser distributed level delivers ITS final iterate on as many ghost planes as the prolongation onto the finer
    // level reads (pib_deep_up, round 4): the exchange of the coarse correction on the way up -- a collective with nothing to
    // hide behind -- goes, the right-hand side of that level is exchanged deeper on the way down instead (the same bytes: at
    // 512^3 / 8 five planes of level 1 instead of 3 + 2, six of level 2 instead of 3 + 3) and a few more ghost planes of two
    // latency-bound levels are relaxed redundantly.  Where the level's slabs are too thin for that depth the exchange stays.
    std::vector<int> fin_l((size_t)nl, 0);
    fin_l[0] = (li[0].dist && li[0].maxd > 1) ? 1 : 0;
    if (s->cfg.deep_up && !cheb && post >= 1)
        for (int l = 1; l < nl; ++l) {
            if (!li[(size_t)l].dist || !li[(size_t)l - 1].dist) continue;
            const int want = coarse_need(s, l - 1, fin_l[(size_t)l - 1] + post);
            if (want > 0 && pre + post - 1 + want <= li[(size_t)l].cdepth) fin_l[(size_t)l] = want;
        }
    auto final_depth = [&](int l) -> int { return fin_l[(size_t)l]; };
    auto down_depth = [&](int l) -> int {
        const LI &I = li[(size_t)l];
        if (!I.dist) return 0;
        return std::min(I.cdepth, std::max(pre + 1, pre + post - 1 + final_depth(l)));
    };
    // prolongation and both post-smoothing steps of level l in one march (k_prolong_smooth2): V(., 2), a level and its coarser
    // one whole on this rank, the conditions of k_prolong_smooth; decided here, before the way down, because the pair of steps
    // is one swap of the level's buffers (level 0 must end in z)
    auto post2_ok = [&](int l) -> bool {
        if (!s->cfg.fuse_post_pair || post != 2 || cheb || !s->cfg.fuse_prolong || l + 1 >= nl || l >= tail0) return false;
        const GridLevel &g = s->levels[(size_t)l];
        const GridLevel &c1 = s->levels[(size_t)l + 1];
        const LI &I = li[(size_t)l];
        const bool whole = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && c1.replicated)) && g.k0 == 0 &&
                           g.k1 == g.n[2] && c1.k0 == 0 && c1.k1 == c1.n[2];
        // ... or both levels in z-slabs of the same ranks: the result on final_depth ghost planes needs the old iterate and the
        // coarse values two planes deeper and b one (what the way down leaves valid; exchanged at the launch if not)
        const int fin0 = fin_l[(size_t)l];  // (final_depth)
        const bool slabs = I.dist && li[(size_t)l + 1].dist && !(g.per & 4) && std::min(I.maxd, I.cdepth) >= fin0 + 2 &&
                           coarse_need(s, l, fin0 + 2) <= li[(size_t)l + 1].maxd;
        if (!(whole || slabs) || g.zring) return false;
        const bool per_ok = g.per == g.tper && (!(g.per & 4) || g.n[2] >= 8);
        if (!g.plain_pair || !per_ok || !fused_run_ok(s, g, 0, I.nk) || g.n[1] % UTY != 0) return false;
        auto al32 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31u) == 0; };
        return al32(g.x + g.pad) && al32(g.x2 + g.pad) && al32(l == 0 ? (const void *)r : (const void *)(g.b + g.pad)) && (l != 0 || al32(z));
    };
    // ---- downward leg
    for (int l = 0; l < nl; ++l) {
        GridLevel &g = s->levels[(size_t)l];
        const LI &I = li[(size_t)l];
        const int64_t pl = g.plane;
        if (l == tail0) {
            TailArgs T;
            std::memset(&T, 0, sizeof T);
            T.nlev = nl - tail0;
            T.omega = omega;
            T.pre = pre;
            T.post = post;
            T.sweeps = s->cfg.coarsest_sweeps;
            for (int q2 = 0; q2 < T.nlev; ++q2) {
                GridLevel &t = s->levels[(size_t)(tail0 + q2)];
                T.lv[q2].L = dev_of(t);
                T.lv[q2].xa = t.x + t.pad;
                T.lv[q2].xb = t.x2 + t.pad;
                T.lv[q2].b = t.b + t.pad;
                T.lv[q2].r = t.r + t.pad;
            }
            const int swaps = (pre - 1) + post;
            cur[(size_t)l] = (swaps % 2 == 0) ? (g.x + g.pad) : (g.x2 + g.pad);
            int64_t need = 0;
            bool fits = s->cfg.coarse_tail_lds != 0;
            for (int q2 = 0; q2 < T.nlev; ++q2) {
                const GridLevel &t = s->levels[(size_t)(tail0 + q2)];
                T.lds_off[q2] = (int)need;
                need += 3 * t.nloc;  // iterate, spare (the residual's place on the way down), right-hand side
                fits = fits && !t.zring && t.nloc == t.n[0] * t.n[1] * t.n[2];
            }
            if (!fits || need > TAIL_POOL)
                for (int q2 = 0; q2 < TAIL_MAX_LEVELS; ++q2) T.lds_off[q2] = -1;
            // the tables of the tail's levels, packed (see TailArgs): offsets first, the block itself when the arguments change
            struct Piece { const void *src; int off, bytes; };
            std::vector<Piece> pieces;
            int used = 0;
            bool tabs_ok = true;
            auto put_d = [&](const double *src, int n) -> int {
                if (src == nullptr) tabs_ok = false;
                const int off = used;
                pieces.push_back({src, off, 8 * n});
                used += n;
                return off;
            };
            auto put_i = [&](const int32_t *src, int n) -> int {
                if (src == nullptr) tabs_ok = false;
                const int off = used;
                pieces.push_back({src, off, 4 * n});
                used += (n + 1) / 2;
                return off;
            };
            for (int q2 = 0; q2 < T.nlev; ++q2) {
                const GridLevel &t = s->levels[(size_t)(tail0 + q2)];
                TailArgs::Tabs &o = T.tt[q2];
                for (int d = 0; d < 3; ++d) {
                    const int n = (int)t.n[d];
                    o.w[d] = put_d(t.w[d], n);
                    o.rw[d] = put_d(t.rw[d], n);
                    o.cm[d] = put_d(t.cm[d], n);
                    o.cp[d] = put_d(t.cp[d], n);
                    if (q2 + 1 < T.nlev) {
                        const int nc = (int)s->levels[(size_t)(tail0 + q2 + 1)].n[d];
                        o.wpar[d] = put_d(t.t_wpar[d], n);
                        o.woth[d] = put_d(t.t_woth[d], n);
                        o.par[d] = put_i(t.t_par[d], n);
                        o.oth[d] = put_i(t.t_oth[d], n);
                        o.fst[d] = put_i(t.t_fst[d], nc);
                    }
                }
            }
            // (vectors and tables go to LDS together or not at all: the kernel has one instance for either)
            const bool tab_lds = T.lds_off[0] >= 0 && tabs_ok && used <= TAIL_TAB;
            if (!tab_lds)
                for (int q2 = 0; q2 < TAIL_MAX_LEVELS; ++q2) T.lds_off[q2] = -1;
            T.tab_used = tab_lds ? used : 0;
            if (s->d_tail_tab == nullptr) PIB_HIP(hipMalloc(&s->d_tail_tab, sizeof(double) * TAIL_TAB));
            T.tab_src = static_cast<const double *>(s->d_tail_tab);
            T.out0 = cur[(size_t)l];
            // the argument block lives in HBM, rewritten only when it changes (levels, sweeps and buffers are fixed per set-up)
            if (s->d_tail_args == nullptr) PIB_HIP(hipMalloc(&s->d_tail_args, sizeof(TailArgs)));
            if (s->h_tail_args.size() != sizeof(TailArgs) || std::memcmp(s->h_tail_args.data(), &T, sizeof(TailArgs)) != 0) {
                s->h_tail_args.assign(reinterpret_cast<const char *>(&T), reinterpret_cast<const char *>(&T) + sizeof(TailArgs));
                PIB_HIP(hipMemcpyAsync(s->d_tail_args, s->h_tail_args.data(), sizeof(TailArgs), hipMemcpyHostToDevice, q));
                if (tab_lds)
                    for (const Piece &pc : pieces)
                        PIB_HIP(hipMemcpyAsync(static_cast<double *>(s->d_tail_tab) + pc.off, pc.src, (size_t)pc.bytes, hipMemcpyDeviceToDevice, q));
            }
            if (T.lds_off[0] >= 0) hipLaunchKernelGGL(k_coarse_tail<true>, dim3(1), dim3(1024), 0, q, S, static_cast<const TailArgs *>(s->d_tail_args));
            else hipLaunchKernelGGL(k_coarse_tail<false>, dim3(1), dim3(1024), 0, q, S, static_cast<const TailArgs *>(s->d_tail_args));
            PIB_HIP(hipGetLastError());
            break;
        }
        const double *b = (l == 0) ? r : g.b + g.pad;
        const double *pin_l = (l == 0) ? pin : nullptr;
        double *xa = g.x + g.pad, *xb = g.x2 + g.pad;
        if (l == nl - 1) {
            double *out = (l == 0) ? z : xa;
            if (g.nloc <= 4096 && !I.dist) {
                hipLaunchKernelGGL(k_coarsest, dim3(1), dim3(256), 0, q, S, dev_of(g), omega, s->cfg.coarsest_sweeps, b, xa, xb,
                                   out);
                PIB_HIP(hipGetLastError());
            } else {
                PIB_CHK(launch_level<1>(s, g, omega, b, nullptr, xa, pin_l, guarded, q, nullptr, 0.0));
                double *a = xa, *c = xb;
                for (int sw = 1; sw < s->cfg.coarsest_sweeps; ++sw) {
                    set_valid(a, 0);
                    PIB_CHK(need(l, a, 1));
                    PIB_CHK(launch_level<2>(s, g, omega, b, a, c, pin_l, guarded, q, nullptr, 0.0));
                    std::swap(a, c);
                }
                if (a != out) PIB_HIP(hipMemcpyAsync(out, a, sizeof(double) * (size_t)g.nloc, hipMemcpyDeviceToDevice, q));
            }
            set_valid(out, 0);
            cur[(size_t)l] = out;
            break;
        }
        // number of ping-pong swaps left on this level: (pre-1) + post; arrange that level 0 ends in z
        double *a = xa, *c = xb;
        if (l == 0) {
            // (both post-smoothing steps in one march, k_prolong_smooth2, are ONE swap)
            const int swaps = (pre - 1) + (post2_ok(0) ? 1 : post);
            // final buffer after `swaps` swaps starting from a: a if even else c
            if (swaps % 2 == 0) { a = z; c = xa; } else { a = xa; c = z; }
        }
        double *rr = g.r + g.pad;
        if (l == 0 && s->gmg_upd.w != nullptr) {
            // PCG left r = r_old - alpha w to this cycle's first march: decided HERE, with the pointers the march would get, by the
            // predicate of its two launch sites below; if it cannot take the update, the update runs as a pass of its own first
            const int FZ0 = march_planes(g, I.nk);
            auto al32 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31u) == 0; };
            // (several ranks: the slab-dependent parts for EVERY rank's slab -- fused_update_slabs_all_ranks -- so that all ranks fall
            // back together or not at all)
            bool site = s->cfg.fuse_residual_update == 1 && !cheb && s->cfg.fuse_presmooth && al32(b) && (pre >= 2 ? al32(c) : (al32(a) && al32(rr)));
            if (I.dist) site = site && fused_update_slabs_all_ranks(s, g);
            else site = site && fused_run_ok(s, g, 0, I.nk) && (g.n[0] / FX) * (g.n[1] / FY) * ((I.nk + FZ0 - 1) / FZ0 + 2) <= PIB_MAXPART;
            // z-slabs: the two-step march only, deep halos (the residual's depth is what w is exchanged to), aligned planes
            if (I.dist)
                site = site && pre >= 2 && !g.zring && down_depth(0) >= 2 && down_depth(0) <= I.maxd && al32(s->gmg_upd.w) && al32(s->gmg_upd.r_old) &&
                       (g.plane % 4) == 0;
            if (!site) {
                if (s->gmg_upd.fallback == nullptr) return fail(PIB_ERR_LIB, "fused residual update: no fallback registered");
                PIB_CHK(s->gmg_upd.fallback(s, const_cast<double *>(b), q));
                s->gmg_upd.w = nullptr;
                s->gmg_upd.used = true;
            }
        }
        // a small level that is whole on this rank: pre-smoothing, residual and restriction in one launch
        {
            GridLevel &cg1 = s->levels[(size_t)l + 1];
            int bc3[3];
            unsigned nblk = 0;
            const bool local_pair = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && cg1.replicated));
            // (level 0 too -- a 2-D case's finest level is such a level, with the pinned row's correction of its right-hand side if
            // there is one -- unless PCG's residual update is folded in; the way up of level 0 keeps its launches: its last step
            // delivers the Krylov sums in a fixed order of workgroups)
            const bool level_ok = l >= 1 || s->gmg_upd.w == nullptr;
            if (level_ok && !cheb && local_pair && small_level_boxes(s, g, cg1, pre, true, bc3, &nblk)) {
                if (l == 0) {  // no swaps on the way down: the swaps of the way up must end in z (the fused pair of
                               // post-smoothing steps, k_prolong_smooth2, is ONE swap -- the same count as above)
                    const int up_swaps = post2_ok(0) ? 1 : post;
                    if (up_swaps % 2 == 0) { a = z; c = xa; } else { a = xa; c = z; }
                }
                hipLaunchKernelGGL(k_small_down, dim3(nblk), dim3(SM_NT), 0, q, S, dev_of(g), dev_of(cg1), omega, pre, b, a, cg1.b + cg1.pad, bc3[0],
                                   bc3[1], bc3[2], pin_l);
                PIB_HIP(hipGetLastError());
                set_valid(a, 0);
                set_valid(cg1.b + cg1.pad, 0);
                cur[(size_t)l] = a;
                s->gmg_spare[(size_t)l] = c;
                continue;
            }
        }
        // the right-hand side on as many ghost planes as the way down (and, on level 0, the way up) consumes
        const int Dd = down_depth(l);
        set_valid(b, 0);
        // (PCG's residual update inside the first march, on slabs: w = A p is what travels; the new residual comes out of the
        // march on the same planes, the neighbours' included)
        const double *xv = (l == 0 && I.dist && s->gmg_upd.w != nullptr) ? s->gmg_upd.w : b;
        set_valid(xv, 0);
        if (I.dist && s->cfg.overlap_halo && I.nk >= 4 && (int64_t)Dd * g.plane * 8 >= (int64_t)s->cfg.overlap_min_bytes)
            PIB_CHK(need_async(l, xv, Dd));
        else
            PIB_CHK(need(l, xv, Dd));
        if (xv != b) set_valid(b, valid(xv));
        if (pre == 1 && !cheb) {
            PIB_CHK(wait_halo());
            // one pre-smoothing step from zero and the residual of its result: x1 kept and r valid on o ghost planes
            const int o = I.dist ? std::max(0, std::min(std::min(Dd - 1, I.maxd - 1), valid(b) - 1)) : 0;
            int64_t ka, kc;
            run(l, o, ka, kc);
            if (s->cfg.fuse_presmooth && fused_run_ok(s, g, ka, kc) && (!I.dist || valid(b) >= o + 1) &&
                ((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(rr)) & 31u) == 0) {
                GridLevel sub = g;  // both in one march (b read once)
                sub.k0 = g.k0 + ka;
                sub.k1 = sub.k0 + kc;
                const int FZ = march_planes(g, kc);
                if (l == 0 && s->gmg_upd.w != nullptr) {
                    if (I.dist || ka != 0 || kc != I.nk) return fail(PIB_ERR_LIB, "fused residual update: level 0 is not a whole single-rank level");
                    const dim3 grid((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ));
                    const int nblk = (int)(grid.x * grid.y * grid.z);
                    if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                    hipLaunchKernelGGL((k_presmooth2<1, 1>), grid, dim3(256), 0, q, S, dev_of(sub), omega, s->gmg_upd.r_old, a, pin_l, FZ, rr,
                                       s->gmg_upd.w, const_cast<double *>(b), s->d_part + 4 * (int64_t)PIB_MAXPART, (int)PIB_MAXPART);
                    PIB_HIP(hipGetLastError());
                    s->gmg_upd.used = true;
                    PIB_CHK(s->gmg_upd.after(s, nblk, q));
                } else {
                hipLaunchKernelGGL(k_presmooth2<1>, dim3((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ)),
                                   dim3(256), 0, q, S, dev_of(sub), omega, b + ka * pl, a + ka * pl, pin_l, FZ, rr + ka * pl);
                PIB_HIP(hipGetLastError());
                }
            } else {
                // x1 one plane deeper than it is kept (pointwise: as deep as b), then its residual
                const int dx = I.dist ? std::min(std::min(o + 1, valid(b)), I.cdepth) : 0;
                int64_t xa2, xc2;
                run(l, dx, xa2, xc2);
                PIB_CHK(launch_level_planes<1>(s, g, xa2, xc2, omega, b, nullptr, a, pin_l, guarded, q));
                set_valid(a, dx);
                if (I.dist && dx < o + 1) PIB_CHK(need(l, a, o + 1));
                PIB_CHK(launch_level_planes<3>(s, g, ka, kc, omega, b, a, rr, pin_l, guarded, q));
            }
            set_valid(a, o);
            set_valid(rr, o);
        } else {
            // the pre-smoothed iterate is kept as deep as the way up wants it (the corrected iterate starts from it)
            const int want_x = I.dist ? std::max(2, post + final_depth(l)) : 0;
            // the two steps, the residual and the restriction in ONE march (k_down_march) where both marches below would run on a
            // level that is whole on this one rank and not periodic
            {
                const GridLevel &c1 = s->levels[(size_t)l + 1];
                const int64_t nkc = c1.k1 - c1.k0;
                const int FZ = march_planes(g, I.nk);
                const bool upd = l == 0 && s->gmg_upd.w != nullptr;
                auto al32 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31u) == 0; };
                const bool fits = s->cfg.fuse_down_march && pre == 2 && !cheb && s->cfg.fuse_presmooth && s->cfg.fuse_residual_restrict &&
                                  s->cfg.march_restrict && !I.dist && !li[(size_t)l + 1].dist && s->comm.nranks == 1 && g.k0 == 0 && g.k1 == g.n[2] &&
                                  c1.k0 == 0 && c1.k1 == c1.n[2] && !g.zring && g.per == 0 && g.tper == 0 && g.plain_pair && !halo_pending &&
                                  fused_run_ok(s, g, 0, I.nk) && g.n[0] % RX == 0 && g.n[1] % 8 == 0 && (g.n[2] & 1) == 0 && 2 * nkc == g.n[2] && nkc >= 4 &&
                                  nkc * c1.plane * 8 >= (int64_t)s->cfg.march_min_cells && (FZ & 1) == 0 && FZ / 2 <= 32 && al32(b) && al32(c) &&
                                  (!upd || (al32(s->gmg_upd.w) && al32(s->gmg_upd.r_old)));
                if (fits) {
                    const int CZ = FZ / 2;
                    // (128 x 8 tiles, 512 threads, 214 registers and no spill: 1.19 ms per 512^3 launch against 0.84 + 0.51; the 128 x 16
                    // tile -- 40 instead of 47 B per cell -- needs 768 threads, i.e. 168 registers, and spilled 65: 3.4 ms)
                    // ... 64 x 16 tiles (42 B per cell, 212 registers): 60.9-61.0 against 61.2-61.4 ms per solve on one box, and the Krylov
                    // sums no longer in k_presmooth2's grouping: not kept)
                    constexpr int TX = 128, TY = 8;
                    const dim3 grid((unsigned)(g.n[0] / TX), (unsigned)(g.n[1] / TY), (unsigned)((nkc + CZ - 1) / CZ));
                    if (upd) {
                        const int nblk = (int)(grid.x * grid.y * grid.z);  // (128 x 8 tiles: k_presmooth2's workgroups)
                        if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                        hipLaunchKernelGGL((k_down_march<1, TX, TY>), grid, dim3(down_threads(TX, TY)), 0, q, S, dev_of(g), dev_of(c1), omega, s->gmg_upd.r_old, c,
                                           c1.b + c1.pad, CZ, pin_l, s->gmg_upd.w, const_cast<double *>(b), s->d_part + 4 * (int64_t)PIB_MAXPART, (int)PIB_MAXPART);
                        PIB_HIP(hipGetLastError());
                        s->gmg_upd.used = true;
                        PIB_CHK(s->gmg_upd.after(s, nblk, q));
                    } else {
                        hipLaunchKernelGGL((k_down_march<0, TX, TY>), grid, dim3(down_threads(TX, TY)), 0, q, S, dev_of(g), dev_of(c1), omega, b, c, c1.b + c1.pad, CZ,
                                           pin_l, (const double *)nullptr, (double *)nullptr, (double *)nullptr, 0);
                        PIB_HIP(hipGetLastError());
                    }
                    set_valid(c, 0);
                    std::swap(a, c);
                    set_valid(c1.b + c1.pad, 0);
                    cur[(size_t)l] = a;
                    s->gmg_spare[(size_t)l] = c;
                    continue;
                }
            }
            PIB_CHK(smooth_seq(l, b, pin_l, a, c, pre, true, want_x));
            // residual and restriction in one march where the marching restriction would run (k_resid_restrict_march): the
            // residual never goes to HBM
            {
                const GridLevel &c1 = s->levels[(size_t)l + 1];
                const int64_t nkc = c1.k1 - c1.k0;
                bool whole = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && c1.replicated)) && g.k0 == 0 &&
                             g.k1 == g.n[2] && c1.k0 == 0 && c1.k1 == c1.n[2] && !g.zring;
                // ... or both levels in z-slabs of the same ranks (an aggregate never straddles a cut): the march reads the iterate
                // two planes and b one plane beyond the slab -- what the way down left valid there -- and the residual's own
                // ghost plane needs no exchange any more
                if (!whole && I.dist && li[(size_t)l + 1].dist && !g.zring && !(g.per & 4) && valid(a) >= 2 && valid(b) >= 1 &&
                    std::min(I.maxd, I.cdepth) >= 2)
                    whole = true;
                if (s->cfg.fuse_residual_restrict && whole && s->cfg.march_restrict && g.plain_pair && g.per == g.tper &&
                    g.n[0] % RX == 0 && g.n[1] % RY == 0 && nkc >= 4 && nkc * c1.plane * 8 >= (int64_t)s->cfg.march_min_cells) {
                    const int CZ = nkc * c1.plane >= ((int64_t)1 << 23) ? 32 : 8;  // (as launch_restrict)
                    hipLaunchKernelGGL(k_resid_restrict_march, dim3((unsigned)(g.n[0] / RX), (unsigned)(g.n[1] / RY), (unsigned)((nkc + CZ - 1) / CZ)),
                                       dim3(256), 0, q, S, dev_of(g), dev_of(c1), b, a, c1.b + c1.pad, CZ, pin_l);
                    PIB_HIP(hipGetLastError());
                    set_valid(c1.b + c1.pad, 0);
                    cur[(size_t)l] = a;
                    s->gmg_spare[(size_t)l] = c;
                    continue;
                }
            }
            int o;
            PIB_CHK(stencil_depth(l, 1, a, b, &o));
            int64_t ka, kc;
            run(l, o, ka, kc);
            PIB_CHK(launch_level_planes<3>(s, g, ka, kc, omega, b, a, rr, pin_l, guarded, q));
            set_valid(rr, o);
        }
        PIB_CHK(need(l, rr, 1));  // the restriction reaches one fine plane beyond the slab
        GridLevel &cg = s->levels[(size_t)l + 1];
        if (cg.replicated && !g.replicated && s->comm.nranks > 1) {
            // restrict the owned coarse planes into a scratch (cg.r), then all-gather into cg.b
            GridLevel own = cg;
            own.k0 = s->gmg_own[(size_t)l + 1][(size_t)s->comm.rank].first;
            own.k1 = s->gmg_own[(size_t)l + 1][(size_t)s->comm.rank].second;
            own.nloc = (own.k1 - own.k0) * cg.plane;
            double *scratch = cg.r + cg.pad;
            PIB_CHK(launch_restrict(s, g, own, rr, scratch, S, q));
            PIB_CHK(gather_level(s, l + 1, cg.plane, scratch, own.nloc, cg.b + cg.pad, q));
        } else {
            PIB_CHK(launch_restrict(s, g, cg, rr, cg.b + cg.pad, S, q));
        }
        cur[(size_t)l] = a;
        // remember the spare buffer in g.scratch for the upward leg
        s->gmg_spare[(size_t)l] = c;
    }
    // ---- upward leg
    for (int l = std::min(nl - 2, tail0 - 1); l >= 0; --l) {
        GridLevel &g = s->levels[(size_t)l];
        GridLevel &cg = s->levels[(size_t)l + 1];
        const LI &I = li[(size_t)l];
        const int64_t pl = g.plane;
        const double *b = (l == 0) ? r : g.b + g.pad;
        const double *pin_l = (l == 0) ? pin : nullptr;
        double *a = cur[(size_t)l], *c = s->gmg_spare[(size_t)l];
        double *xc = cur[(size_t)l + 1];
        const int fin = final_depth(l);
        // the corrected iterate x + P e on e ghost planes: as deep as the post-smoothing consumes, no deeper than the
        // pre-smoothed iterate is valid and than the coarse correction can be had
        int e = 0;
        if (I.dist) {
            // (no deeper than a kernel may compute on: 0 on a periodic slab axis, whose outer ghost planes are the other end of the
            // axis -- V(., 2) on such a level used to take the iterate's exchanged plane for a plane it could correct: plane -1)
            e = std::min(std::min(fin + post, std::min(I.maxd, I.cdepth)), valid(a));
            if (li[(size_t)l + 1].dist)
                while (e > 0 && coarse_need(s, l, e) > li[(size_t)l + 1].maxd) --e;
            if (li[(size_t)l + 1].dist) PIB_CHK(need(l + 1, xc, coarse_need(s, l, e)));
        }
        {
            // a small level whole on this rank: prolongation and the post-smoothing in one launch
            int bc3[3];
            unsigned nblk = 0;
            const bool local_pair = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && cg.replicated));
            if (l >= 1 && !cheb && local_pair && small_level_boxes(s, g, cg, post, false, bc3, &nblk)) {
                hipLaunchKernelGGL(k_small_up, dim3(nblk), dim3(SM_NT), 0, q, S, dev_of(g), dev_of(cg), omega, post, b, a, xc, c, bc3[0], bc3[1],
                                   bc3[2]);
                PIB_HIP(hipGetLastError());
                set_valid(c, 0);
                std::swap(a, c);
                cur[(size_t)l] = a;
                continue;
            }
        }
        const bool dots_l = l == 0 && s->gmg_want_dots && !cheb;
        if (post2_ok(l)) {
            // prolongation + both post-smoothing steps in one march; on level 0 the second one delivers the Krylov sums
            // planes per workgroup: four iterations fill the pipeline, so no fewer than 32 -- except where the Krylov sums are
            // formed, which keep the grouping (and the bits) of k_level_march<8>
            int64_t ka = 0, kc = I.nk;
            if (I.dist) {
                // the run of planes the result is wanted on, and its inputs as deep as the two steps reach
                const int cn = coarse_need(s, l, fin + 2);
                if (li[(size_t)l + 1].dist) PIB_CHK(need(l + 1, xc, cn));
                if (valid(a) < fin + 2) PIB_CHK(need(l, a, fin + 2));
                if (valid(b) < fin + 1) PIB_CHK(need(l, b, fin + 1));
                run(l, fin, ka, kc);
            }
            GridLevel sub = g;
            sub.k0 = g.k0 + ka;
            sub.k1 = sub.k0 + kc;
            const int FZ = dots_l ? march_planes(g, kc) : std::max(march_planes(g, kc), 32);
            const dim3 mg((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / UTY), (unsigned)((kc + FZ - 1) / FZ));
            const double *bq = b + ka * pl;
            double *aq = a + ka * pl, *cq = c + ka * pl;
            if (dots_l) {
                const int64_t needp = 2 * (int64_t)mg.x * mg.y * mg.z;  // (a partial per 128 x 8 tile, as k_level_march<8>)
                if (needp > s->gmg_part_cap) {
                    if (s->d_gmg_part) (void)hipFree(s->d_gmg_part);
                    s->d_gmg_part = nullptr;
                    PIB_HIP(hipMalloc(&s->d_gmg_part, sizeof(double) * (size_t)(3 * needp + 3 * BIG_STAGE)));
                    s->gmg_part_cap = needp;
                }
                double *part = s->d_gmg_part;
                const int part_stride = (int)s->gmg_part_cap;
                hipLaunchKernelGGL(k_prolong_smooth2<1>, mg, dim3(UNT), 0, q, S, dev_of(sub), dev_of(cg), omega, bq, xc, aq, cq, FZ, part, part_stride,
                                   (int)g.k0, (int)g.k1, pin_l);
                PIB_CHK(reduce_dots(s, part, part_stride, (int)needp, q));
                s->gmg_dots_done = true;
            } else
                hipLaunchKernelGGL(k_prolong_smooth2<0>, mg, dim3(UNT), 0, q, S, dev_of(sub), dev_of(cg), omega, bq, xc, aq, cq, FZ, (double *)nullptr, 0,
                                   (int)g.k0, (int)g.k1, pin_l);
            PIB_HIP(hipGetLastError());
            set_valid(c, I.dist ? fin : 0);
            std::swap(a, c);
            cur[(size_t)l] = a;
            if (l == 0 && a != z) return fail(PIB_ERR_LIB, "gmg: internal buffer parity error");
            continue;
        }
        // prolongation + first post-smoothing step in one march (the corrected iterate never goes to HBM)
        bool fused = false;
        // periodic levels: operator and transfers wrap alike, z with >= 8 planes (the ring of coarse planes counts through the seam)
        const bool per_ok = g.per == g.tper && (!(g.per & 4) || g.n[2] >= 8);
        if (s->cfg.fuse_prolong && !cheb && post >= 1 && g.plain_pair && per_ok && (!I.dist || e >= 1)) {
            const int o = I.dist ? std::min(std::min(e - 1, fin + post - 1), valid(b)) : 0;
            int64_t ka, kc;
            run(l, o, ka, kc);
            if (fused_run_ok(s, g, ka, kc) &&
                ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(b)) & 31u) == 0) {
                GridLevel sub = g;
                sub.k0 = g.k0 + ka;
                sub.k1 = sub.k0 + kc;
                const int FZ = march_planes(g, kc);
                const dim3 mg((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ));
                const int dlo = (int)std::max<int64_t>(0, -ka), dhi = (int)std::min<int64_t>(kc, I.nk - ka);
                if (post == 1 && dots_l) {
                    // the only post-smoothing step also delivers z.r, z.z, sum z (what mode 8 does)
                    const int64_t needp = (int64_t)mg.x * mg.y * mg.z;
                    if (needp > s->gmg_part_cap) {
                        if (s->d_gmg_part) (void)hipFree(s->d_gmg_part);
                        s->d_gmg_part = nullptr;
                        PIB_HIP(hipMalloc(&s->d_gmg_part, sizeof(double) * (size_t)(3 * needp + 3 * BIG_STAGE)));
                        s->gmg_part_cap = needp;
                    }
                    double *part = s->d_gmg_part;
                    const int part_stride = (int)s->gmg_part_cap;
                    hipLaunchKernelGGL(k_prolong_smooth<1>, mg, dim3(256), 0, q, S, dev_of(sub), dev_of(cg), omega, b + ka * pl, xc,
                                       a + ka * pl, c + ka * pl, pin_l, FZ, part, part_stride, dlo, dhi);
                    PIB_CHK(reduce_dots(s, part, part_stride, (int)needp, q));
                    s->gmg_dots_done = true;
                } else
                    hipLaunchKernelGGL(k_prolong_smooth<0>, mg, dim3(256), 0, q, S, dev_of(sub), dev_of(cg), omega, b + ka * pl, xc,
                                       a + ka * pl, c + ka * pl, pin_l, FZ, nullptr, 0, dlo, dhi);
                PIB_HIP(hipGetLastError());
                set_valid(c, o);
                std::swap(a, c);
                PIB_CHK(smooth_seq(l, b, pin_l, a, c, post - 1, false, fin, dots_l));
                fused = true;
            }
        }
        if (!fused) {
            int64_t ka, kc;
            run(l, e, ka, kc);
            GridLevel sub = g;
            sub.k0 = g.k0 + ka;
            sub.k1 = sub.k0 + kc;
            PIB_CHK(launch_prolong(sub, cg, xc, a + ka * pl, S, q));
            set_valid(a, e);
            PIB_CHK(smooth_seq(l, b, pin_l, a, c, post, false, fin, dots_l));
        }
        cur[(size_t)l] = a;
        if (l == 0 && a != z) return fail(PIB_ERR_LIB, "gmg: internal buffer parity error");
    }
    s->z_halo_depth = li[0].dist ? valid(z) : 0;
    return 0;
}

// y = A x with the matrix-free stencil (K2); x ghost-padded (halo exchanged here)
// ---- the Krylov product w = A p from the stencil twin (the device time step's Poisson solves on large grids:
// `pib_matrix_free_poisson`): 16 B/row for the product + 16 B/row for the p.w partials instead of the CSR's 104 B/row.
// Same operator to rounding (the twin sums face differences, the CSR row sums products: verified to 1e-10 at registration).
// PINNED: row 0 of the matrix is the identity; its column is zero, which the twin reproduces as long as p[0] = 0 -- the
// Krylov vectors keep that entry at zero.
__global__ __launch_bounds__(64) void k_twin_row0(const Scalars *__restrict__ S, const double *__restrict__ x, double *__restrict__ y)
{
    if (S != nullptr && S->done) return;
    if (threadIdx.x == 0) y[0] = x[0];
}
__global__ __launch_bounds__(256) void k_twin_dot(const Scalars *__restrict__ S, int64_t n, const double *__restrict__ x,
                                                  const double *__restrict__ y, double *__restrict__ part)
{
    if (S != nullptr && S->done) return;
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(lo + chunk, n);
    double v = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += x[i] * y[i];
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// x.y as spmv_launch_blocks() fixed-order partial sums (what the SpMV's fused dot delivers)
int dot_partials(pib_solver *s, const double *x, const double *y, double *part, bool guarded, hipStream_t q)
{
    hipLaunchKernelGGL(k_twin_dot, dim3((unsigned)spmv_launch_blocks()), dim3(256), 0, q, guarded ? s->d_s : nullptr, s->A.n, x, y, part);
    PIB_HIP(hipGetLastError());
    return 0;
}

bool stencil_matmult_ok(const pib_solver *s)
{
    return s->cfg.matrix_free_poisson == 1 && s->has_grid && !s->hint_pc_only && !s->levels.empty() && s->comm.nranks == 1 &&
           s->A.n >= ((int64_t)1 << 20);
}

int stencil_matmult(pib_solver *s, const double *x, double *y, double *dot_part, bool guarded, hipStream_t q)
{
    const GridLevel &g = s->levels[0];
    const Scalars *S = guarded ? s->d_s : nullptr;
    // the LDS-tiled march reads x once (the streaming kernel leans on the caches for the six neighbours) and leaves the
    // x.y partials as it goes; x[0] = 0 under a pinned pressure, so the identity row patched below adds nothing to x.y
    const int64_t nk = g.k1 - g.k0;
    const int FZ = march_planes(g, nk);
    const dim3 mg((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((nk + FZ - 1) / FZ));
    const bool march = s->cfg.march_levels && g.dim == 3 && march_run_ok(s, g, 0, nk) &&
                       ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31u) == 0 &&
                       (int64_t)mg.x * mg.y * mg.z <= spmv_launch_blocks();
    if (march) {
        hipLaunchKernelGGL((k_level_march<0>), mg, dim3(256), 0, q, S, dev_of(g), 0.0, (const double *)nullptr, x, y,
                           (const double *)nullptr, dot_part, spmv_launch_blocks(), FZ, 0, (int)nk);
        PIB_HIP(hipGetLastError());
        if (s->nullspace == PIB_NULLSPACE_PINNED) hipLaunchKernelGGL(k_twin_row0, dim3(1), dim3(64), 0, q, S, x, y);
        PIB_HIP(hipGetLastError());
        s->counters[0]++;
        return 0;
    }
    PIB_CHK(launch_level<0>(s, g, 0.0, nullptr, x, y, nullptr, guarded, q));
    if (s->nullspace == PIB_NULLSPACE_PINNED) hipLaunchKernelGGL(k_twin_row0, dim3(1), dim3(64), 0, q, S, x, y);
    if (dot_part) hipLaunchKernelGGL(k_twin_dot, dim3((unsigned)spmv_launch_blocks()), dim3(256), 0, q, S, s->A.n, x, y, dot_part);
    PIB_HIP(hipGetLastError());
    s->counters[0]++;
    return 0;
}

int stencil_apply(pib_solver *s, double *x_owned, double *y, hipStream_t q)
{
    if (!s->has_grid) return fail(PIB_ERR_ORDER, "stencil apply without grid structure");
    const GridLevel &g = s->levels[0];
    if (s->comm.nranks > 1 && !g.replicated) PIB_CHK(exchange_planes(s, g, x_owned, 1, q));
    return launch_level<0>(s, g, 0.0, nullptr, x_owned, y, nullptr, false, q, nullptr, 0.0);
}

}  // namespace pib
