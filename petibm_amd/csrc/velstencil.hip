// velstencil.hip -- the velocity operator A = I/dt - c nu L applied matrix-free: what the CSR SpMV of the
// matrix assembled by pib_assemble_velocity computes, row for row, from the 1-D mesh tables -- 56 B/row of HBM traffic
// (x with its six neighbours mostly from cache, y) instead of 104 B/row of values, columns and offsets.
//
// Same numbers as the assembled matrix: an entry is (1 / (dLNeg * dLSelf)) * scale (createlaplacian.cpp:134-148 and
// MatScale, navierstokes.cpp:342-344), the diagonal ((-sum) [+ ghost folds]) * scale + shift (:151, :232-243, MatShift),
// and a row is summed by ascending column from 0.0 like the CSR kernels do -- the results are bit-identical to the SpMV
// (tests/test_gpu_navierstokes.py).  Interior points (no ghost, no periodic wrap in the stencil) take a branch-free path
// with j and k workgroup-uniform; the outermost layer goes through the general form, one point per lane in a dense
// enumeration.  Single rank (the device time step is single-GPU); the Krylov solvers use it for their products when the
// matrix came from pib_assemble_velocity and `pib_matrix_free_velocity` is on (default).
#include <algorithm>
#include <cstring>

#include "pib_internal.hpp"

namespace pib {

struct VelDev {
    int dim, per;
    int64_t n[3][3], off[3];
    const double *lneg[3][3], *lpos[3][3];  // [field][direction], index s
    double a0[3][6];
    double scale, shift;
    // y = A (opc * (dinv o x)) when dinv != nullptr: the right-preconditioned products of BiCGStab with the Jacobi sweep
    // applied as the input is read (the same product opc * (dinv[i] * x[i]) a stored copy would hold: same bits)
    const double *dinv;
    double opc;
    // fused Krylov sums (k_vel_product only): dot_mode 1: y . other ; 2: y . x (the RAW input) and y . y -- one partial per
    // workgroup at dot_part[blk] (and dot_part[dot_stride + blk]); 4: those two, then x . x, other . x, other . y (BiCGStab with the
    // residual update merged into the next p-update, krylov.hip: the sums |r|^2 and r . rp are formed from these five)
    int dot_mode;
    const double *dot_other;
    double *dot_part;
    int dot_stride;
    // dot_mode 3: the Chebyshev iteration's update behind the product (krylov.hip: solve_chebyshev) -- with w = A x of a cell:
    // r = ch_b - w ; z = ch_opc * (ch_dinv * r) (or r) ; ch_pm = (a0 ch_pm + omega x) + cz z instead of storing w; sums |r|^2, |z|^2
    const double *ch_b, *ch_dinv;
    double *ch_pm;
    double ch_opc;
    // several ranks (see VelStencil): the slab axis, the ghost-pad places of the neighbours' planes, who has a neighbour
    int slab_axis;
    int64_t pad_lo[3], pad_hi[3];
    int has_lo, has_hi;
};
__device__ __forceinline__ double vel_in(const VelDev &V, const double *__restrict__ x, int64_t idx)
{
    const double v = x[idx];
    return V.dinv != nullptr ? V.opc * (V.dinv[idx] * v) : v;
}

// general form of one row (ghost folds, periodic wraps)
__device__ __forceinline__ double vel_row(const VelDev &V, const double *__restrict__ x, int f, int64_t i, int64_t j, int64_t k)
{
    const int64_t ijk[3] = {i, j, k};
    double v[6] = {0, 0, 0, 0, 0, 0};
    bool interior[6] = {false, false, false, false, false, false};
    bool anywrap = false;
    double acc = 0.0;
    for (int d = 0; d < V.dim; ++d) {
        const int64_t s = ijk[d];
        v[2 * d] = V.lneg[f][d][s];
        v[2 * d + 1] = V.lpos[f][d][s];
        const bool wrap = (V.per >> d) & 1;
        // on a slab the first / last plane's neighbour along the slab axis exists where a neighbour rank does (its plane
        // sits in a ghost pad); elsewhere that plane is a true boundary
        interior[2 * d] = s > 0 || wrap || (d == V.slab_axis && V.has_lo);
        interior[2 * d + 1] = s < V.n[f][d] - 1 || wrap || (d == V.slab_axis && V.has_hi);
        anywrap = anywrap || (wrap && (s == 0 || s == V.n[f][d] - 1));
        acc = acc + v[2 * d];
        acc = acc + v[2 * d + 1];
    }
    double diag = -acc;
    for (int q = 0; q < 2 * V.dim; ++q)
        if (!interior[q]) {
            const double t = v[q] * V.a0[f][q];
            if (t != 0.0) diag = diag + t;
        }
    const double dval = diag * V.scale + V.shift;
    const int64_t st[3] = {1, V.n[f][0], V.n[f][0] * V.n[f][1]};
    const int64_t p = V.off[f] + i + V.n[f][0] * (j + V.n[f][1] * k);
    // the neighbour's place in the vector: in the block, or -- across a slab end -- in a ghost pad (the low pad precedes and
    // the high pad follows every owned entry: the ascending-column order of the CSR row is the natural order still)
    const int sa = V.slab_axis;
    const int64_t inplane = (sa < 0) ? 0 : (p - V.off[f]) % st[sa];
    auto below = [&](int d) { return (d == sa && ijk[d] == 0) ? V.pad_lo[f] + inplane : p - st[d]; };
    auto above = [&](int d) { return (d == sa && ijk[d] == V.n[f][d] - 1) ? V.pad_hi[f] + inplane : p + st[d]; };
    double s = 0.0;
    if (!anywrap) {
        for (int d = V.dim - 1; d >= 0; --d)
            if (interior[2 * d]) s = s + (v[2 * d] * V.scale) * vel_in(V, x, below(d));
        s = s + dval * vel_in(V, x, p);
        for (int d = 0; d < V.dim; ++d)
            if (interior[2 * d + 1]) s = s + (v[2 * d + 1] * V.scale) * vel_in(V, x, above(d));
        return s;
    }
    int64_t ec[7];
    double ev[7];
    int ne = 1;
    ec[0] = p;
    ev[0] = dval;
    for (int q = 0; q < 2 * V.dim; ++q) {
        if (!interior[q]) continue;
        const int d = q >> 1;
        int64_t c;
        if (d == sa) c = (q & 1) ? above(d) : below(d);  // never wraps locally: the pads
        else if (!(q & 1)) c = (ijk[d] == 0) ? p + (V.n[f][d] - 1) * st[d] : p - st[d];
        else c = (ijk[d] == V.n[f][d] - 1) ? p - (V.n[f][d] - 1) * st[d] : p + st[d];
        int t = ne++;
        while (t > 0 && ec[t - 1] > c) {
            ec[t] = ec[t - 1];
            ev[t] = ev[t - 1];
            --t;
        }
        ec[t] = c;
        ev[t] = v[q] * V.scale;
    }
    for (int t = 0; t < ne; ++t) s = s + ev[t] * vel_in(V, x, ec[t]);
    return s;
}

// the Chebyshev update of one cell (VelDev::ch_*; a0, omega, cz: the coefficients of this pass, from the device scalars)
struct VelEpi {
    const double *b, *dinv;
    double *pm;
    double opc, a0, om, cz;
};
__device__ __forceinline__ double vel_epilogue(const VelEpi &E, double w, double bv, double dv, double pmv, double xv, double *acc)
{
    const double r = bv - w;
    const double z = E.dinv != nullptr ? E.opc * (dv * r) : r;
    acc[0] += r * r;
    acc[1] += z * z;
    return (E.a0 * pmv + E.om * xv) + E.cz * z;
}

// the outermost layer of component f (all = 1: every point), dealt to `nblk` workgroups of which this is number `blk`
__device__ __forceinline__ void vel_shell_part(const VelDev &V, int f, int all, const double *__restrict__ x, double *__restrict__ y,
                                               int64_t blk, int64_t nblk, double *acc = nullptr, bool epi = false,
                                               const VelEpi E = VelEpi())
{
    const int64_t nx = V.n[f][0], ny = V.n[f][1], nz = V.n[f][2];
    const bool three = V.dim == 3;
    const int64_t cx = 2 * ny * nz, cy = 2 * (nx - 2) * nz, cz = three ? 2 * (nx - 2) * (ny - 2) : 0;
    const int64_t total = all == 2 ? 2 * nx * ny : (all ? nx * ny * nz : cx + cy + cz);
    for (int64_t t = blk * 256 + threadIdx.x; t < total; t += nblk * 256) {
        int64_t i, j, k;
        if (all == 2) {  // the first and the last plane, whole (contiguous): the tiles of the one-launch product own the x / y edges of the planes between
            const int64_t q = t % (nx * ny);
            i = q % nx;
            j = q / nx;
            k = t < nx * ny ? 0 : nz - 1;
        } else if (all) {
            i = t % nx;
            j = (t / nx) % ny;
            k = t / (nx * ny);
        } else if (t < cx) {
            i = (t & 1) ? nx - 1 : 0;
            j = (t >> 1) % ny;
            k = (t >> 1) / ny;
        } else if (t < cx + cy) {
            const int64_t q = t - cx;
            j = (q & 1) ? ny - 1 : 0;
            i = 1 + (q >> 1) % (nx - 2);
            k = (q >> 1) / (nx - 2);
        } else {
            const int64_t q = t - cx - cy;
            k = (q & 1) ? nz - 1 : 0;
            i = 1 + (q >> 1) % (nx - 2);
            j = 1 + (q >> 1) / (nx - 2);
        }
        const int64_t p = V.off[f] + i + nx * (j + ny * k);
        const double v = vel_row(V, x, f, i, j, k);
        if (epi) {
            E.pm[p] = vel_epilogue(E, v, E.b[p], E.dinv != nullptr ? E.dinv[p] : 1.0, E.pm[p], x[p], acc);
            continue;
        }
        y[p] = v;
        if (acc != nullptr) {
            if (V.dot_mode == 1) acc[0] += v * V.dot_other[p];
            else {
                const double xv = x[p];
                acc[0] += v * xv;
                acc[1] += v * v;
                if (V.dot_mode == 4) {
                    const double ov = V.dot_other[p];
                    acc[2] += xv * xv;
                    acc[3] += ov * xv;
                    acc[4] += ov * v;
                }
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_vel_shell(const Scalars *__restrict__ S, VelDev V, int f, int all,
                                                   const double *__restrict__ x, double *__restrict__ y)
{
    if (S != nullptr && S->done) return;
    vel_shell_part(V, f, all, x, y, blockIdx.x, gridDim.x);
}

// interior rows (j, k) of component f, cells 1 + bx * 256 + lane (+ stride)
template <int DIM>
__device__ __forceinline__ void vel_interior_part(const VelDev &V, int f, const double *__restrict__ x, double *__restrict__ y, int bx,
                                                  int nbx, int j, int k);
template <int DIM>
__global__ __launch_bounds__(256) void k_vel_interior(const Scalars *__restrict__ S, VelDev V, int f, const double *__restrict__ x,
                                                      double *__restrict__ y)
{
    if (S != nullptr && S->done) return;
    vel_interior_part<DIM>(V, f, x, y, blockIdx.x, gridDim.x, blockIdx.y + 1, (DIM == 3) ? blockIdx.z + 1 : 0);
}
template <int DIM>
__device__ __forceinline__ void vel_interior_part(const VelDev &V, int f, const double *__restrict__ x, double *__restrict__ y, int bx,
                                                  int nbx, int j, int k)
{
    const int nx = (int)V.n[f][0];
    const int64_t sy = V.n[f][0], sz = sy * V.n[f][1];
    const int64_t base = V.off[f] + sy * j + sz * k;
    const double yneg = V.lneg[f][1][j], ypos = V.lpos[f][1][j];
    const double zneg = (DIM == 3) ? V.lneg[f][2][k] : 0.0, zpos = (DIM == 3) ? V.lpos[f][2][k] : 0.0;
    const double *__restrict__ xn = V.lneg[f][0], *__restrict__ xp = V.lpos[f][0];
    for (int i = 1 + bx * 256 + threadIdx.x; i < nx - 1; i += nbx * 256) {
        const int64_t p = base + i;
        const double xneg = xn[i], xpos = xp[i];
        double acc = 0.0;
        acc = acc + xneg;
        acc = acc + xpos;
        acc = acc + yneg;
        acc = acc + ypos;
        if (DIM == 3) {
            acc = acc + zneg;
            acc = acc + zpos;
        }
        const double diag = -acc;
        const double dval = diag * V.scale + V.shift;
        double s = 0.0;
        if (DIM == 3) s = s + (zneg * V.scale) * x[p - sz];
        s = s + (yneg * V.scale) * x[p - sy];
        s = s + (xneg * V.scale) * x[p - 1];
        s = s + dval * x[p];
        s = s + (xpos * V.scale) * x[p + 1];
        s = s + (ypos * V.scale) * x[p + sy];
        if (DIM == 3) s = s + (zpos * V.scale) * x[p + sz];
        y[p] = s;
    }
}

static VelDev vel_dev(const VelStencil &h)
{
    VelDev V;
    V.dim = h.dim;
    V.per = h.per;
    V.scale = h.scale;
    V.shift = h.shift;
    V.dinv = nullptr;
    V.opc = 1.0;
    V.dot_mode = 0;
    V.dot_other = nullptr;
    V.dot_part = nullptr;
    V.dot_stride = 0;
    V.slab_axis = h.slab_axis;
    V.has_lo = h.has_lo ? 1 : 0;
    V.has_hi = h.has_hi ? 1 : 0;
    for (int f = 0; f < 3; ++f) {
        V.pad_lo[f] = h.pad_lo[f];
        V.pad_hi[f] = h.pad_hi[f];
    }
    for (int f = 0; f < 3; ++f) {
        V.off[f] = h.off[f];
        for (int d = 0; d < 3; ++d) {
            V.n[f][d] = h.n[f][d];
            V.lneg[f][d] = h.lneg[f][d];
            V.lpos[f][d] = h.lpos[f][d];
        }
        for (int q = 0; q < 6; ++q) V.a0[f][q] = h.a0[f][q];
    }
    return V;
}

// Small operators (the 2-D cases of the reference, a few 10^5 rows): the whole product in ONE launch of the streaming forms --
// the shells of the components first, their interior rows behind -- instead of two launches per component; a time step of
// those cases is a chain of ~5 us launches.  Same device functions, same bits.
struct VelSmallPlan {
    int first[7];   // first workgroup of: shell of component 0, 1, 2, interior of component 0, 1, 2, end
    int all[3];     // the component has no interior: its shell part takes every row
    int nbx[3];     // workgroups per interior grid line
};
template <int DIM>
__global__ __launch_bounds__(256) void k_vel_product_small(const Scalars *__restrict__ S, VelDev V, VelSmallPlan P,
                                                           const double *__restrict__ x, double *__restrict__ y)
{
    if (S != nullptr && S->done) return;
    const int b = blockIdx.x;
    if (b < P.first[3]) {
        const int f = (b >= P.first[1]) + (b >= P.first[2]);
        vel_shell_part(V, f, P.all[f], x, y, b - P.first[f], P.first[f + 1] - P.first[f]);
        return;
    }
    const int f = (b >= P.first[4]) + (b >= P.first[5]);
    const int lb = b - P.first[3 + f], nbx = P.nbx[f];
    const int bx = lb % nbx, row = lb / nbx;
    const int nyi = (int)V.n[f][1] - 2;
    vel_interior_part<DIM>(V, f, x, y, bx, nbx, row % nyi + 1, (DIM == 3) ? row / nyi + 1 : 0);
}

// The same rows, four cells per lane (grid lines of a multiple of four points that start on a 32-byte boundary: every
// periodic direction, every component across its own direction with an even count): the centre and the +-y / +-z
// neighbours are one 32-byte access each, the x coefficients one vector load per table -- eleven vector-memory
// instructions for four rows instead of thirty-six (the one-cell form is bound by their issue rate: 2.1 TB/s).  The
// arithmetic per row is unchanged: bit-identical.  A wave owns one grid line; the cells i = 0 and nx - 1 of the first
// and last group belong to the shell kernel and are not stored.
template <int DIM>
__global__ __launch_bounds__(256) void k_vel_interior4(const Scalars *__restrict__ S, VelDev V, int f, const double *__restrict__ x,
                                                       double *__restrict__ y)
{
    if (S != nullptr && S->done) return;
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int nx = (int)V.n[f][0], ny = (int)V.n[f][1];
    const int j = blockIdx.y * 4 + threadIdx.y + 1, k = (DIM == 3) ? blockIdx.z + 1 : 0;
    if (j >= ny - 1) return;
    const int64_t sy = V.n[f][0], sz = sy * V.n[f][1];
    const int64_t base = V.off[f] + sy * j + sz * k;
    const double yneg = V.lneg[f][1][j], ypos = V.lpos[f][1][j];
    const double zneg = (DIM == 3) ? V.lneg[f][2][k] : 0.0, zpos = (DIM == 3) ? V.lpos[f][2][k] : 0.0;
    const double *__restrict__ xn = V.lneg[f][0], *__restrict__ xp = V.lpos[f][0];
    for (int i0 = 4 * (blockIdx.x * 64 + threadIdx.x); i0 < nx; i0 += gridDim.x * 256) {
        const int64_t p = base + i0;
        const v4 xc = *reinterpret_cast<const v4 *>(x + p);
        const v4 ym = *reinterpret_cast<const v4 *>(x + p - sy), yp = *reinterpret_cast<const v4 *>(x + p + sy);
        v4 zm = {0, 0, 0, 0}, zp = {0, 0, 0, 0};
        if (DIM == 3) {
            zm = *reinterpret_cast<const v4 *>(x + p - sz);
            zp = *reinterpret_cast<const v4 *>(x + p + sz);
        }
        const double xl = x[p - 1], xr = x[p + 4];
        const v4 vn = *reinterpret_cast<const v4 *>(xn + i0), vp = *reinterpret_cast<const v4 *>(xp + i0);
        v4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double xneg = vn[c], xpos = vp[c];
            double acc = 0.0;
            acc = acc + xneg;
            acc = acc + xpos;
            acc = acc + yneg;
            acc = acc + ypos;
            if (DIM == 3) {
                acc = acc + zneg;
                acc = acc + zpos;
            }
            const double diag = -acc;
            const double dval = diag * V.scale + V.shift;
            const double left = (c == 0) ? xl : xc[c - 1], right = (c == 3) ? xr : xc[c + 1];
            double s = 0.0;
            if (DIM == 3) s = s + (zneg * V.scale) * zm[c];
            s = s + (yneg * V.scale) * ym[c];
            s = s + (xneg * V.scale) * left;
            s = s + dval * xc[c];
            s = s + (xpos * V.scale) * right;
            s = s + (ypos * V.scale) * yp[c];
            if (DIM == 3) s = s + (zpos * V.scale) * zp[c];
            out[c] = s;
        }
        if (i0 > 0 && i0 + 4 < nx)
            *reinterpret_cast<v4 *>(y + p) = out;
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (i0 + c >= 1 && i0 + c <= nx - 2) y[p + c] = out[c];
        }
    }
}

// ---- the same interior rows, 2.5-D blocked like the multigrid's level kernels (gmg.hip k_level_march): a workgroup owns
// a 128 x 8 tile of the component's planes and marches through MZ planes; a thread keeps its four cells' z neighbours in
// registers and only the current plane (tile + one halo cell in x and y) sits in LDS, double-buffered -- x is read once
// (1.27 x with the tile halos) and y written once, 16 B/row, where the streaming forms above lean on the caches for six
// of their seven reads (2.0 TB/s of algorithmic traffic at 256^3).  Same expressions in the same order as
// k_vel_interior: bit-identical.  Components whose grid lines are a multiple of 128 points on a 32-byte boundary (every
// component of a periodic box, the components across their own direction of a wall-bounded one); 3-D.
constexpr int VX = 128, VY = 8, VSX = VX + 4, VSY = VY + 2;
// LDS column of cell t (-1 .. VX) of a tile row.  With four CONSECUTIVE cells per thread the plain layout t + 1 makes every
// LDS access of a wave a stride of 32 bytes -- an 8-way bank conflict: SQ counters had the LDS pipe busy for the whole
// kernel, 58 % of it conflicts (profiles/r03_velocity256_pmc_SQ.md).  Cells are therefore dealt to four groups of 33 columns
// by t mod 4: the lanes of a wave (consecutive tx, the same c) then touch consecutive columns for the centre, the left and
// the right neighbour alike.  Cells 32 apart per thread (the other form) are lane-consecutive in the plain layout already.
template <bool V4>
__device__ __forceinline__ int vcol(int t)
{
    return V4 ? (t & 3) * 33 + (t >> 2) + 1 : t + 1;
}
// V4: a thread's four cells are consecutive (aligned 32-byte accesses; grid lines a multiple of 128 points on a 32-byte
// boundary).  Otherwise they are 32 cells apart (lane-consecutive 8-byte accesses, any line length and alignment: the
// 255-point lines of a wall-bounded component along its own direction; the last tile of a line is partial).
// EDGES (wall-bounded x and y; the one-launch product): the tile also produces the rows of its cells on the component's x / y
// boundaries -- the same sums with the missing neighbour left out and its ghost fold on the diagonal, in vel_row's order: the
// bits the shell computes -- so that the shell workgroups only own the first and the last plane.  (The x faces of a shell are
// one strided point per lane: 40 of the product's 240 us at 256^3.)
template <bool V4, bool EDGES = false, bool EPI = false, bool D5 = false>
__device__ __forceinline__ void vel_march_tile(const VelDev &V, int f, const double *__restrict__ x, double *__restrict__ y, int MZ,
                                               int bx, int by, int bz, double (&sp)[2][VSY][VSX], double *acc = nullptr,
                                               const VelEpi E = VelEpi())
{
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int nx = (int)V.n[f][0], ny = (int)V.n[f][1], nz = (int)V.n[f][2];
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const int i0 = bx * VX, j0 = by * VY;
    const int k0 = 1 + bz * MZ, kend = min(k0 + MZ, nz - 1);  // interior planes [1, nz - 1)
    const int64_t sy = nx, sz = (int64_t)nx * ny;
    const int j = j0 + ty;
    const int jc = min(j, ny - 1);  // a partial tile's rows / cells beyond the component are clamped for the loads, never stored
    int ci[4], lx[4], lxm[4], lxp[4];  // global index (clamped), LDS column of the thread's cells and of their x neighbours
    bool cin[4];                    // interior cell of the line
    bool exm[4], exq[4], cok[4];    // EDGES: first / last cell of the line, a cell of the component at all
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int t = V4 ? 4 * tx + c : tx + 32 * c;
        exm[c] = i0 + t == 0;
        exq[c] = i0 + t == nx - 1;
        cok[c] = i0 + t < nx;
        lx[c] = vcol<V4>(t);
        lxm[c] = vcol<V4>(t - 1);
        lxp[c] = vcol<V4>(t + 1);
        cin[c] = i0 + t >= 1 && i0 + t <= nx - 2;
        ci[c] = min(i0 + t, nx - 1);
    }
    const int hy_row = (tid < 128) ? -1 : VY, hy_x = tid & 127;
    const int hx_col = (tid & 1) ? VX : -1, hx_y = (tid >> 1) & 7;
    const int hy_lx = vcol<V4>(hy_x), hx_lx = vcol<V4>(hx_col);
    const int hyj = j0 + hy_row, hyi = i0 + hy_x, hxj = min(j0 + hx_y, ny - 1), hxi = i0 + hx_col;
    const bool hy_ok = hyj >= 0 && hyj < ny && hyi < nx, hx_ok = tid < 16 && hxi >= 0 && hxi < nx;
    const int64_t base = V.off[f];
    const int64_t row = base + (int64_t)jc * sy, off_hy = base + (int64_t)hyj * sy + hyi, off_hx = base + (int64_t)hxj * sy + hxi;
    const bool jin = j >= 1 && j <= ny - 2;
    const bool eym = EDGES && j == 0, eyq = EDGES && j == ny - 1, jok = j < ny;
    const double fxm = V.a0[f][0], fxq = V.a0[f][1], fym = V.a0[f][2], fyq = V.a0[f][3];
    const double yneg = V.lneg[f][1][jc], ypos = V.lpos[f][1][jc];
    double vn[4], vp[4], zm[4], xc[4], zp[4];
    const bool scaled = V.dinv != nullptr;
    // pl: the plane of x, kp: its index (the plane of dinv)
    auto load4 = [&](const double *pl, int kp, double (&o)[4]) {
        if (V4) {
            const v4 t = *reinterpret_cast<const v4 *>(pl + row + ci[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = t[c];
            if (scaled) {
                const v4 d = *reinterpret_cast<const v4 *>(V.dinv + (int64_t)kp * sz + row + ci[0]);
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = V.opc * (d[c] * o[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = pl[row + ci[c]];
            if (scaled) {
                const double *pd = V.dinv + (int64_t)kp * sz;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = V.opc * (pd[row + ci[c]] * o[c]);
            }
        }
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        vn[c] = V.lneg[f][0][ci[c]];
        vp[c] = V.lpos[f][0][ci[c]];
    }
    // the second factor of the fused sums (V.dot_other, or the raw input) for the thread's cells of plane kp: fetched a
    // plane ahead with the input planes -- behind the barrier it would be a load consumed at once, a memory round trip
    // on every plane's critical path
    const double *pbase = (acc != nullptr) ? (V.dot_mode == 1 ? V.dot_other : x) : nullptr;
    auto loadp = [&](int kp, double (&o)[4]) {
        const double *pl = pbase + (int64_t)kp * sz;
        if (V4) {
            const v4 t = *reinterpret_cast<const v4 *>(pl + row + ci[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = t[c];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = pl[row + ci[c]];
        }
    };
    double pc[4] = {0.0, 0.0, 0.0, 0.0}, pn[4] = {0.0, 0.0, 0.0, 0.0};
    double qc[4] = {0.0, 0.0, 0.0, 0.0}, qn[4] = {0.0, 0.0, 0.0, 0.0};  // D5 (dot_mode 4): V.dot_other beside the raw input
    double la[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // the tile's sums, in registers (added to acc[] -- zero on entry -- at the end)
    // EPI: b, 1 / a_ii and the vector the update overwrites, for the thread's cells of a plane -- fetched a plane ahead like the
    // sums' second factor
    double eb[4] = {0.0, 0.0, 0.0, 0.0}, ed[4] = {1.0, 1.0, 1.0, 1.0}, em[4] = {0.0, 0.0, 0.0, 0.0};
    double nb4[4] = {0.0, 0.0, 0.0, 0.0}, nd4[4] = {1.0, 1.0, 1.0, 1.0}, nm4[4] = {0.0, 0.0, 0.0, 0.0};
    auto loade1 = [&](const double *v, int kp, double (&o)[4]) {
        const double *pl = v + (int64_t)kp * sz;
        if (V4) {
            const v4 t = *reinterpret_cast<const v4 *>(pl + row + ci[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = t[c];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = pl[row + ci[c]];
        }
    };
    auto loade = [&](int kp, double (&ob)[4], double (&od)[4], double (&om)[4]) {
        loade1(E.b, kp, ob);
        if (E.dinv != nullptr) loade1(E.dinv, kp, od);
        loade1(E.pm, kp, om);
    };
    // Every load of a step is consumed in the NEXT step: the plane two ahead (it becomes the upper neighbour), the halo cells
    // and the fused sums' second factor of the plane one ahead.  Nothing a step issues is waited for in that step, so the
    // loads stay in flight across the barrier (a load consumed in the step that issues it drains the in-order memory counter
    // and with it every prefetch behind it: SQ counters had 77 % of the wave cycles waiting with < 3 memory instructions in
    // flight per CU).
    auto load_halo = [&](int kp, double &hyv, double &hxv) {
        const double *pl = x + (int64_t)kp * sz;
        hyv = hy_ok ? pl[off_hy] : 0.0;
        hxv = hx_ok ? pl[off_hx] : 0.0;
        if (scaled) {
            const double *pd = V.dinv + (int64_t)kp * sz;
            if (hy_ok) hyv = V.opc * (pd[off_hy] * hyv);
            if (hx_ok) hxv = V.opc * (pd[off_hx] * hxv);
        }
    };
    // With the Jacobi sweep folded into the read (scaled: the input is M^-1 p, formed as the planes come in) the products
    // opc (dinv x) used to stand right behind their loads, i.e. every load of a step was waited for on the spot after all (ISA:
    // four to five s_waitcnt vmcnt(0) per plane).  The raw values are kept instead -- x and 1 / a_ii of the plane two ahead and of
    // the next plane's halo cells -- and multiplied at the END of the step, a whole step after their loads went out (same
    // expression, same bits).  The plane's two z coefficients, read behind the barrier as vector loads, travel a step ahead too.
    auto raw4 = [&](const double *v, int kp, double (&o)[4]) {
        const double *pl = v + (int64_t)kp * sz;
        if (V4) {
            const v4 t = *reinterpret_cast<const v4 *>(pl + row + ci[0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = t[c];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = pl[row + ci[c]];
        }
    };
    double zn[4] = {0.0, 0.0, 0.0, 0.0}, znd[4] = {1.0, 1.0, 1.0, 1.0}, hyc, hxc, hyn = 0.0, hxn = 0.0, hynd = 1.0, hxnd = 1.0;
    load4(x + (int64_t)(k0 - 1) * sz, k0 - 1, zm);
    load4(x + (int64_t)k0 * sz, k0, xc);
    load_halo(k0, hyc, hxc);
    load4(x + (int64_t)(k0 + 1) * sz, k0 + 1, zp);
    if (!EPI && acc != nullptr) loadp(k0, pc);
    if (D5) loade1(V.dot_other, k0, qc);
    if (EPI) loade(k0, eb, ed, em);
    double zneg = V.lneg[f][2][k0], zpos = V.lpos[f][2][k0], znegn = 0.0, zposn = 0.0;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing pending on entry either
    for (int k = k0; k < kend; ++k) {
        const int slot = k & 1;
        if (k + 1 < kend) {
            raw4(x, k + 2, zn);
            if (scaled) raw4(V.dinv, k + 2, znd);
            {
                const double *pl = x + (int64_t)(k + 1) * sz;
                hyn = hy_ok ? pl[off_hy] : 0.0;
                hxn = hx_ok ? pl[off_hx] : 0.0;
                if (scaled) {
                    const double *pd = V.dinv + (int64_t)(k + 1) * sz;
                    hynd = hy_ok ? pd[off_hy] : 1.0;
                    hxnd = hx_ok ? pd[off_hx] : 1.0;
                }
            }
            if (!EPI && acc != nullptr) loadp(k + 1, pn);
            if (D5) loade1(V.dot_other, k + 1, qn);
            if (EPI) loade(k + 1, nb4, nd4, nm4);
            znegn = V.lneg[f][2][k + 1];
            zposn = V.lpos[f][2][k + 1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) sp[slot][ty + 1][lx[c]] = xc[c];
        sp[slot][hy_row + 1][hy_lx] = hyc;
        if (tid < 16) sp[slot][hx_y + 1][hx_lx] = hxc;
        __syncthreads();
        double out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double xneg = vn[c], xpos = vp[c];
            double acc = 0.0;
            acc = acc + xneg;
            acc = acc + xpos;
            acc = acc + yneg;
            acc = acc + ypos;
            acc = acc + zneg;
            acc = acc + zpos;
            double diag = -acc;
            const bool em = EDGES && exm[c], eq = EDGES && exq[c];
            if (EDGES) {  // a wall's ghost fold (vel_row: x-, x+, y-, y+; exact zeros are not added)
                double t = xneg * fxm;
                diag = (em && t != 0.0) ? diag + t : diag;
                t = xpos * fxq;
                diag = (eq && t != 0.0) ? diag + t : diag;
                t = yneg * fym;
                diag = (eym && t != 0.0) ? diag + t : diag;
                t = ypos * fyq;
                diag = (eyq && t != 0.0) ? diag + t : diag;
            }
            const double dval = diag * V.scale + V.shift;
            double s2 = 0.0;
            s2 = s2 + (zneg * V.scale) * zm[c];
            if (!eym) s2 = s2 + (yneg * V.scale) * sp[slot][ty][lx[c]];
            if (!em) s2 = s2 + (xneg * V.scale) * sp[slot][ty + 1][lxm[c]];
            s2 = s2 + dval * xc[c];
            if (!eq) s2 = s2 + (xpos * V.scale) * sp[slot][ty + 1][lxp[c]];
            if (!eyq) s2 = s2 + (ypos * V.scale) * sp[slot][ty + 2][lx[c]];
            s2 = s2 + (zpos * V.scale) * zp[c];
            out[c] = s2;
        }
        if (EDGES) {  // every cell of the component in this tile is stored here
#pragma unroll
            for (int c = 0; c < 4; ++c) cin[c] = cok[c];
        }
        if (EPI) {  // the update of the stored cells takes the product's place
            if (EDGES ? jok : jin) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (cin[c]) out[c] = vel_epilogue(E, out[c], eb[c], ed[c], em[c], xc[c], acc);
            }
        } else if ((EDGES ? jok : jin) && acc != nullptr) {  // the stored rows only (the shell owns the others)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (cin[c]) {
                    la[0] += out[c] * pc[c];
                    if (D5 || V.dot_mode == 2) la[1] += out[c] * out[c];
                    if (D5) {
                        la[2] += pc[c] * pc[c];
                        la[3] += qc[c] * pc[c];
                        la[4] += qc[c] * out[c];
                    }
                }
        }
        // Everything this step requested is waited for HERE, ahead of the stores (and on every path, through the builtin, which the
        // compiler's own wait insertion takes into account): loads and stores share one counter and may complete out of order
        // with each other, so a wait behind the stores -- where the compiler would put it, at the first use of the loaded values
        // -- waits for the stores too, a write latency on every plane.  Like this they are in flight during the whole next step.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (scaled && k + 1 < kend) {  // the sweep on what came in during this step (load4 / load_halo's expression)
#pragma unroll
            for (int c = 0; c < 4; ++c) zn[c] = V.opc * (znd[c] * zn[c]);
            if (hy_ok) hyn = V.opc * (hynd * hyn);
            if (hx_ok) hxn = V.opc * (hxnd * hxn);
        }
        if (EDGES ? jok : jin) {
            double *py = (EPI ? E.pm : y) + (int64_t)k * sz + row;
            if (V4 && cin[0] && cin[3]) {
                v4 t;
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = out[c];
                *reinterpret_cast<v4 *>(py + ci[0]) = t;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (cin[c]) py[ci[c]] = out[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            zm[c] = xc[c];
            xc[c] = zp[c];
            zp[c] = zn[c];
            pc[c] = pn[c];
            if (D5) qc[c] = qn[c];
            if (EPI) {
                eb[c] = nb4[c];
                ed[c] = nd4[c];
                em[c] = nm4[c];
            }
        }
        hyc = hyn;
        hxc = hxn;
        zneg = znegn;
        zpos = zposn;
    }
    if (!EPI && acc != nullptr) {
        acc[0] += la[0];
        acc[1] += la[1];
        if (D5) {
            acc[2] += la[2];
            acc[3] += la[3];
            acc[4] += la[4];
        }
    }
}

template <bool V4>
__global__ __launch_bounds__(256) void k_vel_march(const Scalars *__restrict__ S, VelDev V, int f, const double *__restrict__ x,
                                                   double *__restrict__ y, int MZ)
{
    if (S != nullptr && S->done) return;
    __shared__ double sp[2][VSY][VSX];
    vel_march_tile<V4>(V, f, x, y, MZ, blockIdx.x, blockIdx.y, blockIdx.z, sp);
}

// ---- the whole product in ONE launch (3-D, every component on the marching path): six back-to-back launches -- three
// marches of ~70 us and three shells of ~30 us at 256^3 -- each ramp up and drain the chip on their own, and the shells
// (one strided point per lane) are latency-bound.  Here the shell workgroups of all components come first in the grid and
// the tiles follow, so the shells' loads are in flight while the tiles stream.  Same device functions, same bits.
struct VelPlan {
    int edges;          // wall-bounded x and y: the tiles produce their own boundary cells, the shell is the first and the last plane
    int first[7];       // first workgroup of: shell of component 0, 1, 2, tiles of component 0, 1, 2, end
    int gx[3], gy[3];   // tiles per plane of a component
    int v4[3];
};
template <int DOT>  // DOT = 1: with the fused Krylov sums (V.dot_mode 1, 2), 2: with the Chebyshev update (dot_mode 3), 3: the five sums of dot_mode 4; separate instantiations keep the plain product's registers
__global__ __launch_bounds__(256) void k_vel_product(const Scalars *__restrict__ S, VelDev V, VelPlan P, const double *__restrict__ x,
                                                     double *__restrict__ y, int MZ)
{
    if (S != nullptr && S->done) return;
    __shared__ double sp[2][VSY][VSX];
    const int b = blockIdx.x;
    int tile_slot = b;
    constexpr int NACC = (DOT == 3) ? 5 : 2;
    double acc[NACC];
#pragma unroll
    for (int k2 = 0; k2 < NACC; ++k2) acc[k2] = 0.0;
    double *pa = DOT ? acc : nullptr;
    VelEpi epi = {V.ch_b, V.ch_dinv, V.ch_pm, V.ch_opc, 0.0, 0.0, 0.0};
    if (DOT == 2) {  // (S is never null here: the coefficients of the pass live in the device scalars)
        epi.a0 = S->cheb_a0;
        epi.om = S->omega;
        epi.cz = S->cheb_cz;
    }
    if (b < P.first[3]) {
        const int f = (b >= P.first[1]) + (b >= P.first[2]);
        vel_shell_part(V, f, P.edges ? 2 : 0, x, y, b - P.first[f], P.first[f + 1] - P.first[f], pa, DOT == 2, epi);
    } else {
        const int f = (b >= P.first[4]) + (b >= P.first[5]);
        const int lb = b - P.first[3 + f];
        int bx = lb % P.gx[f], by = (lb / P.gx[f]) % P.gy[f], bz = lb / (P.gx[f] * P.gy[f]);
#ifndef PIB_NO_XCD_BANDS
        // workgroups b, b + 8, ... share an XCD and its L2: each of the eight classes takes a contiguous band of y-tiles (all
        // x-tiles of it, chunk of planes after chunk), so that the halo rows / columns neighbouring tiles both read are
        // fetched from HBM once (gmg.hip: tile_of_block)
        if ((P.gy[f] & 7) == 0) {
            const int xcd = lb & 7, m = lb >> 3, band = P.gy[f] >> 3;
            const int r = m / P.gx[f];
            bx = m - r * P.gx[f];
            by = xcd * band + r % band;
            bz = r / band;
        }
#endif
        tile_slot = P.first[3 + f] + bx + P.gx[f] * (by + P.gy[f] * bz);  // the sums stay with the tile: same order as ever
        if (P.edges) {
            if (P.v4[f]) vel_march_tile<true, true, DOT == 2, DOT == 3>(V, f, x, y, MZ, bx, by, bz, sp, pa, epi);
            else vel_march_tile<false, true, DOT == 2, DOT == 3>(V, f, x, y, MZ, bx, by, bz, sp, pa, epi);
        } else {
            if (P.v4[f]) vel_march_tile<true, false, DOT == 2, DOT == 3>(V, f, x, y, MZ, bx, by, bz, sp, pa, epi);
            else vel_march_tile<false, false, DOT == 2, DOT == 3>(V, f, x, y, MZ, bx, by, bz, sp, pa, epi);
        }
    }
    if (DOT) {  // one partial per workgroup and sum, fixed order
        __shared__ double sh[NACC][4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int k2 = 0; k2 < NACC; ++k2) {
            double v = acc[k2];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) sh[k2][w] = v;
        }
        __syncthreads();
        if (threadIdx.x < NACC && (threadIdx.x == 0 || V.dot_mode >= 2))
            V.dot_part[(int64_t)threadIdx.x * V.dot_stride + tile_slot] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
    }
}

// second stage of the fused sums: the per-workgroup partials of slot y -> 64 partial sums in the Krylov solver's partial
// array (d_part[slot0 + y][0..63]), which its finalize kernels take from there
constexpr int VEL_STAGE = 64;
__global__ __launch_bounds__(256) void k_vel_reduce(const Scalars *__restrict__ S, const double *__restrict__ part, int stride, int count,
                                                    double *__restrict__ out, int slot0)
{
    if (S != nullptr && S->done) return;
    const double *p = part + (int64_t)blockIdx.y * stride;
    const int chunk = (count + VEL_STAGE - 1) / VEL_STAGE;
    const int lo = blockIdx.x * chunk, hi = min(lo + chunk, count);
    double v = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) v += p[i];
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[(int64_t)(slot0 + blockIdx.y) * PIB_MAXPART + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- verification of a recovered structure (structure.cpp): y = A x from the tables against the CSR SpMV
__global__ __launch_bounds__(256) void k_vel_verify_fill(int64_t n, double *__restrict__ x)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        x[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
__global__ __launch_bounds__(256) void k_vel_verify_diff(int64_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                         unsigned long long *__restrict__ out /* bits of max |a - b|, max |a| */)
{
    double d = 0.0, m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double e = fabs(a[i] - b[i]);
        d = (e > d || e != e) ? (e != e ? 1e300 : e) : d;
        m = fmax(m, fabs(a[i]));
    }
    // non-negative doubles order like their bit patterns
    atomicMax(&out[0], (unsigned long long)__double_as_longlong(d));
    atomicMax(&out[1], (unsigned long long)__double_as_longlong(m));
}
int vel_stencil_verify(pib_solver *s)
{
    const int64_t n = s->A.n;
    double *buf = nullptr;
    unsigned long long *out = nullptr;
    hipStream_t q = s->stream;
    double *x, *y1, *y2;
    if (s->comm.nranks > 1) {  // ghost-padded vectors, the neighbours' planes exchanged like a Krylov product's (collective)
        PIB_CHK(ensure_work(s, 3));
        x = s->vec(0);
        y1 = s->vec(1);
        y2 = s->vec(2);
    } else {
        PIB_HIP(hipMalloc(&buf, sizeof(double) * 3 * (size_t)n));
        x = buf;
        y1 = buf + n;
        y2 = buf + 2 * n;
    }
    PIB_HIP(hipMalloc(&out, 2 * sizeof(unsigned long long)));
    PIB_HIP(hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), q));
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(k_vel_verify_fill, dim3(nb), dim3(256), 0, q, n, x);
    if (s->comm.nranks > 1) PIB_CHK(halo_exchange(s, x, q));
    int err = spmv_rows(s, x, y1, 0, n, nullptr, false, q);
    if (!err) err = vel_stencil_apply(s, x, y2, false, q);
    if (!err) {
        hipLaunchKernelGGL(k_vel_verify_diff, dim3(nb), dim3(256), 0, q, n, y1, y2, out);
        unsigned long long h[2] = {0, 0};
        if (hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, q) != hipSuccess || hipStreamSynchronize(q) != hipSuccess)
            err = fail(PIB_ERR_LIB, "velocity structure check: HIP error");
        else {
            double d, m;
            std::memcpy(&d, &h[0], 8);
            std::memcpy(&m, &h[1], 8);
            if (!(d <= 1e-12 * m) || !(m > 0.0)) s->vel.valid = false;
        }
    }
    if (buf) (void)hipFree(buf);
    (void)hipFree(out);
    return err;
}

void vel_stencil_release(pib_solver *s)
{
    drop_iteration_graph(s);  // (krylov.hip: the captured iteration goes before the memory it points at)
    if (s->d_vel_part) (void)hipFree(s->d_vel_part);
    s->d_vel_part = nullptr;
    s->vel_part_cap = 0;
    for (double *p : s->vel.owned) (void)hipFree(p);
    s->vel = VelStencil();
}

// y = A x on the whole (single-rank) vector
// the one-launch product serves this operator (3-D, every component on the marching path): the form that can take its
// input through the Jacobi sweep (vel_stencil_apply's dinv / opc)
bool vel_stencil_fused_ok(const pib_solver *s)
{
    const VelStencil &h = s->vel;
    if (!h.valid || h.dim != 3 || !s->cfg.march || !s->cfg.fuse_velocity_product) return false;
    for (int f = 0; f < 3; ++f) {
        const int64_t nx = h.n[f][0], ny = h.n[f][1], nz = h.n[f][2];
        if (!(ny >= 3 && nz >= 3 && nx >= VX - 1 && nx * ny * nz >= std::min<int64_t>(s->cfg.march_min_cells, (int64_t)1 << 22))) return false;
    }
    return true;
}

// One pass of the Chebyshev iteration in one launch: with w = A x, pm <- (a0 pm + omega x) + cz M^-1 (b - w) and the partial
// sums of |b - w|^2, |M^-1 (b - w)|^2 in d_part[slot0], [slot0 + 1] (VEL_DOT_PARTIALS each); a0, omega, cz from the device scalars.
int vel_stencil_apply_cheb(pib_solver *s, const double *x, double *pm, const double *b, const double *dinv, double opc, hipStream_t q, int slot0)
{
    if (!vel_stencil_fused_ok(s)) return fail(PIB_ERR_ORDER, "velocity product with the Chebyshev update folded in: the one-launch form does not serve this operator");
    s->vel_epi_b = b;
    s->vel_epi_dinv = dinv;
    s->vel_epi_pm = pm;
    s->vel_epi_opc = opc;
    const int err = vel_stencil_apply(s, x, pm, true, q, nullptr, 1.0, 3, nullptr, slot0);
    s->vel_epi_pm = nullptr;
    return err;
}

int vel_stencil_apply(pib_solver *s, const double *x, double *y, bool guarded, hipStream_t q, const double *dinv, double opc,
                      int dot_mode, const double *dot_other, int dot_slot0)
{
    const VelStencil &h = s->vel;
    VelDev V = vel_dev(h);
    V.ch_b = V.ch_dinv = nullptr;
    V.ch_pm = nullptr;
    V.ch_opc = 1.0;
    if (dot_mode == 3) {
        V.ch_b = s->vel_epi_b;
        V.ch_dinv = s->vel_epi_dinv;
        V.ch_pm = s->vel_epi_pm;
        V.ch_opc = s->vel_epi_opc;
    }
    if (dinv != nullptr || dot_mode != 0) {
        if (!vel_stencil_fused_ok(s)) return fail(PIB_ERR_ORDER, "velocity product with the Jacobi sweep / the Krylov sums folded in: the one-launch form does not serve this operator");
        V.dinv = dinv;
        V.opc = opc;
        V.dot_mode = dot_mode;
        V.dot_other = dot_other;
    }
    const Scalars *S = guarded ? s->d_s : nullptr;
    const int MZ = 16;  // planes a workgroup of the march walks through (8 ... 64 measured flat within 2 %)
    auto marches = [&](int f) {
        const int64_t nx = h.n[f][0], ny = h.n[f][1], nz = h.n[f][2];
        return h.dim == 3 && s->cfg.march && ny >= 3 && nz >= 3 && nx >= VX - 1 &&
               nx * ny * nz >= std::min<int64_t>(s->cfg.march_min_cells, (int64_t)1 << 22);
    };
    if (h.dim == 3 && s->cfg.fuse_velocity_product && marches(0) && marches(1) && marches(2)) {
        VelPlan P;
        P.edges = ((h.per & 3) == 0) ? 1 : 0;  // wall-bounded x and y: the tiles produce their own x / y boundary cells, the shell is two planes
        int nb = 0;
        for (int f = 0; f < 3; ++f) {
            const int64_t nx = h.n[f][0], ny = h.n[f][1], nz = h.n[f][2];
            const int64_t shell = P.edges ? 2 * nx * ny : 2 * (ny * nz + (nx - 2) * nz + (nx - 2) * (ny - 2));
            P.first[f] = nb;
            nb += (int)std::min<int64_t>(4096, (shell + 255) / 256);
        }
        for (int f = 0; f < 3; ++f) {
            const int64_t nx = h.n[f][0], ny = h.n[f][1], nz = h.n[f][2];
            P.first[3 + f] = nb;
            P.gx[f] = (int)((nx + VX - 1) / VX);
            P.gy[f] = (int)((ny + VY - 1) / VY);
            P.v4[f] = (nx % VX == 0 && h.off[f] % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31u) == 0) ? 1 : 0;
            if (dot_mode == 3 && ((reinterpret_cast<uintptr_t>(V.ch_b) | reinterpret_cast<uintptr_t>(V.ch_dinv)) & 31u) != 0) P.v4[f] = 0;
            nb += P.gx[f] * P.gy[f] * (int)((nz - 2 + MZ - 1) / MZ);
        }
        P.first[6] = nb;
        if (dot_mode != 0) {
            if (s->vel_part_cap < nb) {
                if (s->d_vel_part) PIB_HIP(hipFree(s->d_vel_part));
                s->d_vel_part = nullptr;
                PIB_HIP(hipMalloc(&s->d_vel_part, sizeof(double) * 5 * (size_t)nb));
                s->vel_part_cap = nb;
            }
            V.dot_part = s->d_vel_part;
            V.dot_stride = s->vel_part_cap;
        }
        if (dot_mode == 3) hipLaunchKernelGGL(k_vel_product<2>, dim3((unsigned)nb), dim3(256), 0, q, s->d_s, V, P, x, y, MZ);
        else if (dot_mode == 4) hipLaunchKernelGGL(k_vel_product<3>, dim3((unsigned)nb), dim3(256), 0, q, S, V, P, x, y, MZ);
        else if (dot_mode != 0) hipLaunchKernelGGL(k_vel_product<1>, dim3((unsigned)nb), dim3(256), 0, q, S, V, P, x, y, MZ);
        else hipLaunchKernelGGL(k_vel_product<0>, dim3((unsigned)nb), dim3(256), 0, q, S, V, P, x, y, MZ);
        if (dot_mode != 0)
            hipLaunchKernelGGL(k_vel_reduce, dim3(VEL_STAGE, dot_mode == 4 ? 5 : dot_mode >= 2 ? 2 : 1), dim3(256), 0, q, S, s->d_vel_part, s->vel_part_cap, nb,
                               s->d_part, dot_slot0);
        PIB_HIP(hipGetLastError());
        s->counters[0]++;
        return 0;
    }
    if (s->cfg.fuse_velocity_product && h.off[h.dim - 1] + h.n[h.dim - 1][0] * h.n[h.dim - 1][1] * h.n[h.dim - 1][2] <= ((int64_t)1 << 21)) {
        VelSmallPlan P;
        int nb = 0;
        int64_t rows[3] = {0, 0, 0};
        for (int f = 0; f < 3; ++f) {
            P.first[f] = nb;
            P.all[f] = 0;
            P.nbx[f] = 1;
            if (f >= h.dim) continue;
            const int64_t nx = h.n[f][0], ny = h.n[f][1], nz = h.n[f][2];
            const bool inner = nx >= 3 && ny >= 3 && (h.dim == 2 || nz >= 3);
            P.all[f] = inner ? 0 : 1;
            const int64_t shell = inner ? 2 * (ny * nz + (nx - 2) * nz + (h.dim == 3 ? (nx - 2) * (ny - 2) : 0)) : nx * ny * nz;
            nb += (int)std::min<int64_t>(4096, (shell + 255) / 256);
            if (inner) {
                P.nbx[f] = (int)((nx - 2 + 255) / 256);
                rows[f] = (ny - 2) * (h.dim == 3 ? nz - 2 : 1);
            }
        }
        for (int f = 0; f < 3; ++f) {
            P.first[3 + f] = nb;
            nb += (int)(rows[f] * P.nbx[f]);
        }
        P.first[6] = nb;
        if (h.dim == 3) hipLaunchKernelGGL(k_vel_product_small<3>, dim3((unsigned)nb), dim3(256), 0, q, S, V, P, x, y);
        else hipLaunchKernelGGL(k_vel_product_small<2>, dim3((unsigned)nb), dim3(256), 0, q, S, V, P, x, y);
        PIB_HIP(hipGetLastError());
        s->counters[0]++;
        return 0;
    }
    for (int f = 0; f < h.dim; ++f) {
        const int64_t nx = h.n[f][0], ny = h.n[f][1], nz = h.n[f][2];
        const bool inner = nx >= 3 && ny >= 3 && (h.dim == 2 || nz >= 3);
        if (inner) {
            const bool vec4 = nx % 4 == 0 && h.off[f] % 4 == 0 &&
                              ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31u) == 0;
            if (marches(f)) {
                const dim3 grid((unsigned)((nx + VX - 1) / VX), (unsigned)((ny + VY - 1) / VY), (unsigned)((nz - 2 + MZ - 1) / MZ));
                if (vec4 && nx % VX == 0) hipLaunchKernelGGL(k_vel_march<true>, grid, dim3(256), 0, q, S, V, f, x, y, MZ);
                else hipLaunchKernelGGL(k_vel_march<false>, grid, dim3(256), 0, q, S, V, f, x, y, MZ);
            } else if (vec4) {
                const dim3 grid((unsigned)((nx / 4 + 63) / 64), (unsigned)((ny - 2 + 3) / 4), (unsigned)(h.dim == 3 ? nz - 2 : 1));
                if (h.dim == 3) hipLaunchKernelGGL(k_vel_interior4<3>, grid, dim3(64, 4), 0, q, S, V, f, x, y);
                else hipLaunchKernelGGL(k_vel_interior4<2>, grid, dim3(64, 4), 0, q, S, V, f, x, y);
            } else {
                const dim3 grid((unsigned)((nx - 2 + 255) / 256), (unsigned)(ny - 2), (unsigned)(h.dim == 3 ? nz - 2 : 1));
                if (h.dim == 3) hipLaunchKernelGGL(k_vel_interior<3>, grid, dim3(256), 0, q, S, V, f, x, y);
                else hipLaunchKernelGGL(k_vel_interior<2>, grid, dim3(256), 0, q, S, V, f, x, y);
            }
        }
        const int64_t shell = inner ? 2 * (ny * nz + (nx - 2) * nz + (h.dim == 3 ? (nx - 2) * (ny - 2) : 0)) : nx * ny * nz;
        hipLaunchKernelGGL(k_vel_shell, dim3((unsigned)std::min<int64_t>(4096, (shell + 255) / 256)), dim3(256), 0, q, S, V, f,
                           inner ? 0 : 1, x, y);
    }
    PIB_HIP(hipGetLastError());
    s->counters[0]++;
    return 0;
}

}  // namespace pib
