// gmg_up_kernels.hpp -- geometric multigrid (gmg.hip), device side: the way up: prolongation by rows, prolongation + one / both post-smoothing steps in one march.
// Included by gmg.hip only (one translation unit: the launches there instantiate these templates).
#pragma once
#include "pib_internal.hpp"

namespace pib {
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
}  // namespace pib
