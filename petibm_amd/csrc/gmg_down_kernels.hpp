// gmg_down_kernels.hpp -- geometric multigrid (gmg.hip), device side: the way down: restriction by rows / z-march, residual + restriction in one march, the whole way down of a V(2,.) level in one march.
// Included by gmg.hip only (one translation unit: the launches there instantiate these templates).
#pragma once
#include "pib_internal.hpp"

namespace pib {
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
}  // namespace pib
