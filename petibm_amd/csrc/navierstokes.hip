// navierstokes.hip -- SURVEY.md 8f-1: device-resident right-hand-side assembly and projection around the two
// linear solves, i.e. one NavierStokesSolver::advance (applications/navierstokes/navierstokes.cpp:240-266) with
// every vector in HBM and no matrix but the two system matrices:
//
//   assembleRHSVelocity (:432-521)  rhs1 = -G p + u/dt + sum c_i conv_i + sum d_i diff_i + c nu Lbc
//   solveVelocity       (:524-537)  A u* = rhs1                      -> pib_solve (BiCGStab / CG + Jacobi)
//   assembleRHSPoisson  (:540-563)  rhs2 = D u* + Dbc ; rhs2[0] = 0 when the pressure is pinned
//   solvePoisson        (:566-580)  DBNG dP = rhs2                   -> pib_solve (multigrid-PCG)
//   applyDivergenceFreeVelocity (:583-598)  u = u* - BNG dP ; updatePressure (:601-615)  p += dP
//
// G, D, L, BNG are applied matrix-free from the 1-D mesh arrays, entry for entry and in the same summation
// order as the assembled AIJ matrices of the reference (creategradient.cpp:64-128, createdivergence.cpp:135-223,
// createlaplacian.cpp:108-263); N(u) is src/operators/createconvection.cpp:40-195 with ghost values computed on
// the stored ghost values; the BC correction shells (createlaplacian.cpp:45-78, createdivergence.cpp:45-78) are
// sum coeff*a1 over the ghost points of a row.  Ghost points carry the reference's per-point state (a0, a1, value:
// type.h GhostPointInfo) in face arrays: Dirichlet / Neumann equations are constant
// (singleboundarydirichlet.cpp:22-46, singleboundaryneumann.cpp:22-33), the convective outlet re-derives a1
// every step (singleboundaryconvective.cpp:12-38).
// AB2 convection {1.5,-0.5} + Crank-Nicolson diffusion (timeintegration.h:107-166), BN order 1.  Single GPU.
// The oracle (oracle/navierstokes.py) performs the same VecScale/VecAXPY sequence; the explicit parts agree
// bit for bit, the solves to solver tolerance.
#include <cstring>

#include "ns_internal.hpp"

namespace pib {

// velocity value at (i,j,k) of field f; an index one step outside is the stored ghost value at a wall and the point
// at the other end on a periodic direction (DMGlobalToLocal of the BOX-stencil DMDA, cartesianmesh.cpp:507-517).
// A corner that is a wall ghost in one direction AND a periodic wrap in another is written by neither the scatter nor
// copyValues2LocalVecs in the reference: its local vectors keep their initial zero there, and so does this.
// (i, j, k) of point q of an n0 x n1 x . box.  The index arithmetic in 64 bits -- two divisions and two remainders per point -- was
// most of what the streaming kernels of a step executed (k_ns_project: 0.41 ms per 256^3 step for 1.2 GB); every field of a rank
// fits 32 bits (like the CSR's columns), where a division is a fifth of the instructions.
__device__ __forceinline__ void split3(int64_t q, int64_t n0, int64_t n1, int64_t &i, int64_t &j, int64_t &k)
{
    if ((uint64_t)q < ((uint64_t)1 << 32) && n0 > 0 && n1 > 0) {
        const uint32_t u = (uint32_t)q, r = u / (uint32_t)n0, kk = r / (uint32_t)n1;
        i = u - r * (uint32_t)n0;
        j = r - kk * (uint32_t)n1;
        k = kk;
    } else {
        i = q % n0;
        j = (q / n0) % n1;
        k = q / (n0 * n1);
    }
}
__device__ __forceinline__ double vel(const NsDev &D, const double *__restrict__ U, int f, int64_t i, int64_t j, int64_t k)
{
    const NsField &F = D.f[f];
    int loc = -1, nghost = 0;  // a corner between two wall ghosts is written by nobody either (the vorticity utility reads some)
    bool wrapped = false;
    if (i < 0) { if (D.per & 1) { i = F.n[0] - 1; wrapped = true; } else { loc = 0; i = 0; ++nghost; } }
    else if (i >= F.n[0]) { if (D.per & 1) { i = 0; wrapped = true; } else { loc = 1; i = F.n[0] - 1; ++nghost; } }
    if (j < 0) { if (D.per & 2) { j = F.n[1] - 1; wrapped = true; } else { loc = 2; j = 0; ++nghost; } }
    else if (j >= F.n[1]) { if (D.per & 2) { j = 0; wrapped = true; } else { loc = 3; j = F.n[1] - 1; ++nghost; } }
    if (D.dim == 3) {
        if (k < 0) { if (D.per & 4) { k = F.n[2] - 1; wrapped = true; } else { loc = 4; k = 0; ++nghost; } }
        else if (k >= F.n[2]) { if (D.per & 4) { k = 0; wrapped = true; } else { loc = 5; k = F.n[2] - 1; ++nghost; } }
    }
    if (loc < 0) return U[fidx(F, i, j, k)];
    if (D.seam_axis >= 0 && (loc >> 1) != D.seam_axis) {
        // on a rank next to the periodic seam of the slab axis the same corner (a wall ghost seen from a plane across the
        // seam) has a ghost value here -- the cut is a wall to this engine -- but must read as on one rank
        const int64_t q = D.seam_axis == 0 ? i : (D.seam_axis == 1 ? j : k);
        if (q < D.seam_lo || q >= D.seam_hi) return 0.0;
    }
    return (wrapped || nghost > 1) ? 0.0 : D.gv[face_index(F, loc, i, j, k)];
}

// -N(u) of createconvection.cpp at one velocity point (the caller scales by -1)
__device__ __forceinline__ double convection_at(const NsDev &D, const double *__restrict__ U, int f, int64_t i, int64_t j,
                                                int64_t k)
{
    const NsField &F = D.f[f];
    const bool three = D.dim == 3;
    const double self = vel(D, U, f, i, j, k);
    const double W = (self + vel(D, U, f, i - 1, j, k)) / 2.0, E = (self + vel(D, U, f, i + 1, j, k)) / 2.0;
    const double S = (self + vel(D, U, f, i, j - 1, k)) / 2.0, N = (self + vel(D, U, f, i, j + 1, k)) / 2.0;
    double B = 0.0, Fw = 0.0;
    if (three) {
        B = (self + vel(D, U, f, i, j, k - 1)) / 2.0;
        Fw = (self + vel(D, U, f, i, j, k + 1)) / 2.0;
    }
    const double dLx = F.dl[0][i + 1], dLy = F.dl[1][j + 1];
    double r;
    if (f == 0) {
        const double vS = (vel(D, U, 1, i, j - 1, k) + vel(D, U, 1, i + 1, j - 1, k)) / 2.0;
        const double vN = (vel(D, U, 1, i, j, k) + vel(D, U, 1, i + 1, j, k)) / 2.0;
        r = (E * E - W * W) / dLx + (vN * N - vS * S) / dLy;
        if (three) {
            const double wB = (vel(D, U, 2, i, j, k - 1) + vel(D, U, 2, i + 1, j, k - 1)) / 2.0;
            const double wF = (vel(D, U, 2, i, j, k) + vel(D, U, 2, i + 1, j, k)) / 2.0;
            r = r + (wF * Fw - wB * B) / F.dl[2][k + 1];
        }
    } else if (f == 1) {
        const double uW = (vel(D, U, 0, i - 1, j, k) + vel(D, U, 0, i - 1, j + 1, k)) / 2.0;
        const double uE = (vel(D, U, 0, i, j, k) + vel(D, U, 0, i, j + 1, k)) / 2.0;
        r = (uE * E - uW * W) / dLx + (N * N - S * S) / dLy;
        if (three) {
            const double wB = (vel(D, U, 2, i, j, k - 1) + vel(D, U, 2, i, j + 1, k - 1)) / 2.0;
            const double wF = (vel(D, U, 2, i, j, k) + vel(D, U, 2, i, j + 1, k)) / 2.0;
            r = r + (wF * Fw - wB * B) / F.dl[2][k + 1];
        }
    } else {
        const double uW = (vel(D, U, 0, i - 1, j, k) + vel(D, U, 0, i - 1, j, k + 1)) / 2.0;
        const double uE = (vel(D, U, 0, i, j, k) + vel(D, U, 0, i, j, k + 1)) / 2.0;
        const double vS = (vel(D, U, 1, i, j - 1, k) + vel(D, U, 1, i, j - 1, k + 1)) / 2.0;
        const double vN = (vel(D, U, 1, i, j, k) + vel(D, U, 1, i, j, k + 1)) / 2.0;
        r = (uE * E - uW * W) / dLx + (vN * N - vS * S) / dLy + (Fw * Fw - B * B) / F.dl[2][k + 1];
    }
    return r;
}

// (L u)_row in the CSR order of createLaplacian (z-, y-, x-, diag, x+, y+, z+) and the BC correction
// sum coeff*a1 with the current (lc) and the updated (lcn) ghost equations
__device__ __forceinline__ void laplacian_at(const NsDev &D, const double *__restrict__ U, int f, int64_t i, int64_t j,
                                             int64_t k, double *lu, double *lc, double *lcn)
{
    const NsField &F = D.f[f];
    const int64_t ijk[3] = {i, j, k};
    double v[6] = {0, 0, 0, 0, 0, 0};
    bool interior[6] = {false, false, false, false, false, false};
    bool anywrap = false;
    double acc = 0.0;
    for (int d = 0; d < D.dim; ++d) {
        const int64_t s = ijk[d];
        const double dLSelf = F.dl[d][s + 1];
        const double dLNeg = F.co[d][s + 1] - F.co[d][s];
        const double dLPos = F.co[d][s + 2] - F.co[d][s + 1];
        v[2 * d] = 1.0 / (dLNeg * dLSelf);
        v[2 * d + 1] = 1.0 / (dLPos * dLSelf);
        const bool wrap = (D.per >> d) & 1;
        interior[2 * d] = s > 0 || wrap;
        interior[2 * d + 1] = s < F.n[d] - 1 || wrap;
        anywrap = anywrap || (wrap && (s == 0 || s == F.n[d] - 1));
        acc = acc + v[2 * d];
        acc = acc + v[2 * d + 1];
    }
    double diag = -acc, corr = 0.0, corrn = 0.0;
    for (int q = 0; q < 2 * D.dim; ++q)
        if (!interior[q]) {
            const double t = v[q] * F.a0[q];
            if (t != 0.0) diag = diag + t;
            const int64_t g = face_index(F, q, i, j, k);
            corr = corr + v[q] * D.a1[g];
            corrn = corrn + v[q] * D.a1n[g];
        }
    const int64_t st[3] = {1, F.n[0], F.n[0] * F.n[1]};
    const int64_t p = fidx(F, i, j, k);
    double s = 0.0;
    if (!anywrap) {
        for (int d = D.dim - 1; d >= 0; --d)
            if (interior[2 * d]) s = s + v[2 * d] * U[p - st[d]];
        s = s + diag * U[p];
        for (int d = 0; d < D.dim; ++d)
            if (interior[2 * d + 1]) s = s + v[2 * d + 1] * U[p + st[d]];
    } else {
        // a neighbour across the periodic seam has the wrapped column: the row is summed by ascending column
        int64_t ec[7];
        double ev[7];
        int ne = 1;
        ec[0] = p;
        ev[0] = diag;
        for (int q = 0; q < 2 * D.dim; ++q) {
            if (!interior[q]) continue;
            const int d = q >> 1;
            int64_t c;
            if (!(q & 1)) c = (ijk[d] == 0) ? p + (F.n[d] - 1) * st[d] : p - st[d];
            else c = (ijk[d] == F.n[d] - 1) ? p - (F.n[d] - 1) * st[d] : p + st[d];
            int t = ne++;
            while (t > 0 && ec[t - 1] > c) {
                ec[t] = ec[t - 1];
                ev[t] = ev[t - 1];
                --t;
            }
            ec[t] = c;
            ev[t] = v[q];
        }
        for (int t = 0; t < ne; ++t) s = s + ev[t] * U[ec[t]];
    }
    *lu = s;
    *lc = corr;
    *lcn = corrn;
}

// rhs1 and the new convective term (navierstokes.cpp:432-521) at one velocity point, general form (ghost values, ghost
// equations, periodic wraps)
__device__ __forceinline__ void rhs_velocity_point(const NsDev &D, double dt, double nu, const NsTime &T,
                                                   const double *__restrict__ U, const double *__restrict__ p,
                                                   const double *__restrict__ conv1, double *__restrict__ conv0,
                                                   double *__restrict__ rhs1, double *__restrict__ diff0,
                                                   const double *__restrict__ diff1, int f, int64_t i, int64_t j, int64_t k)
{
    const NsField &F = D.f[f];
    const int64_t g = fidx(F, i, j, k);
    // G p: row {-1/dL at the cell, +1/dL at the + neighbour}, dL = dL[f][f][idx]
    const int64_t ijk[3] = {i, j, k};
    const double gv = 1.0 / F.dl[f][ijk[f] + 1];
    const int64_t pst[3] = {1, D.pn[0], D.pn[0] * D.pn[1]};
    const int64_t pc = i + D.pn[0] * (j + D.pn[1] * k);
    double r;
    if (ijk[f] < D.pn[f] - 1) {
        r = 0.0 + (-gv) * p[pc];
        r = r + gv * p[pc + pst[f]];
    } else {  // last point of a periodic direction: the + neighbour is cell 0, the smaller column of G's row
        r = 0.0 + gv * p[pc - (D.pn[f] - 1) * pst[f]];
        r = r + (-gv) * p[pc];
    }
    r = -1.0 * r;
    r = r + (1.0 / dt) * U[g];
    if (T.nconv > 0) {
        const double cn = -1.0 * convection_at(D, U, f, i, j, k);
        conv0[g] = cn;
        r = r + T.cc[0] * cn;
        if (T.nconv > 1) r = r + T.cc[1] * conv1[g];
    }
    // explicit diffusion with the ghost equations of the previous step, implicit correction with the updated
    // ones (bc->updateEqs sits between the two, navierstokes.cpp:492-514)
    double lu, lc, lcn;
    laplacian_at(D, U, f, i, j, k, &lu, &lc, &lcn);
    if (T.ndiff > 0) {
        double df = lu + lc;
        df = nu * df;
        diff0[g] = df;
        r = r + T.dc[0] * df;
        if (T.ndiff > 1) r = r + T.dc[1] * diff1[g];
    }
    const double b1 = nu * lcn;
    r = r + T.cimpl * b1;
    rhs1[g] = r;
}

// The outermost layer of a component (any index 0 or n-1: ghost values, ghost equations or periodic wraps in the
// stencil), one point per lane in a dense enumeration -- x faces, then y faces without the x faces, then z faces without
// both; `all` != 0: every point (a component with fewer than three points in some direction has no interior).
__global__ __launch_bounds__(256) void k_ns_rhs_velocity_shell(NsDev D, int f, int all, double dt, double nu, NsTime T,
                                                               const double *__restrict__ U,
                                                               const double *__restrict__ p, const double *__restrict__ conv1,
                                                               double *__restrict__ conv0, double *__restrict__ rhs1,
                                                               double *__restrict__ diff0, const double *__restrict__ diff1)
{
    const NsField &F = D.f[f];
    const int64_t nx = F.n[0], ny = F.n[1], nz = F.n[2];
    const bool three = D.dim == 3;
    const int64_t cx = 2 * ny * nz, cy = 2 * (nx - 2) * nz, cz = three ? 2 * (nx - 2) * (ny - 2) : 0;
    const int64_t total = all ? nx * ny * nz : cx + cy + cz;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        int64_t i, j, k;
        if (all) {
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
        rhs_velocity_point(D, dt, nu, T, U, p, conv1, conv0, rhs1, diff0, diff1, f, i, j, k);
    }
}

// rhs1, -N(u) and nu (L u) at an INTERIOR point of component F from the values its stencils read (no ghost value, no ghost
// equation, no periodic wrap: the boundary terms are exact zeros): the operations of rhs_velocity_point in the same order.  Shared
// by the per-component interior kernel and the three-component march below -- one text, so both give the same bits.
//   a0..a3: the first of the other components' values of N(u) (v for u; u for v and w), c0..c3 the second (w for u and v; v for w)
template <int DIM, int F>
__device__ __forceinline__ void rhs_interior_arith(const NsTime &T, double dt, double nu, double p0, double p1, double self, double uxm, double uxp,
                                                   double uym, double uyp, double uzm, double uzp, double a0, double a1, double a2, double a3,
                                                   double c0, double c1, double c2, double c3, double cold, double dold, double dLx, double dLy,
                                                   double dLz, double gv, double xNeg, double xPos, double yNeg, double yPos, double zNeg,
                                                   double zPos, double &cn, double &df, double &r)
{
    // ---- G p
    r = 0.0 + (-gv) * p0;
    r = r + gv * p1;
    r = -1.0 * r;
    r = r + (1.0 / dt) * self;
    // ---- N(u)  (createconvection.cpp:40-195)
    cn = 0.0;
    if (T.nconv > 0) {
        const double W = (self + uxm) / 2.0, E = (self + uxp) / 2.0;
        const double S = (self + uym) / 2.0, N = (self + uyp) / 2.0;
        double B = 0.0, Fw = 0.0;
        if (DIM == 3) {
            B = (self + uzm) / 2.0;
            Fw = (self + uzp) / 2.0;
        }
        double cv;
        if (F == 0) {
            const double vS = (a0 + a1) / 2.0;
            const double vN = (a2 + a3) / 2.0;
            cv = (E * E - W * W) / dLx + (vN * N - vS * S) / dLy;
            if (DIM == 3) {
                const double wB = (c0 + c1) / 2.0;
                const double wF = (c2 + c3) / 2.0;
                cv = cv + (wF * Fw - wB * B) / dLz;
            }
        } else if (F == 1) {
            const double uW = (a0 + a1) / 2.0;
            const double uE = (a2 + a3) / 2.0;
            cv = (uE * E - uW * W) / dLx + (N * N - S * S) / dLy;
            if (DIM == 3) {
                const double wB = (c0 + c1) / 2.0;
                const double wF = (c2 + c3) / 2.0;
                cv = cv + (wF * Fw - wB * B) / dLz;
            }
        } else {
            const double uW = (a0 + a1) / 2.0;
            const double uE = (a2 + a3) / 2.0;
            const double vS = (c0 + c1) / 2.0;
            const double vN = (c2 + c3) / 2.0;
            cv = (uE * E - uW * W) / dLx + (vN * N - vS * S) / dLy + (Fw * Fw - B * B) / dLz;
        }
        cn = -1.0 * cv;
        r = r + T.cc[0] * cn;
        if (T.nconv > 1) r = r + T.cc[1] * cold;
    }
    // ---- L u in the row's column order z-, y-, x-, diag, x+, y+, z+ ; no ghost point: the corrections are zero
    double acc = 0.0;
    acc = acc + xNeg;
    acc = acc + xPos;
    acc = acc + yNeg;
    acc = acc + yPos;
    if (DIM == 3) {
        acc = acc + zNeg;
        acc = acc + zPos;
    }
    const double diag = -acc;
    double lu = 0.0;
    if (DIM == 3) lu = lu + zNeg * uzm;
    lu = lu + yNeg * uym;
    lu = lu + xNeg * uxm;
    lu = lu + diag * self;
    lu = lu + xPos * uxp;
    lu = lu + yPos * uyp;
    if (DIM == 3) lu = lu + zPos * uzp;
    const double lc = 0.0, lcn = 0.0;
    df = 0.0;
    if (T.ndiff > 0) {
        df = lu + lc;
        df = nu * df;
        r = r + T.dc[0] * df;
        if (T.ndiff > 1) r = r + T.dc[1] * dold;
    }
    const double b1 = nu * lcn;
    r = r + T.cimpl * b1;
}

// The interior of component F (every index in [1, n-2]: no ghost value, no ghost equation, no periodic wrap in the
// stencil): grid (x chunks, j-1, k-1), so j and k are workgroup-uniform (their mesh coefficients come through the scalar
// path, no per-point division) and a lane walks i; branch-free, direct loads, the operations of rhs_velocity_point in
// the same order (bit-identical results; the boundary terms are exact zeros here).  The generic one-lane-per-point form
// spent 4.8 ms per step on the 256^3 Taylor-Green case (5.0e7 points: 64-bit div/mod, twenty branchy vel() calls and
// scratch-resident index arrays per point); mixing the two forms in one kernel left half of the waves paying for both.
template <int DIM, int F>
__global__ __launch_bounds__(256) void k_ns_rhs_velocity(NsDev D, double dt, double nu, NsTime T,
                                                         const double *__restrict__ U,
                                                         const double *__restrict__ p, const double *__restrict__ conv1,
                                                         double *__restrict__ conv0, double *__restrict__ rhs1,
                                                         double *__restrict__ diff0, const double *__restrict__ diff1, int gx,
                                                         int band, int KZ)
{
    const NsField &Fd = D.f[F];
    const int nx = (int)Fd.n[0];
    // band > 0 (3-D, 1-D grid of gx * 8 * band * (nz - 2) workgroups): workgroups L, L + 8, ... share an XCD and its L2, so each
    // of the eight classes takes a contiguous band of rows and walks it plane after plane -- a row's y neighbours and the three
    // planes of every component the stencils reach (band * nx * 8 B each) then stay in THAT L2 from one plane to the next.
    // Dealt out row by row, every XCD touched every row: nine planes of 512 KB at 256^3 per 4 MB L2, refetched for every plane.
    int bxi = blockIdx.x, j = blockIdx.y + 1, k = (DIM == 3) ? blockIdx.z + 1 : 0;
    if (band > 0) {
        const int L = blockIdx.x, c = L & 7;
        int m = L >> 3;
        bxi = m % gx;
        m /= gx;
        j = c * band + m % band + 1;
        k = (m / band) * KZ + 1;
        if (j > (int)Fd.n[1] - 2) return;
    }
    const int kend = (DIM == 3 && band > 0) ? min(k + KZ, (int)Fd.n[2] - 1) : k + 1;
    for (; k < kend; ++k) {
    // bases of every component at (0, j, k) and their strides
    int64_t b[DIM], sy[DIM], sz[DIM];
#pragma unroll
    for (int ff = 0; ff < DIM; ++ff) {
        sy[ff] = D.f[ff].n[0];
        sz[ff] = sy[ff] * D.f[ff].n[1];
        b[ff] = D.f[ff].off + sy[ff] * j + sz[ff] * k;
    }
    const int64_t pbase = D.pn[0] * (j + D.pn[1] * (int64_t)k);
    const int64_t pstF = (F == 0) ? 1 : (F == 1 ? D.pn[0] : D.pn[0] * D.pn[1]);
    const double dLy = Fd.dl[1][j + 1];
    const double dLz = (DIM == 3) ? Fd.dl[2][k + 1] : 1.0;
    // Laplacian coefficients of the y and z directions (createlaplacian.cpp:134-148), from the tables
    const double yNeg = Fd.lneg[1][j], yPos = Fd.lpos[1][j];
    const double zNeg = (DIM == 3) ? Fd.lneg[2][k] : 0.0, zPos = (DIM == 3) ? Fd.lpos[2][k] : 0.0;
    const double gvyz = (F == 1) ? Fd.ginv[j] : ((F == 2) ? Fd.ginv[k] : 0.0);
#define PIB_V(ff, di, dj, dk) U[b[ff] + i + (di) + (dj) * sy[ff] + (dk) * sz[ff]]
    for (int i = 1 + bxi * 256 + threadIdx.x; i < nx - 1; i += gx * 256) {
        const int64_t g = b[F] + i;
        // Every value the point needs is requested HERE, before the first use, and its three results are stored at the end: written
        // in the order of the formulae (p, then the convective stencil, a store, conv1, the Laplacian's stencil again, a store,
        // diff1, a store) the compiler kept five memory round trips one after the other per wave -- the stores between them forbid
        // moving the loads up, and each wait behind a store waited for the store too -- and the kernel ran at the latency of those
        // (3 TB/s whatever the L2 hit rate: PMC had 2.9 x the algorithmic bytes fetched without the XCD bands, 1.0 x with them, at
        // the same 370 us).  Same operations on the same values in the same order: same bits.
        const int64_t pc = pbase + i;
        const double p0 = p[pc], p1 = p[pc + pstF];
        const double self = U[g];
        const double uxm = PIB_V(F, -1, 0, 0), uxp = PIB_V(F, 1, 0, 0), uym = PIB_V(F, 0, -1, 0), uyp = PIB_V(F, 0, 1, 0);
        const double uzm = (DIM == 3) ? PIB_V(F, 0, 0, -1) : 0.0, uzp = (DIM == 3) ? PIB_V(F, 0, 0, 1) : 0.0;
        // the other components' values of N(u): a0..a3 the first of them (v for u, u for v and w), c0..c3 the second (w for u and v,
        // v for w), in the order the formulae below read them
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
        if (T.nconv > 0) {
            if (F == 0) {
                a0 = PIB_V(1, 0, -1, 0), a1 = PIB_V(1, 1, -1, 0), a2 = PIB_V(1, 0, 0, 0), a3 = PIB_V(1, 1, 0, 0);
                if (DIM == 3) c0 = PIB_V(DIM - 1, 0, 0, -1), c1 = PIB_V(DIM - 1, 1, 0, -1), c2 = PIB_V(DIM - 1, 0, 0, 0), c3 = PIB_V(DIM - 1, 1, 0, 0);
            } else if (F == 1) {
                a0 = PIB_V(0, -1, 0, 0), a1 = PIB_V(0, -1, 1, 0), a2 = PIB_V(0, 0, 0, 0), a3 = PIB_V(0, 0, 1, 0);
                if (DIM == 3) c0 = PIB_V(DIM - 1, 0, 0, -1), c1 = PIB_V(DIM - 1, 0, 1, -1), c2 = PIB_V(DIM - 1, 0, 0, 0), c3 = PIB_V(DIM - 1, 0, 1, 0);
            } else {
                a0 = PIB_V(0, -1, 0, 0), a1 = PIB_V(0, -1, 0, 1), a2 = PIB_V(0, 0, 0, 0), a3 = PIB_V(0, 0, 0, 1);
                c0 = PIB_V(1, 0, -1, 0), c1 = PIB_V(1, 0, -1, 1), c2 = PIB_V(1, 0, 0, 0), c3 = PIB_V(1, 0, 0, 1);
            }
        }
        const double cold = (T.nconv > 1) ? conv1[g] : 0.0;
        const double dold = (T.ndiff > 1) ? diff1[g] : 0.0;
        const double dLx = Fd.dl[0][i + 1];
        const double gv = (F == 0) ? Fd.ginv[i] : gvyz;
        const double xNeg = Fd.lneg[0][i], xPos = Fd.lpos[0][i];
        double cn, df, r;
        rhs_interior_arith<DIM, F>(T, dt, nu, p0, p1, self, uxm, uxp, uym, uyp, uzm, uzp, a0, a1, a2, a3, c0, c1, c2, c3, cold, dold, dLx, dLy, dLz,
                                   gv, xNeg, xPos, yNeg, yPos, zNeg, zPos, cn, df, r);
        if (T.nconv > 0) conv0[g] = cn;
        if (T.ndiff > 0) diff0[g] = df;
        rhs1[g] = r;
    }
    }
#undef PIB_V
}

// The interiors of ALL THREE components in one z-march (3-D; round 6).  The per-component kernels above read every value of U
// three times from HBM / L2 (once as the component's own stencil, twice as the other components' cross terms of N(u)) and p three
// times: 64 B per velocity point where 42.7 suffice.  Here a workgroup owns a RHS_TX x RHS_TY tile of cells and marches through
// RHS planes; planes k-1, k, k+1 of u, v and w (tile + one ring of halo points, each component in its own index space: point
// (i,j,k) of a component sits on the + face of cell (i,j,k)) and planes k, k+1 of p live in LDS, plane k+2 is on its way in
// registers while plane k is computed, and the three points of a cell are computed by the cell's thread from LDS reads only.
// Four LDS slots per component (three for p): the plane that arrives replaces the one nobody reads any more, so ONE barrier per
// plane.  HBM traffic per cell: U 24 B (+ the ring, from L2: tiles next to each other in y share an XCD), p 8, the old N(u) 24,
// three results x 3 components 72 (48 when the diffusive term is not kept) = 128 (104) instead of 192.
// Same values through rhs_interior_arith: the bits of k_ns_rhs_velocity.  The boundary layer stays with k_ns_rhs_velocity_shell.
#ifndef PIB_RHS_WAVES
#define PIB_RHS_WAVES 4  // waves per SIMD the march is compiled for (128 registers: two workgroups per CU)
#endif
constexpr int RHS_TX = 64, RHS_TY = 8, RHS_PX = RHS_TX + 2, RHS_TILE = 664;  // (66 x 10 = 660 points, padded)
constexpr int RHS_HALO = 2 * RHS_PX + 2 * RHS_TY;                            // 148 ring points
static_assert(RHS_TX == 64, "a wave is one row of the tile: j is wave-uniform (readfirstlane)");
static_assert(RHS_HALO <= RHS_TX * RHS_TY, "one ring point per thread");
template <bool STORE_DIFF>
__global__ __launch_bounds__(RHS_TX * RHS_TY, PIB_RHS_WAVES) void k_ns_rhs_march(NsDev D, double dt, double nu, NsTime T, const double *__restrict__ U,
                                                               const double *__restrict__ p, const double *__restrict__ conv1,
                                                               double *__restrict__ conv0, double *__restrict__ rhs1,
                                                               double *__restrict__ diff0, const double *__restrict__ diff1, int ntx,
                                                               int nty, int band, int KZ)
{
    __shared__ double su[3][4][RHS_TILE];
    __shared__ double sp[3][RHS_TILE];
    // workgroups L, L + 8, ... share an XCD: each of the eight classes takes a band of tile rows (tiles that share ring points
    // run side by side on one L2), x tiles fastest, then the band's rows, then the z chunks
    const int L = blockIdx.x, xcd = L & 7;
    int m = L >> 3;
    const int txi = m % ntx;
    m /= ntx;
    const int tyi = xcd * band + m % band;
    const int zc = m / band;
    if (tyi >= nty) return;
    const int t = threadIdx.x, tx = t & (RHS_TX - 1), ty = t / RHS_TX;
    const int i0 = txi * RHS_TX, j0 = tyi * RHS_TY;
    // (a wave is one row of the tile: j is wave-uniform, and so is everything looked up by it -- scalar registers)
    const int i = i0 + tx, j = __builtin_amdgcn_readfirstlane(j0 + ty);
    const int kmax = (int)D.pn[2] - 2;  // last cell plane on which some component has an interior point
    const int ka = 1 + zc * KZ, kb = min(ka + KZ, kmax + 1);
    if (ka >= kb) return;
    // this thread's second point of a plane: ring point h = t (t < RHS_HALO)
    int hx = 0, hy = 0;
    if (t < RHS_PX) hx = t, hy = 0;
    else if (t < 2 * RHS_PX) hx = t - RHS_PX, hy = RHS_TY + 1;
    else if (t < 2 * RHS_PX + RHS_TY) hx = 0, hy = t - 2 * RHS_PX + 1;
    else hx = RHS_PX - 1, hy = t - 2 * RHS_PX - RHS_TY + 1;
    const bool has_h = t < RHS_HALO;
    const int lm = (ty + 1) * RHS_PX + tx + 1, lh = hy * RHS_PX + hx;  // LDS positions of the two
    const int hi_ = i0 - 1 + hx, hj_ = j0 - 1 + hy;                    // ... and the ring point's indices
    // global offsets (plane 0) and validity of the two points in every component's index space and in the cells'
    int64_t gm[4], gh[4], pl[4];
    bool vm[4], vh[4];
    int nz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t n0 = c < 3 ? D.f[c].n[0] : D.pn[0], n1 = c < 3 ? D.f[c].n[1] : D.pn[1];
        nz[c] = (int)(c < 3 ? D.f[c].n[2] : D.pn[2]);
        const int64_t off = c < 3 ? D.f[c].off : 0;
        pl[c] = n0 * n1;
        vm[c] = i < n0 && j < n1;
        vh[c] = has_h && hi_ >= 0 && hi_ < n0 && hj_ >= 0 && hj_ < n1;
        gm[c] = off + i + n0 * (int64_t)j;
        gh[c] = off + hi_ + n0 * (int64_t)hj_;
    }
    // interior of every component in x and y, and the k-independent coefficients of the thread's three points
    bool in_xy[3];
    double dLx[3], xNeg[3], xPos[3], dLy[3], yNeg[3], yPos[3], gvxy[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const NsField &Fd = D.f[f];
        const bool in_x = i >= 1 && i <= (int)Fd.n[0] - 2, in_y = j >= 1 && j <= (int)Fd.n[1] - 2;
        in_xy[f] = in_x && in_y;
        dLx[f] = xNeg[f] = xPos[f] = dLy[f] = yNeg[f] = yPos[f] = gvxy[f] = 0.0;
        if (in_y) dLy[f] = Fd.dl[1][j + 1], yNeg[f] = Fd.lneg[1][j], yPos[f] = Fd.lpos[1][j];  // (uniform)
        if (in_x) dLx[f] = Fd.dl[0][i + 1], xNeg[f] = Fd.lneg[0][i], xPos[f] = Fd.lpos[0][i];
        if (f == 0 && in_x) gvxy[f] = Fd.ginv[i];
        if (f == 1 && in_y) gvxy[f] = Fd.ginv[j];
    }
    double rm[4], rh[4];  // plane on its way: the thread's tile point and ring point of u, v, w, p
    auto fetch = [&](int kk) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double *src = c < 3 ? U : p;
            const bool zin = kk >= 0 && kk < nz[c];
            rm[c] = (zin && vm[c]) ? src[gm[c] + pl[c] * kk] : 0.0;
            rh[c] = (zin && vh[c]) ? src[gh[c] + pl[c] * kk] : 0.0;
        }
    };
    auto put = [&](int kk) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            su[c][kk & 3][lm] = rm[c];
            if (has_h) su[c][kk & 3][lh] = rh[c];
        }
        sp[kk % 3][lm] = rm[3];
        if (has_h) sp[kk % 3][lh] = rh[3];
    };
    for (int kk = ka - 1; kk <= ka + 1; ++kk) {
        fetch(kk);
        put(kk);
    }
    __syncthreads();
    for (int k = ka; k < kb; ++k) {
        fetch(k + 2);
        const int sm = (k - 1) & 3, s0 = k & 3, s1 = (k + 1) & 3;
        const double *pk = sp[k % 3], *pk1 = sp[(k + 1) % 3];
#define PIB_L(c, sl, dx, dy) su[c][sl][lm + (dx) + (dy) * RHS_PX]
        // ---- u
        if (in_xy[0] && k <= nz[0] - 2) {
            const NsField &Fd = D.f[0];
            const int64_t g = gm[0] + pl[0] * k;
            const double cold = (T.nconv > 1) ? conv1[g] : 0.0, dold = (T.ndiff > 1) ? diff1[g] : 0.0;
            double cn, df, r;
            rhs_interior_arith<3, 0>(T, dt, nu, pk[lm], pk[lm + 1], PIB_L(0, s0, 0, 0), PIB_L(0, s0, -1, 0), PIB_L(0, s0, 1, 0), PIB_L(0, s0, 0, -1),
                                     PIB_L(0, s0, 0, 1), PIB_L(0, sm, 0, 0), PIB_L(0, s1, 0, 0), PIB_L(1, s0, 0, -1), PIB_L(1, s0, 1, -1),
                                     PIB_L(1, s0, 0, 0), PIB_L(1, s0, 1, 0), PIB_L(2, sm, 0, 0), PIB_L(2, sm, 1, 0), PIB_L(2, s0, 0, 0),
                                     PIB_L(2, s0, 1, 0), cold, dold, dLx[0], dLy[0], Fd.dl[2][k + 1], gvxy[0], xNeg[0], xPos[0], yNeg[0], yPos[0],
                                     Fd.lneg[2][k], Fd.lpos[2][k], cn, df, r);
            if (T.nconv > 0) conv0[g] = cn;
            if (STORE_DIFF && T.ndiff > 0) diff0[g] = df;
            rhs1[g] = r;
        }
        // ---- v
        if (in_xy[1] && k <= nz[1] - 2) {
            const NsField &Fd = D.f[1];
            const int64_t g = gm[1] + pl[1] * k;
            const double cold = (T.nconv > 1) ? conv1[g] : 0.0, dold = (T.ndiff > 1) ? diff1[g] : 0.0;
            double cn, df, r;
            rhs_interior_arith<3, 1>(T, dt, nu, pk[lm], pk[lm + RHS_PX], PIB_L(1, s0, 0, 0), PIB_L(1, s0, -1, 0), PIB_L(1, s0, 1, 0), PIB_L(1, s0, 0, -1),
                                     PIB_L(1, s0, 0, 1), PIB_L(1, sm, 0, 0), PIB_L(1, s1, 0, 0), PIB_L(0, s0, -1, 0), PIB_L(0, s0, -1, 1),
                                     PIB_L(0, s0, 0, 0), PIB_L(0, s0, 0, 1), PIB_L(2, sm, 0, 0), PIB_L(2, sm, 0, 1), PIB_L(2, s0, 0, 0),
                                     PIB_L(2, s0, 0, 1), cold, dold, dLx[1], dLy[1], Fd.dl[2][k + 1], gvxy[1], xNeg[1], xPos[1], yNeg[1], yPos[1],
                                     Fd.lneg[2][k], Fd.lpos[2][k], cn, df, r);
            if (T.nconv > 0) conv0[g] = cn;
            if (STORE_DIFF && T.ndiff > 0) diff0[g] = df;
            rhs1[g] = r;
        }
        // ---- w
        if (in_xy[2] && k <= nz[2] - 2) {
            const NsField &Fd = D.f[2];
            const int64_t g = gm[2] + pl[2] * k;
            const double cold = (T.nconv > 1) ? conv1[g] : 0.0, dold = (T.ndiff > 1) ? diff1[g] : 0.0;
            double cn, df, r;
            rhs_interior_arith<3, 2>(T, dt, nu, pk[lm], pk1[lm], PIB_L(2, s0, 0, 0), PIB_L(2, s0, -1, 0), PIB_L(2, s0, 1, 0), PIB_L(2, s0, 0, -1),
                                     PIB_L(2, s0, 0, 1), PIB_L(2, sm, 0, 0), PIB_L(2, s1, 0, 0), PIB_L(0, s0, -1, 0), PIB_L(0, s1, -1, 0),
                                     PIB_L(0, s0, 0, 0), PIB_L(0, s1, 0, 0), PIB_L(1, s0, 0, -1), PIB_L(1, s1, 0, -1), PIB_L(1, s0, 0, 0),
                                     PIB_L(1, s1, 0, 0), cold, dold, dLx[2], dLy[2], Fd.dl[2][k + 1], Fd.ginv[k], xNeg[2], xPos[2], yNeg[2], yPos[2],
                                     Fd.lneg[2][k], Fd.lpos[2][k], cn, df, r);
            if (T.nconv > 0) conv0[g] = cn;
            if (STORE_DIFF && T.ndiff > 0) diff0[g] = df;
            rhs1[g] = r;
        }
#undef PIB_L
        put(k + 2);  // (slot (k + 2) & 3 held plane k - 2, last read before the previous barrier)
        __syncthreads();
    }
}

// rhs2 = D u + Dbc (navierstokes.cpp:540-563) at one pressure cell; D row in packed-column order u(i-1), u(i), v(j-1), v(j), w(k-1), w(k)
__device__ __forceinline__ double rhs_poisson_cell(const NsDev &D, const double *__restrict__ U, int64_t i, int64_t j, int64_t k)
{
    {
        const int64_t ijk[3] = {i, j, k};
        const double wx = D.pw[0][i], wy = D.pw[1][j], wz = (D.dim == 3) ? D.pw[2][k] : 1.0;
        const double area[3] = {wy * wz, wx * wz, wx * wy};
        double s = 0.0, corr = 0.0;
        for (int f = 0; f < D.dim; ++f) {
            const NsField &F = D.f[f];
            const int64_t st = (f == 0) ? 1 : (f == 1 ? F.n[0] : F.n[0] * F.n[1]);
            const int64_t s_ = ijk[f];
            // velocity point with the cell's own index = the + face; index s-1 = the - face
            int64_t fi[3] = {i, j, k};
            if ((D.per >> f) & 1) {
                // periodic: both faces are velocity points, + face = index s, - face = index s-1 (point n-1 for cell 0,
                // which then sorts after the + face in D's row)
                const int64_t base = fidx(F, i, j, k);
                const int64_t im = (s_ > 0) ? base - st : base + (F.n[f] - 1) * st;
                if (s_ > 0) {
                    s = s + (-area[f]) * U[im];
                    s = s + area[f] * U[base];
                } else {
                    s = s + area[f] * U[base];
                    s = s + (-area[f]) * U[im];
                }
                continue;
            }
            const bool has_m = s_ > 0, has_p = s_ < F.n[f];
            fi[f] = has_p ? s_ : s_ - 1;
            const int64_t base = fidx(F, fi[0], fi[1], fi[2]);
            double vm = -area[f], vp = area[f];
            if (!has_m) {  // ghost - face folds onto the + face (its target): D[row,target] += coeff*a0
                const double t = (-area[f]) * F.a0[2 * f];
                if (t != 0.0) vp = vp + t;
                corr = corr + (-area[f]) * D.a1[face_index(F, 2 * f, fi[0], fi[1], fi[2])];
            }
            if (!has_p) {
                const double t = area[f] * F.a0[2 * f + 1];
                if (t != 0.0) vm = vm + t;
            }
            if (has_m) s = s + vm * U[has_p ? base - st : base];
            if (has_p) s = s + vp * U[base];
            if (!has_p) corr = corr + area[f] * D.a1[face_index(F, 2 * f + 1, fi[0], fi[1], fi[2])];
        }
        return s + corr;
    }
}
__global__ __launch_bounds__(256) void k_ns_rhs_poisson(NsDev D, int64_t pin_cell, const double *__restrict__ U,
                                                        double *__restrict__ rhs2)
{
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < D.pN; c += (int64_t)gridDim.x * 256) {
        int64_t i, j, k;
        split3(c, D.pn[0], D.pn[1], i, j, k);
        double r = rhs_poisson_cell(D, U, i, j, k);
        if (c == pin_cell) r = 0.0;  // the pinned pressure's row (global cell 0; -1: none on this rank)
        rhs2[c] = r;
    }
}
// ... walked by grid line (round 5, as k_ns_project_rows): j, k workgroup-uniform, no index division, the lanes' loads coalesced
__global__ __launch_bounds__(256) void k_ns_rhs_poisson_rows(NsDev D, int64_t pin_cell, const double *__restrict__ U,
                                                             double *__restrict__ rhs2)
{
    const int j = blockIdx.y, k = blockIdx.z;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (int)D.pn[0]) return;
    const int64_t c = i + D.pn[0] * (j + D.pn[1] * (int64_t)k);
    double r = rhs_poisson_cell(D, U, i, j, k);
    if (c == pin_cell) r = 0.0;
    rhs2[c] = r;
}

// Ghost-point state, one ghost point per lane.  MODE 0: setGhostICs (navierstokes.cpp:142), 1: updateEqs into
// a1n (navierstokes.cpp:508), 2: updateGhostValues (navierstokes.cpp:263).
template <int MODE>
__global__ __launch_bounds__(256) void k_ns_ghosts(NsDev D, double dt, const double *__restrict__ U)
{
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < D.nghost; g += (int64_t)gridDim.x * 256) {
        int f = 0, loc = 0;
        for (int ff = 0; ff < D.dim; ++ff)
            for (int q = 0; q < 2 * D.dim; ++q)
                if (g >= D.f[ff].goff[q] && g < D.f[ff].goff[q] + D.f[ff].gcnt[q]) {
                    f = ff;
                    loc = q;
                }
        const NsField &F = D.f[f];
        const int axis = loc >> 1;
        const int64_t r = g - F.goff[loc];
        int64_t ijk[3];
        const int64_t na = (axis == 0) ? F.n[1] : F.n[0];
        const int64_t a = r % na, b = r / na;
        if (axis == 0) { ijk[1] = a; ijk[2] = b; } else if (axis == 1) { ijk[0] = a; ijk[2] = b; } else { ijk[0] = a; ijk[1] = b; }
        ijk[axis] = (loc & 1) ? F.n[axis] - 1 : 0;
        const double target = U[fidx(F, ijk[0], ijk[1], ijk[2])];
        const bool same = (axis == f);
        const double normal = (loc & 1) ? 1.0 : -1.0;
        const double value = F.bcv[loc];
        if (MODE == 2) {
            D.gv[g] = F.a0[loc] * target + D.a1[g];
            continue;
        }
        if (F.type[loc] == 2) {  // singleboundaryconvective.cpp:12-38
            double pv = D.gv[g];
            double tdt = dt;
            if (MODE == 0) {
                pv = target;  // t = 0: the ghost takes the target's value, dt = 0 in the kernel (:69-85)
                tdt = 0.0;
                D.gv[g] = pv;
            }
            double a1;
            if (same)
                a1 = pv - normal * tdt * value * (pv - target) / F.gdl[loc];
            else
                a1 = pv + target - 2.0 * normal * tdt * value * (pv - target) / F.gdl[loc];
            if (MODE == 0) D.a1[g] = a1;
            D.a1n[g] = a1;
        } else {
            if (MODE == 0) {
                double a1;
                if (F.type[loc] == 0)
                    a1 = same ? value : 2.0 * value;  // singleboundarydirichlet.cpp:35-44
                else
                    a1 = normal * F.gdl[loc] * value;  // singleboundaryneumann.cpp:27-28
                D.a1[g] = a1;
                D.a1n[g] = a1;
                D.gv[g] = F.a0[loc] * target + a1;
            } else {
                D.a1n[g] = D.a1[g];
            }
        }
    }
}

// u = u - BNG dP (BNG = dt*G) ; p = p + dP
__global__ __launch_bounds__(256) void k_ns_project(NsDev D, double dt, const double *__restrict__ dP, double *__restrict__ U,
                                                    double *__restrict__ p)
{
    const int64_t total = D.UN > D.pN ? D.UN : D.pN;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        if (g < D.UN) {
            int f = 0;
            if (D.dim > 1 && g >= D.f[1].off) f = 1;
            if (D.dim > 2 && g >= D.f[2].off) f = 2;
            const NsField &F = D.f[f];
            const int64_t q = g - F.off;
            int64_t i, j, k;
            split3(q, F.n[0], F.n[1], i, j, k);
            const int64_t ijk[3] = {i, j, k};
            const double gv = 1.0 / F.dl[f][ijk[f] + 1];
            const int64_t pst[3] = {1, D.pn[0], D.pn[0] * D.pn[1]};
            const int64_t pc = i + D.pn[0] * (j + D.pn[1] * k);
            double r;
            if (ijk[f] < D.pn[f] - 1) {
                r = 0.0 + (dt * (-gv)) * dP[pc];
                r = r + (dt * gv) * dP[pc + pst[f]];
            } else {  // periodic seam: column order of BNG's row
                r = 0.0 + (dt * gv) * dP[pc - (D.pn[f] - 1) * pst[f]];
                r = r + (dt * (-gv)) * dP[pc];
            }
            U[g] = U[g] + (-1.0) * r;
        }
        if (g < D.pN) p[g] = p[g] + 1.0 * dP[g];
    }
}

// The same projection walked by PRESSURE CELL (round 5): a workgroup takes a piece of a grid line (j, k workgroup-uniform: no
// index division, the y / z gradient entries through the scalar path, the x one from the 1 / dL table instead of a division per
// point), a lane its cell -- p += dP there, and the velocity point of every component that carries the cell's own index (its +
// face): one load of dP[c] serves all four updates, the + neighbours come from the cache.  Every velocity point has such a cell
// (a component has n - 1 points along its own direction, n when periodic).  Same expressions as k_ns_project: same bits.
template <int DIM>
__global__ __launch_bounds__(256) void k_ns_project_rows(NsDev D, double dt, const double *__restrict__ dP, double *__restrict__ U,
                                                         double *__restrict__ p)
{
    const int j = blockIdx.y, k = (DIM == 3) ? blockIdx.z : 0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nx = (int)D.pn[0];
    if (i >= nx) return;
    const int64_t sy = D.pn[0], sz = D.pn[0] * D.pn[1];
    const int64_t c = i + sy * j + sz * (int64_t)k;
    const double d0 = dP[c];
    p[c] = p[c] + 1.0 * d0;
    const int ijk[3] = {i, j, k};
    const int64_t pst[3] = {1, sy, sz};
#pragma unroll
    for (int f = 0; f < DIM; ++f) {
        const NsField &F = D.f[f];
        if (i >= (int)F.n[0] || j >= (int)F.n[1] || k >= (int)F.n[2]) continue;
        const int64_t g = F.off + i + F.n[0] * (j + F.n[1] * (int64_t)k);
        const double gv = F.ginv[ijk[f]];
        double r;
        if (ijk[f] < (int)D.pn[f] - 1) {
            r = 0.0 + (dt * (-gv)) * d0;
            r = r + (dt * gv) * dP[c + pst[f]];
        } else {  // periodic seam: column order of BNG's row
            r = 0.0 + (dt * gv) * dP[c - (D.pn[f] - 1) * pst[f]];
            r = r + (dt * (-gv)) * d0;
        }
        U[g] = U[g] + (-1.0) * r;
    }
}

// Vorticity as the reference's post-processing utility defines it (applications/vorticity/main.cpp:185-372), from the
// velocity and the stored ghost values.  2-D: wz at the vertices.  3-D: wx on (x centres, y vertices, z vertices), wy on
// (x vertices, y centres, z vertices), wz on (x vertices, y vertices, z centres) -- with the utility's index convention,
// which reads the two face-centred components one point lower along the vorticity component's own centred direction
// (w[k-1][j][i-1] for wx at centre i, :300-306; likewise j-1 for wy, :322-332).
//   comp 0 wx, 1 wy, 2 wz;  n[3]: points of the vorticity field;  coord[f][d][s] is F.co[d][s+1]
__global__ __launch_bounds__(256) void k_ns_vorticity(NsDev D, int comp, int64_t n0, int64_t n1, int64_t n2,
                                                      const double *__restrict__ U, double *__restrict__ out)
{
    const int64_t total = n0 * n1 * n2;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        int64_t i, j, k;
        split3(t, n0, n1, i, j, k);
        double w;
        if (D.dim == 2) {
            w = (vel(D, U, 1, i, j - 1, 0) - vel(D, U, 1, i - 1, j - 1, 0)) / (D.f[1].co[0][i + 1] - D.f[1].co[0][i]) -
                (vel(D, U, 0, i - 1, j, 0) - vel(D, U, 0, i - 1, j - 1, 0)) / (D.f[0].co[1][j + 1] - D.f[0].co[1][j]);
        } else if (comp == 0) {
            w = (vel(D, U, 2, i - 1, j, k - 1) - vel(D, U, 2, i - 1, j - 1, k - 1)) / (D.f[2].co[1][j + 1] - D.f[2].co[1][j]) -
                (vel(D, U, 1, i - 1, j - 1, k) - vel(D, U, 1, i - 1, j - 1, k - 1)) / (D.f[1].co[2][k + 1] - D.f[1].co[2][k]);
        } else if (comp == 1) {
            w = (vel(D, U, 0, i - 1, j - 1, k) - vel(D, U, 0, i - 1, j - 1, k - 1)) / (D.f[0].co[2][k + 1] - D.f[0].co[2][k]) -
                (vel(D, U, 2, i, j - 1, k - 1) - vel(D, U, 2, i - 1, j - 1, k - 1)) / (D.f[2].co[0][i + 1] - D.f[2].co[0][i]);
        } else {
            w = (vel(D, U, 1, i, j - 1, k) - vel(D, U, 1, i - 1, j - 1, k)) / (D.f[1].co[0][i + 1] - D.f[1].co[0][i]) -
                (vel(D, U, 0, i - 1, j, k) - vel(D, U, 0, i - 1, j - 1, k)) / (D.f[0].co[1][j + 1] - D.f[0].co[1][j]);
        }
        out[t] = w;
    }
}

// out = BNG phi (BNG = dt G): the row of k_ns_project, stored instead of subtracted
__global__ __launch_bounds__(256) void k_ns_bng(NsDev D, double dt, const double *__restrict__ phi, double *__restrict__ out)
{
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < D.UN; g += (int64_t)gridDim.x * 256) {
        int f = 0;
        if (D.dim > 1 && g >= D.f[1].off) f = 1;
        if (D.dim > 2 && g >= D.f[2].off) f = 2;
        const NsField &F = D.f[f];
        const int64_t q = g - F.off;
        int64_t i, j, k;
        split3(q, F.n[0], F.n[1], i, j, k);
        const int64_t ijk[3] = {i, j, k};
        const double gv = 1.0 / F.dl[f][ijk[f] + 1];
        const int64_t pst[3] = {1, D.pn[0], D.pn[0] * D.pn[1]};
        const int64_t pc = i + D.pn[0] * (j + D.pn[1] * k);
        double r;
        if (ijk[f] < D.pn[f] - 1) {
            r = 0.0 + (dt * (-gv)) * phi[pc];
            r = r + (dt * gv) * phi[pc + pst[f]];
        } else {
            r = 0.0 + (dt * gv) * phi[pc - (D.pn[f] - 1) * pst[f]];
            r = r + (dt * (-gv)) * phi[pc];
        }
        out[g] = r;
    }
}

// w = w - D t over the pressure cells (D without the boundary correction: the rows of k_ns_rhs_poisson); the identity
// row of a pinned pressure is left alone
__global__ __launch_bounds__(256) void k_ns_div_sub(NsDev D, int64_t pin_cell, const double *__restrict__ t, double *__restrict__ w)
{
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < D.pN; c += (int64_t)gridDim.x * 256) {
        if (c == pin_cell) continue;
        int64_t i, j, k;
        split3(c, D.pn[0], D.pn[1], i, j, k);
        const int64_t ijk[3] = {i, j, k};
        const double wx = D.pw[0][i], wy = D.pw[1][j], wz = (D.dim == 3) ? D.pw[2][k] : 1.0;
        const double area[3] = {wy * wz, wx * wz, wx * wy};
        double s = 0.0;
        for (int f = 0; f < D.dim; ++f) {
            const NsField &F = D.f[f];
            const int64_t st = (f == 0) ? 1 : (f == 1 ? F.n[0] : F.n[0] * F.n[1]);
            const int64_t s_ = ijk[f];
            const bool per = (D.per >> f) & 1;
            const bool has_m = s_ > 0 || per, has_p = s_ < F.n[f];
            int64_t fi[3] = {i, j, k};
            fi[f] = has_p ? s_ : s_ - 1;
            const int64_t base = fidx(F, fi[0], fi[1], fi[2]);
            if (per) {
                const int64_t im = (s_ > 0) ? base - st : base + (F.n[f] - 1) * st;
                s = s + (-area[f]) * t[im];
                s = s + area[f] * t[base];
                continue;
            }
            // ghost faces: a0 = 0 for the normal component with Dirichlet / convective boundaries (no column); NEUMANN folds the
            // ghost's coefficient onto the cell's other face (the entries of k_ns_rhs_poisson's row)
            double vm = -area[f], vp = area[f];
            if (!has_m) {
                const double tt = (-area[f]) * F.a0[2 * f];
                if (tt != 0.0) vp = vp + tt;
            }
            if (!has_p) {
                const double tt = area[f] * F.a0[2 * f + 1];
                if (tt != 0.0) vm = vm + tt;
            }
            if (has_m) s = s + vm * t[has_p ? base - st : base];
            if (has_p) s = s + vp * t[base];
        }
        w[c] = w[c] - s;
    }
}

// u = u - BNG dP with the assembled BNG (BN order > 1: bn.hip), row sums in column order ; p = p + dP
__global__ __launch_bounds__(256) void k_ns_project_csr(int64_t UN, int64_t pN, const int32_t *__restrict__ rp,
                                                        const int32_t *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ dP, double *__restrict__ U, double *__restrict__ p)
{
    const int64_t total = UN > pN ? UN : pN;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        if (g < UN) {
            double r = 0.0;
            for (int32_t q = rp[g]; q < rp[g + 1]; ++q) r = r + val[q] * dP[col[q]];
            U[g] = U[g] + (-1.0) * r;
        }
        if (g < pN) p[g] = p[g] + 1.0 * dP[g];
    }
}

}  // namespace pib

namespace pib {
int ns_bng_apply(pib_ns *ns, const double *phi, double *out, hipStream_t q)
{
    const NsDev &D = ns->D;
    hipLaunchKernelGGL(k_ns_bng, dim3((unsigned)std::min<int64_t>(4096, (D.UN + 255) / 256)), dim3(256), 0, q, D, ns->dt, phi, out);
    PIB_HIP(hipGetLastError());
    return 0;
}
// the pinned pressure's cell (global cell 0) in this rank's (extended) slab, -1 where there is none
static int64_t ns_pin_cell(const pib_ns *ns)
{
    if (!ns->pinned || ns->rank != 0) return -1;
    return (ns->slab_pk0 - ns->slab_e0) * ns->p_plane;  // 0 unless the slab starts with a plane of the last rank (periodic slab axis)
}
int ns_div_sub(pib_ns *ns, const double *t, double *w, hipStream_t q)
{
    const NsDev &D = ns->D;
    hipLaunchKernelGGL(k_ns_div_sub, dim3((unsigned)std::min<int64_t>(4096, (D.pN + 255) / 256)), dim3(256), 0, q, D, ns_pin_cell(ns), t, w);
    PIB_HIP(hipGetLastError());
    return 0;
}
}  // namespace pib

static int ghost_blocks(const pib::NsDev &D) { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (D.nghost + 255) / 256)); }

namespace pib {
int ns_before_solve(pib_ns *ns, pib_solver *sol)
{
    PIB_HIP(hipEventRecord(ns->ev_dep, ns->stream));
    PIB_HIP(hipStreamWaitEvent(sol->stream, ns->ev_dep, 0));
    return 0;
}

}  // namespace pib

extern "C" {

int pib_ns_destroy(pib_ns *ns)
try {
    if (ns == nullptr) return 0;
    (void)hipSetDevice(ns->device);
    // (captured iterations first: the Poisson solver's may hold the Schur hook's pointers into the immersed-boundary state)
    pib::drop_iteration_graph(ns->psol);
    pib::drop_iteration_graph(ns->vsol);
    pib::ib_release(ns->ib);
    ns->ib = nullptr;
    for (int k = 0; k < 7; ++k)
        if (ns->ev_stage[k]) (void)hipEventDestroy(ns->ev_stage[k]);
    if (ns->vsol) pib_destroy(ns->vsol);
    if (ns->psol) pib_destroy(ns->psol);
    for (double *q : ns->owned) (void)hipFree(q);
    if (ns->bng_rowptr) (void)hipFree(ns->bng_rowptr);
    if (ns->bng_col) (void)hipFree(ns->bng_col);
    if (ns->bng_val) (void)hipFree(ns->bng_val);
    if (ns->bn_rowptr) (void)hipFree(ns->bn_rowptr);
    if (ns->bn_col) (void)hipFree(ns->bn_col);
    if (ns->bn_val) (void)hipFree(ns->bn_val);
    if (ns->ev_dep) (void)hipEventDestroy(ns->ev_dep);
    if (ns->stream) (void)hipStreamDestroy(ns->stream);
    delete ns;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

}  // extern "C"

// The engine on one rank's z-slab (y-slab in 2-D) of the mesh -- SURVEY.md 8e, NavierStokesSolver on the DMDA
// decomposition (src/mesh/cartesianmesh.cpp:492-538).  A rank keeps, of every field, its owned planes plus one plane of
// each neighbour (and one more, unused, plane above: the component along the slab axis lives on the faces BETWEEN the
// planes), all in the packed [u | v | w] layout: to the kernels this EXTENDED slab is simply a smaller mesh whose 1-D
// arrays along the slab axis are slices of the global ones -- the "ghost" entries at a cut are the neighbour's real
// coordinates -- so every kernel of the single-rank engine runs unchanged and computes, on the owned planes, the bits
// the single rank computes (their stencils reach one plane; what the kernels write on the neighbours' planes from the
// dummy boundary condition at a cut is overwritten by the next halo exchange).  The two linear solvers work on the
// GLOBAL systems (this rank's rows: pib_assemble_velocity / pib_assemble_poisson on slabs) and share one communicator.
static int ns_create_impl(pib_ns **out, int dim, const int64_t n_global[3], const double *wx, const double *wy,
                          const double *wz, const double lo[3], const double hi[3], const int bc_type_in[18],
                          const double bc_value_in[18], double dt, double nu, const char *velocity_cfg,
                          const char *poisson_cfg, int device, int rank, int nranks, const void *uid)
{
    using namespace pib;
    if (out == nullptr || n_global == nullptr || lo == nullptr || hi == nullptr || bc_type_in == nullptr || bc_value_in == nullptr)
        return fail(PIB_ERR_ARG_NULL, "pib_ns_create: null argument");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_create: dim must be 2 or 3");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_create: bad rank %d / %d", rank, nranks);
    const double *w[3] = {wx, wy, wz};
    const int64_t *n = n_global;
    int bc_type[18];
    double bc_value[18];
    std::memcpy(bc_type, bc_type_in, sizeof bc_type);
    std::memcpy(bc_value, bc_value_in, sizeof bc_value);
    std::vector<double> hdl[3][3], hco[3][3];
    int64_t fn[3][3];
    // periodic directions: PERIODIC at both ends for every component (misc.cpp:17-85 checkPeriodicBC)
    int periodic[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
        int cnt = 0;
        for (int f = 0; f < dim; ++f)
            for (int e = 0; e < 2; ++e) cnt += (bc_type[6 * f + 2 * d + e] == 3) ? 1 : 0;
        if (cnt != 0 && cnt != 2 * dim)
            return fail(PIB_ERR_ARG_WRONG, "pib_ns_create: direction %d is periodic for some components / ends only "
                                           "(a periodic boundary needs PERIODIC at both ends for every velocity component)", d);
        periodic[d] = cnt ? 1 : 0;
    }
    velocity_mesh_arrays(dim, n, w, lo, hi, periodic, hdl, hco, fn);
    std::vector<double> vtx_global[3], dlu_global[3];  // before the slab axis is cut: the bodies' background cells are global
    for (int d = 0; d < dim; ++d) {
        vtx_global[d] = hco[d][d];
        dlu_global[d] = hdl[0][d];
    }
    // ---- this rank's extended slab: cells [e0, e1) of the slab axis = owned [pk0, pk1) + one plane below + two above
    const int sd = dim - 1;
    int64_t pk0 = 0, pk1 = n[sd], e0 = 0, e1 = n[sd];
    int64_t fn_global[3] = {1, 1, 1};
    for (int f = 0; f < dim; ++f) fn_global[f] = fn[f][sd];
    std::vector<double> w_slab;
    int64_t n_local[3] = {n[0], n[1], n[2]};
    // a periodic slab axis on several ranks (the Taylor-Green box on several GPUs): every rank's extended slab has a cut at
    // both ends -- rank 0's lower neighbour plane is the last rank's top plane -- and the engine's kernels see a wall-bounded
    // direction there; the wrap lives in the plane exchanges (a ring) and in the two linear systems
    const bool ring = nranks > 1 && periodic[sd] != 0;
    const int64_t n_global_sd = n[sd];
    int periodic_local[3] = {periodic[0], periodic[1], periodic[2]};
    if (nranks > 1) {
        slab_range(n[sd], nranks, rank, &pk0, &pk1);
        if (pk1 - pk0 < 2) return fail(PIB_ERR_SUP, "pib_ns_create_slab: rank %d owns %lld plane(s); every rank needs >= 2", rank, (long long)(pk1 - pk0));
        e0 = ring ? pk0 - 1 : std::max<int64_t>(pk0 - 1, 0);            // with a ring: cell -1 is cell n - 1, cells n, n + 1 are 0, 1
        e1 = ring ? pk1 + 2 : std::min<int64_t>(pk1 + 2, n[sd]);
        const double length = hi[sd] - lo[sd];
        for (int f = 0; f < dim; ++f) {
            const int64_t cnt = (f == sd) ? (e1 - e0 - 1) : (e1 - e0);  // faces between the local planes / the local planes
            std::vector<double> &a = hdl[f][sd], &c = hco[f][sd];
            // entry s + 1 <-> point s, one entry beyond each end
            if (!ring) {
                a = std::vector<double>(a.begin() + e0, a.begin() + e0 + cnt + 2);
                c = std::vector<double>(c.begin() + e0, c.begin() + e0 + cnt + 2);
            } else {
                // the single-rank arrays hold the points -1 .. np (their two ghost entries are the wrapped neighbours): those
                // entries are taken as they are (same bits as on one rank); a point further out -- only the dummy entries beyond
                // the cuts -- is its periodic image, a domain length away
                const int64_t np = fn[f][sd];
                std::vector<double> a2, c2;
                for (int64_t j = e0; j < e0 + cnt + 2; ++j) {
                    if (j >= 0 && j <= np + 1) {
                        a2.push_back(a[(size_t)j]);
                        c2.push_back(c[(size_t)j]);
                    } else {
                        const int64_t pt = j - 1, img = ((pt % np) + np) % np, turns = (pt - img) / np;
                        a2.push_back(a[(size_t)img + 1]);
                        c2.push_back(c[(size_t)img + 1] + (double)turns * length);
                    }
                }
                a = a2;
                c = c2;
            }
            fn[f][sd] = cnt;
        }
        w_slab.clear();
        for (int64_t q = e0; q < e1; ++q) w_slab.push_back(w[sd][(size_t)(((q % n[sd]) + n[sd]) % n[sd])]);
        w[sd] = w_slab.data();
        n_local[sd] = e1 - e0;
        // a cut is a dummy wall (the values the kernels derive from it land on the neighbours' planes only)
        for (int f = 0; f < dim; ++f) {
            if (e0 > 0 || ring) { bc_type[6 * f + 2 * sd] = 0; bc_value[6 * f + 2 * sd] = 0.0; }
            if (e1 < n[sd] || ring) { bc_type[6 * f + 2 * sd + 1] = 0; bc_value[6 * f + 2 * sd + 1] = 0.0; }
        }
        if (ring) periodic_local[sd] = 0;
    }
    pib_ns *ns = new pib_ns();
    ns->nranks = nranks;
    ns->rank = rank;
    ns->slab_pk0 = pk0;
    ns->slab_pk1 = pk1;
    ns->slab_e0 = e0;
    for (int d = 0; d < 3; ++d) ns->periodic[d] = periodic[d];
    ns->ring = ring;
    ns->dt = dt;
    ns->nu = nu;
    for (int d = 0; d < dim; ++d) {
        ns->h_vtx[d] = vtx_global[d];  // coord[component d][direction d] = the vertices (cartesianmesh.cpp:246)
        ns->h_dlu[d] = dlu_global[d];
        ns->lo[d] = lo[d];
        ns->hi[d] = hi[d];
    }
    int err = 0;
    auto bail = [&](int e) {
        pib_ns_destroy(ns);
        return e;
    };
    // the two linear solvers (same device)
    if ((err = pib_create_from_string(&ns->vsol, "velocity", velocity_cfg, rank, nranks, uid, device))) return bail(err);
    // one communicator (an RCCL id makes one): the Poisson solver borrows the velocity solver's
    if ((err = create_sharing_comm(&ns->psol, "poisson", poisson_cfg, ns->vsol))) return bail(err);
    if (ns->psol->cfg.matrix_free_poisson < 0) ns->psol->cfg.matrix_free_poisson = 1;  // auto: on inside the time step
    ns->device = ns->vsol->device;
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamCreateWithFlags(&ns->stream, hipStreamNonBlocking));
    PIB_HIP(hipEventCreateWithFlags(&ns->ev_dep, hipEventDisableTiming));
    // ghost-equation tables: ghost = a0*target + a1
    double a0[18], gdl[18];
    int btype[18];
    for (int f = 0; f < 3; ++f)
        for (int loc = 0; loc < 6; ++loc) {
            a0[6 * f + loc] = gdl[6 * f + loc] = 0.0;
            btype[6 * f + loc] = 0;
            if (f >= dim || loc >= 2 * dim) continue;
            const int t = bc_type[6 * f + loc];
            const int axis = loc / 2;
            btype[6 * f + loc] = t;
            {   // distance ghost - target (misc.cpp:187-190)
                const std::vector<double> &c = hco[f][axis];
                const int64_t nf = fn[f][axis];
                gdl[6 * f + loc] = (loc % 2 == 1) ? c[(size_t)nf + 1] - c[(size_t)nf] : c[1] - c[0];
            }
            if (t == 3) {  // PERIODIC: no ghost point, no fold (singleboundaryperiodic.cpp)
                gdl[6 * f + loc] = 0.0;
            } else if (t == 0 || t == 2) {  // DIRICHLET (singleboundarydirichlet.cpp:35-44), CONVECTIVE (singleboundaryconvective.cpp:20-36)
                a0[6 * f + loc] = (axis == f) ? 0.0 : -1.0;
            } else if (t == 1) {  // NEUMANN (singleboundaryneumann.cpp:27-28) -- on the NORMAL component too (round 5): a0 = 1 then
                // folds into D (createdivergence.cpp:231-242) and hence into DBNG, which loses its symmetry at that face:
                // the Poisson operator comes from the reference's product chain with the folded D (bn.hip), see below
                a0[6 * f + loc] = 1.0;
                if (axis == f) ns->neumann_normal = true;
            } else {
                return bail(fail(PIB_ERR_SUP, "pib_ns_create: boundary type %d is not supported (0 DIRICHLET, 1 NEUMANN, "
                                              "2 CONVECTIVE, 3 PERIODIC)", t));
            }
        }
    // matrices: A = I/dt - c nu L (CN: c = 1/2), DBNG; null-space convention from the Poisson solver's flavour
    // (navierstokes.cpp:395-429)
    if ((err = pib_set_periodic(ns->vsol, periodic)) || (err = pib_set_periodic(ns->psol, periodic))) return bail(err);
    // the global systems (this rank's rows); a0 of the TRUE boundaries (the dummy walls at the cuts are Dirichlet too, so
    // the table built above already has the true values wherever a true boundary exists)
    double a0_global[18];
    std::memcpy(a0_global, a0, sizeof a0);
    for (int f = 0; f < dim && nranks > 1; ++f)
        for (int e = 0; e < 2; ++e) {
            const int t = bc_type_in[6 * f + 2 * sd + e];
            a0_global[6 * f + 2 * sd + e] = (t == 1) ? 1.0 : ((sd == f) ? 0.0 : -1.0);
        }
    if ((err = pib_assemble_velocity(ns->vsol, dim, n, wx, wy, wz, lo, hi, a0_global, dt, 0.5 * nu))) return bail(err);
    char tbuf[64];
    pib_get_type(ns->psol, tbuf, sizeof tbuf);
    ns->pinned = (std::strcmp(tbuf, "NVIDIA AmgX") == 0) ? 1 : 0;
    const double *wg[3] = {wx, wy, wz};
    if (ns->neumann_normal) {
        // D carries a ghost fold: DBNG = D (dt I) G through the chain of sparse products with the folded D (bn.hip, order 1; on
        // slabs the window chain), the multigrid of the unfolded operator registered as preconditioner only.  The matrix is no
        // longer symmetric at the Neumann face, so a solver file that says CG gets BiCGStab with the same preconditioner -- said
        // once on stderr, never silently
        if (ns->psol->cfg.method == Method::CG) {
            ns->psol->cfg.method = Method::BICGSTAB;
            ns->psol->departures.push_back("method: the file's cg runs as bicgstab (same preconditioner): a NEUMANN condition on a normal velocity "
                                           "component makes DBNG non-symmetric (createdivergence.cpp:231-242)");
            std::fprintf(stderr, "[petibm_amd] poisson solver: a NEUMANN condition on a normal velocity component makes DBNG non-symmetric "
                                 "(createdivergence.cpp:231-242): CG of the solver file replaced by BiCGStab, same preconditioner\n");
        }
        if ((err = assemble_poisson_bn(ns->psol, dim, n, wg, lo, hi, a0_global, dt, 0.5 * nu, 1, ns->pinned ? PIB_NULLSPACE_PINNED : PIB_NULLSPACE_CONSTANT,
                                       nullptr, nullptr, nullptr, nullptr)))
            return bail(err);
    } else if ((err = pib_assemble_poisson(ns->psol, dim, n, wx, wy, wz, dt, ns->pinned ? PIB_NULLSPACE_PINNED : PIB_NULLSPACE_CONSTANT)))
        return bail(err);
    for (int d = 0; d < dim; ++d) {
        ns->h_n[d] = n[d];
        ns->h_w[d].assign(wg[d], wg[d] + n[d]);
    }
    std::memcpy(ns->h_a0, a0_global, sizeof a0_global);
    n = n_local;  // from here on: the (extended) slab as the kernels' mesh
    // device mesh arrays
    NsDev &D = ns->D;
    D.dim = dim;
    D.per = (periodic_local[0] ? 1 : 0) | (periodic_local[1] ? 2 : 0) | (periodic_local[2] ? 4 : 0);
    D.seam_axis = ring ? sd : -1;
    D.seam_lo = (ring && e0 < 0) ? -e0 : 0;
    D.seam_hi = ring ? n_global_sd - e0 : 0;
    int64_t off = 0;
    for (int f = 0; f < 3; ++f) {
        NsField &F = D.f[f];
        for (int d = 0; d < 3; ++d) {
            F.n[d] = fn[f][d];
            F.dl[d] = F.co[d] = nullptr;
            F.lneg[d] = F.lpos[d] = nullptr;
            if (d == 0) F.ginv = nullptr;
            if (f < dim && d < dim) {
                double *p1 = nullptr, *p2 = nullptr;
                if ((err = upload_vec(hdl[f][d], &p1)) || (err = upload_vec(hco[f][d], &p2))) return bail(err);
                ns->owned.push_back(p1);
                ns->owned.push_back(p2);
                F.dl[d] = p1;
                F.co[d] = p2;
                // quotient tables: the expressions of laplacian_at / the gradient row, evaluated once (IEEE division
                // and multiplication round identically on the host)
                const int64_t nfd = fn[f][d];
                std::vector<double> tn((size_t)nfd), tp((size_t)nfd), tg((size_t)nfd);
                for (int64_t q = 0; q < nfd; ++q) {
                    const double dLSelf = hdl[f][d][(size_t)q + 1];
                    const double dLNeg = hco[f][d][(size_t)q + 1] - hco[f][d][(size_t)q];
                    const double dLPos = hco[f][d][(size_t)q + 2] - hco[f][d][(size_t)q + 1];
                    tn[(size_t)q] = 1.0 / (dLNeg * dLSelf);
                    tp[(size_t)q] = 1.0 / (dLPos * dLSelf);
                    tg[(size_t)q] = 1.0 / dLSelf;
                }
                double *p3 = nullptr, *p4 = nullptr;
                if ((err = upload_vec(tn, &p3)) || (err = upload_vec(tp, &p4))) return bail(err);
                ns->owned.push_back(p3);
                ns->owned.push_back(p4);
                F.lneg[d] = p3;
                F.lpos[d] = p4;
                if (d == f) {
                    double *p5 = nullptr;
                    if ((err = upload_vec(tg, &p5))) return bail(err);
                    ns->owned.push_back(p5);
                    F.ginv = p5;
                }
            }
        }
        F.off = off;
        if (f < dim) off += fn[f][0] * fn[f][1] * fn[f][2];
        for (int q = 0; q < 6; ++q) {
            F.a0[q] = a0[6 * f + q];
            F.type[q] = btype[6 * f + q];
            F.bcv[q] = bc_value[6 * f + q];
            F.gdl[q] = gdl[6 * f + q];
            F.goff[q] = F.gcnt[q] = 0;
        }
    }
    D.UN = off;
    D.nghost = 0;
    for (int f = 0; f < dim; ++f)
        for (int q = 0; q < 2 * dim; ++q) {
            NsField &F = D.f[f];
            F.goff[q] = D.nghost;
            F.gcnt[q] = periodic_local[q / 2] ? 0 : fn[f][0] * fn[f][1] * fn[f][2] / fn[f][q / 2];
            D.nghost += F.gcnt[q];
        }
    D.pN = 1;
    for (int d = 0; d < 3; ++d) {
        D.pn[d] = (d < dim) ? n[d] : 1;
        D.pN *= D.pn[d];
        D.pw[d] = nullptr;
        if (d < dim) {
            std::vector<double> hw(w[d], w[d] + n[d]);
            double *p1 = nullptr;
            if ((err = upload_vec(hw, &p1))) return bail(err);
            ns->owned.push_back(p1);
            D.pw[d] = p1;
        }
    }
    auto alloc = [&](double **q, int64_t cnt) -> int {
        PIB_HIP(hipMalloc(q, sizeof(double) * (size_t)cnt));
        PIB_MEMSET(*q, 0, sizeof(double) * (size_t)cnt);
        ns->owned.push_back(*q);
        return 0;
    };
    if ((err = alloc(&ns->U, D.UN)) || (err = alloc(&ns->rhs1, D.UN)) || (err = alloc(&ns->conv[0], D.UN)) ||
        (err = alloc(&ns->conv[1], D.UN)) || (err = alloc(&ns->diff0, D.UN)) || (err = alloc(&ns->diff1, D.UN)) || (err = alloc(&ns->p, D.pN)) || (err = alloc(&ns->dP, D.pN)) ||
        (err = alloc(&ns->rhs2, D.pN)) || (err = alloc(&D.a1, D.nghost)) || (err = alloc(&D.a1n, D.nghost)) ||
        (err = alloc(&D.gv, D.nghost)))
        return bail(err);
    // slab bookkeeping: the owned planes of every field inside the extended slab, and the packed [u | v | w] vector of
    // the owned points the velocity solver works on (what DMCompositeGetAccess hands out on a rank)
    ns->UN_owned = 0;
    for (int f = 0; f < 3; ++f) {
        ns->fld_plane[f] = ns->fld_own_lo[f] = ns->fld_own_cnt[f] = ns->fld_pk_off[f] = 0;
        if (f >= dim) continue;
        ns->fld_plane[f] = (dim == 3) ? fn[f][0] * fn[f][1] : fn[f][0];
        ns->fld_own_lo[f] = pk0 - e0;
        ns->fld_own_cnt[f] = std::min(pk1, fn_global[f]) - pk0;
        ns->fld_pk_off[f] = ns->UN_owned;
        ns->UN_owned += ns->fld_plane[f] * ns->fld_own_cnt[f];
    }
    ns->p_plane = (dim == 3) ? n[0] * n[1] : n[0];
    ns->pN_owned = ns->p_plane * (pk1 - pk0);
    if (nranks > 1) {
        if (ns->UN_owned != ns->vsol->A.n || ns->pN_owned != ns->psol->A.n)
            return bail(fail(PIB_ERR_LIB, "pib_ns_create_slab: slab sizes disagree with the assembled systems"));
        if ((err = alloc(&ns->Upk, ns->UN_owned)) || (err = alloc(&ns->rhs1pk, ns->UN_owned))) return bail(err);
    }
    // bc->setGhostICs(solution) for the zero initial state (navierstokes.cpp:142)
    hipLaunchKernelGGL(k_ns_ghosts<0>, dim3(ghost_blocks(D)), dim3(256), 0, ns->stream, D, 0.0, ns->U);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipStreamSynchronize(ns->stream));
    *out = ns;
    return 0;
}

// ---- slab helpers (no-ops on one rank)
namespace pib {
// owned points of the extended-slab vector `ext` <-> the packed vector `pk` (three contiguous pieces)
static int ns_pack(pib_ns *ns, const double *ext, double *pk)
{
    for (int f = 0; f < ns->D.dim; ++f)
        PIB_HIP(hipMemcpyAsync(pk + ns->fld_pk_off[f], ext + ns->D.f[f].off + ns->fld_own_lo[f] * ns->fld_plane[f],
                               sizeof(double) * (size_t)(ns->fld_own_cnt[f] * ns->fld_plane[f]), hipMemcpyDeviceToDevice, ns->stream));
    return 0;
}
static int ns_unpack(pib_ns *ns, const double *pk, double *ext)
{
    for (int f = 0; f < ns->D.dim; ++f)
        PIB_HIP(hipMemcpyAsync(ext + ns->D.f[f].off + ns->fld_own_lo[f] * ns->fld_plane[f], pk + ns->fld_pk_off[f],
                               sizeof(double) * (size_t)(ns->fld_own_cnt[f] * ns->fld_plane[f]), hipMemcpyDeviceToDevice, ns->stream));
    return 0;
}
// one plane of every velocity component from each neighbour (C1: the VecScatter of DMGlobalToLocal, navierstokes.cpp:246)
static int ns_halo_velocity(pib_ns *ns, double *ext)
{
    if (ns->nranks <= 1) return 0;
    const int r = ns->rank, P = ns->nranks;
    for (int f = 0; f < ns->D.dim; ++f) {
        const int64_t pl = ns->fld_plane[f];
        const int64_t lo_n = (r > 0 || ns->ring) ? pl : 0, hi_n = (r < P - 1 || ns->ring) ? pl : 0;
        PIB_CHK(halo_exchange_planes(ns->vsol, ext + ns->D.f[f].off + ns->fld_own_lo[f] * pl, ns->fld_own_cnt[f] * pl, lo_n, hi_n, lo_n,
                                     hi_n, ns->stream));
    }
    return 0;
}
static int ns_halo_cells(pib_ns *ns, double *ext)
{
    if (ns->nranks <= 1) return 0;
    const int r = ns->rank, P = ns->nranks;
    const int64_t pl = ns->p_plane;
    const int64_t lo_n = (r > 0 || ns->ring) ? pl : 0, hi_n = (r < P - 1 || ns->ring) ? pl : 0;
    return halo_exchange_planes(ns->vsol, ext + (ns->slab_pk0 - ns->slab_e0) * pl, ns->pN_owned, lo_n, hi_n, lo_n, hi_n, ns->stream);
}
__global__ __launch_bounds__(256) void k_ns_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = y[i] + a * x[i];
}
// u = u - BN G dP, p = p + dP on z-slabs with BN order N > 1 (navierstokes.cpp:583-615).  One rank multiplies by the
// assembled BNG; a row of it reaches N planes along the slab axis, so here BN = sum_k dt^k (c nu)^(k-1) L^(k-1)
// (createbn.cpp:47-92) is APPLIED term by term to t = dt G dP instead: L is the velocity solver's matrix-free operator with
// MatScale(1), MatShift(0) -- its tables cover this rank's planes, the neighbours' planes come through the solver's own
// halo exchange -- one exchange per extra term.  The same operator to rounding (the sums run in another order than the
// assembled rows'): tests/test_gpu_bn.py compares 2 / 3 loopback ranks with the single rank at the solver tolerance.
// t (extended slab; its owned points) <- sum_{k = 1 .. N} (dt c nu)^(k-1) L^(k-1) t : BN without its leading dt, term by term
// (createBnHead's series, createbn.cpp:59-92).  Collective over the ranks; returns behind a synchronisation of both streams.
int ns_bn_series_slab(pib_ns *ns, double *t)
{
    pib_solver *vs = ns->vsol;
    if (!vs->vel.valid || vs->vel.slab_axis < 0)
        return fail(PIB_ERR_SUP, "BN order > 1 on z-slabs needs the matrix-free velocity operator (pib_matrix_free_velocity=1)");
    const NsDev &D = ns->D;
    hipStream_t q = ns->stream;
    const int64_t no = ns->UN_owned;
    const unsigned gb = (unsigned)std::min<int64_t>(4096, (no + 255) / 256);
    PIB_CHK(ensure_work(vs, 2));
    PIB_HIP(hipStreamSynchronize(vs->stream));  // (a first allocation clears the vectors on the SOLVER's stream: not behind the pack below)
    double *Y = vs->vec(0), *Z = vs->vec(1), *acc = ns->rhs1pk;  // the velocity solve is over: its vectors are free
    PIB_CHK(ns_pack(ns, t, Y));
    PIB_HIP(hipMemcpyAsync(acc, Y, sizeof(double) * (size_t)no, hipMemcpyDeviceToDevice, q));  // term 1: the identity
    const double scale0 = vs->vel.scale, shift0 = vs->vel.shift;
    const double cnu = ns->T.cimpl * ns->nu;
    int err = 0;
    for (int term = 2; term <= ns->bn_order && !err; ++term) {
        PIB_HIP(hipStreamSynchronize(q));  // the exchange below runs on the solver's stream
        if ((err = halo_exchange(vs, Y, vs->stream))) break;
        vs->vel.scale = 1.0;  // L itself
        vs->vel.shift = 0.0;
        err = vel_stencil_apply(vs, Y, Z, false, vs->stream);
        vs->vel.scale = scale0;
        vs->vel.shift = shift0;
        if (err) break;
        PIB_HIP(hipStreamSynchronize(vs->stream));
        std::swap(Y, Z);
        hipLaunchKernelGGL(k_ns_axpy, dim3(gb), dim3(256), 0, q, no, std::pow(ns->dt, term - 1) * std::pow(cnu, term - 1), Y, acc);
        PIB_HIP(hipGetLastError());
    }
    if (err) return err;
    PIB_HIP(hipMemsetAsync(t, 0, sizeof(double) * (size_t)D.UN, q));
    PIB_CHK(ns_unpack(ns, acc, t));
    return 0;
}

static int ns_project_bn_slab(pib_ns *ns)
{
    const NsDev &D = ns->D;
    hipStream_t q = ns->stream;
    if (ns->bn_tmp == nullptr) {
        PIB_HIP(hipMalloc(&ns->bn_tmp, sizeof(double) * (size_t)D.UN));
        ns->owned.push_back(ns->bn_tmp);
    }
    double *t = ns->bn_tmp;
    PIB_CHK(ns_bng_apply(ns, ns->dP, t, q));  // t = dt G dP on the extended slab
    PIB_CHK(ns_bn_series_slab(ns, t));        // t = BN G dP
    hipLaunchKernelGGL(k_ns_axpy, dim3((unsigned)std::min<int64_t>(4096, (D.UN + 255) / 256)), dim3(256), 0, q, D.UN, -1.0, t, ns->U);
    hipLaunchKernelGGL(k_ns_axpy, dim3((unsigned)std::min<int64_t>(4096, (D.pN + 255) / 256)), dim3(256), 0, q, D.pN, 1.0, ns->dP, ns->p);
    PIB_HIP(hipGetLastError());
    return 0;
}
}  // namespace pib

extern "C" {

int pib_ns_create(pib_ns **out, int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                  const double lo[3], const double hi[3], const int bc_type[18], const double bc_value[18], double dt,
                  double nu, const char *velocity_cfg, const char *poisson_cfg, int device)
try {
    return ns_create_impl(out, dim, n, wx, wy, wz, lo, hi, bc_type, bc_value, dt, nu, velocity_cfg, poisson_cfg, device, 0, 1, nullptr);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_create_slab(pib_ns **out, int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                       const double lo[3], const double hi[3], const int bc_type[18], const double bc_value[18], double dt,
                       double nu, const char *velocity_cfg, const char *poisson_cfg, int rank, int nranks,
                       const void *uid_or_null, int device)
try {
    return ns_create_impl(out, dim, n, wx, wy, wz, lo, hi, bc_type, bc_value, dt, nu, velocity_cfg, poisson_cfg, device, rank, nranks,
                          uid_or_null);
} catch (...) {
    return pib::fail_exception(__func__);
}

/* parameters.BN (navierstokes.cpp:349-356): order N of the approximate inverse BN.  N > 1 re-assembles the Poisson
 * operator through the product chain D * BN * G (bn.hip) and keeps the assembled BNG for the projection. */
int pib_ns_set_bn_order(pib_ns *ns, int order)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    if (order < 1) return fail(PIB_ERR_SUP, "The order of Bn can not be smaller than 1.");
    if (order == ns->bn_order) return 0;
    if (ns->ib) return fail(PIB_ERR_ORDER, "pib_ns_set_bn_order: call it before pib_ns_set_bodies (BNH = BN H and E BN H are built from BN)");
    PIB_HIP(hipSetDevice(ns->device));
    if (ns->bng_rowptr) (void)hipFree(ns->bng_rowptr);
    if (ns->bng_col) (void)hipFree(ns->bng_col);
    if (ns->bng_val) (void)hipFree(ns->bng_val);
    if (ns->bn_rowptr) (void)hipFree(ns->bn_rowptr);
    if (ns->bn_col) (void)hipFree(ns->bn_col);
    if (ns->bn_val) (void)hipFree(ns->bn_val);
    ns->bng_rowptr = ns->bng_col = ns->bn_rowptr = ns->bn_col = nullptr;
    ns->bng_val = ns->bn_val = nullptr;
    ns->bng_nnz = ns->bn_nnz = 0;
    const int dim = ns->D.dim;
    const double *w[3] = {ns->h_w[0].data(), ns->h_w[1].data(), dim == 3 ? ns->h_w[2].data() : nullptr};
    const int nullspace = ns->pinned ? PIB_NULLSPACE_PINNED : PIB_NULLSPACE_CONSTANT;
    if (order == 1 && !ns->neumann_normal) {
        PIB_CHK(pib_assemble_poisson(ns->psol, dim, ns->h_n, w[0], w[1], w[2], ns->dt, nullspace));
    } else if (order == 1) {  // (the folded D: the chain, with the matrix-free projection of order 1)
        ns->psol->has_matrix = false;
        ns->psol->has_grid = false;
        gmg_release(ns->psol);
        PIB_CHK(assemble_poisson_bn(ns->psol, dim, ns->h_n, w, ns->lo, ns->hi, ns->h_a0, ns->dt, ns->T.cimpl * ns->nu, 1, nullspace, nullptr, nullptr,
                                    nullptr, nullptr));
    } else {
        ns->psol->has_matrix = false;
        ns->psol->has_grid = false;
        gmg_release(ns->psol);
        if (ns->nranks > 1)  // z-slabs: the operator from the window chain (bn.hip), the projection applies BN term by term
            PIB_CHK(assemble_poisson_bn(ns->psol, dim, ns->h_n, w, ns->lo, ns->hi, ns->h_a0, ns->dt, ns->T.cimpl * ns->nu, order, nullspace,
                                        nullptr, nullptr, nullptr, nullptr));
        else
        PIB_CHK(assemble_poisson_bn(ns->psol, dim, ns->h_n, w, ns->lo, ns->hi, ns->h_a0, ns->dt, ns->T.cimpl * ns->nu, order, nullspace,
                                    &ns->bng_rowptr, &ns->bng_col, &ns->bng_val, &ns->bng_nnz, &ns->bn_rowptr, &ns->bn_col, &ns->bn_val,
                                    &ns->bn_nnz));
    }
    ns->bn_order = order;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* parameters.convection / parameters.diffusion (src/timeintegration/timeintegration.cpp:41-80): EULER_EXPLICIT,
 * EULER_IMPLICIT, ADAMS_BASHFORTH_2, CRANK_NICOLSON for either term; the convective term uses the scheme's explicit
 * coefficients, the diffusive one its implicit coefficient (the velocity operator A = I/dt - c nu L is re-assembled)
 * and its explicit coefficients (navierstokes.cpp:342-344,448-514). */
static int scheme_coeffs(const char *name, int *nexp, double ce[2], double *cimpl)
{
    const std::string sname = name ? name : "";
    ce[0] = ce[1] = 0.0;
    if (sname == "EULER_EXPLICIT") { *cimpl = 0.0; *nexp = 1; ce[0] = 1.0; }
    else if (sname == "EULER_IMPLICIT") { *cimpl = 1.0; *nexp = 0; }
    else if (sname == "ADAMS_BASHFORTH_2") { *cimpl = 0.0; *nexp = 2; ce[0] = 1.5; ce[1] = -0.5; }
    else if (sname == "CRANK_NICOLSON") { *cimpl = 0.5; *nexp = 1; ce[0] = 0.5; }
    else return pib::fail(PIB_ERR_ARG_OUTOFRANGE, "The time integration scheme \"%s\" does not exist.", sname.c_str());
    return 0;
}

int pib_ns_set_time_integration(pib_ns *ns, const char *convection, const char *diffusion)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    if (ns->ib) return fail(PIB_ERR_ORDER, "pib_ns_set_time_integration: call before pib_ns_set_bodies");
    NsTime T = ns->T;
    double unused = 0.0;
    PIB_CHK(scheme_coeffs(convection, &T.nconv, T.cc, &unused));
    PIB_CHK(scheme_coeffs(diffusion, &T.ndiff, T.dc, &T.cimpl));
    const bool reassemble = T.cimpl != ns->T.cimpl;
    ns->T = T;
    if (!reassemble) return 0;
    PIB_HIP(hipSetDevice(ns->device));
    const int dim = ns->D.dim;
    PIB_CHK(pib_assemble_velocity(ns->vsol, dim, ns->h_n, ns->h_w[0].data(), ns->h_w[1].data(), dim == 3 ? ns->h_w[2].data() : nullptr,
                                  ns->lo, ns->hi, ns->h_a0, ns->dt, T.cimpl * ns->nu));
    if (ns->bn_order > 1) {  // BN depends on the implicit coefficient
        const int order = ns->bn_order;
        ns->bn_order = 0;
        PIB_CHK(pib_ns_set_bn_order(ns, order));
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* One explicit term kept between steps (restart files: /convection/<index>, /diffusion/<index>,
 * navierstokes.cpp:637-686,689-746).  kind 0: convection, 1: diffusion; host array of UN entries. */
int pib_ns_history_term(pib_ns *ns, int kind, int index, int set, double *host)
try {
    using namespace pib;
    if (ns == nullptr || host == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_ns_history_term: null argument");
    const int count = (kind == 0) ? ns->T.nconv : ns->T.ndiff;
    if ((kind != 0 && kind != 1) || index < 0 || index >= count)
        return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_history_term: the scheme keeps %d term(s) of kind %d", count, kind);
    PIB_HIP(hipSetDevice(ns->device));
    double *dev = (kind == 0) ? ns->conv[index] : (index == 0 ? ns->diff0 : ns->diff1);
    if (ns->nranks > 1) {  // the packed owned points (only those are ever read back by the right-hand side)
        const size_t ub = sizeof(double) * (size_t)ns->UN_owned;
        if (set) {
            PIB_HIP(hipMemcpy(ns->rhs1pk, host, ub, hipMemcpyHostToDevice));
            PIB_CHK(ns_unpack(ns, ns->rhs1pk, dev));
            PIB_HIP(hipStreamSynchronize(ns->stream));
        } else {
            PIB_CHK(ns_pack(ns, dev, ns->rhs1pk));
            PIB_HIP(hipStreamSynchronize(ns->stream));
            PIB_HIP(hipMemcpy(host, ns->rhs1pk, ub, hipMemcpyDeviceToHost));
        }
        return 0;
    }
    const size_t bytes = sizeof(double) * (size_t)ns->D.UN;
    if (set) PIB_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    else PIB_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* The vorticity field of the reference's petibm-vorticity utility (applications/vorticity/main.cpp) from the current
 * velocity and ghost values.  comp: 0 wx, 1 wy, 2 wz (2-D: only 2).  n_out[3] receives the field's point counts
 * (:384-470: vertices along the two directions the component is not centred on); out may be NULL to query them. */
int pib_ns_get_vorticity(pib_ns *ns, int comp, int64_t n_out[3], double *out)
try {
    using namespace pib;
    if (ns == nullptr || n_out == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_ns_get_vorticity: null argument");
    const NsDev &D = ns->D;
    if (comp < 0 || comp > 2 || (D.dim == 2 && comp != 2)) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_get_vorticity: component %d", comp);
    if (ns->nranks > 1) return fail(PIB_ERR_SUP, "pib_ns_get_vorticity: post-processing runs on one rank");
    for (int d = 0; d < 3; ++d) n_out[d] = (d < D.dim) ? ((d == comp) ? D.pn[d] : D.pn[d] + 1) : 1;
    if (out == nullptr) return 0;
    PIB_HIP(hipSetDevice(ns->device));
    const int64_t total = n_out[0] * n_out[1] * n_out[2];
    double *d_out = nullptr;
    PIB_HIP(hipMalloc(&d_out, sizeof(double) * (size_t)total));
    hipLaunchKernelGGL(k_ns_vorticity, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, ns->stream, D, comp,
                       n_out[0], n_out[1], n_out[2], ns->U, d_out);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipStreamSynchronize(ns->stream));
    PIB_HIP(hipMemcpy(out, d_out, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost));
    PIB_HIP(hipFree(d_out));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_sizes(pib_ns *ns, int64_t *UN, int64_t *pN)
try {
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    // on a slab: this rank's part of the distributed vectors (packed owned [u | v | w]; owned pressure cells)
    if (UN) *UN = (ns->nranks > 1) ? ns->UN_owned : ns->D.UN;
    if (pN) *pN = (ns->nranks > 1) ? ns->pN_owned : ns->D.pN;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_set_state(pib_ns *ns, const double *U, const double *p)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    const bool slab = ns->nranks > 1;
    if (U) {
        if (slab) {
            PIB_HIP(hipMemcpy(ns->Upk, U, sizeof(double) * (size_t)ns->UN_owned, hipMemcpyHostToDevice));
            PIB_CHK(ns_unpack(ns, ns->Upk, ns->U));
            PIB_CHK(ns_halo_velocity(ns, ns->U));
        } else
            PIB_HIP(hipMemcpy(ns->U, U, sizeof(double) * (size_t)ns->D.UN, hipMemcpyHostToDevice));
        // initial data are followed by bc->setGhostICs(solution) (navierstokes.cpp:139-142, restart :743)
        hipLaunchKernelGGL(k_ns_ghosts<0>, dim3(ghost_blocks(ns->D)), dim3(256), 0, ns->stream, ns->D, 0.0, ns->U);
        PIB_HIP(hipGetLastError());
        PIB_HIP(hipStreamSynchronize(ns->stream));
    }
    if (p) {
        if (slab) {
            PIB_HIP(hipMemcpy(ns->p + (ns->slab_pk0 - ns->slab_e0) * ns->p_plane, p, sizeof(double) * (size_t)ns->pN_owned, hipMemcpyHostToDevice));
            PIB_CHK(ns_halo_cells(ns, ns->p));
            PIB_HIP(hipStreamSynchronize(ns->stream));
        } else
            PIB_HIP(hipMemcpy(ns->p, p, sizeof(double) * (size_t)ns->D.pN, hipMemcpyHostToDevice));
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_get_state(pib_ns *ns, double *U, double *p, double *rhs1, double *rhs2)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamSynchronize(ns->stream));
    if (ns->nranks > 1) {  // this rank's owned points
        const int64_t own = (ns->slab_pk0 - ns->slab_e0) * ns->p_plane;
        const size_t ub = sizeof(double) * (size_t)ns->UN_owned, pb = sizeof(double) * (size_t)ns->pN_owned;
        if (U) {
            PIB_CHK(ns_pack(ns, ns->U, ns->Upk));
            PIB_HIP(hipStreamSynchronize(ns->stream));
            PIB_HIP(hipMemcpy(U, ns->Upk, ub, hipMemcpyDeviceToHost));
        }
        if (rhs1) {
            PIB_CHK(ns_pack(ns, ns->rhs1, ns->rhs1pk));
            PIB_HIP(hipStreamSynchronize(ns->stream));
            PIB_HIP(hipMemcpy(rhs1, ns->rhs1pk, ub, hipMemcpyDeviceToHost));
        }
        if (p) PIB_HIP(hipMemcpy(p, ns->p + own, pb, hipMemcpyDeviceToHost));
        if (rhs2) PIB_HIP(hipMemcpy(rhs2, ns->rhs2 + own, pb, hipMemcpyDeviceToHost));
        return 0;
    }
    if (U) PIB_HIP(hipMemcpy(U, ns->U, sizeof(double) * (size_t)ns->D.UN, hipMemcpyDeviceToHost));
    if (p) PIB_HIP(hipMemcpy(p, ns->p, sizeof(double) * (size_t)ns->D.pN, hipMemcpyDeviceToHost));
    if (rhs1) PIB_HIP(hipMemcpy(rhs1, ns->rhs1, sizeof(double) * (size_t)ns->D.UN, hipMemcpyDeviceToHost));
    if (rhs2) PIB_HIP(hipMemcpy(rhs2, ns->rhs2, sizeof(double) * (size_t)ns->D.pN, hipMemcpyDeviceToHost));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* the explicit terms a restart needs (navierstokes.cpp:637-686: /convection/0, /convection/1, /diffusion/0); any may be NULL */
int pib_ns_get_history(pib_ns *ns, double *conv0, double *conv1, double *diff0)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamSynchronize(ns->stream));
    if (ns->nranks > 1) {
        if (conv0 && ns->T.nconv > 0) PIB_CHK(pib_ns_history_term(ns, 0, 0, 0, conv0));
        if (conv1 && ns->T.nconv > 1) PIB_CHK(pib_ns_history_term(ns, 0, 1, 0, conv1));
        if (diff0 && ns->T.ndiff > 0) PIB_CHK(pib_ns_history_term(ns, 1, 0, 0, diff0));
        return 0;
    }
    const size_t bytes = sizeof(double) * (size_t)ns->D.UN;
    if (conv0) PIB_HIP(hipMemcpy(conv0, ns->conv[0], bytes, hipMemcpyDeviceToHost));
    if (conv1) PIB_HIP(hipMemcpy(conv1, ns->conv[1], bytes, hipMemcpyDeviceToHost));
    if (diff0) PIB_HIP(hipMemcpy(diff0, ns->diff0, bytes, hipMemcpyDeviceToHost));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_set_history(pib_ns *ns, const double *conv0, const double *conv1)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    if (ns->nranks > 1) {
        if (conv0 && ns->T.nconv > 0) PIB_CHK(pib_ns_history_term(ns, 0, 0, 1, const_cast<double *>(conv0)));
        if (conv1 && ns->T.nconv > 1) PIB_CHK(pib_ns_history_term(ns, 0, 1, 1, const_cast<double *>(conv1)));
        return 0;
    }
    const size_t bytes = sizeof(double) * (size_t)ns->D.UN;
    if (conv0) PIB_HIP(hipMemcpy(ns->conv[0], conv0, bytes, hipMemcpyHostToDevice));
    if (conv1) PIB_HIP(hipMemcpy(ns->conv[1], conv1, bytes, hipMemcpyHostToDevice));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_advance(pib_ns *ns, int nsteps)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    NsDev &D = ns->D;
    const int gg = ghost_blocks(D);
    const int gu = (int)std::min<int64_t>(4096, (D.UN + 255) / 256), gp = (int)std::min<int64_t>(4096, (D.pN + 255) / 256);
    const int gt = gu > gp ? gu : gp;
    // stage boundaries (pib_ns_stage_timers): an event on the engine's stream; a solve runs on its solver's stream and returns
    // behind its own host synchronisation, so the event recorded after it is stamped when it has finished
    auto mark = [&](int k) -> int {
        if (ns->stage_timing) PIB_HIP(hipEventRecord(ns->ev_stage[k], ns->stream));
        return 0;
    };
    auto close_step = [&]() -> int {
        if (!ns->stage_timing) return 0;
        PIB_CHK(mark(6));  // end of update
        PIB_HIP(hipEventSynchronize(ns->ev_stage[6]));
        for (int k = 0; k < 6; ++k) {
            float ms = 0.0f;
            PIB_HIP(hipEventElapsedTime(&ms, ns->ev_stage[k], ns->ev_stage[k + 1]));
            ns->stage_ms[k] += (double)ms;
        }
        ns->stage_steps++;
        return 0;
    };
    for (int it = 0; it < nsteps; ++it) {
        PIB_CHK(mark(0));
        // VecSwap chain of the convective terms (navierstokes.cpp:452-458)
        if (ns->T.nconv > 1) std::swap(ns->conv[0], ns->conv[1]);
        if (ns->T.ndiff > 1) std::swap(ns->diff0, ns->diff1);
        // bc->updateEqs(solution, dt) (:508) into a1n; the right-hand side needs both generations
        hipLaunchKernelGGL(k_ns_ghosts<1>, dim3(gg), dim3(256), 0, ns->stream, D, ns->dt, ns->U);
        // interior kernel of the explicit terms: rows dealt to the XCDs in bands (a short component: row by row, the 3-D grid)
        constexpr int rhs_bands = 1;
        // (planes a workgroup walks: 1 / 2 / 4 / 8 measured 1.09 / 1.02 / 1.03 / 1.06 ms of rhsVelocity per 256^3 step)
        constexpr int rhs_kz = 2;
        // 3-D, every component with an interior: the three interiors in ONE z-march over tiles of cells (k_ns_rhs_march: U and p
        // from HBM once per step instead of three times); the shells as before.  With a one-term diffusive scheme (Crank-Nicolson,
        // explicit Euler) nothing reads the stored diffusive term but the restart files (navierstokes.cpp:672-680), which are
        // written between two calls of advance: it is stored in the last step of a call only.
        bool all_inner = D.dim == 3;
        for (int f = 0; f < 3 && all_inner; ++f) all_inner = D.f[f].n[0] >= 3 && D.f[f].n[1] >= 3 && D.f[f].n[2] >= 3;
        const bool march = all_inner && D.pn[0] >= 32 && D.pn[1] >= 8 && D.pn[2] >= 4;
        if (march) {
            const int ntx = (int)((D.pn[0] + RHS_TX - 1) / RHS_TX), nty = (int)((D.pn[1] + RHS_TY - 1) / RHS_TY), band = (nty + 7) / 8;
            const int KZ = 32, nzc = (int)((D.pn[2] - 2 + KZ - 1) / KZ);
            const dim3 grid_((unsigned)(8 * ntx * band * nzc));
            const bool keep_diff = ns->T.ndiff > 1 || it == nsteps - 1;
            if (keep_diff)
                hipLaunchKernelGGL(k_ns_rhs_march<true>, grid_, dim3(RHS_TX * RHS_TY), 0, ns->stream, D, ns->dt, ns->nu, ns->T, ns->U, ns->p,
                                   ns->conv[1], ns->conv[0], ns->rhs1, ns->diff0, ns->diff1, ntx, nty, band, KZ);
            else
                hipLaunchKernelGGL(k_ns_rhs_march<false>, grid_, dim3(RHS_TX * RHS_TY), 0, ns->stream, D, ns->dt, ns->nu, ns->T, ns->U, ns->p,
                                   ns->conv[1], ns->conv[0], ns->rhs1, ns->diff0, ns->diff1, ntx, nty, band, KZ);
        }
#define PIB_RHS(DIM_, F_)                                                                                                   \
    {                                                                                                                       \
        const NsField &Fq = D.f[F_];                                                                                        \
        const bool inner = Fq.n[0] >= 3 && Fq.n[1] >= 3 && (DIM_ == 2 || Fq.n[2] >= 3);                                     \
        if (inner && !march) {                                                                                              \
            const int gx_ = (int)((Fq.n[0] - 2 + 255) / 256);                                                               \
            const int band_ = (DIM_ == 3 && rhs_bands && Fq.n[1] - 2 >= 64) ? (int)((Fq.n[1] - 2 + 7) / 8) : 0;              \
            const dim3 grid_ = band_ ? dim3((unsigned)(gx_ * 8 * band_ * ((Fq.n[2] - 2 + rhs_kz - 1) / rhs_kz)))            \
                                     : dim3((unsigned)gx_, (unsigned)(Fq.n[1] - 2), (unsigned)(DIM_ == 3 ? Fq.n[2] - 2 : 1)); \
            hipLaunchKernelGGL((k_ns_rhs_velocity<DIM_, F_>), grid_, dim3(256), 0, ns->stream, D, ns->dt, ns->nu, ns->T,     \
                               ns->U, ns->p, ns->conv[1], ns->conv[0], ns->rhs1, ns->diff0, ns->diff1, gx_, band_, rhs_kz);  \
        }                                                                                                                   \
        const int64_t shell = inner ? 2 * (Fq.n[1] * Fq.n[2] + (Fq.n[0] - 2) * Fq.n[2] +                                    \
                                           (DIM_ == 3 ? (Fq.n[0] - 2) * (Fq.n[1] - 2) : 0))                                 \
                                    : Fq.n[0] * Fq.n[1] * Fq.n[2];                                                          \
        hipLaunchKernelGGL(k_ns_rhs_velocity_shell, dim3((unsigned)std::min<int64_t>(4096, (shell + 255) / 256)), dim3(256), 0, \
                           ns->stream, D, F_, inner ? 0 : 1, ns->dt, ns->nu, ns->T, ns->U, ns->p, ns->conv[1],                \
                           ns->conv[0], ns->rhs1, ns->diff0, ns->diff1);                                                    \
    }
        if (D.dim == 2) {
            PIB_RHS(2, 0)
            PIB_RHS(2, 1)
        } else {
            PIB_RHS(3, 0)
            PIB_RHS(3, 1)
            PIB_RHS(3, 2)
        }
#undef PIB_RHS
        PIB_HIP(hipGetLastError());
        std::swap(D.a1, D.a1n);
        if (ns->ib) PIB_CHK(ib_spread_forces(ns));  // rhs1 += H f  (decoupledibpm.cpp:243)
        PIB_CHK(mark(1));  // end of rhsVelocity
        if (ns->nranks > 1) {
            // the solver's vectors are the packed owned points; afterwards the neighbours' planes of u* (DMGlobalToLocal)
            PIB_CHK(ns_pack(ns, ns->rhs1, ns->rhs1pk));
            PIB_CHK(ns_pack(ns, ns->U, ns->Upk));
            PIB_CHK(ns_before_solve(ns, ns->vsol));
            PIB_CHK(pib_solve(ns->vsol, ns->Upk, ns->rhs1pk));
            PIB_CHK(ns_unpack(ns, ns->Upk, ns->U));
            PIB_CHK(ns_halo_velocity(ns, ns->U));
        } else {
        PIB_CHK(ns_before_solve(ns, ns->vsol));
        PIB_CHK(pib_solve(ns->vsol, ns->U, ns->rhs1));  // vSolver->solve(UGlobal, rhs1)  (:532)
        }
        PIB_CHK(mark(2));  // end of solveVelocity
        const bool coupled = ib_is_coupled(ns);
        if (ns->ib && !coupled) {
            PIB_CHK(ib_solve_forces(ns));   // assembleRHSForces, solveForces, applyNoSlip (decoupledibpm.cpp:116-118)
            PIB_CHK(ns_halo_velocity(ns, ns->U));  // u += BNH df changed owned points next to the neighbours' planes
        }
        PIB_CHK(mark(3));  // end of rhsForces + solveForces (nothing between the two marks without bodies)
        {
            const dim3 pg((unsigned)((D.pn[0] + 255) / 256), (unsigned)D.pn[1], (unsigned)(D.dim == 3 ? D.pn[2] : 1));
            if (D.pn[1] <= 65535 && (D.dim == 2 || D.pn[2] <= 65535))
                hipLaunchKernelGGL(k_ns_rhs_poisson_rows, pg, dim3(256), 0, ns->stream, D, ns_pin_cell(ns), ns->U, ns->rhs2);
            else
                hipLaunchKernelGGL(k_ns_rhs_poisson, dim3(gp), dim3(256), 0, ns->stream, D, ns_pin_cell(ns), ns->U, ns->rhs2);
        }
        PIB_HIP(hipGetLastError());
        PIB_CHK(mark(4));  // end of rhsPoisson
        if (coupled) {
            // IBPMSolver (applications/ibpm): pressure and forces are one unknown; solved here through the Schur
            // complement on the pressure, then u = u* - BN [G, -H] [dP; df], p += dP, f += df
            PIB_CHK(ib_coupled_solve_and_project(ns));
            PIB_CHK(mark(5));  // (the coupled pressure / forces solve and its projection count as solvePoisson)
            hipLaunchKernelGGL(k_ns_ghosts<2>, dim3(gg), dim3(256), 0, ns->stream, D, ns->dt, ns->U);
            PIB_HIP(hipGetLastError());
            ns->steps++;
            PIB_CHK(close_step());
            continue;
        }
        PIB_CHK(ns_before_solve(ns, ns->psol));
        {
            const int64_t own = (ns->slab_pk0 - ns->slab_e0) * ns->p_plane;  // the owned cells are contiguous in the extended slab
            PIB_CHK(pib_solve(ns->psol, ns->dP + own, ns->rhs2 + own));  // pSolver->solve(dP, rhs2)      (:575)
            PIB_CHK(ns_halo_cells(ns, ns->dP));  // G dP at the faces towards the neighbours
        }
        PIB_CHK(mark(5));  // end of solvePoisson
        if (ns->bn_order > 1 && ns->nranks > 1)
            PIB_CHK(ns_project_bn_slab(ns));
        else if (ns->bn_order > 1)
            hipLaunchKernelGGL(k_ns_project_csr, dim3(gt), dim3(256), 0, ns->stream, D.UN, D.pN, ns->bng_rowptr, ns->bng_col,
                               ns->bng_val, ns->dP, ns->U, ns->p);
        else
        {
            const dim3 pg((unsigned)((D.pn[0] + 255) / 256), (unsigned)D.pn[1], (unsigned)(D.dim == 3 ? D.pn[2] : 1));
            if (D.pn[1] <= 65535 && (D.dim == 2 || D.pn[2] <= 65535)) {
                if (D.dim == 3) hipLaunchKernelGGL(k_ns_project_rows<3>, pg, dim3(256), 0, ns->stream, D, ns->dt, ns->dP, ns->U, ns->p);
                else hipLaunchKernelGGL(k_ns_project_rows<2>, pg, dim3(256), 0, ns->stream, D, ns->dt, ns->dP, ns->U, ns->p);
            } else
            hipLaunchKernelGGL(k_ns_project, dim3(gt), dim3(256), 0, ns->stream, D, ns->dt, ns->dP, ns->U, ns->p);
        }
        if (ns->ib) PIB_CHK(ib_update_forces(ns));  // f += df  (decoupledibpm.cpp:125)
        PIB_CHK(ns_halo_velocity(ns, ns->U));       // the projected velocity on the neighbours' planes
        hipLaunchKernelGGL(k_ns_ghosts<2>, dim3(gg), dim3(256), 0, ns->stream, D, ns->dt, ns->U);  // bc->updateGhostValues (:263)
        PIB_HIP(hipGetLastError());
        ns->steps++;
        PIB_CHK(close_step());
    }
    PIB_HIP(hipStreamSynchronize(ns->stream));
    pib_get_iters(ns->vsol, &ns->v_iters);
    pib_get_residual(ns->vsol, &ns->v_res);
    pib_get_iters(ns->psol, &ns->p_iters);
    pib_get_residual(ns->psol, &ns->p_res);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* Stage timers under the reference's PetscLogStage names (navierstokes.cpp:186-199; decoupledibpm.cpp:93-97): HIP events on
 * the engine's stream around the stages of NavierStokesSolver::advance / DecoupledIBPMSolver::advance. */
static const char *const kStageNames[6] = {"rhsVelocity", "solveVelocity", "solveForces", "rhsPoisson", "solvePoisson", "update"};
const char *pib_ns_stage_name(int stage) { return (stage >= 0 && stage < 6) ? kStageNames[stage] : ""; }
int pib_ns_stage_timers(pib_ns *ns, int enable)
try {
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    if (enable)
        for (int k = 0; k < 7; ++k)
            if (ns->ev_stage[k] == nullptr) PIB_HIP(hipEventCreate(&ns->ev_stage[k]));
    ns->stage_timing = enable != 0;
    for (int k = 0; k < 6; ++k) ns->stage_ms[k] = 0.0;
    ns->stage_steps = 0;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
int pib_ns_get_stage_times(pib_ns *ns, double ms[6], int64_t *steps)
try {
    if (ns == nullptr || ms == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "pib_ns_get_stage_times: null argument");
    for (int k = 0; k < 6; ++k) ms[k] = ns->stage_ms[k];
    if (steps) *steps = ns->stage_steps;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* the columns of iterations-<start>.txt (navierstokes.cpp:766-794): vIters, vRes, pIters, pRes of the last step */
int pib_ns_get_solver_info(pib_ns *ns, int *v_iters, double *v_res, int *p_iters, double *p_res)
try {
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    if (v_iters) *v_iters = ns->v_iters;
    if (v_res) *v_res = ns->v_res;
    if (p_iters) *p_iters = ns->p_iters;
    if (p_res) *p_res = ns->p_res;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* NavierStokesSolver::writeLinSolversInfo (navierstokes.cpp:780-787 prints both solvers' banners): what each solver of the engine
 * RUNS -- pib_describe of the velocity (which = 0) or Poisson (1) solver, departures from their files included */
int pib_ns_describe_solver(pib_ns *ns, int which, char *buf, int buflen)
try {
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    pib_solver *s = which == 0 ? ns->vsol : (which == 1 ? ns->psol : nullptr);
    if (s == nullptr) return pib::fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_describe_solver: no solver %d in this engine", which);
    return pib_describe(s, buf, buflen);
} catch (...) {
    return pib::fail_exception(__func__);
}

}  // extern "C"
