// structure.cpp -- recover the mesh structure of a Poisson matrix handed over as a plain CSR.
//
// The reference's `type: GPU` path gives the solver nothing but the assembled matrix: LinSolverAmgX::setMatrix ->
// AmgXSolver::setA(A) (src/linsolver/linsolveramgx.cpp:84).  AmgX builds its algebraic hierarchy from those entries;
// this backend's multigrid is geometric and wants the 1-D width arrays of the mesh.  For PetIBM's Poisson operator they
// can be read back from the entries: DBNG = D (dt I) G (applications/navierstokes/navierstokes.cpp:349-356) in natural
// ordering has, toward +x of cell (i,j,k), the entry
//        dt dy_j dz_k / (0.5 (dx_i + dx_{i+1}))  =  gx_i wy_j wz_k          (createdivergence.cpp:140-151,
//                                                                           creategradient.cpp:70-86, createbn.cpp:49)
// i.e. a product of three 1-D factors, and likewise toward +y and +z.  Lines of entries through one base cell give the
// factors up to one scale per direction; the three conditions g_d[0] * 0.5 (w_d[0] + w_d[1]) = dt (the same dt in every
// direction) fix the scales up to ONE common length unit, which the operator -- on every multigrid level, because the
// coarse operators are rediscretised from the widths -- does not depend on.  A periodic direction shows as the wrapped
// neighbour of its first cell (one more offset in the first rows) and contributes one more face, the one across the seam.
//
// The recovered arrays go through grid_register, which verifies the matrix-free twin against the CSR on the device
// (1e-10 relative): a matrix that is not such an operator (velocity system, BN order > 1, arbitrary CSR)
// simply stays without grid structure.  Several ranks: z-slabs (y-slabs in 2-D); the in-plane lines come from the rank
// that owns slab plane 1, the lines along the slab axis from every rank's planes, all-gathered as RAW matrix entries so
// that every rank does the same arithmetic on the same numbers (the level hierarchy must come out identical everywhere).
#include <algorithm>
#include <cmath>
#include <set>

#include "pib_internal.hpp"

namespace pib {

namespace {
struct HostCsr {
    int64_t n_local, row0, n_global;
    const int64_t *rp64, *cl64;
    const int32_t *rp32, *cl32;
    const double *val;
    int64_t RP(int64_t i) const { return rp64 ? rp64[i] : (int64_t)rp32[i]; }
    int64_t CL(int64_t p) const { return cl64 ? cl64[p] : (int64_t)cl32[p]; }
    // entry (global row, global column) of a LOCAL row; false if the row is not local or the entry is not stored
    bool entry(int64_t row, int64_t col, double *v) const
    {
        const int64_t l = row - row0;
        if (l < 0 || l >= n_local) return false;
        for (int64_t p = RP(l); p < RP(l + 1); ++p)
            if (CL(p) == col) {
                *v = val[p];
                return true;
            }
        return false;
    }
};
}  // namespace

int comm_allgather_host(pib_solver *s, const std::vector<double> &mine, std::vector<double> &all);

// The LOCAL rows detect_grid_structure reads of a rank's natural z-slab (y-slab) [k0, k1) of an n[0] x n[1] x n[2] grid (2-D:
// n[1] = 1, slab axis = n[2]): the first eight rows, the two in-plane lines through base index 1 on slab plane 1 (its owner only)
// and the line along the slab axis through the in-plane base cell.  A caller that holds the rows on the device (the box route,
// redistribute.hip) downloads only these -- a thousand rows instead of the slab's 1.4 GB at 512^3 / 8; every `take` and every
// RP / CL access below must stay inside this set (a row outside it reads as "entry not stored": no structure, never a wrong one).
void grid_detection_rows(const int64_t n[3], int64_t k0, int64_t k1, std::vector<int64_t> &local_rows)
{
    const bool three = n[1] > 1;
    const int64_t pl = n[0] * n[1], n_local = (k1 - k0) * pl;
    std::set<int64_t> rows;
    for (int64_t l = 0; l < std::min<int64_t>(n_local, 8); ++l) rows.insert(l);
    auto add = [&](int64_t global) {
        const int64_t l = global - k0 * pl;
        if (l >= 0 && l < n_local) rows.insert(l);
    };
    if (k0 <= 1 && 1 < k1) {
        if (three) {
            for (int64_t i = 0; i < n[0]; ++i) add(pl + n[0] + i);
            for (int64_t j = 0; j < n[1]; ++j) add(pl + j * n[0] + 1);
        } else
            for (int64_t i = 0; i < n[0]; ++i) add(pl + i);
    }
    for (int64_t k = k0; k < k1; ++k) add(k * pl + (three ? n[0] + 1 : 1));
    local_rows.assign(rows.begin(), rows.end());
}

// returns 0 always unless a collective / HIP call fails; success shows in s->has_grid
int detect_grid_structure(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64,
                          const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val)
{
    const HostCsr A{n_local, row0, n_global, rp64, cl64, rp32, cl32, val};
    const int P = s->comm.nranks, rank = s->comm.rank;
    // every rank walks through the same collectives whatever it finds locally: `ok` only gates the local work
    bool ok = n_local > 0;
    // ---- the distinct |column - row| of the first rows: {1, nx[, nx ny]}, and for a periodic direction the wrapped
    // neighbour of its first cell: nx - 1 (next to nx: the only two consecutive values), nx (ny - 1), nx ny (nz - 1)
    int dim = 0;
    int64_t n[3] = {1, 1, 1};
    bool per[3] = {false, false, false};
    if (ok) {
        std::set<int64_t> offs;
        for (int64_t l = 0; l < std::min<int64_t>(n_local, 8); ++l)
            for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p) {
                const int64_t d = std::llabs(A.CL(p) - (row0 + l));
                if (d > 0) offs.insert(d);
            }
        const std::vector<int64_t> S(offs.begin(), offs.end());
        ok = S.size() >= 2 && S.size() <= 6 && S[0] == 1;
        size_t q = 2;
        if (ok) {
            if (S.size() > 2 && S[2] == S[1] + 1) {
                per[0] = true;
                n[0] = S[2];
                q = 3;
            } else
                n[0] = S[1];
            ok = n[0] >= 3 && n_global % n[0] == 0;
        }
        if (ok) {
            const int64_t rows = n_global / n[0];  // ny, or ny nz
            std::vector<int64_t> m;                // the larger offsets in grid lines: {ny - 1 (y periodic), ny (3-D), ny (nz - 1) (z periodic)}
            for (size_t t = q; t < S.size() && ok; ++t) {
                ok = S[t] % n[0] == 0;
                m.push_back(S[t] / n[0]);
            }
            auto has = [&](int64_t v) { return std::find(m.begin(), m.end(), v) != m.end(); };
            if (!ok) {
            } else if (m.empty() || (m.size() == 1 && m[0] == rows - 1)) {
                dim = 2;
                n[1] = rows;
                per[1] = !m.empty();
            } else {
                dim = 3;
                ok = false;
                for (int64_t y : m) {
                    if (y < 3 || rows % y != 0 || rows / y < 3) continue;
                    const int64_t z = rows / y;
                    bool fits = true;
                    for (int64_t v : m) fits = fits && (v == y || v == y - 1 || v == y * (z - 1));
                    if (!fits) continue;
                    n[1] = y;
                    n[2] = z;
                    per[1] = has(y - 1);
                    per[2] = has(y * (z - 1));
                    ok = true;
                    break;
                }
            }
        }
        for (int d = 0; d < dim; ++d) ok = ok && n[d] >= 3;
        // directions the caller declared periodic (pib_set_periodic) must be the ones found
        for (int d = 0; d < 3 && ok; ++d)
            if (s->periodic_user[d] && !per[d] && !(d == dim - 1 && P > 1)) ok = false;
    }
    const int sd = dim - 1;                               // slab axis
    const int64_t st[3] = {1, n[0], n[0] * n[1]};         // strides
    const int64_t pl = ok ? st[sd] : 1;                   // cells per slab plane
    int64_t k0 = 0, k1 = 0;
    if (ok) {
        slab_range(n[sd], P, rank, &k0, &k1);
        ok = k0 * pl == row0 && (k1 - k0) * pl == n_local;  // rows = the DMDA z-slab (y-slab) of this rank
    }
    // ---- raw lines of entries.  Buffer layout per rank (fixed length so one all-gather serves):
    //   [0] status: -1 an expected entry is missing, 1 fine, 2 fine and row 0 is the identity (rank 0)
    //   in-plane lines (owner of slab plane 1): X[0..mx) A[0..mx) Y[0..mx) B[0..mx)      mx = max in-plane extent
    //   slab-axis lines (own planes):           Cs[0..mp) Zs[0..mp)                       mp = max planes per rank
    // first the locally detected sizes (every rank must have found the same grid), then the lines themselves
    {
        // [5]: the in-plane periodic directions (every rank sees them); [6]: the wrap of the slab axis, which only the ranks
        // at the seam see -- it holds if any rank found it
        const double inplane = (per[0] ? 1.0 : 0.0) + ((dim == 3 && per[1]) ? 2.0 : 0.0);
        std::vector<double> head = {ok ? 1.0 : 0.0, (double)dim, (double)n[0], (double)n[1], (double)n[2], inplane,
                                    (ok && per[sd]) ? 1.0 : 0.0, 0.0};
        std::vector<double> heads;
        PIB_CHK(comm_allgather_host(s, head, heads));
        bool all_ok = true, seam = false;
        for (int r = 0; r < P; ++r) {
            const double *h = &heads[8 * (size_t)r];
            all_ok = all_ok && h[0] == 1.0 && h[1] == heads[1] && h[2] == heads[2] && h[3] == heads[3] && h[4] == heads[4] && h[5] == heads[5];
            seam = seam || h[6] == 1.0;
        }
        if (!all_ok) return 0;  // every rank sees the same heads: all leave together
        per[sd] = seam;
    }
    const int64_t mx_all = std::max(n[0], dim == 3 ? n[1] : (int64_t)0);  // longest in-plane line
    const int64_t mp_all = (n[sd] + P - 1) / P;                           // most planes on one rank
    const size_t L = 1 + 4 * (size_t)mx_all + 2 * (size_t)mp_all;
    std::vector<double> mine(L, 0.0), all;
    mine[0] = 1.0;
    double *X = &mine[1], *Ax = X + mx_all, *Y = Ax + mx_all, *B = Y + mx_all, *Cs = B + mx_all, *Zs = Cs + mp_all;
    bool found = true;
    auto take = [&](int64_t row, int64_t col, double *dst) {
        double v = 0.0;
        if (!A.entry(row, col, &v)) found = false;
        *dst = v;
    };
    if (k0 <= 1 && 1 < k1) {  // owner of slab plane 1: the in-plane lines through base index 1
        if (dim == 3) {
            const int64_t b = 1 * st[2];
            for (int64_t i = 0; i < n[0]; ++i) {
                const int64_t c = b + 1 * st[1] + i;
                if (i + 1 < n[0]) take(c, c + 1, &X[i]);   // gx_i wy_1 wz_1
                else if (per[0]) take(c, c - (n[0] - 1), &X[i]);  // the face across the periodic seam
                take(c, c + st[1], &Ax[i]);                 // wx_i gy_1 wz_1
            }
            for (int64_t j = 0; j < n[1]; ++j) {
                const int64_t c = b + j * st[1] + 1;
                if (j + 1 < n[1]) take(c, c + st[1], &Y[j]);  // wx_1 gy_j wz_1
                else if (per[1]) take(c, c - (n[1] - 1) * st[1], &Y[j]);
                take(c, c + 1, &B[j]);                        // gx_1 wy_j wz_1
            }
        } else {
            for (int64_t i = 0; i < n[0]; ++i) {
                const int64_t c = 1 * st[1] + i;
                if (i + 1 < n[0]) take(c, c + 1, &X[i]);   // gx_i wy_1
                else if (per[0]) take(c, c - (n[0] - 1), &X[i]);
                take(c, c + st[1], &Ax[i]);                 // wx_i gy_1
            }
        }
    }
    for (int64_t k = k0; k < k1; ++k) {  // the lines along the slab axis through the in-plane base cell (1[,1])
        const int64_t c = k * st[sd] + (dim == 3 ? st[1] + 1 : 1);
        take(c, c + 1, &Cs[k - k0]);                          // gx_1 wy_1 wz_k     (2-D: gx_1 wy_k)
        if (k + 1 < n[sd]) take(c, c + st[sd], &Zs[k - k0]);  // wx_1 wy_1 gz_k     (2-D: wx_1 gy_k)
        else if (per[sd]) take(c, c - (n[sd] - 1) * st[sd], &Zs[k - k0]);
    }
    if (rank == 0) {  // pinned pressure: row 0 is the identity (MatZeroRowsColumns, navierstokes.cpp:416-418)
        bool pinned = true;
        for (int64_t p = A.RP(0); p < A.RP(1); ++p) pinned = pinned && (A.CL(p) == 0 ? val[p] == 1.0 : val[p] == 0.0);
        if (pinned) mine[0] = 2.0;
    }
    if (!found) mine[0] = -1.0;
    PIB_CHK(comm_allgather_host(s, mine, all));
    for (int r = 0; r < P; ++r)
        if (all[L * (size_t)r] < 0.0) return 0;  // some expected entry is not stored: not this kind of matrix
    const int nullspace = (all[0] == 2.0) ? PIB_NULLSPACE_PINNED : PIB_NULLSPACE_CONSTANT;
    // ---- assemble the global lines (identical on every rank)
    int owner1 = 0;
    for (int r = 0; r < P; ++r) {
        int64_t b, e;
        slab_range(n[sd], P, r, &b, &e);
        if (b <= 1 && 1 < e) owner1 = r;
    }
    const double *o1 = &all[L * (size_t)owner1 + 1];
    std::vector<double> gX(o1, o1 + mx_all), gA(o1 + mx_all, o1 + 2 * mx_all), gY(o1 + 2 * mx_all, o1 + 3 * mx_all),
        gB(o1 + 3 * mx_all, o1 + 4 * mx_all), gCs((size_t)n[sd], 0.0), gZs((size_t)n[sd], 0.0);
    for (int r = 0; r < P; ++r) {
        int64_t b, e;
        slab_range(n[sd], P, r, &b, &e);
        const double *src = &all[L * (size_t)r + 1 + 4 * (size_t)mx_all];
        for (int64_t k = b; k < e; ++k) {
            gCs[(size_t)k] = src[k - b];
            gZs[(size_t)k] = src[mp_all + (k - b)];
        }
    }
    // ---- scales: wx_1 = 1; the same dt from face 0 of every direction
    std::vector<double> w[3], g[3];
    double dt = 0.0;
    const int64_t ng[3] = {per[0] ? n[0] : n[0] - 1, per[1] ? n[1] : n[1] - 1, per[2] ? n[2] : n[2] - 1};  // faces: one more across a seam
    auto positive = [](const std::vector<double> &v, int64_t cnt) {
        for (int64_t q = 0; q < cnt; ++q)
            if (!(v[(size_t)q] > 0.0) || !std::isfinite(v[(size_t)q])) return false;
        return true;
    };
    if (dim == 3) {
        if (!positive(gX, ng[0]) || !positive(gA, n[0]) || !positive(gY, ng[1]) || !positive(gB, n[1]) ||
            !positive(gCs, n[2]) || !positive(gZs, ng[2]))
            return 0;
        const double xh = gX[0] * (0.5 * (gA[0] / gA[1] + 1.0)), yh = gY[0] * (0.5 * (gB[0] / gB[1] + 1.0)),
                     zh = gZs[0] * (0.5 * (gCs[0] / gCs[1] + 1.0));
        const double beta = std::sqrt(xh / yh), gamma = std::sqrt(xh / zh);
        dt = xh / (beta * gamma);
        w[0].resize((size_t)n[0]);
        w[1].resize((size_t)n[1]);
        w[2].resize((size_t)n[2]);
        g[0].resize((size_t)ng[0]);
        g[1].resize((size_t)ng[1]);
        g[2].resize((size_t)ng[2]);
        for (int64_t i = 0; i < n[0]; ++i) w[0][(size_t)i] = gA[(size_t)i] / gA[1];
        for (int64_t j = 0; j < n[1]; ++j) w[1][(size_t)j] = beta * (gB[(size_t)j] / gB[1]);
        for (int64_t k = 0; k < n[2]; ++k) w[2][(size_t)k] = gamma * (gCs[(size_t)k] / gCs[1]);
        for (int64_t i = 0; i < ng[0]; ++i) g[0][(size_t)i] = gX[(size_t)i] / (beta * gamma);
        for (int64_t j = 0; j < ng[1]; ++j) g[1][(size_t)j] = gY[(size_t)j] / gamma;
        for (int64_t k = 0; k < ng[2]; ++k) g[2][(size_t)k] = gZs[(size_t)k] / beta;
    } else {
        if (!positive(gX, ng[0]) || !positive(gA, n[0]) || !positive(gCs, n[1]) || !positive(gZs, ng[1])) return 0;
        const double xh = gX[0] * (0.5 * (gA[0] / gA[1] + 1.0)), zh = gZs[0] * (0.5 * (gCs[0] / gCs[1] + 1.0));
        const double beta = std::sqrt(xh / zh);
        dt = xh / beta;
        w[0].resize((size_t)n[0]);
        w[1].resize((size_t)n[1]);
        g[0].resize((size_t)ng[0]);
        g[1].resize((size_t)ng[1]);
        for (int64_t i = 0; i < n[0]; ++i) w[0][(size_t)i] = gA[(size_t)i] / gA[1];
        for (int64_t k = 0; k < n[1]; ++k) w[1][(size_t)k] = beta * (gCs[(size_t)k] / gCs[1]);
        for (int64_t i = 0; i < ng[0]; ++i) g[0][(size_t)i] = gX[(size_t)i] / beta;
        for (int64_t k = 0; k < ng[1]; ++k) g[1][(size_t)k] = gZs[(size_t)k];
    }
    if (!(dt > 0.0) || !std::isfinite(dt)) return 0;
    const double *cw[3] = {w[0].data(), w[1].data(), dim == 3 ? w[2].data() : nullptr};
    const double *cg[3] = {g[0].data(), g[1].data(), dim == 3 ? g[2].data() : nullptr};
    const double one = 1.0;
    if (dim == 2) cw[2] = &one;
    // registration verifies the recovered operator against the CSR on the device; a mismatch leaves the solver without
    // grid structure (the outcome is the same on every rank: the check is a global sum)
    int was[3] = {s->periodic[0], s->periodic[1], s->periodic[2]};
    for (int d = 0; d < 3; ++d) s->periodic[d] = per[d] ? 1 : 0;  // what pib_set_periodic would have said (periodic_user stays)
    const int e = grid_register(s, dim, n, cw, cg, nullspace, dt);
    if (e != 0) {
        s->gmg_error.clear();
        s->has_grid = false;
        for (int d = 0; d < 3; ++d) s->periodic[d] = was[d];
    } else {
        s->structure_detected = true;
    }
    return 0;
}

// ---- the velocity operator A = I/dt - c nu L handed over as a plain CSR (an unchanged PetIBM: vSolver->setMatrix(A),
// navierstokes.cpp:345): recover what the matrix-free product (velstencil.hip) needs, so that the drop-in route gets the
// 16 B/row products instead of the CSR's 104.  In the packed [u | v | w] ordering the entry of a row towards -d / +d is a
// function of the point's index along d only (createlaplacian.cpp:134-148: 1 / (dLNeg dLSelf), times MatScale), so one
// line of entries per field and direction IS the table; the diagonal is shift - (sum of the six table values) where a
// missing neighbour (a wall) contributes an effective value read off the diagonal of one boundary point (it carries the
// ghost fold a0 with it, whatever the boundary type).  scale = 1, a0 = 0 in the recovered description; the product is
// verified against the CSR SpMV on the device before it is used (1e-12), so anything that is not such an operator keeps
// its CSR products.  One rank here; rows in packed z-slabs on several ranks: detect_velocity_structure_slabs below.
// sizes and periodic directions of the velocity system from the distinct |column - row| of u's first rows and the total
// row count (`slab`: rows on several ranks -- only the in-range offsets were collected, the slab axis must not be periodic)
static bool parse_velocity_sizes(const std::vector<int64_t> &S, int64_t n_global, bool slab, int *dim_out, int64_t n[3], bool per[3])
{
    if (S.size() < 2 || S.size() > 6 || S[0] != 1) return false;
    per[0] = per[1] = per[2] = false;
    int64_t nxu = 0;
    size_t q = 2;
    if (S.size() > 2 && S[2] == S[1] + 1) {
        per[0] = true;
        nxu = S[2];
        q = 3;
    } else
        nxu = S[1];
    if (nxu < 3) return false;
    std::vector<int64_t> m;
    for (size_t t = q; t < S.size(); ++t) {
        if (S[t] % nxu != 0) return false;
        m.push_back(S[t] / nxu);
    }
    auto has = [&](int64_t v) { return std::find(m.begin(), m.end(), v) != m.end(); };
    n[0] = per[0] ? nxu : nxu + 1;
    n[1] = n[2] = 1;
    int dim = 0;
    // candidates for ny (u has ny points along y): 2-D when u's block is all there is besides v's
    for (int pass = 0; pass < 2 && dim == 0; ++pass) {
        if (pass == 0) {  // 2-D: offsets {ny - 1 (y periodic)} only; n_global = nxu ny + nx nyv
            for (int py = 0; py < 2 && dim == 0; ++py) {
                if (slab && py) continue;  // 2-D slabs run along y
                // n_global = nxu ny + nx (ny - (py ? 0 : 1))
                const int64_t num = n_global + (py ? 0 : n[0]), den = nxu + n[0];
                if (num % den != 0) continue;
                const int64_t ny = num / den;
                if (ny < 3) continue;
                if (m.empty() ? py == 0 : (m.size() == 1 && m[0] == ny - 1 && py == 1)) {
                    dim = 2;
                    n[1] = ny;
                    per[1] = py != 0;
                }
            }
        } else {
            for (int64_t y : m) {
                if (y < 3) continue;
                for (int py = 0; py < 2 && dim == 0; ++py)
                    for (int pz = 0; pz < 2 && dim == 0; ++pz) {
                        if (slab && pz) continue;
                        // n_global = nz [nxu y + nx (y - !py)] + nx y (nz - !pz)
                        const int64_t a = nxu * y + n[0] * (y - (py ? 0 : 1)) + n[0] * y;
                        const int64_t num = n_global + (pz ? 0 : n[0] * y);
                        if (num % a != 0) continue;
                        const int64_t z = num / a;
                        if (z < 3) continue;
                        bool fits = has(y);
                        for (int64_t v : m) fits = fits && (v == y || (py && v == y - 1) || (pz && v == y * (z - 1)));
                        fits = fits && (py == (has(y - 1) ? 1 : 0)) && (pz == (has(y * (z - 1)) ? 1 : 0));
                        if (!fits) continue;
                        dim = 3;
                        n[1] = y;
                        n[2] = z;
                        per[1] = py != 0;
                        per[2] = pz != 0;
                    }
            }
        }
    }
    *dim_out = dim;
    return dim != 0;
}

static int detect_velocity_structure_slabs(pib_solver *s, const HostCsr &A);

int detect_velocity_structure(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64,
                              const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val)
{
    const HostCsr A{n_local, row0, n_global, rp64, cl64, rp32, cl32, val};
    if (s->comm.nranks > 1) return detect_velocity_structure_slabs(s, A);
    if (n_local != n_global || row0 != 0 || n_local < 27) return 0;
    // ---- field u: the offsets of its first rows
    std::set<int64_t> offs;
    for (int64_t l = 0; l < std::min<int64_t>(n_local, 8); ++l)
        for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p) {
            const int64_t d = std::llabs(A.CL(p) - l);
            if (d > 0) offs.insert(d);
        }
    bool per[3] = {false, false, false};
    int64_t n[3] = {1, 1, 1};
    int dim = 0;
    if (!parse_velocity_sizes(std::vector<int64_t>(offs.begin(), offs.end()), n_global, false, &dim, n, per)) return 0;
    const int64_t nxu = per[0] ? n[0] : n[0] - 1;
    (void)nxu;
    // ---- field sizes and offsets
    int64_t fn[3][3], off[3] = {0, 0, 0};
    int64_t total = 0;
    for (int f = 0; f < 3; ++f)
        for (int d = 0; d < 3; ++d) fn[f][d] = 1;
    for (int f = 0; f < dim; ++f) {
        off[f] = total;
        int64_t cnt = 1;
        for (int d = 0; d < dim; ++d) {
            fn[f][d] = n[d] - ((d == f && !per[d]) ? 1 : 0);
            if (fn[f][d] < 3) return 0;
            cnt *= fn[f][d];
        }
        total += cnt;
    }
    if (total != n_global) return 0;
    // ---- shift and tables
    VelStencil V;
    V.dim = dim;
    V.per = (per[0] ? 1 : 0) | (per[1] ? 2 : 0) | (per[2] ? 4 : 0);
    V.scale = 1.0;
    bool found = true;
    auto get = [&](int64_t row, int64_t col) {
        double v = 0.0;
        if (!A.entry(row, col, &v)) found = false;
        return v;
    };
    std::vector<double> tn[3][3], tp[3][3];
    double shift = 0.0;
    for (int f = 0; f < dim && found; ++f) {
        const int64_t st[3] = {1, fn[f][0], fn[f][0] * fn[f][1]};
        auto point = [&](int d, int64_t sidx) {  // index 1 in the other directions
            int64_t p = off[f];
            for (int e = 0; e < dim; ++e) p += st[e] * (e == d ? sidx : 1);
            return p;
        };
        // neighbour column of point p (index sidx along d), or -1 at a wall
        auto nbr = [&](int64_t p, int d, int64_t sidx, int side) -> int64_t {
            const int64_t nd = fn[f][d];
            if (side < 0) return sidx > 0 ? p - st[d] : (per[d] ? p + (nd - 1) * st[d] : -1);
            return sidx < nd - 1 ? p + st[d] : (per[d] ? p - (nd - 1) * st[d] : -1);
        };
        if (f == 0) {
            const int64_t b = point(0, 1);
            double sum = 0.0;
            for (int d = 0; d < dim; ++d) sum += get(b, nbr(b, d, 1, -1)) + get(b, nbr(b, d, 1, +1));
            shift = get(b, b) + sum;  // diag = shift - (sum of the six table values)
        }
        for (int d = 0; d < dim && found; ++d) {
            const int64_t nd = fn[f][d];
            tn[f][d].assign((size_t)nd, 0.0);
            tp[f][d].assign((size_t)nd, 0.0);
            for (int64_t sidx = 0; sidx < nd; ++sidx) {
                const int64_t p = point(d, sidx);
                const int64_t cm = nbr(p, d, sidx, -1), cp = nbr(p, d, sidx, +1);
                if (cm >= 0) tn[f][d][(size_t)sidx] = get(p, cm);
                if (cp >= 0) tp[f][d][(size_t)sidx] = get(p, cp);
            }
            // walls: the effective value of the missing neighbour from the diagonal of the boundary point of this line
            // (the other directions' neighbours of that point exist: it sits at index 1 there)
            for (int side = -1; side <= 1 && !per[d]; side += 2) {
                const int64_t sidx = side < 0 ? 0 : nd - 1, p = point(d, sidx);
                double present = 0.0;
                for (int e = 0; e < dim; ++e)
                    for (int sd2 = -1; sd2 <= 1; sd2 += 2) {
                        const int64_t c = nbr(p, e, e == d ? sidx : 1, sd2);
                        if (c >= 0) present += get(p, c);
                    }
                const double eff = (shift - get(p, p)) - present;
                (side < 0 ? tn : tp)[f][d][(size_t)sidx] = eff;
            }
        }
    }
    if (!found || !std::isfinite(shift)) return 0;
    V.shift = shift;
    for (int f = 0; f < dim; ++f) {
        V.off[f] = off[f];
        for (int d = 0; d < 3; ++d) V.n[f][d] = fn[f][d];
        for (int qq = 0; qq < 6; ++qq) V.a0[f][qq] = 0.0;
        for (int d = 0; d < dim; ++d) {
            double *p1 = nullptr, *p2 = nullptr;
            PIB_CHK(upload_vec(tn[f][d], &p1));
            PIB_CHK(upload_vec(tp[f][d], &p2));
            V.owned.push_back(p1);
            V.owned.push_back(p2);
            V.lneg[f][d] = p1;
            V.lpos[f][d] = p2;
        }
    }
    V.valid = true;
    vel_stencil_release(s);
    s->vel = V;
    // ---- the recovered product against the CSR's, on the device
    int err = vel_stencil_verify(s);
    if (err != 0 || !s->vel.valid) {
        vel_stencil_release(s);
    } else
        s->vel_detected = true;
    return 0;
}

// ---- the same on several ranks, rows in z-slabs (y-slabs in 2-D) of the DMComposite's packed ordering: every rank hands
// over [u-slab | v-slab | w-slab] (an unchanged PetIBM on two ranks -- PETSC_DECIDE gives (1,1,2) on a cube --, or one
// whose process grid was set to (1,1,P)).  Such rows come through the general halo plan (partition.cpp): the low ghost
// pad holds the previous rank's last plane of u, v, w back to back, the high pad the next rank's first planes -- the layout
// pib_assemble_velocity's segmented plan has, so the matrix-free product of velstencil.hip serves them as it serves the
// rows assembled on the device.  Global sizes from u's first rows as on one rank; a field's planes on this rank from
// walking its +z neighbours (the DMDA splits a component's planes on its own: nz - 1 planes of w over P ranks are not the
// pressure split); the in-plane tables from rank 0's lines, the tables along the slab axis gathered from every rank's
// planes; the walls' effective values from the diagonals as on one rank.  Every rank walks the same collectives; the
// recovered product is compared with the CSR's on all ranks before it is used.  The slab axis must not be periodic.
static int detect_velocity_structure_slabs(pib_solver *s, const HostCsr &A)
{
    const int P = s->comm.nranks, rank = s->comm.rank;
    const int64_t n_local = A.n_local, row0 = A.row0;
    bool ok = s->A.general && n_local >= 27;
    int dim = 0;
    int64_t n[3] = {1, 1, 1};
    bool per[3] = {false, false, false};
    auto in_range = [&](int64_t c) { return c >= row0 && c < row0 + n_local; };
    auto has_col = [&](int64_t l, int64_t c) {
        for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p)
            if (A.CL(p) == c) return true;
        return false;
    };
    if (ok) {
        std::set<int64_t> offs;
        for (int64_t l = 0; l < std::min<int64_t>(n_local, 8); ++l)
            for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p) {
                const int64_t c = A.CL(p);
                if (in_range(c) && c != row0 + l) offs.insert(std::llabs(c - (row0 + l)));
            }
        ok = parse_velocity_sizes(std::vector<int64_t>(offs.begin(), offs.end()), A.n_global, true, &dim, n, per);
    }
    const int sd = ok ? dim - 1 : 0;
    int64_t fn[3][3], pl[3] = {0, 0, 0}, cnt[3] = {0, 0, 0}, loff[4] = {0, 0, 0, 0};
    for (int f = 0; f < 3; ++f)
        for (int d = 0; d < 3; ++d) fn[f][d] = 1;
    if (ok) {
        for (int f = 0; f < dim && ok; ++f) {
            for (int d = 0; d < dim; ++d) {
                fn[f][d] = n[d] - ((d == f && !per[d]) ? 1 : 0);
                ok = ok && fn[f][d] >= 3;
            }
            pl[f] = dim == 3 ? fn[f][0] * fn[f][1] : fn[f][0];
        }
        // planes of every field on this rank: walk the +slab-axis neighbour of the block's first point
        for (int f = 0; f < dim && ok; ++f) {
            int64_t l = loff[f];
            ok = l < n_local;
            cnt[f] = 1;
            while (ok && l + pl[f] < n_local && has_col(l, row0 + l + pl[f])) {
                l += pl[f];
                ++cnt[f];
            }
            loff[f + 1] = loff[f] + pl[f] * cnt[f];
        }
        ok = ok && loff[dim] == n_local;
    }
    // ---- what every rank found
    const size_t HL = 12;
    std::vector<double> head = {ok ? 1.0 : 0.0, (double)dim, (double)n[0], (double)n[1], (double)n[2],
                                (double)((per[0] ? 1 : 0) | (per[1] ? 2 : 0) | (per[2] ? 4 : 0)), (double)cnt[0], (double)cnt[1], (double)cnt[2],
                                (double)row0, (double)n_local, 0.0},
                        heads;
    PIB_CHK(comm_allgather_host(s, head, heads));
    auto H = [&](int q, int k) { return (int64_t)heads[HL * (size_t)q + (size_t)k]; };
    for (int q = 0; q < P; ++q)
        for (int k = 0; k < 6; ++k)
            if (H(q, 0) != 1 || H(q, k) != H(0, k)) return 0;
    dim = (int)H(0, 1);
    // first plane of every field on every rank; the ranks' blocks must tile the fields and the row ranges
    std::vector<int64_t> kb[3];
    for (int f = 0; f < dim; ++f) {
        kb[f].assign((size_t)P + 1, 0);
        for (int q = 0; q < P; ++q) kb[f][(size_t)q + 1] = kb[f][(size_t)q] + H(q, 6 + f);
        if (kb[f][(size_t)P] != fn[f][sd]) return 0;
    }
    {
        int64_t expect = 0;
        for (int q = 0; q < P; ++q) {
            if (H(q, 9) != expect) return 0;
            expect += H(q, 10);
        }
    }
    // global column of point (i, j, k) of field f (k: index along the slab axis; 2-D: j is that index)
    auto gcol = [&](int f, int64_t inplane, int64_t k) -> int64_t {
        int q = (int)(std::upper_bound(kb[f].begin(), kb[f].end(), k) - kb[f].begin()) - 1;
        if (q < 0 || q >= P) return -1;
        int64_t o = H(q, 9);
        for (int e = 0; e < f; ++e) o += pl[e] * H(q, 6 + e);
        return o + (k - kb[f][(size_t)q]) * pl[f] + inplane;
    };
    bool found = true;
    auto get = [&](int64_t row, int64_t col) {
        double v = 0.0;
        if (col < 0 || !A.entry(row, col, &v)) found = false;
        return v;
    };
    // ---- this rank's lines.  Buffer: [status, shift, per field: in-plane tables (tn, tp per in-plane direction), slab-axis
    // tables of the own planes (tn, tp, padded to the most planes a rank holds)]
    int64_t maxcnt = 0, inlen = 0;
    for (int f = 0; f < dim; ++f) {
        for (int q = 0; q < P; ++q) maxcnt = std::max(maxcnt, H(q, 6 + f));
        for (int d = 0; d < dim; ++d)
            if (d != sd) inlen += 2 * fn[f][d];
    }
    const size_t L = 2 + (size_t)inlen + 2 * (size_t)dim * (size_t)maxcnt;
    std::vector<double> mine(L, 0.0), all;
    double shift = 0.0;
    {
        size_t w = 2;
        // rank 0 finds the shift first (an interior point of u: plane 1)
        if (rank == 0 && cnt[0] >= 2) {
            const int64_t st0[3] = {1, fn[0][0], pl[0]};
            int64_t inpl = 0;
            for (int e = 0; e < sd; ++e) inpl += st0[e];
            const int64_t b = gcol(0, inpl, 1);
            double sum = 0.0;
            for (int e = 0; e < sd; ++e) sum += get(b, b - st0[e]) + get(b, b + st0[e]);
            sum += get(b, gcol(0, inpl, 0)) + get(b, gcol(0, inpl, 2));
            shift = get(b, b) + sum;
        }
        mine[1] = shift;
        for (int f = 0; f < dim; ++f) {
            const int64_t st[3] = {1, fn[f][0], pl[f]};
            const int64_t k_lo = kb[f][(size_t)rank], k_hi = k_lo + cnt[f];  // own planes
            // a plane of this rank that is no global boundary plane (its slab-axis neighbours exist)
            const int64_t kg = std::max<int64_t>(k_lo, 1);
            const bool have_kg = kg < k_hi && kg <= fn[f][sd] - 2;
            for (int d = 0; d < dim; ++d) {
                if (d == sd) continue;
                const int64_t nd = fn[f][d];
                double *tn = &mine[w], *tp = tn + nd;
                w += 2 * (size_t)nd;
                if (!have_kg || rank != 0) continue;  // the in-plane tables are rank 0's (the same entries on every rank)
                auto inplane_of = [&](int64_t sidx) {
                    int64_t p = 0;
                    for (int e = 0; e < sd; ++e) p += st[e] * (e == d ? sidx : 1);
                    return p;
                };
                for (int64_t sidx = 0; sidx < nd; ++sidx) {
                    const int64_t p = gcol(f, inplane_of(sidx), kg);
                    const int64_t cm = sidx > 0 ? p - st[d] : (per[d] ? p + (nd - 1) * st[d] : -1);
                    const int64_t cp = sidx < nd - 1 ? p + st[d] : (per[d] ? p - (nd - 1) * st[d] : -1);
                    if (cm >= 0) tn[sidx] = get(p, cm);
                    if (cp >= 0) tp[sidx] = get(p, cp);
                }
                for (int side = -1; side <= 1 && !per[d]; side += 2) {  // a wall's effective value from the boundary point's diagonal
                    const int64_t sidx = side < 0 ? 0 : nd - 1, p = gcol(f, inplane_of(sidx), kg);
                    double present = get(p, gcol(f, inplane_of(sidx), kg - 1)) + get(p, gcol(f, inplane_of(sidx), kg + 1));
                    for (int e = 0; e < sd; ++e) {
                        const int64_t se = e == d ? sidx : 1, ne = fn[f][e];
                        if (se > 0 || per[e]) present += get(p, se > 0 ? p - st[e] : p + (ne - 1) * st[e]);
                        if (se < ne - 1 || per[e]) present += get(p, se < ne - 1 ? p + st[e] : p - (ne - 1) * st[e]);
                    }
                    (side < 0 ? tn : tp)[sidx] = (shift - get(p, p)) - present;
                }
            }
        }
        // the slab-axis tables of the own planes (every rank): the line through the in-plane point (1[, 1])
        for (int f = 0; f < dim; ++f) {
            const int64_t st[3] = {1, fn[f][0], pl[f]};
            int64_t inpl = 0;
            for (int e = 0; e < sd; ++e) inpl += st[e];
            double *tn = &mine[w], *tp = tn + maxcnt;
            w += 2 * (size_t)maxcnt;
            const int64_t k_lo = kb[f][(size_t)rank];
            for (int64_t t = 0; t < cnt[f]; ++t) {
                const int64_t k = k_lo + t, p = gcol(f, inpl, k);
                if (k > 0) tn[t] = get(p, gcol(f, inpl, k - 1));
                if (k < fn[f][sd] - 1) tp[t] = get(p, gcol(f, inpl, k + 1));
            }
        }
        mine[0] = found ? 1.0 : -1.0;
    }
    PIB_CHK(comm_allgather_host(s, mine, all));
    for (int q = 0; q < P; ++q)
        if (all[L * (size_t)q] < 0.0) return 0;
    shift = all[1];  // rank 0's
    if (!std::isfinite(shift)) return 0;
    // the walls of the slab axis: effective values from the boundary planes' diagonals, computed by their owners with the
    // gathered shift and sent round in a second (small) gather
    std::vector<double> ends(2 * (size_t)dim, 0.0), all_ends;
    found = true;
    for (int f = 0; f < dim; ++f) {
        const int64_t st[3] = {1, fn[f][0], pl[f]};
        int64_t inpl = 0;
        for (int e = 0; e < sd; ++e) inpl += st[e];
        const int64_t k_lo = kb[f][(size_t)rank], k_hi = k_lo + cnt[f];
        for (int side = 0; side < 2; ++side) {
            const int64_t k = side == 0 ? 0 : fn[f][sd] - 1;
            if (k < k_lo || k >= k_hi) continue;
            const int64_t p = gcol(f, inpl, k);
            double present = side == 0 ? get(p, gcol(f, inpl, k + 1)) : get(p, gcol(f, inpl, k - 1));
            for (int e = 0; e < sd; ++e) present += get(p, p - st[e]) + get(p, p + st[e]);
            ends[2 * (size_t)f + (size_t)side] = (shift - get(p, p)) - present;
        }
    }
    if (!found) ends[0] = std::nan("");
    PIB_CHK(comm_allgather_host(s, ends, all_ends));
    for (double v : all_ends)
        if (std::isnan(v)) return 0;
    // ---- assemble the global tables (identical on every rank) and build the slab description
    VelStencil V;
    V.dim = dim;
    V.per = ((per[0] ? 1 : 0) | (per[1] ? 2 : 0) | (per[2] ? 4 : 0)) & ~(1 << sd);
    V.scale = 1.0;
    V.shift = shift;
    V.slab_axis = sd;
    V.has_lo = rank > 0;
    V.has_hi = rank < P - 1;
    const DeviceCsr &M = s->A;
    int64_t glo = 0, ghi = 0;
    {
        size_t w0 = 2;
        const double *r0 = &all[0];  // rank 0's buffer: the in-plane tables
        std::vector<std::vector<double>> inpl_tabs;
        for (int f = 0; f < dim; ++f)
            for (int d = 0; d < dim; ++d) {
                if (d == sd) continue;
                const int64_t nd = fn[f][d];
                inpl_tabs.emplace_back(r0 + w0, r0 + w0 + nd);
                inpl_tabs.emplace_back(r0 + w0 + nd, r0 + w0 + 2 * nd);
                w0 += 2 * (size_t)nd;
            }
        size_t it = 0;
        for (int f = 0; f < dim; ++f) {
            V.off[f] = loff[f];
            for (int d = 0; d < 3; ++d) V.n[f][d] = fn[f][d];
            V.n[f][sd] = cnt[f];
            for (int qq = 0; qq < 6; ++qq) V.a0[f][qq] = 0.0;
            V.pad_lo[f] = V.has_lo ? -M.ghost_lo + glo : 0;
            V.pad_hi[f] = V.has_hi ? n_local + ghi : 0;
            if (V.has_lo) glo += pl[f];
            if (V.has_hi) ghi += pl[f];
            for (int d = 0; d < dim; ++d) {
                std::vector<double> tn, tp;
                int64_t first = 0;
                if (d != sd) {
                    tn = inpl_tabs[it++];
                    tp = inpl_tabs[it++];
                } else {
                    tn.assign((size_t)fn[f][sd], 0.0);
                    tp.assign((size_t)fn[f][sd], 0.0);
                    for (int q = 0; q < P; ++q) {
                        const double *src = &all[L * (size_t)q + 2 + (size_t)inlen + 2 * (size_t)f * (size_t)maxcnt];
                        for (int64_t t = 0; t < H(q, 6 + f); ++t) {
                            tn[(size_t)(kb[f][(size_t)q] + t)] = src[t];
                            tp[(size_t)(kb[f][(size_t)q] + t)] = src[maxcnt + t];
                        }
                    }
                    for (int q = 0; q < P; ++q) {  // the two walls: whoever owns the plane has the value
                        const double lo_v = all_ends[2 * (size_t)dim * (size_t)q + 2 * (size_t)f], hi_v = all_ends[2 * (size_t)dim * (size_t)q + 2 * (size_t)f + 1];
                        if (kb[f][(size_t)q] == 0) tn[0] = lo_v;
                        if (kb[f][(size_t)q + 1] == fn[f][sd]) tp[(size_t)fn[f][sd] - 1] = hi_v;
                    }
                    first = kb[f][(size_t)rank];
                }
                double *p1 = nullptr, *p2 = nullptr;
                PIB_CHK(upload_vec(tn, &p1));
                PIB_CHK(upload_vec(tp, &p2));
                V.owned.push_back(p1);
                V.owned.push_back(p2);
                V.lneg[f][d] = p1 + first;
                V.lpos[f][d] = p2 + first;
            }
        }
    }
    if ((V.has_lo && glo != M.ghost_lo) || (V.has_hi && ghi != M.ghost_hi) || (!V.has_lo && M.ghost_lo != 0) || (!V.has_hi && M.ghost_hi != 0)) {
        for (double *p : V.owned) (void)hipFree(p);  // the ghost pads are not one plane of every field: not this layout
        V.owned.clear();
        V.valid = false;
    } else
        V.valid = true;
    // every rank must agree before anything collective follows
    std::vector<double> okv = {V.valid ? 1.0 : 0.0}, okall;
    PIB_CHK(comm_allgather_host(s, okv, okall));
    bool all_valid = true;
    for (double v : okall) all_valid = all_valid && v == 1.0;
    vel_stencil_release(s);
    if (!all_valid) {
        for (double *p : V.owned) (void)hipFree(p);
        return 0;
    }
    s->vel = V;
    // ---- the recovered product against the CSR's on every rank's rows (halo planes exchanged like a Krylov product's)
    int err = vel_stencil_verify(s);
    std::vector<double> vv = {(err == 0 && s->vel.valid) ? 1.0 : 0.0}, vall;
    PIB_CHK(comm_allgather_host(s, vv, vall));
    bool good = true;
    for (double v : vall) good = good && v == 1.0;
    if (!good) vel_stencil_release(s);
    else s->vel_detected = true;
    return 0;
}

}  // namespace pib
