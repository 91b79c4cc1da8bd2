// partition.cpp -- rows in ANY partition: what an unchanged PetIBM hands setMatrix on P > 1 ranks.
//
// The reference creates its DMDAs with nProc = PETSC_DECIDE (src/mesh/cartesianmesh.cpp:97, 503-519) and never calls
// DMSetFromOptions, so from 4 ranks up PETSc cuts a 3-D mesh into BOXES ((1,2,2) at 4 ranks, (2,2,2) at 8) and numbers
// the unknowns rank after rank, every box in its own natural order (cartesianmesh.cpp:700-738); the velocity unknowns
// are packed per rank as [u box | v box | w box] (cartesianmesh.cpp:741-779).  `vSolver->setMatrix(A)` /
// `pSolver->setMatrix(DBNG)` (applications/navierstokes/navierstokes.cpp:163-164) pass those rows on;
// AmgXSolver::setA (src/linsolver/linsolveramgx.cpp:84) takes any row partition, and so does this file:
//
//  * classify_partition: z-slabs in natural ordering (ghost columns = contiguous runs at the ends of the two adjacent
//    ranks' ranges) keep the contiguous-plane halo plan of halo.hip; anything else is "general".
//  * upload_csr_general: the ghost columns are the sorted distinct off-rank columns (low pad: those below row0, high pad:
//    the rest), their owners come from the all-gathered row ranges, every owner learns which of its rows its peers
//    need (one exchange of index lists at set-up) and packs them before every product (redistribute.hip).  Krylov with
//    Jacobi / no preconditioner runs on that directly.
//  * redist_setup (multigrid asked for): every rank's box is read off the column offsets of its first rows
//    (1, xm, xm ym) and the owners across its three + faces, the process grid follows from walking those owners, and the
//    rows are moved ONCE to the z-slabs in natural ordering the geometric multigrid works on -- inside an inner solver
//    sharing this one's communicator, which recovers the mesh from the entries as it does for slab callers
//    (structure.cpp) and verifies it against the CSR.  Per solve b (and the guess) go the same way and x comes back
//    (redistribute.hip): 8 B/row each way, against the ~3000 B/row a multigrid-PCG solve moves.
#include <algorithm>
#include <cmath>
#include <set>

#include "pib_internal.hpp"

namespace pib {

namespace {
struct View {
    int64_t n_local, row0, n_global;
    const int64_t *rp64, *cl64;
    const int32_t *rp32, *cl32;
    int64_t RP(int64_t i) const { return rp64 ? rp64[i] : (int64_t)rp32[i]; }
    int64_t CL(int64_t p) const { return cl64 ? cl64[p] : (int64_t)cl32[p]; }
    bool mine(int64_t c) const { return c >= row0 && c < row0 + n_local; }
};
// ranges: [P + 1] first row of every rank (+ n_global)
inline int owner_of(const std::vector<int64_t> &ranges, int64_t c)
{
    return (int)(std::upper_bound(ranges.begin(), ranges.end(), c) - ranges.begin()) - 1;
}
// sorted distinct off-rank columns
std::vector<int64_t> ghost_columns(const View &A)
{
    std::vector<int64_t> g;
    const int64_t base = A.RP(0), end = A.RP(A.n_local);
    for (int64_t p = base; p < end; ++p) {
        const int64_t c = A.CL(p);
        if (!A.mine(c)) g.push_back(c);
    }
    std::sort(g.begin(), g.end());
    g.erase(std::unique(g.begin(), g.end()), g.end());
    return g;
}
}  // namespace

int classify_partition(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                       const int32_t *rp32, const int32_t *cl32, std::vector<int64_t> &ranges, bool *general)
{
    *general = false;
    const int P = s->comm.nranks, r = s->comm.rank;
    ranges.assign((size_t)P + 1, 0);
    ranges[(size_t)P] = n_global;
    if (P <= 1) return 0;
    const View A{n_local, row0, n_global, rp64, cl64, rp32, cl32};
    std::vector<double> mine = {(double)row0, (double)n_local, 0.0}, all;
    // local verdict: every ghost column belongs to an adjacent rank (the ring counts) and the ghosts of one owner are at
    // most two contiguous runs, each touching an end of the owner's range -- planes of a slab next to this one
    bool local_general = false;
    {
        // owners need the ranges: gather them first
        std::vector<double> head = {(double)row0, (double)n_local}, heads;
        PIB_CHK(comm_allgather_host(s, head, heads));
        int64_t expect = 0;
        for (int q = 0; q < P; ++q) {
            ranges[(size_t)q] = (int64_t)heads[2 * (size_t)q];
            if (ranges[(size_t)q] != expect) return fail(PIB_ERR_ARG_WRONG, "set_csr: the ranks' row ranges are not consecutive (rank %d starts at %lld, expected %lld)", q,
                                                         (long long)ranges[(size_t)q], (long long)expect);
            expect += (int64_t)heads[2 * (size_t)q + 1];
        }
        if (expect != n_global) return fail(PIB_ERR_ARG_WRONG, "set_csr: the ranks' rows add up to %lld, n_global is %lld", (long long)expect, (long long)n_global);
        const std::vector<int64_t> g = ghost_columns(A);
        size_t a = 0;
        while (a < g.size() && !local_general) {
            const int q = owner_of(ranges, g[a]);
            size_t b = a;
            while (b < g.size() && g[b] < ranges[(size_t)q + 1]) ++b;  // [a, b): the ghosts owned by q
            const bool adjacent = q == r - 1 || q == r + 1 || (r == 0 && q == P - 1) || (r == P - 1 && q == 0);
            if (!adjacent) local_general = true;
            // runs
            int runs = 0;
            bool touches = true;
            for (size_t t = a; t < b;) {
                size_t u = t + 1;
                while (u < b && g[u] == g[u - 1] + 1) ++u;
                ++runs;
                touches = touches && (g[t] == ranges[(size_t)q] || g[u - 1] == ranges[(size_t)q + 1] - 1);
                t = u;
            }
            if (runs > 2 || !touches) local_general = true;
            a = b;
        }
    }
    mine[2] = local_general ? 1.0 : 0.0;
    PIB_CHK(comm_allgather_host(s, mine, all));
    for (int q = 0; q < P; ++q) *general = *general || all[3 * (size_t)q + 2] != 0.0;
    return 0;
}

int upload_csr_general(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                       const int32_t *rp32, const int32_t *cl32, const double *val, const std::vector<int64_t> &ranges)
{
    const int P = s->comm.nranks, r = s->comm.rank;
    const View V{n_local, row0, n_global, rp64, cl64, rp32, cl32};
    const int64_t base = V.RP(0), nnz = V.RP(n_local) - base;
    if (nnz < 0) return fail(PIB_ERR_ARG_OUTOFRANGE, "set_csr: negative nnz");
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t c = V.CL(base + p);
        if (c < 0 || c >= n_global) return fail(PIB_ERR_ARG_OUTOFRANGE, "set_csr: column %lld out of range", (long long)c);
    }
    DeviceCsr &A = s->A;
    A.release();
    s->comm.ring = false;
    vel_stencil_release(s);
    A.general = true;
    A.n = n_local;
    A.nnz = nnz;
    A.row0 = row0;
    A.n_global = n_global;
    A.ghost_cols = ghost_columns(V);
    const std::vector<int64_t> &g = A.ghost_cols;
    A.ghost_lo = (int64_t)(std::lower_bound(g.begin(), g.end(), row0) - g.begin());
    A.ghost_hi = (int64_t)g.size() - A.ghost_lo;
    if (A.ghost_lo + A.n + A.ghost_hi >= (int64_t)INT32_MAX) return fail(PIB_ERR_SUP, "set_csr: local column range does not fit 32-bit indices");
    A.ghost_off.assign((size_t)P + 1, 0);
    for (int q = 0; q < P; ++q)
        A.ghost_off[(size_t)q + 1] = (int64_t)(std::lower_bound(g.begin(), g.end(), ranges[(size_t)q + 1]) - g.begin());
    A.rp64 = nnz >= (int64_t)INT32_MAX;
    // local columns: [low ghosts | owned | high ghosts]
    std::vector<int32_t> c32((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t c = V.CL(base + p);
        if (V.mine(c))
            c32[(size_t)p] = (int32_t)(A.ghost_lo + (c - row0));
        else {
            const int64_t idx = (int64_t)(std::lower_bound(g.begin(), g.end(), c) - g.begin());
            c32[(size_t)p] = (int32_t)(idx + (c > row0 ? n_local : 0));
        }
    }
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(nnz + 4)));
    PIB_MEMSET(A.col, 0, sizeof(int32_t) * (size_t)(nnz + 4));
    PIB_MEMSET(A.val, 0, sizeof(double) * (size_t)(nnz + 4));
    PIB_HIP(hipMemcpy(A.col, c32.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
    PIB_HIP(hipMemcpy(A.val, val + base, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
    if (A.rp64) {
        std::vector<int64_t> rp((size_t)n_local + 1);
        for (int64_t i = 0; i <= n_local; ++i) rp[(size_t)i] = V.RP(i) - base;
        PIB_HIP(hipMalloc(&A.rowptr, sizeof(int64_t) * ((size_t)n_local + 1)));
        PIB_HIP(hipMemcpy(A.rowptr, rp.data(), sizeof(int64_t) * ((size_t)n_local + 1), hipMemcpyHostToDevice));
    } else {
        std::vector<int32_t> rp((size_t)n_local + 1);
        for (int64_t i = 0; i <= n_local; ++i) rp[(size_t)i] = (int32_t)(V.RP(i) - base);
        PIB_HIP(hipMalloc(&A.rowptr, sizeof(int32_t) * ((size_t)n_local + 1)));
        PIB_HIP(hipMemcpy(A.rowptr, rp.data(), sizeof(int32_t) * ((size_t)n_local + 1), hipMemcpyHostToDevice));
    }
    // ---- who needs what: every rank's request counts, all-gathered; then the index lists themselves travel to the owners
    std::vector<double> req((size_t)P), reqs;
    for (int q = 0; q < P; ++q) req[(size_t)q] = (double)(A.ghost_off[(size_t)q + 1] - A.ghost_off[(size_t)q]);
    if (req[(size_t)r] != 0.0) return fail(PIB_ERR_LIB, "set_csr: internal error (a ghost column inside the own range)");
    PIB_CHK(comm_allgather_host(s, req, reqs));
    ExchangePlan ask;  // requester -> owner
    ask.cnt.assign((size_t)P * P, 0);
    A.xplan.cnt.assign((size_t)P * P, 0);
    for (int a = 0; a < P; ++a)
        for (int b = 0; b < P; ++b) {
            const int64_t c = (int64_t)reqs[(size_t)a * P + b];  // a needs c entries of b
            ask.cnt[(size_t)a * P + b] = c;
            A.xplan.cnt[(size_t)b * P + a] = c;
        }
    ask.finish(P, r);
    A.xplan.finish(P, r);
    const int64_t ns = A.xplan.send_total, ng = (int64_t)g.size();
    double *d_req = nullptr, *d_got = nullptr;
    PIB_HIP(hipMalloc(&d_req, sizeof(double) * (size_t)std::max<int64_t>(ng, 1)));
    PIB_HIP(hipMalloc(&d_got, sizeof(double) * (size_t)std::max<int64_t>(ns, 1)));
    {
        std::vector<double> gd((size_t)std::max<int64_t>(ng, 1), 0.0);
        for (int64_t i = 0; i < ng; ++i) gd[(size_t)i] = (double)g[(size_t)i];  // exact below 2^53
        PIB_HIP(hipMemcpyAsync(d_req, gd.data(), sizeof(double) * (size_t)std::max<int64_t>(ng, 1), hipMemcpyHostToDevice, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));
    }
    std::vector<double *> recv((size_t)P, nullptr);
    for (int q = 0; q < P; ++q) recv[(size_t)q] = d_got + A.xplan.send_off[(size_t)q];
    PIB_CHK(comm_exchange_v(s, ask, d_req, recv.data(), s->stream));
    std::vector<double> got((size_t)std::max<int64_t>(ns, 1), 0.0);
    PIB_HIP(hipMemcpyAsync(got.data(), d_got, sizeof(double) * (size_t)std::max<int64_t>(ns, 1), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipFree(d_req));
    PIB_HIP(hipFree(d_got));
    std::vector<int32_t> idx((size_t)std::max<int64_t>(ns, 1), 0);
    for (int64_t i = 0; i < ns; ++i) {
        const int64_t l = (int64_t)got[(size_t)i] - row0;
        if (l < 0 || l >= n_local) return fail(PIB_ERR_LIB, "set_csr: a peer asked rank %d for row %lld, which it does not own", r, (long long)got[(size_t)i]);
        idx[(size_t)i] = (int32_t)l;
    }
    PIB_HIP(hipMalloc(&A.send_idx, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
    PIB_HIP(hipMalloc(&A.send_buf, sizeof(double) * (size_t)std::max<int64_t>(ns, 1)));
    PIB_HIP(hipMemcpy(A.send_idx, idx.data(), sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1), hipMemcpyHostToDevice));
    PIB_MEMSET(A.send_buf, 0, sizeof(double) * (size_t)std::max<int64_t>(ns, 1));
    return 0;
}

// ------------------------------------------------------------------------------------------------ boxes -> slabs
void redist_release(pib_solver *s)
{
    Redist &R = s->redist;
    if (R.inner) (void)pib_destroy(R.inner);
    if (R.d_split) (void)hipFree(R.d_split);
    if (R.d_src) (void)hipFree(R.d_src);
    if (R.stage) (void)hipFree(R.stage);
    if (R.b_nat) (void)hipFree(R.b_nat);
    if (R.x_nat) (void)hipFree(R.x_nat);
    R = Redist();
}

namespace {
// {1, xm[, xm ym]} and the in-box periodic wraps from the distinct positive in-range offsets of the first rows: the
// arithmetic of detect_grid_structure (structure.cpp) on a box of `total` cells
bool parse_box_offsets(const std::vector<int64_t> &S, int64_t total, int *dim, int64_t n[3])
{
    n[0] = n[1] = n[2] = 1;
    *dim = 0;
    if (S.size() < 2 || S.size() > 6 || S[0] != 1) return false;
    size_t q = 2;
    if (S.size() > 2 && S[2] == S[1] + 1) {
        n[0] = S[2];
        q = 3;
    } else
        n[0] = S[1];
    if (n[0] < 3 || total % n[0] != 0) return false;
    const int64_t rows = total / n[0];
    std::vector<int64_t> m;
    for (size_t t = q; t < S.size(); ++t) {
        if (S[t] % n[0] != 0) return false;
        m.push_back(S[t] / n[0]);
    }
    if (m.empty() || (m.size() == 1 && m[0] == rows - 1)) {
        *dim = 2;
        n[1] = rows;
        return n[1] >= 3;
    }
    for (int64_t y : m) {
        if (y < 3 || rows % y != 0 || rows / y < 3) continue;
        const int64_t z = rows / y;
        bool fits = true;
        for (int64_t v : m) fits = fits && (v == y || v == y - 1 || v == y * (z - 1));
        if (!fits) continue;
        *dim = 3;
        n[1] = y;
        n[2] = z;
        return true;
    }
    return false;
}
}  // namespace

int redist_setup(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                 const int32_t *rp32, const int32_t *cl32, const double *val, const std::vector<int64_t> &ranges)
{
    redist_release(s);
    const int P = s->comm.nranks, rank = s->comm.rank;
    const View A{n_local, row0, n_global, rp64, cl64, rp32, cl32};
    const int64_t base = A.RP(0);
    // ---- this rank's box: dims from the in-range offsets of its first rows, the owners across its + faces from the
    // off-range columns of the cells (xm - 1, 1, 1), (1, ym - 1, 1), (1, 1, zm - 1)
    bool ok = n_local > 0;
    int dim = 0;
    int64_t bx[3] = {1, 1, 1}, own[3] = {-1, -1, -1}, maxlen = 0;
    if (ok) {
        std::set<int64_t> offs;
        for (int64_t l = 0; l < std::min<int64_t>(n_local, 8); ++l)
            for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p) {
                const int64_t c = A.CL(p);
                if (A.mine(c) && c != row0 + l) offs.insert(std::llabs(c - (row0 + l)));
            }
        ok = parse_box_offsets(std::vector<int64_t>(offs.begin(), offs.end()), n_local, &dim, bx);
    }
    for (int64_t l = 0; l < n_local; ++l) maxlen = std::max(maxlen, A.RP(l + 1) - A.RP(l));
    if (ok) {
        const int64_t st[3] = {1, bx[0], bx[0] * bx[1]};
        for (int d = 0; d < dim && ok; ++d) {
            int64_t l = 0;
            for (int e = 0; e < dim; ++e) l += st[e] * (e == d ? bx[e] - 1 : 1);
            int found = 0;
            for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p) {
                const int64_t c = A.CL(p);
                if (c < 0 || c >= n_global) ok = false;
                else if (!A.mine(c)) {
                    own[d] = owner_of(ranges, c);
                    ++found;
                }
            }
            ok = ok && found <= 1;
        }
    }
    std::vector<double> head = {ok ? 1.0 : 0.0, (double)dim, (double)bx[0], (double)bx[1], (double)bx[2], (double)own[0], (double)own[1],
                                (double)own[2], (double)maxlen},
                        heads;
    const size_t HL = head.size();
    PIB_CHK(comm_allgather_host(s, head, heads));
    auto H = [&](int q, int k) { return (int64_t)heads[HL * (size_t)q + (size_t)k]; };
    int64_t W = 0;
    for (int q = 0; q < P; ++q) {
        if (H(q, 0) != 1 || H(q, 1) != H(0, 1)) return 0;  // every rank sees the same heads: all leave together
        W = std::max(W, H(q, 8));
    }
    dim = (int)H(0, 1);
    // ---- the process grid from walking the + owners (rank = px + m (py + n pz), cartesianmesh.cpp / DMDA)
    int g3[3] = {1, 1, 1};
    {
        int cur = 0, stride = 1;
        for (int d = 0; d < dim; ++d) {
            cur = 0;
            while (cur + stride < P && H(cur, 5 + d) == cur + stride) {
                cur += stride;
                ++g3[d];
            }
            stride *= g3[d];
        }
        if (g3[0] * g3[1] * g3[2] != P) return 0;
    }
    // box sizes per process coordinate, every rank checked against them; origins are the running sums
    std::vector<int64_t> ext[3];
    int64_t N[3] = {1, 1, 1};
    for (int d = 0; d < 3; ++d) {
        const int stride = d == 0 ? 1 : (d == 1 ? g3[0] : g3[0] * g3[1]);
        for (int c = 0; c < g3[d]; ++c) ext[d].push_back(H(c * stride, 2 + d));
        N[d] = 0;
        for (int64_t e : ext[d]) N[d] += e;
    }
    if (N[0] * N[1] * N[2] != n_global) return 0;
    Redist &R = s->redist;
    R.box.assign(6 * (size_t)P, 0);
    {
        int64_t expect = 0;
        for (int q = 0; q < P; ++q) {
            const int pc[3] = {q % g3[0], (q / g3[0]) % g3[1], q / (g3[0] * g3[1])};
            int64_t cells = 1;
            for (int d = 0; d < 3; ++d) {
                if (H(q, 2 + d) != ext[d][(size_t)pc[d]]) return 0;
                int64_t o = 0;
                for (int c = 0; c < pc[d]; ++c) o += ext[d][(size_t)c];
                R.box[6 * (size_t)q + d] = o;
                R.box[6 * (size_t)q + 3 + d] = ext[d][(size_t)pc[d]];
                cells *= ext[d][(size_t)pc[d]];
            }
            if (ranges[(size_t)q] != expect || ranges[(size_t)q + 1] - ranges[(size_t)q] != cells) return 0;
            expect += cells;
        }
    }
    // internal layout: the slab axis is the last one -- a 2-D grid (nx, ny) becomes (nx, 1, ny), its process grid (m, 1, n)
    if (dim == 2) {
        std::swap(N[1], N[2]);
        std::swap(g3[1], g3[2]);
        for (int q = 0; q < P; ++q) {
            std::swap(R.box[6 * (size_t)q + 1], R.box[6 * (size_t)q + 2]);
            std::swap(R.box[6 * (size_t)q + 4], R.box[6 * (size_t)q + 5]);
        }
    }
    R.dim = dim;
    for (int d = 0; d < 3; ++d) {
        R.n[d] = N[d];
        R.grid[d] = g3[d];
    }
    const int64_t pl = N[0] * N[1];
    slab_range(N[2], P, rank, &R.k0, &R.k1);
    R.n_slab = (R.k1 - R.k0) * pl;
    auto B = [&](int q, int k) { return R.box[6 * (size_t)q + (size_t)k]; };
    // rows (= vector entries) box s -> slab d
    R.fwd.cnt.assign((size_t)P * P, 0);
    R.bwd.cnt.assign((size_t)P * P, 0);
    for (int a = 0; a < P; ++a)
        for (int d = 0; d < P; ++d) {
            int64_t kb, ke;
            slab_range(N[2], P, d, &kb, &ke);
            const int64_t lo = std::max(kb, B(a, 2)), hi = std::min(ke, B(a, 2) + B(a, 5));
            const int64_t c = hi > lo ? (hi - lo) * B(a, 3) * B(a, 4) : 0;
            R.fwd.cnt[(size_t)a * P + d] = c;
            R.bwd.cnt[(size_t)d * P + a] = c;
        }
    R.fwd.finish(P, rank);
    R.bwd.finish(P, rank);
    // ---- the rows themselves: fixed-width records [length | W columns (natural numbering) | W values]
    const int64_t RW = 1 + 2 * W;
    ExchangePlan rows = R.fwd;
    for (auto &c : rows.cnt) c *= RW;
    rows.finish(P, rank);
    std::vector<double> rec((size_t)std::max<int64_t>(n_local * RW, 1), 0.0);
    auto natural = [&](int64_t c) {
        const int q = A.mine(c) ? rank : owner_of(ranges, c);
        const int64_t l = c - ranges[(size_t)q], x = B(q, 3), y = B(q, 4);
        const int64_t i = l % x, j = (l / x) % y, k = l / (x * y);
        return (B(q, 0) + i) + N[0] * ((B(q, 1) + j) + N[1] * (B(q, 2) + k));
    };
    for (int64_t l = 0; l < n_local; ++l) {
        double *rr = &rec[(size_t)(l * RW)];
        const int64_t a = A.RP(l), len = A.RP(l + 1) - a;
        rr[0] = (double)len;
        for (int64_t t = 0; t < len; ++t) {
            rr[1 + t] = (double)natural(A.CL(a + t));
            rr[1 + W + t] = val[a + t];
        }
    }
    (void)base;
    double *d_send = nullptr, *d_recv = nullptr;
    const int64_t nrecv = R.n_slab * RW;
    PIB_HIP(hipMalloc(&d_send, sizeof(double) * rec.size()));
    PIB_HIP(hipMalloc(&d_recv, sizeof(double) * (size_t)std::max<int64_t>(nrecv, 1)));
    PIB_HIP(hipMemcpyAsync(d_send, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    std::vector<double *> recv((size_t)P, nullptr);
    std::vector<int64_t> roff((size_t)P + 1, 0);
    for (int q = 0; q < P; ++q) {
        roff[(size_t)q + 1] = roff[(size_t)q] + R.fwd.from(q);
        recv[(size_t)q] = d_recv + roff[(size_t)q] * RW;
    }
    if (roff[(size_t)P] != R.n_slab) return fail(PIB_ERR_LIB, "set_csr: internal error (the boxes do not cover this rank's slab)");
    PIB_CHK(comm_exchange_v(s, rows, d_send, recv.data(), s->stream));
    std::vector<double> got((size_t)std::max<int64_t>(nrecv, 1), 0.0);
    PIB_HIP(hipMemcpyAsync(got.data(), d_recv, sizeof(double) * got.size(), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipFree(d_send));
    PIB_HIP(hipFree(d_recv));
    rec.clear();
    rec.shrink_to_fit();
    // natural local row of record t of source q
    std::vector<int64_t> rp((size_t)R.n_slab + 1, 0);
    auto local_row = [&](int q, int64_t t) {
        const int64_t x = B(q, 3), y = B(q, 4), kf = std::max(B(q, 2), R.k0);
        const int64_t i = t % x, j = (t / x) % y, k = kf + t / (x * y);
        return (B(q, 0) + i) + N[0] * ((B(q, 1) + j) + N[1] * (k - R.k0));
    };
    for (int q = 0; q < P; ++q)
        for (int64_t t = 0; t < R.fwd.from(q); ++t) rp[(size_t)local_row(q, t) + 1] = (int64_t)got[(size_t)((roff[(size_t)q] + t) * RW)];
    for (int64_t l = 0; l < R.n_slab; ++l) rp[(size_t)l + 1] += rp[(size_t)l];
    std::vector<int64_t> cl((size_t)std::max<int64_t>(rp[(size_t)R.n_slab], 1));
    std::vector<double> vl((size_t)std::max<int64_t>(rp[(size_t)R.n_slab], 1));
    for (int q = 0; q < P; ++q)
        for (int64_t t = 0; t < R.fwd.from(q); ++t) {
            const double *rr = &got[(size_t)((roff[(size_t)q] + t) * RW)];
            const int64_t l = local_row(q, t), len = (int64_t)rr[0];
            // by ascending natural column, as MatMPIAIJGetLocalMat would deliver the row on slabs
            std::pair<int64_t, double> e[64];
            std::vector<std::pair<int64_t, double>> big;
            std::pair<int64_t, double> *ep = e;
            if (len > 64) {
                big.resize((size_t)len);
                ep = big.data();
            }
            for (int64_t u = 0; u < len; ++u) ep[u] = {(int64_t)rr[1 + u], rr[1 + W + u]};
            std::sort(ep, ep + len, [](const std::pair<int64_t, double> &a, const std::pair<int64_t, double> &b) { return a.first < b.first; });
            for (int64_t u = 0; u < len; ++u) {
                cl[(size_t)(rp[(size_t)l] + u)] = ep[u].first;
                vl[(size_t)(rp[(size_t)l] + u)] = ep[u].second;
            }
        }
    got.clear();
    got.shrink_to_fit();
    // ---- the inner solver: same configuration, same communicator, z-slabs in natural ordering
    PIB_CHK(create_sharing_comm(&R.inner, s->name.c_str(), s->cfg.raw.c_str(), s));
    pib_solver *in = R.inner;
    in->cfg = s->cfg;
    for (int d = 0; d < 3; ++d) in->periodic[d] = in->periodic_user[d] = s->periodic_user[d];
    PIB_CHK(upload_csr(in, R.n_slab, R.k0 * pl, n_global, rp.data(), cl.data(), nullptr, nullptr, vl.data()));
    PIB_CHK(after_set_matrix(in));
    PIB_CHK(detect_grid_structure(in, R.n_slab, R.k0 * pl, n_global, rp.data(), cl.data(), nullptr, nullptr, vl.data()));
    if (!in->has_grid) {  // not PetIBM's Poisson operator after all (the same verdict on every rank: the check is a global sum)
        redist_release(s);
        return 0;
    }
    PIB_CHK(redist_tables(s));
    R.active = true;
    return 0;
}

}  // namespace pib
