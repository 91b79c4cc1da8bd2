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
// a scratch device buffer of the set-up paths: released on every exit, the error ones included
struct DevScratch {
    double *p = nullptr;
    ~DevScratch()
    {
        if (p) (void)hipFree(p);
    }
};
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
    par_ranges(nnz, [&](int64_t pb, int64_t pe) {
        for (int64_t p = pb; p < pe; ++p) {
            const int64_t c = V.CL(base + p);
            if (V.mine(c))
                c32[(size_t)p] = (int32_t)(A.ghost_lo + (c - row0));
            else {
                const int64_t idx = (int64_t)(std::lower_bound(g.begin(), g.end(), c) - g.begin());
                c32[(size_t)p] = (int32_t)(idx + (c > row0 ? n_local : 0));
            }
        }
    });
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(nnz + 4)));
    // (on the solver's own stream: null-stream fills and copies wait for every stream of the process -- the other ranks' set-up,
    // when the ranks of a process are threads on one device)
    hipStream_t st = s->stream;
    PIB_HIP(hipMemsetAsync(A.col, 0, sizeof(int32_t) * (size_t)(nnz + 4), st));
    PIB_HIP(hipMemsetAsync(A.val, 0, sizeof(double) * (size_t)(nnz + 4), st));
    PIB_HIP(hipMemcpyAsync(A.col, c32.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, st));
    PIB_HIP(hipMemcpyAsync(A.val, val + base, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, st));
    if (A.rp64) {
        std::vector<int64_t> rp((size_t)n_local + 1);
        for (int64_t i = 0; i <= n_local; ++i) rp[(size_t)i] = V.RP(i) - base;
        PIB_HIP(hipMalloc(&A.rowptr, sizeof(int64_t) * ((size_t)n_local + 1)));
        PIB_HIP(hipMemcpyAsync(A.rowptr, rp.data(), sizeof(int64_t) * ((size_t)n_local + 1), hipMemcpyHostToDevice, st));
        PIB_HIP(hipStreamSynchronize(st));
    } else {
        std::vector<int32_t> rp((size_t)n_local + 1);
        for (int64_t i = 0; i <= n_local; ++i) rp[(size_t)i] = (int32_t)(V.RP(i) - base);
        PIB_HIP(hipMalloc(&A.rowptr, sizeof(int32_t) * ((size_t)n_local + 1)));
        PIB_HIP(hipMemcpyAsync(A.rowptr, rp.data(), sizeof(int32_t) * ((size_t)n_local + 1), hipMemcpyHostToDevice, st));
        PIB_HIP(hipStreamSynchronize(st));
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
    DevScratch req_buf, got_buf;
    PIB_HIP(hipMalloc(&req_buf.p, sizeof(double) * (size_t)std::max<int64_t>(ng, 1)));
    PIB_HIP(hipMalloc(&got_buf.p, sizeof(double) * (size_t)std::max<int64_t>(ns, 1)));
    double *const d_req = req_buf.p, *const d_got = got_buf.p;
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
    std::vector<int32_t> idx((size_t)std::max<int64_t>(ns, 1), 0);
    for (int64_t i = 0; i < ns; ++i) {
        const int64_t l = (int64_t)got[(size_t)i] - row0;
        if (l < 0 || l >= n_local) return fail(PIB_ERR_LIB, "set_csr: a peer asked rank %d for row %lld, which it does not own", r, (long long)got[(size_t)i]);
        idx[(size_t)i] = (int32_t)l;
    }
    PIB_HIP(hipMalloc(&A.send_idx, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
    PIB_HIP(hipMalloc(&A.send_buf, sizeof(double) * (size_t)std::max<int64_t>(ns, 1)));
    PIB_HIP(hipMemcpyAsync(A.send_idx, idx.data(), sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1), hipMemcpyHostToDevice, s->stream));
    PIB_HIP(hipMemsetAsync(A.send_buf, 0, sizeof(double) * (size_t)std::max<int64_t>(ns, 1), s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ boxes -> slabs
void redist_release(pib_solver *s)
{
    drop_iteration_graph(s);  // (krylov.hip: the captured iteration goes before the memory it points at)
    Redist &R = s->redist;
    if (R.inner) (void)pib_destroy(R.inner);
    for (int f = 0; f < 3; ++f) {
        if (R.f[f].d_split) (void)hipFree(R.f[f].d_split);
        if (R.f[f].d_src) (void)hipFree(R.f[f].d_src);
        if (R.f[f].stage) (void)hipFree(R.f[f].stage);
    }
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
    SetupTrace tr("redist_setup", rank);
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
    tr.mark("own box read off the rows");
    PIB_CHK(comm_allgather_host(s, head, heads));
    tr.mark("heads gathered");
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
    R.nf = 1;
    RedistField &F = R.f[0];
    F.box.assign(6 * (size_t)P, 0);
    {
        int64_t expect = 0;
        for (int q = 0; q < P; ++q) {
            const int pc[3] = {q % g3[0], (q / g3[0]) % g3[1], q / (g3[0] * g3[1])};
            int64_t cells = 1;
            for (int d = 0; d < 3; ++d) {
                if (H(q, 2 + d) != ext[d][(size_t)pc[d]]) {
                    redist_release(s);  // (R.nf / F.box are filled by now: leave no half-built state behind, like redist_velocity_setup)
                    return 0;
                }
                int64_t o = 0;
                for (int c = 0; c < pc[d]; ++c) o += ext[d][(size_t)c];
                F.box[6 * (size_t)q + d] = o;
                F.box[6 * (size_t)q + 3 + d] = ext[d][(size_t)pc[d]];
                cells *= ext[d][(size_t)pc[d]];
            }
            if (ranges[(size_t)q] != expect || ranges[(size_t)q + 1] - ranges[(size_t)q] != cells) {
                redist_release(s);
                return 0;
            }
            expect += cells;
        }
    }
    // internal layout: the slab axis is the last one -- a 2-D grid (nx, ny) becomes (nx, 1, ny), its process grid (m, 1, n)
    if (dim == 2) {
        std::swap(N[1], N[2]);
        std::swap(g3[1], g3[2]);
        for (int q = 0; q < P; ++q) {
            std::swap(F.box[6 * (size_t)q + 1], F.box[6 * (size_t)q + 2]);
            std::swap(F.box[6 * (size_t)q + 4], F.box[6 * (size_t)q + 5]);
        }
    }
    R.dim = dim;
    for (int d = 0; d < 3; ++d) {
        F.n[d] = N[d];
        R.grid[d] = g3[d];
    }
    const int64_t pl = N[0] * N[1];
    slab_range(N[2], P, rank, &F.k0, &F.k1);
    F.n_slab = (F.k1 - F.k0) * pl;
    auto B = [&](int q, int k) { return F.box[6 * (size_t)q + (size_t)k]; };
    // rows (= vector entries) box s -> slab d
    F.fwd.cnt.assign((size_t)P * P, 0);
    F.bwd.cnt.assign((size_t)P * P, 0);
    for (int a = 0; a < P; ++a)
        for (int d = 0; d < P; ++d) {
            int64_t kb, ke;
            slab_range(N[2], P, d, &kb, &ke);
            const int64_t lo = std::max(kb, B(a, 2)), hi = std::min(ke, B(a, 2) + B(a, 5));
            const int64_t c = hi > lo ? (hi - lo) * B(a, 3) * B(a, 4) : 0;
            F.fwd.cnt[(size_t)a * P + d] = c;
            F.bwd.cnt[(size_t)d * P + a] = c;
        }
    F.fwd.finish(P, rank);
    F.bwd.finish(P, rank);
    // ---- the rows themselves, on the device (redistribute.hip: records, exchange, per-row sort, the slab's CSR); shapes that path
    // does not take (rows wider than 16 entries, 64-bit sizes) and PIB_BOX_ROWS_ON_DEVICE=0 go through the host loops below
    {
        const char *e = std::getenv("PIB_BOX_ROWS_ON_DEVICE");
        if (!(e && e[0] == '0')) {
            tr.mark("plans");
            PIB_CHK(create_sharing_comm(&R.inner, s->name.c_str(), s->cfg.raw.c_str(), s));
            tr.mark("inner solver created");
            pib_solver *in = R.inner;
            in->cfg = s->cfg;
            for (int d = 0; d < 3; ++d) in->periodic[d] = in->periodic_user[d] = s->periodic_user[d];
            int32_t *h_rp = nullptr, *h_cl = nullptr;
            double *h_vl = nullptr;
            int64_t nnz_s = 0;
            int err = redist_rows_on_device(s, in, F, N, n_local, row0, n_global, rp64, cl64, rp32, cl32, val, W, ranges, &h_rp, &h_cl, &h_vl, &nnz_s);
            tr.mark("rows moved to the slab on the device");
            if (err == 0) {
                err = after_set_matrix(in);
                tr.mark("inner after_set_matrix");
                if (!err) err = detect_grid_structure(in, F.n_slab, F.k0 * pl, n_global, nullptr, nullptr, h_rp, h_cl, h_vl);
                tr.mark("inner detect_grid_structure");
            }
            std::free(h_rp);
            std::free(h_cl);
            std::free(h_vl);
            if (err == 0) {
                if (!in->has_grid) {
                    redist_release(s);
                    return 0;
                }
                PIB_CHK(redist_tables(s));
                tr.mark("redist_tables");
                R.active = true;
                return 0;
            }
            if (err != PIB_ERR_SUP) return err;
            pib_destroy(R.inner);  // (not taken: the host path makes its own)
            R.inner = nullptr;
        }
    }
    // ---- the host path: fixed-width records [length | W columns (natural numbering) | W values]
    const int64_t RW = 1 + 2 * W;
    ExchangePlan rows = F.fwd;
    for (auto &c : rows.cnt) c *= RW;
    rows.finish(P, rank);
    std::vector<double> rec((size_t)std::max<int64_t>(n_local * RW, 1), 0.0);
    auto natural = [&](int64_t c) {
        const int q = A.mine(c) ? rank : owner_of(ranges, c);
        const int64_t l = c - ranges[(size_t)q], x = B(q, 3), y = B(q, 4);
        const int64_t i = l % x, j = (l / x) % y, k = l / (x * y);
        return (B(q, 0) + i) + N[0] * ((B(q, 1) + j) + N[1] * (B(q, 2) + k));
    };
    par_ranges(n_local, [&](int64_t lb, int64_t le) {
        for (int64_t l = lb; l < le; ++l) {
            double *rr = &rec[(size_t)(l * RW)];
            const int64_t a = A.RP(l), len = A.RP(l + 1) - a;
            rr[0] = (double)len;
            for (int64_t t = 0; t < len; ++t) {
                rr[1 + t] = (double)natural(A.CL(a + t));
                rr[1 + W + t] = val[a + t];
            }
        }
    });
    (void)base;
    tr.mark("records built");
    DevScratch send_buf, recv_buf;
    const int64_t nrecv = F.n_slab * RW;
    PIB_HIP(hipMalloc(&send_buf.p, sizeof(double) * rec.size()));
    PIB_HIP(hipMalloc(&recv_buf.p, sizeof(double) * (size_t)std::max<int64_t>(nrecv, 1)));
    double *const d_send = send_buf.p, *const d_recv = recv_buf.p;
    PIB_HIP(hipMemcpyAsync(d_send, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    std::vector<double *> recv((size_t)P, nullptr);
    std::vector<int64_t> roff((size_t)P + 1, 0);
    for (int q = 0; q < P; ++q) {
        roff[(size_t)q + 1] = roff[(size_t)q] + F.fwd.from(q);
        recv[(size_t)q] = d_recv + roff[(size_t)q] * RW;
    }
    if (roff[(size_t)P] != F.n_slab) return fail(PIB_ERR_LIB, "set_csr: internal error (the boxes do not cover this rank's slab)");
    tr.mark("records uploaded");
    PIB_CHK(comm_exchange_v(s, rows, d_send, recv.data(), s->stream));
    std::vector<double> got((size_t)std::max<int64_t>(nrecv, 1), 0.0);
    PIB_HIP(hipMemcpyAsync(got.data(), d_recv, sizeof(double) * got.size(), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    rec.clear();
    rec.shrink_to_fit();
    tr.mark("records exchanged and fetched");
    // natural local row of record t of source q
    std::vector<int64_t> rp((size_t)F.n_slab + 1, 0);
    auto local_row = [&](int q, int64_t t) {
        const int64_t x = B(q, 3), y = B(q, 4), kf = std::max(B(q, 2), F.k0);
        const int64_t i = t % x, j = (t / x) % y, k = kf + t / (x * y);
        return (B(q, 0) + i) + N[0] * ((B(q, 1) + j) + N[1] * (k - F.k0));
    };
    for (int q = 0; q < P; ++q)
        par_ranges(F.fwd.from(q), [&](int64_t tb, int64_t te) {
            for (int64_t t = tb; t < te; ++t) rp[(size_t)local_row(q, t) + 1] = (int64_t)got[(size_t)((roff[(size_t)q] + t) * RW)];
        });
    for (int64_t l = 0; l < F.n_slab; ++l) rp[(size_t)l + 1] += rp[(size_t)l];
    std::vector<int64_t> cl((size_t)std::max<int64_t>(rp[(size_t)F.n_slab], 1));
    std::vector<double> vl((size_t)std::max<int64_t>(rp[(size_t)F.n_slab], 1));
    for (int q = 0; q < P; ++q)
        par_ranges(F.fwd.from(q), [&](int64_t tb, int64_t te) {
        for (int64_t t = tb; t < te; ++t) {
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
        });
    got.clear();
    got.shrink_to_fit();
    tr.mark("rows sorted into the slab's CSR");
    // ---- the inner solver: same configuration, same communicator, z-slabs in natural ordering
    PIB_CHK(create_sharing_comm(&R.inner, s->name.c_str(), s->cfg.raw.c_str(), s));
    pib_solver *in = R.inner;
    in->cfg = s->cfg;
    for (int d = 0; d < 3; ++d) in->periodic[d] = in->periodic_user[d] = s->periodic_user[d];
    PIB_CHK(upload_csr(in, F.n_slab, F.k0 * pl, n_global, rp.data(), cl.data(), nullptr, nullptr, vl.data()));
    tr.mark("inner upload_csr");
    PIB_CHK(after_set_matrix(in));
    tr.mark("inner after_set_matrix");
    PIB_CHK(detect_grid_structure(in, F.n_slab, F.k0 * pl, n_global, rp.data(), cl.data(), nullptr, nullptr, vl.data()));
    tr.mark("inner detect_grid_structure");
    if (!in->has_grid) {  // not PetIBM's Poisson operator after all (the same verdict on every rank: the check is a global sum)
        redist_release(s);
        return 0;
    }
    PIB_CHK(redist_tables(s));
    R.active = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------ velocity boxes -> slabs
// vSolver->setMatrix(A) of an unchanged PetIBM from 4 ranks up: every rank holds [u box | v box | w box], the boxes of the
// three velocity DMDAs on the pressure grid's process grid (cartesianmesh.cpp:551-553, 741-779).  Rows and, per solve,
// vectors are moved to the packed z-slabs [u slab | v slab | w slab] (each component's planes split over the ranks by the
// DMDA's default rule), where the inner solver recovers the operator's structure from the entries
// (structure.cpp: detect_velocity_structure_slabs) and runs the matrix-free products of velstencil.hip: 16 B/row and product
// instead of the CSR's 104.  A field's box is read off by WALKING from its first point: +1 neighbours give xm, +xm
// neighbours ym, +xm ym neighbours zm (a row's columns are points of its own component only, so the walk stops at the
// box's end); the next field starts behind it.  The process grid comes from the owners across u's + faces as for the
// pressure boxes.  Anything that does not fit leaves the solver on its general plan with CSR products.
int redist_velocity_setup(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                          const int32_t *rp32, const int32_t *cl32, const double *val, const std::vector<int64_t> &ranges)
{
    redist_release(s);
    const int P = s->comm.nranks, rank = s->comm.rank;
    const View A{n_local, row0, n_global, rp64, cl64, rp32, cl32};
    auto has_col = [&](int64_t l, int64_t c) {
        if (l < 0 || l >= n_local) return false;
        for (int64_t p = A.RP(l); p < A.RP(l + 1); ++p)
            if (A.CL(p) == c) return true;
        return false;
    };
    // ---- this rank's boxes
    bool ok = n_local >= 8;
    int nf = 0;
    int64_t bx[3][3] = {{1, 1, 1}, {1, 1, 1}, {1, 1, 1}}, boff[4] = {0, 0, 0, 0}, own[3] = {-1, -1, -1}, maxlen = 0;
    for (int64_t l = 0; l < n_local; ++l) maxlen = std::max(maxlen, A.RP(l + 1) - A.RP(l));
    while (ok && boff[nf] < n_local) {
        if (nf == 3) {
            ok = false;
            break;
        }
        const int64_t l0 = boff[nf];
        int64_t xm = 1, ym = 1, zm = 1;
        while (l0 + xm < n_local && has_col(l0 + xm - 1, row0 + l0 + xm)) ++xm;
        while (l0 + ym * xm < n_local && has_col(l0 + (ym - 1) * xm, row0 + l0 + ym * xm)) ++ym;
        while (l0 + zm * xm * ym < n_local && has_col(l0 + (zm - 1) * xm * ym, row0 + l0 + zm * xm * ym)) ++zm;
        bx[nf][0] = xm;
        bx[nf][1] = ym;
        bx[nf][2] = zm;
        boff[nf + 1] = l0 + xm * ym * zm;
        ok = xm >= 3 && ym >= 3 && boff[nf + 1] <= n_local;
        ++nf;
    }
    ok = ok && (nf == 2 || nf == 3) && boff[nf] == n_local;
    const int dim = nf;
    for (int f = 0; f < nf && ok; ++f) ok = (dim == 2) ? bx[f][2] == 1 : bx[f][2] >= 3;
    if (ok) {  // owners across u's + faces
        const int64_t st[3] = {1, bx[0][0], bx[0][0] * bx[0][1]};
        for (int d = 0; d < dim && ok; ++d) {
            int64_t l = 0;
            for (int e = 0; e < dim; ++e) l += st[e] * (e == d ? bx[0][e] - 1 : 1);
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
    std::vector<double> head = {ok ? 1.0 : 0.0, (double)dim, (double)own[0], (double)own[1], (double)own[2], (double)maxlen};
    for (int f = 0; f < 3; ++f)
        for (int d = 0; d < 3; ++d) head.push_back((double)bx[f][d]);
    const size_t HL = head.size();
    std::vector<double> heads;
    PIB_CHK(comm_allgather_host(s, head, heads));
    auto H = [&](int q, int k) { return (int64_t)heads[HL * (size_t)q + (size_t)k]; };
    int64_t W = 0;
    for (int q = 0; q < P; ++q) {
        if (H(q, 0) != 1 || H(q, 1) != H(0, 1)) return 0;
        W = std::max(W, H(q, 5));
    }
    int g3[3] = {1, 1, 1};
    {
        int cur = 0, stride = 1;
        for (int d = 0; d < dim; ++d) {
            cur = 0;
            while (cur + stride < P && H(cur, 2 + d) == cur + stride) {
                cur += stride;
                ++g3[d];
            }
            stride *= g3[d];
        }
        if (g3[0] * g3[1] * g3[2] != P) return 0;
    }
    Redist &R = s->redist;
    R.nf = nf;
    R.dim = dim;
    int64_t total = 0;
    std::vector<int64_t> cells_of((size_t)P, 0);
    for (int f = 0; f < nf; ++f) {
        RedistField &F = R.f[f];
        std::vector<int64_t> ext[3];
        int64_t N[3] = {1, 1, 1};
        for (int d = 0; d < 3; ++d) {
            const int stride = d == 0 ? 1 : (d == 1 ? g3[0] : g3[0] * g3[1]);
            for (int c = 0; c < g3[d]; ++c) ext[d].push_back(H(c * stride, 6 + 3 * f + d));
            N[d] = 0;
            for (int64_t e : ext[d]) N[d] += e;
        }
        F.box.assign(6 * (size_t)P, 0);
        for (int q = 0; q < P; ++q) {
            const int pc[3] = {q % g3[0], (q / g3[0]) % g3[1], q / (g3[0] * g3[1])};
            int64_t cells = 1;
            for (int d = 0; d < 3; ++d) {
                if (H(q, 6 + 3 * f + d) != ext[d][(size_t)pc[d]]) {
                    redist_release(s);
                    return 0;
                }
                int64_t o = 0;
                for (int c = 0; c < pc[d]; ++c) o += ext[d][(size_t)c];
                F.box[6 * (size_t)q + d] = o;
                F.box[6 * (size_t)q + 3 + d] = ext[d][(size_t)pc[d]];
                cells *= ext[d][(size_t)pc[d]];
            }
            cells_of[(size_t)q] += cells;
        }
        for (int d = 0; d < 3; ++d) F.n[d] = N[d];
        total += N[0] * N[1] * N[2];
    }
    bool fits = total == n_global;
    for (int q = 0; q < P && fits; ++q) fits = ranges[(size_t)q + 1] - ranges[(size_t)q] == cells_of[(size_t)q];
    if (!fits) {
        redist_release(s);
        return 0;
    }
    // internal layout: the slab axis last (2-D: (nx, ny) -> (nx, 1, ny), process grid (m, n) -> (m, 1, n))
    if (dim == 2) {
        std::swap(g3[1], g3[2]);
        for (int f = 0; f < nf; ++f) {
            RedistField &F = R.f[f];
            std::swap(F.n[1], F.n[2]);
            for (int q = 0; q < P; ++q) {
                std::swap(F.box[6 * (size_t)q + 1], F.box[6 * (size_t)q + 2]);
                std::swap(F.box[6 * (size_t)q + 4], F.box[6 * (size_t)q + 5]);
            }
        }
    }
    for (int d = 0; d < 3; ++d) R.grid[d] = g3[d];
    // a field's block in every rank's box-ordered vector / slab-ordered vector; the slab split of every field: the DMDA's
    // default rule on its own plane count
    std::vector<int64_t> bofq[3], sofq[3], srow0((size_t)P + 1, 0);
    for (int f = 0; f < nf; ++f) {
        bofq[f].assign((size_t)P, 0);
        sofq[f].assign((size_t)P, 0);
    }
    for (int q = 0; q < P; ++q) {
        int64_t bo = 0, so = 0;
        for (int f = 0; f < nf; ++f) {
            const RedistField &F = R.f[f];
            bofq[f][(size_t)q] = bo;
            sofq[f][(size_t)q] = so;
            bo += F.box[6 * (size_t)q + 3] * F.box[6 * (size_t)q + 4] * F.box[6 * (size_t)q + 5];
            int64_t kb, ke;
            slab_range(F.n[2], P, q, &kb, &ke);
            so += (ke - kb) * F.n[0] * F.n[1];
        }
        srow0[(size_t)q + 1] = srow0[(size_t)q] + so;
    }
    R.n_slab = 0;
    for (int f = 0; f < nf; ++f) {
        RedistField &F = R.f[f];
        slab_range(F.n[2], P, rank, &F.k0, &F.k1);
        F.n_slab = (F.k1 - F.k0) * F.n[0] * F.n[1];
        F.boff = bofq[f][(size_t)rank];
        F.soff = sofq[f][(size_t)rank];
        F.fwd.cnt.assign((size_t)P * P, 0);
        F.bwd.cnt.assign((size_t)P * P, 0);
        for (int a = 0; a < P; ++a)
            for (int d = 0; d < P; ++d) {
                int64_t kb, ke;
                slab_range(F.n[2], P, d, &kb, &ke);
                const int64_t zs = F.box[6 * (size_t)a + 2], zm = F.box[6 * (size_t)a + 5];
                const int64_t lo = std::max(kb, zs), hi = std::min(ke, zs + zm);
                const int64_t c = hi > lo ? (hi - lo) * F.box[6 * (size_t)a + 3] * F.box[6 * (size_t)a + 4] : 0;
                F.fwd.cnt[(size_t)a * P + d] = c;
                F.bwd.cnt[(size_t)d * P + a] = c;
            }
        F.fwd.finish(P, rank);
        F.bwd.finish(P, rank);
        R.n_slab += F.n_slab;
    }
    // the slab-packed global index of box-packed global column c
    auto slab_index = [&](int64_t c) -> int64_t {
        const int q = A.mine(c) ? rank : owner_of(ranges, c);
        int64_t l = c - ranges[(size_t)q];
        int f = nf - 1;
        while (f > 0 && l < bofq[f][(size_t)q]) --f;
        l -= bofq[f][(size_t)q];
        const RedistField &F = R.f[f];
        const int64_t x = F.box[6 * (size_t)q + 3], y = F.box[6 * (size_t)q + 4];
        const int64_t I = F.box[6 * (size_t)q] + l % x, J = F.box[6 * (size_t)q + 1] + (l / x) % y, K = F.box[6 * (size_t)q + 2] + l / (x * y);
        // owner of plane K of field f: the DMDA split is monotone, find it by its range
        int d = (int)std::min<int64_t>(P - 1, K * P / std::max<int64_t>(F.n[2], 1));
        for (;;) {
            int64_t kb, ke;
            slab_range(F.n[2], P, d, &kb, &ke);
            if (K < kb) --d;
            else if (K >= ke) ++d;
            else return srow0[(size_t)d] + sofq[f][(size_t)d] + (K - kb) * F.n[0] * F.n[1] + J * F.n[0] + I;
        }
    };
    // ---- the rows, field by field: fixed-width records [length | W columns (slab-packed numbering) | W values]
    const int64_t RW = 1 + 2 * W;
    std::vector<int64_t> rp((size_t)R.n_slab + 1, 0);
    std::vector<std::vector<double>> got_f((size_t)nf);
    std::vector<std::vector<int64_t>> roff_f((size_t)nf);
    for (int f = 0; f < nf; ++f) {
        RedistField &F = R.f[f];
        ExchangePlan rows = F.fwd;
        for (auto &c : rows.cnt) c *= RW;
        rows.finish(P, rank);
        const int64_t nb = F.box[6 * (size_t)rank + 3] * F.box[6 * (size_t)rank + 4] * F.box[6 * (size_t)rank + 5];
        std::vector<double> rec((size_t)std::max<int64_t>(nb * RW, 1), 0.0);
        for (int64_t t = 0; t < nb; ++t) {
            const int64_t l = F.boff + t;
            double *rr = &rec[(size_t)(t * RW)];
            const int64_t a = A.RP(l), len = A.RP(l + 1) - a;
            rr[0] = (double)len;
            for (int64_t u = 0; u < len; ++u) {
                rr[1 + u] = (double)slab_index(A.CL(a + u));
                rr[1 + W + u] = val[a + u];
            }
        }
        DevScratch send_buf, recv_buf;
        const int64_t nrecv = F.n_slab * RW;
        PIB_HIP(hipMalloc(&send_buf.p, sizeof(double) * rec.size()));
        PIB_HIP(hipMalloc(&recv_buf.p, sizeof(double) * (size_t)std::max<int64_t>(nrecv, 1)));
        double *const d_send = send_buf.p, *const d_recv = recv_buf.p;
        PIB_HIP(hipMemcpyAsync(d_send, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));
        std::vector<double *> recv((size_t)P, nullptr);
        std::vector<int64_t> &roff = roff_f[(size_t)f];
        roff.assign((size_t)P + 1, 0);
        for (int q = 0; q < P; ++q) {
            roff[(size_t)q + 1] = roff[(size_t)q] + F.fwd.from(q);
            recv[(size_t)q] = d_recv + roff[(size_t)q] * RW;
        }
        if (roff[(size_t)P] != F.n_slab) return fail(PIB_ERR_LIB, "set_csr: internal error (the velocity boxes do not cover this rank's slab)");
        PIB_CHK(comm_exchange_v(s, rows, d_send, recv.data(), s->stream));
        got_f[(size_t)f].assign((size_t)std::max<int64_t>(nrecv, 1), 0.0);
        PIB_HIP(hipMemcpyAsync(got_f[(size_t)f].data(), d_recv, sizeof(double) * got_f[(size_t)f].size(), hipMemcpyDeviceToHost, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));
    }
    auto local_row = [&](int f, int q, int64_t t) {
        const RedistField &F = R.f[f];
        const int64_t x = F.box[6 * (size_t)q + 3], y = F.box[6 * (size_t)q + 4], kf = std::max(F.box[6 * (size_t)q + 2], F.k0);
        const int64_t i = t % x, j = (t / x) % y, k = kf + t / (x * y);
        return F.soff + (F.box[6 * (size_t)q] + i) + F.n[0] * ((F.box[6 * (size_t)q + 1] + j) + F.n[1] * (k - F.k0));
    };
    for (int f = 0; f < nf; ++f)
        for (int q = 0; q < P; ++q)
            for (int64_t t = 0; t < R.f[f].fwd.from(q); ++t)
                rp[(size_t)local_row(f, q, t) + 1] = (int64_t)got_f[(size_t)f][(size_t)((roff_f[(size_t)f][(size_t)q] + t) * RW)];
    for (int64_t l = 0; l < R.n_slab; ++l) rp[(size_t)l + 1] += rp[(size_t)l];
    std::vector<int64_t> cl((size_t)std::max<int64_t>(rp[(size_t)R.n_slab], 1));
    std::vector<double> vl((size_t)std::max<int64_t>(rp[(size_t)R.n_slab], 1));
    for (int f = 0; f < nf; ++f)
        for (int q = 0; q < P; ++q)
            for (int64_t t = 0; t < R.f[f].fwd.from(q); ++t) {
                const double *rr = &got_f[(size_t)f][(size_t)((roff_f[(size_t)f][(size_t)q] + t) * RW)];
                const int64_t l = local_row(f, q, t), len = (int64_t)rr[0];
                std::vector<std::pair<int64_t, double>> e((size_t)len);
                for (int64_t u = 0; u < len; ++u) e[(size_t)u] = {(int64_t)rr[1 + u], rr[1 + W + u]};
                std::sort(e.begin(), e.end(), [](const std::pair<int64_t, double> &a, const std::pair<int64_t, double> &b) { return a.first < b.first; });
                for (int64_t u = 0; u < len; ++u) {
                    cl[(size_t)(rp[(size_t)l] + u)] = e[(size_t)u].first;
                    vl[(size_t)(rp[(size_t)l] + u)] = e[(size_t)u].second;
                }
            }
    got_f.clear();
    // ---- the inner solver on the packed slabs: general plan, structure recovered from the entries
    PIB_CHK(create_sharing_comm(&R.inner, s->name.c_str(), s->cfg.raw.c_str(), s));
    pib_solver *in = R.inner;
    in->cfg = s->cfg;
    const int64_t in_row0 = srow0[(size_t)rank];
    std::vector<int64_t> in_ranges;
    bool in_general = false;
    PIB_CHK(classify_partition(in, R.n_slab, in_row0, n_global, rp.data(), cl.data(), nullptr, nullptr, in_ranges, &in_general));
    if (in_general) {
        PIB_CHK(upload_csr_general(in, R.n_slab, in_row0, n_global, rp.data(), cl.data(), nullptr, nullptr, vl.data(), in_ranges));
        PIB_CHK(after_set_matrix(in));
        PIB_CHK(detect_velocity_structure(in, R.n_slab, in_row0, n_global, rp.data(), cl.data(), nullptr, nullptr, vl.data()));
    }
    if (!in_general || !in->vel.valid) {  // not PetIBM's velocity operator after all (the same verdict on every rank)
        redist_release(s);
        return 0;
    }
    PIB_CHK(redist_tables(s));
    R.active = true;
    return 0;
}

}  // namespace pib
