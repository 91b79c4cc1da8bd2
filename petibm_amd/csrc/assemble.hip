// assemble.hip -- matrix ingestion and on-device operator assembly.
//
//  * upload_csr: LinSolverBase::setMatrix / AmgXSolver::setA
//    (src/linsolver/linsolveramgx.cpp:84): local rows, global columns ->
//    HBM-resident CSR with int32 ghost-shifted local columns.
//  * assemble_poisson: DBNG = D * (dt I) * G built directly in HBM from the
//    1-D mesh width arrays, reproducing -- in the same floating-point order --
//    what createDivergence(normalize=FALSE) (src/operators/createdivergence.cpp:
//    135-223), createBnHead(N=1) (src/operators/createbn.cpp:49,53),
//    createGradient(normalize=FALSE) (src/operators/creategradient.cpp:64-128)
//    and the two MatMatMult calls of applications/navierstokes/navierstokes.cpp:
//    349-356 produce, so that the 512^3 operator (9.4e8 non-zeros, 11.8 GB)
//    never crosses PCIe.  Row offsets come from a closed-form prefix count, so
//    assembly is a single streaming kernel (no scan).
#include <algorithm>
#include <limits>

#include <mutex>

#include "pib_internal.hpp"

namespace pib {

void DeviceCsr::release()
{
    if (rowptr) (void)hipFree(rowptr);
    if (col) (void)hipFree(col);
    if (val) (void)hipFree(val);
    if (dinv) (void)hipFree(dinv);
    if (code) (void)hipFree(code);
    if (dict) (void)hipFree(dict);
    code = nullptr;
    dict = nullptr;
    coded = false;
    if (pat_id) (void)hipFree(pat_id);
    if (pat_tab) (void)hipFree(pat_tab);
    if (pat_len) (void)hipFree(pat_len);
    if (pat_blk) (void)hipFree(pat_blk);
    pat_blk = nullptr;
    pat_tables = 0;
    pat_id = nullptr;
    pat_tab = nullptr;
    pat_len = nullptr;
    patterned = false;
    rowptr = nullptr;
    col = nullptr;
    val = nullptr;
    dinv = nullptr;
    n = nnz = 0;
    seg_send_prev.clear();
    seg_send_next.clear();
    seg_recv_lo.clear();
    seg_recv_hi.clear();
    segmented = false;
    if (send_idx) (void)hipFree(send_idx);
    if (send_buf) (void)hipFree(send_buf);
    send_idx = nullptr;
    send_buf = nullptr;
    general = false;
    ghost_cols.clear();
    ghost_off.clear();
    xplan = ExchangePlan();
}

// DMDA default ownership along one axis (cartesianmesh.cpp:492-538 via
// DMDACreate3d with l* = nullptr): rank r of P owns N/P + ((N % P) > r) planes.
void slab_range(int64_t nplanes, int nranks, int rank, int64_t *b, int64_t *e)
{
    int64_t beg = 0;
    for (int r = 0; r < rank; ++r) beg += nplanes / nranks + ((nplanes % nranks) > r ? 1 : 0);
    *b = beg;
    *e = beg + nplanes / nranks + ((nplanes % nranks) > rank ? 1 : 0);
}

int upload_csr(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rowptr,
               const int64_t *col, const int32_t *rowptr32, const int32_t *col32, const double *val)
{
    if (n_local < 0 || row0 < 0 || n_global < n_local) return fail(PIB_ERR_ARG_OUTOFRANGE, "set_csr: bad sizes");
    if ((rowptr == nullptr && rowptr32 == nullptr) || (col == nullptr && col32 == nullptr && n_local > 0) ||
        (val == nullptr && n_local > 0))
        return fail(PIB_ERR_ARG_NULL, "set_csr: null array");
    auto RP = [&](int64_t i) -> int64_t { return rowptr ? rowptr[i] : (int64_t)rowptr32[i]; };
    auto CL = [&](int64_t p) -> int64_t { return col ? col[p] : (int64_t)col32[p]; };
    const int64_t base = RP(0);
    const int64_t nnz = RP(n_local) - base;
    if (nnz < 0) return fail(PIB_ERR_ARG_OUTOFRANGE, "set_csr: negative nnz");
    // On several ranks a column may belong to a row across a periodic seam of the slab axis (rank 0's first plane couples
    // to the last rank's last plane and back): such a column is taken a whole vector length away, next to this rank's rows
    // -- the previous rank of rank 0 is rank P - 1 -- and the halo exchange becomes a ring (comm_setup_halo)
    const bool may_wrap = s->comm.nranks > 1;
    auto near = [&](int64_t c) -> int64_t {
        if (!may_wrap) return c;
        const int64_t lo = row0, hi = row0 + n_local - 1;
        auto dist = [&](int64_t x) { return x < lo ? lo - x : (x > hi ? x - hi : 0); };
        int64_t best = c;
        if (dist(c - n_global) < dist(best)) best = c - n_global;
        if (dist(c + n_global) < dist(best)) best = c + n_global;
        return best;
    };
    int64_t cmin = std::numeric_limits<int64_t>::max(), cmax = std::numeric_limits<int64_t>::min();
    {
        // (ranges of entries on a few host threads: the per-range minima / maxima and the first bad column are merged below)
        std::mutex mu;
        int64_t bad = -1;
        par_ranges(nnz, [&](int64_t pb, int64_t pe) {
            int64_t lo = std::numeric_limits<int64_t>::max(), hi = std::numeric_limits<int64_t>::min(), b = -1;
            for (int64_t p = pb; p < pe; ++p) {
                const int64_t c = CL(base + p);
                if (c < 0 || c >= n_global) {
                    b = c;
                    break;
                }
                const int64_t cn = near(c);
                lo = std::min(lo, cn);
                hi = std::max(hi, cn);
            }
            std::lock_guard<std::mutex> lk(mu);
            cmin = std::min(cmin, lo);
            cmax = std::max(cmax, hi);
            if (b != -1 && bad == -1) bad = b;
        });
        if (bad != -1) return fail(PIB_ERR_ARG_OUTOFRANGE, "set_csr: column %lld out of range", (long long)bad);
    }
    DeviceCsr &A = s->A;
    A.release();
    s->comm.ring = false;
    vel_stencil_release(s);
    A.n = n_local;
    A.nnz = nnz;
    A.row0 = row0;
    A.n_global = n_global;
    A.ghost_lo = (nnz > 0 && cmin < row0) ? row0 - cmin : 0;
    A.ghost_hi = (nnz > 0 && cmax > row0 + n_local - 1) ? cmax - (row0 + n_local - 1) : 0;
    if (A.ghost_lo + A.n + A.ghost_hi >= (int64_t)std::numeric_limits<int32_t>::max())
        return fail(PIB_ERR_SUP, "set_csr: local column range does not fit 32-bit indices");
    A.rp64 = nnz >= (int64_t)std::numeric_limits<int32_t>::max();
    const int64_t shift = row0 - A.ghost_lo;
    std::vector<int32_t> c32((size_t)std::max<int64_t>(nnz, 1));
    par_ranges(nnz, [&](int64_t pb, int64_t pe) {
        for (int64_t p = pb; p < pe; ++p) c32[(size_t)p] = (int32_t)(near(CL(base + p)) - shift);
    });
    // +4 entries of padding: the SpMV reads val/col in aligned pairs
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(nnz + 4)));
    PIB_MEMSET(A.col, 0, sizeof(int32_t) * (size_t)(nnz + 4));
    PIB_MEMSET(A.val, 0, sizeof(double) * (size_t)(nnz + 4));
    PIB_HIP(hipMemcpy(A.col, c32.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
    PIB_HIP(hipMemcpy(A.val, val + base, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
    if (A.rp64) {
        std::vector<int64_t> rp((size_t)n_local + 1);
        for (int64_t i = 0; i <= n_local; ++i) rp[(size_t)i] = RP(i) - base;
        PIB_HIP(hipMalloc(&A.rowptr, sizeof(int64_t) * ((size_t)n_local + 1)));
        PIB_HIP(hipMemcpy(A.rowptr, rp.data(), sizeof(int64_t) * ((size_t)n_local + 1), hipMemcpyHostToDevice));
    } else {
        std::vector<int32_t> rp((size_t)n_local + 1);
        for (int64_t i = 0; i <= n_local; ++i) rp[(size_t)i] = (int32_t)(RP(i) - base);
        PIB_HIP(hipMalloc(&A.rowptr, sizeof(int32_t) * ((size_t)n_local + 1)));
        PIB_HIP(hipMemcpy(A.rowptr, rp.data(), sizeof(int32_t) * ((size_t)n_local + 1), hipMemcpyHostToDevice));
    }
    return 0;
}

int adopt_device_csr(pib_solver *s, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col, const double *val)
{
    PIB_HIP(hipSetDevice(s->device));
    s->has_matrix = false;
    s->has_grid = false;
    gmg_release(s);
    DeviceCsr &A = s->A;
    A.release();
    vel_stencil_release(s);
    A.n = n;
    A.nnz = nnz;
    A.row0 = 0;
    A.n_global = n;
    A.ghost_lo = A.ghost_hi = 0;
    A.rp64 = false;
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(nnz + 4)));
    PIB_MEMSET(A.col, 0, sizeof(int32_t) * (size_t)(nnz + 4));
    PIB_MEMSET(A.val, 0, sizeof(double) * (size_t)(nnz + 4));
    PIB_HIP(hipMalloc(&A.rowptr, sizeof(int32_t) * ((size_t)n + 1)));
    // on the solver's own stream: a device-to-device hipMemcpy on the null stream may return before the copy has run,
    // and the non-blocking streams do not wait for the null stream (the caller has synchronised the producer of the source)
    PIB_HIP(hipMemsetAsync(A.col, 0, sizeof(int32_t) * (size_t)(nnz + 4), s->stream));
    PIB_HIP(hipMemsetAsync(A.val, 0, sizeof(double) * (size_t)(nnz + 4), s->stream));
    PIB_HIP(hipMemcpyAsync(A.col, col, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, s->stream));
    PIB_HIP(hipMemcpyAsync(A.val, val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, s->stream));
    PIB_HIP(hipMemcpyAsync(A.rowptr, rowptr, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToDevice, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    return after_set_matrix(s);
}

// ------------------------------------------------------------------------
// closed-form number of non-zeros in rows [0, g) of the natural-order
// (2*dim+1)-point operator with no-neighbour walls; per: bit d set = direction d periodic (every cell has both
// neighbours in that direction: the columns wrap, cartesianmesh.cpp:595-681)
__host__ __device__ inline int64_t nnz_before(int64_t g, int dim, int64_t nx, int64_t ny, int64_t nz, int per)
{
    const int64_t pl = nx * ny;
    int64_t c = g;                               // diagonals
    if (per & 1)
        c += 2 * g;
    else {
        c += g - (g + nx - 1) / nx;              // has i-1  (cells with i == 0: ceil(g/nx))
        c += g - g / nx;                         // has i+1  (cells with i == nx-1: floor(g/nx))
    }
    const int64_t kq = g / pl, rem = g % pl;
    if (per & 2)
        c += 2 * g;
    else {
        c += g - (kq * nx + (rem < nx ? rem : nx));  // has j-1
        const int64_t top = rem - (ny - 1) * nx;
        c += g - (kq * nx + (top > 0 ? top : 0));    // has j+1
    }
    if (dim == 3) {
        if (per & 4)
            c += 2 * g;
        else {
            c += g - (g < pl ? g : pl);              // has k-1
            const int64_t last = g - (nz - 1) * pl;
            c += g - (last > 0 ? last : 0);          // has k+1
        }
    }
    return c;
}

// One row = up to 7 entries.  Values are the face terms D_cf * (dt * G_fc'); the diagonal is what the sparse accumulator
// of MatMatMult leaves: the face terms in the column order of D's row (first one assigned, the rest added).  D's
// columns are the packed velocity indices u(i-1), u(i), v(j-1), v(j), w(k-1), w(k); on a periodic direction the minus
// face of cell 0 is velocity point n-1, which sorts AFTER the plus face (point 0).  The row's entries are stored by
// ascending column (a wrapped neighbour changes its place).
template <typename RP>
__global__ __launch_bounds__(256) void k_assemble_poisson(int dim, int64_t nx, int64_t ny, int64_t nz, int64_t row0,
                                                          int64_t n_local, int64_t ghost_lo, int64_t nnz0,
                                                          const double *__restrict__ wx, const double *__restrict__ wy,
                                                          const double *__restrict__ wz, const double *__restrict__ gx,
                                                          const double *__restrict__ gy, const double *__restrict__ gz,
                                                          int pinned, int per, int ring, RP *__restrict__ rowptr,
                                                          int32_t *__restrict__ col, double *__restrict__ val)
{
    const int64_t pl = nx * ny;
    const bool px = per & 1, py = per & 2, pz = per & 4;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= n_local; r += (int64_t)gridDim.x * 256) {
        const int64_t g = row0 + r;
        int64_t p = nnz_before(g, dim, nx, ny, nz, per) - nnz0;
        rowptr[r] = (RP)p;
        if (r == n_local) break;
        const int64_t i = g % nx, j = (g / nx) % ny, k = g / pl;
        const double ax = wy[j] * wz[k];  // dL[0][1][j]*dL[0][2][k]  (createdivergence.cpp:140-143)
        const double ay = wx[i] * wz[k];  // dL[1][0][i]*dL[1][2][k]
        const double az = wx[i] * wy[j];  // dL[2][0][i]*dL[2][1][j]
        // minus / plus face of each direction: present?, value, column offset (wrapped on a periodic direction)
        bool has[6];
        double o[6];
        int64_t off[6];
        has[0] = i > 0 || px;
        has[1] = i < nx - 1 || px;
        has[2] = j > 0 || py;
        has[3] = j < ny - 1 || py;
        has[4] = (dim == 3) && (k > 0 || pz);
        has[5] = (dim == 3) && (k < nz - 1 || pz);
        o[0] = has[0] ? ax * gx[i > 0 ? i - 1 : nx - 1] : 0.0;
        o[1] = has[1] ? ax * gx[i] : 0.0;
        o[2] = has[2] ? ay * gy[j > 0 ? j - 1 : ny - 1] : 0.0;
        o[3] = has[3] ? ay * gy[j] : 0.0;
        o[4] = has[4] ? az * gz[k > 0 ? k - 1 : nz - 1] : 0.0;
        o[5] = has[5] ? az * gz[k] : 0.0;
        off[0] = i > 0 ? -1 : nx - 1;
        off[1] = i < nx - 1 ? 1 : -(nx - 1);
        // ring: the slab axis (y in 2-D, z in 3-D) wraps through the ghost pads, i.e. the plain neighbour offsets
        off[2] = (j > 0 || (ring && dim == 2)) ? -nx : (ny - 1) * nx;
        off[3] = (j < ny - 1 || (ring && dim == 2)) ? nx : -(ny - 1) * nx;
        off[4] = (k > 0 || ring) ? -pl : (nz - 1) * pl;
        off[5] = (k < nz - 1 || ring) ? pl : -(nz - 1) * pl;
        // diagonal: first contribution assigned, the rest added (sparse accumulator)
        double d = 0.0;
        bool first = true;
#define PIB_ACC(q)                                 \
    if (has[q]) {                                  \
        if (first) { d = -(o[q]); first = false; } \
        else d = d + (-(o[q]));                    \
    }
        if (i == 0 && px) { PIB_ACC(1) PIB_ACC(0) } else { PIB_ACC(0) PIB_ACC(1) }
        if (j == 0 && py) { PIB_ACC(3) PIB_ACC(2) } else { PIB_ACC(2) PIB_ACC(3) }
        if (k == 0 && pz) { PIB_ACC(5) PIB_ACC(4) } else { PIB_ACC(4) PIB_ACC(5) }
#undef PIB_ACC
        const int64_t lc = r + ghost_lo;  // local column of the diagonal
        const bool row_pinned = pinned && g == 0;
        // entries by ascending column offset: insertion sort of (offset, value) with the diagonal at offset 0
        int64_t eo[7];
        double ev[7];
        int ne = 1;
        eo[0] = 0;
        ev[0] = row_pinned ? 1.0 : d;
        for (int q = 0; q < 6; ++q) {
            if (!has[q]) continue;
            int64_t gc = g + off[q];  // global column (ring: the ghost pads stand for the planes at the other end)
            if (gc < 0) gc += pl * nz;
            else if (gc >= pl * nz) gc -= pl * nz;
            const double v = (row_pinned || (pinned && gc == 0)) ? 0.0 : o[q];
            int t = ne++;
            while (t > 0 && eo[t - 1] > off[q]) {
                eo[t] = eo[t - 1];
                ev[t] = ev[t - 1];
                --t;
            }
            eo[t] = off[q];
            ev[t] = v;
        }
        for (int t = 0; t < ne; ++t) {
            col[p] = (int32_t)(lc + eo[t]);
            val[p] = ev[t];
            ++p;
        }
    }
}

int upload_vec(const std::vector<double> &h, double **d)
{
    PIB_HIP(hipMalloc(d, sizeof(double) * std::max<size_t>(h.size(), 1)));
    if (!h.empty()) PIB_HIP(hipMemcpy(*d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

int assemble_poisson(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], double dt, int nullspace)
{
    if (dim != 2 && dim != 3) return fail(PIB_ERR_ARG_OUTOFRANGE, "assemble_poisson: dim must be 2 or 3");
    const int64_t nx = n[0], ny = n[1], nz = (dim == 3) ? n[2] : 1;
    if (nx < 2 || ny < 2 || (dim == 3 && nz < 2)) return fail(PIB_ERR_ARG_SIZ, "assemble_poisson: need >= 2 cells per direction");
    if (w[0] == nullptr || w[1] == nullptr || (dim == 3 && w[2] == nullptr)) return fail(PIB_ERR_ARG_NULL, "assemble_poisson: null widths");
    // host: 1-D arrays.  g_d[s] = dt * (1 / (0.5*(w[s+1] + w[s])))
    //   velocity-cell width: cartesianmesh.cpp:246-258 (adjacent_difference with 0.5*(x+y))
    //   G value 1/dL:        creategradient.cpp:70-86 ; BN = dt*I: createbn.cpp:49
    std::vector<double> hw[3], hg[3];
    int per = 0;
    for (int d = 0; d < 3; ++d) {
        const int64_t nd = (d < dim) ? n[d] : 1;
        hw[d].resize((size_t)nd);
        for (int64_t q = 0; q < nd; ++q) hw[d][(size_t)q] = (d < dim) ? w[d][q] : 1.0;
        const bool wrap = d < dim && s->periodic[d] != 0;
        if (wrap) per |= 1 << d;
        if (wrap && nd < 3) return fail(PIB_ERR_SUP, "assemble_poisson: a periodic direction needs >= 3 cells");
        hg[d].resize((size_t)std::max<int64_t>(nd - 1, 0) + (wrap ? 1 : 0));
        for (int64_t q = 0; q + 1 < nd; ++q) {
            const double dl = 0.5 * (hw[d][(size_t)q + 1] + hw[d][(size_t)q]);
            const double v = 1.0 / dl;
            hg[d][(size_t)q] = dt * v;
        }
        if (wrap) {  // dL[d][d] of velocity point n-1 on a periodic axis: 0.5*(w[0] + w[n-1])  (cartesianmesh.cpp:259-266)
            const double dl = 0.5 * (hw[d][0] + hw[d][(size_t)nd - 1]);
            const double v = 1.0 / dl;
            hg[d][(size_t)nd - 1] = dt * v;
        }
    }
    const int P = s->comm.nranks, r = s->comm.rank;
    // periodic slab axis on several ranks: rank 0 and rank P-1 are neighbours, every rank has both ghost planes and the
    // wrapped columns of the outer planes point into them
    const bool ring = P > 1 && (per & (1 << (dim - 1)));
    s->comm.ring = ring;
    const int64_t nlast = (dim == 3) ? nz : ny;
    const int64_t plane = (dim == 3) ? nx * ny : nx;
    int64_t k0, k1;
    slab_range(nlast, P, r, &k0, &k1);
    if (k1 - k0 < 1) return fail(PIB_ERR_SUP, "assemble_poisson: a rank owns no plane (%lld planes on %d ranks)", (long long)nlast, P);

    DeviceCsr &A = s->A;
    A.release();
    vel_stencil_release(s);
    A.n = (k1 - k0) * plane;
    A.row0 = k0 * plane;
    A.n_global = nx * ny * nz;
    A.ghost_lo = (r > 0 || ring) ? plane : 0;
    A.ghost_hi = (r < P - 1 || ring) ? plane : 0;
    const int64_t nnz0 = nnz_before(A.row0, dim, nx, ny, nz, per);
    A.nnz = nnz_before(A.row0 + A.n, dim, nx, ny, nz, per) - nnz0;
    A.rp64 = A.nnz >= (int64_t)std::numeric_limits<int32_t>::max();
    if (A.ghost_lo + A.n + A.ghost_hi >= (int64_t)std::numeric_limits<int32_t>::max())
        return fail(PIB_ERR_SUP, "assemble_poisson: local slab too large for 32-bit column indices");
    PIB_HIP(hipMalloc(&A.rowptr, (A.rp64 ? 8 : 4) * ((size_t)A.n + 1)));
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(A.nnz + 4)));  // +4: the SpMV reads aligned pairs
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(A.nnz + 4)));
    PIB_HIP(hipMemsetAsync(A.col + A.nnz, 0, sizeof(int32_t) * 4, s->stream));
    PIB_HIP(hipMemsetAsync(A.val + A.nnz, 0, sizeof(double) * 4, s->stream));
    double *dw[3] = {nullptr, nullptr, nullptr}, *dg[3] = {nullptr, nullptr, nullptr};
    for (int d = 0; d < 3; ++d) {
        PIB_CHK(upload_vec(hw[d], &dw[d]));
        PIB_CHK(upload_vec(hg[d], &dg[d]));
    }
    const int pinned = (nullspace == PIB_NULLSPACE_PINNED) ? 1 : 0;
    const int nb = (int)std::min<int64_t>(8192, (A.n + 1 + 255) / 256);
    if (A.rp64)
        hipLaunchKernelGGL(k_assemble_poisson<int64_t>, dim3(nb), dim3(256), 0, s->stream, dim, nx, ny, nz, A.row0, A.n,
                           A.ghost_lo, nnz0, dw[0], dw[1], dw[2], dg[0], dg[1], dg[2], pinned, per, ring ? 1 : 0, (int64_t *)A.rowptr, A.col,
                           A.val);
    else
        hipLaunchKernelGGL(k_assemble_poisson<int32_t>, dim3(nb), dim3(256), 0, s->stream, dim, nx, ny, nz, A.row0, A.n,
                           A.ghost_lo, nnz0, dw[0], dw[1], dw[2], dg[0], dg[1], dg[2], pinned, per, ring ? 1 : 0, (int32_t *)A.rowptr, A.col,
                           A.val);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipStreamSynchronize(s->stream));
    for (int d = 0; d < 3; ++d) {
        (void)hipFree(dw[d]);
        (void)hipFree(dg[d]);
    }
    // the caller registers the structure (grid_register) AFTER the halo plan exists: the hint
    // verification exchanges ghost planes.
    for (int d = 0; d < 3; ++d) {
        s->asm_w[d] = hw[d];
        s->asm_g[d] = hg[d];
    }
    s->asm_dt = dt;
    return 0;
}

}  // namespace pib

// ============================================================================
// Velocity operator A = I/dt - c nu L assembled directly in HBM (K9, a-8, a-9).
//
//   L: petibm::operators::createLaplacian (src/operators/createlaplacian.cpp:108-263) on the packed
//      (u,v[,w]) ordering of one rank (src/mesh/cartesianmesh.cpp:741-779): per velocity point and direction
//      1/(dLNeg*dLSelf), 1/(dLPos*dLSelf) with dLSelf = dL[f][dir][self] and dLNeg/dLPos = coordinate
//      differences including the ghost coordinates (:134-148); diagonal = -accumulate(values, 0.0) in stencil
//      order x-,x+,y-,y+,z-,z+ (:151); ghost columns dropped; then L[row,row] += coeff*a0 per ghost (:232-243).
//   A: MatDuplicate(L); MatScale(A, -c nu); MatShift(A, 1/dt)   (applications/navierstokes/navierstokes.cpp:342-344)
//   mesh arithmetic: CartesianMesh::createVelocityMesh (src/mesh/cartesianmesh.cpp:213-355), non-periodic.
// Same floating-point evaluation order as the oracle (oracle/operators.py): bit-identical entries.
// Single rank only (the packed ordering interleaves the fields per rank: its halo is not a contiguous plane).
namespace pib {

struct FieldDev {
    int64_t n[3];        // points of this field (global)
    int64_t row_off;     // first packed LOCAL row of the field's block
    int64_t nnz_off;     // first nnz of the field's block
    // slab of this rank along the last axis: planes [kb, ke) of the field; the neighbours' planes kb-1 / ke live in the
    // ghost pads at ghost_lo_off / ghost_hi_off (local column indices: [ghost_lo | owned | ghost_hi])
    int64_t kb, ke, col_base, ghost_lo_off, ghost_hi_off;
    const double *dl[3];     // dL[f][d], index s+1 (ghost at 0)
    const double *co[3];     // coord[f][d], index s+1
    double a0[6];            // ghost coefficient per boundary location (0 where periodic / unused)
    int per;                 // bit d: direction d periodic (neighbour indices wrap, no ghost fold)
    int ring;                // the slab axis is periodic AND distributed: its wrap goes through the ghost pads (rank 0's
                             // lower neighbour plane is the last rank's top plane)
};


// Coordinates and widths (with one ghost entry each side, index s+1) of the velocity fields:
// CartesianMesh::createPressureMesh / createVertexMesh / createVelocityMesh (src/mesh/cartesianmesh.cpp:136-355),
// including the periodic variants (:259-266 one more point of component d along a periodic d, :300-318 ghost entries
// taken from the other end).  Same evaluation order as oracle/mesh.py.
void velocity_mesh_arrays(int dim, const int64_t n[3], const double *const w[3], const double mn[3], const double mx[3],
                          const int per[3], std::vector<double> hdl[3][3], std::vector<double> hco[3][3], int64_t fn[3][3],
                          const MeshWindow *win)
{
    std::vector<double> c3[3], c4[3];
    for (int d = 0; d < dim; ++d) {
        const bool windowed = win != nullptr && win->active && win->axis == d;
        const int64_t nd = windowed ? win->n_global : n[d];
        const double *wd = windowed ? win->w_global : w[d];
        const double m0 = windowed ? win->lo : mn[d];
        c3[d].resize((size_t)nd);
        c4[d].resize((size_t)nd + 1);
        double run = 0.0;
        c4[d][0] = 0.0 + m0;
        for (int64_t q = 0; q < nd; ++q) {
            run = (q == 0) ? wd[0] : run + wd[q];  // std::partial_sum
            c3[d][(size_t)q] = (run + m0) - 0.5 * wd[q];
            c4[d][(size_t)q + 1] = run + m0;
        }
        if (windowed) {  // the window's planes of the whole mesh's coordinates
            std::vector<double> a((size_t)n[d]), b((size_t)n[d] + 1);
            const double period = win->hi - win->lo;
            auto turn = [&](int64_t k) { return (double)((k >= 0) ? k / nd : -((-k + nd - 1) / nd)); };
            for (int64_t q = 0; q <= n[d]; ++q) {
                const int64_t k = win->first + q, km = ((k % nd) + nd) % nd;
                if (q < n[d]) a[(size_t)q] = (turn(k) == 0.0) ? c3[d][(size_t)km] : c3[d][(size_t)km] + turn(k) * period;
                // (vertex nd of the whole mesh is its upper end, not vertex 0 a period on)
                b[(size_t)q] = (k >= 0 && k <= nd) ? c4[d][(size_t)k] : c4[d][(size_t)km] + turn(k) * period;
            }
            c3[d].swap(a);
            c4[d].swap(b);
        }
    }
    for (int f = 0; f < 3; ++f)
        for (int d = 0; d < 3; ++d) {
            fn[f][d] = 1;
            hdl[f][d].clear();
            hco[f][d].clear();
            if (f >= dim || d >= dim) continue;
            const int64_t n3 = n[d];
            const bool wrap = per != nullptr && per[d] != 0;
            if (d == f) {
                const int64_t nf = wrap ? n3 : n3 - 1;
                fn[f][d] = nf;
                hco[f][d] = c4[d];  // n3+1 entries: vertices, ghosts = the walls
                hdl[f][d].assign((size_t)n3 + 1, 0.0);
                hdl[f][d][0] = w[d][0];
                for (int64_t q = 1; q < n3; ++q) hdl[f][d][(size_t)q] = 0.5 * (w[d][q] + w[d][q - 1]);
                if (wrap) {
                    hdl[f][d][(size_t)n3] = 0.5 * (w[d][0] + w[d][n3 - 1]);
                    hdl[f][d][0] = hdl[f][d][(size_t)n3];
                    hdl[f][d].push_back(hdl[f][d][1]);
                    hco[f][d].push_back(mx[d] + w[d][0]);
                } else
                    hdl[f][d][(size_t)n3] = w[d][n3 - 1];
            } else {
                fn[f][d] = n3;
                hco[f][d].assign((size_t)n3 + 2, 0.0);
                hdl[f][d].assign((size_t)n3 + 2, 0.0);
                for (int64_t q = 0; q < n3; ++q) {
                    hco[f][d][(size_t)q + 1] = c3[d][(size_t)q];
                    hdl[f][d][(size_t)q + 1] = w[d][q];
                }
                if (wrap) {
                    hco[f][d][0] = mn[d] - w[d][n3 - 1] / 2.0;
                    hco[f][d][(size_t)n3 + 1] = mx[d] + w[d][0] / 2.0;
                    hdl[f][d][0] = w[d][n3 - 1];
                    hdl[f][d][(size_t)n3 + 1] = w[d][0];
                } else {
                    hco[f][d][0] = mn[d] - w[d][0] / 2.0;
                    hco[f][d][(size_t)n3 + 1] = mx[d] + w[d][n3 - 1] / 2.0;
                    hdl[f][d][0] = w[d][0];
                    hdl[f][d][(size_t)n3 + 1] = w[d][n3 - 1];
                }
            }
        }
}

template <typename RP>
__global__ __launch_bounds__(256) void k_assemble_velocity(int dim, FieldDev F, double scale, double shift,
                                                           RP *__restrict__ rowptr, int32_t *__restrict__ col,
                                                           double *__restrict__ val, int last_field)
{
    // the slab axis is the last one: z in 3-D, y in 2-D (planes of a 2-D field are its grid lines)
    const int64_t nx = F.n[0], ny = F.n[1], nz = F.n[2];
    const int64_t pl = (dim == 3) ? nx * ny : nx;  // entries per plane of the slab axis
    const int64_t g0 = pl * F.kb, nf = pl * (F.ke - F.kb);
    const int64_t base_nnz = nnz_before(g0, dim, nx, ny, nz, F.per);
    for (int64_t lr = (int64_t)blockIdx.x * 256 + threadIdx.x; lr <= nf; lr += (int64_t)gridDim.x * 256) {
        const int64_t r = g0 + lr;  // row in the field's global natural order
        int64_t p = F.nnz_off + (nnz_before(r, dim, nx, ny, nz, F.per) - base_nnz);
        if (lr < nf || last_field) rowptr[F.row_off + lr] = (RP)p;
        if (lr == nf) break;
        const int64_t ijk[3] = {r % nx, (r / nx) % ny, r / (nx * ny)};
        double v[6] = {0, 0, 0, 0, 0, 0};
        bool interior[6] = {false, false, false, false, false, false};
        double acc = 0.0;
        for (int d = 0; d < dim; ++d) {
            const int64_t sidx = ijk[d];
            const double dLSelf = F.dl[d][sidx + 1];
            const double dLNeg = F.co[d][sidx + 1] - F.co[d][sidx];
            const double dLPos = F.co[d][sidx + 2] - F.co[d][sidx + 1];
            v[2 * d] = 1.0 / (dLNeg * dLSelf);
            v[2 * d + 1] = 1.0 / (dLPos * dLSelf);
            const bool wrap = (F.per >> d) & 1;
            interior[2 * d] = sidx > 0 || wrap;
            interior[2 * d + 1] = sidx < F.n[d] - 1 || wrap;
            acc = acc + v[2 * d];
            acc = acc + v[2 * d + 1];
        }
        double diag = -acc;
        for (int q = 0; q < 2 * dim; ++q)
            if (!interior[q]) {
                const double t = v[q] * F.a0[q];
                if (t != 0.0) diag = diag + t;  // MAT_IGNORE_ZERO_ENTRIES: a zero fold is not added
            }
        const int64_t lc = F.col_base + F.row_off + lr;  // local column of this row's own entry
        const int64_t st[3] = {1, nx, nx * ny};
        const int sd = dim - 1;
        const int64_t ks = ijk[sd], inplane = lr % pl;
        // columns ascending: z-, y-, x-, diag, x+, y+, z+ (a neighbour rank's plane sits in a ghost pad: the low pad
        // precedes and the high pad follows every owned column, so the order is kept); a neighbour across a periodic
        // seam has the wrapped column and takes its sorted place
        int64_t ec[7];
        double ev[7];
        int ne = 1;
        ec[0] = lc;
        ev[0] = diag * scale + shift;
        for (int q = 0; q < 2 * dim; ++q) {
            if (!interior[q]) continue;
            const int d = q >> 1;
            int64_t c;
            const bool via_pads = d == sd && F.ring;
            if (!(q & 1)) {
                if (ijk[d] == 0 && !via_pads) c = lc + (F.n[d] - 1) * st[d];
                else c = (d == sd && ks - 1 < F.kb) ? F.ghost_lo_off + inplane : lc - st[d];
            } else {
                if (ijk[d] == F.n[d] - 1 && !via_pads) c = lc - (F.n[d] - 1) * st[d];
                else c = (d == sd && ks + 1 >= F.ke) ? F.ghost_hi_off + inplane : lc + st[d];
            }
            int t = ne++;
            while (t > 0 && ec[t - 1] > c) {
                ec[t] = ec[t - 1];
                ev[t] = ev[t - 1];
                --t;
            }
            ec[t] = c;
            ev[t] = v[q] * scale;
        }
        for (int t = 0; t < ne; ++t) {
            col[p] = (int32_t)ec[t];
            val[p] = ev[t];
            ++p;
        }
    }
}

int assemble_velocity(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double mn[3],
                      const double mx[3], const double a0[18], double dt, double coeff_nu)
{
    if (dim != 2 && dim != 3) return fail(PIB_ERR_ARG_OUTOFRANGE, "assemble_velocity: dim must be 2 or 3");
    for (int d = 0; d < dim; ++d)
        if (n[d] < 2 || w[d] == nullptr) return fail(PIB_ERR_ARG_SIZ, "assemble_velocity: need >= 2 cells per direction");
    // ---- host mesh arithmetic (cartesianmesh.cpp:136-355, non-periodic)
    std::vector<double> hdl[3][3], hco[3][3];
    int64_t fn[3][3];
    int per = 0;
    for (int d = 0; d < dim; ++d)
        if (s->periodic[d]) {
            if (n[d] < 3) return fail(PIB_ERR_SUP, "assemble_velocity: a periodic direction needs >= 3 cells");
            per |= 1 << d;
        }
    velocity_mesh_arrays(dim, n, w, mn, mx, s->periodic, hdl, hco, fn, &s->mesh_window);
    // ---- sizes.  Decomposition: slabs along the last axis with the pressure grid's ownership (the velocity DMDAs
    // reuse the pressure process grid, cartesianmesh.cpp:516-535): rank r owns the planes of its pressure cells; the
    // component along the slab axis has one plane fewer, taken from the last rank.  Each rank's vector is the packed
    // [u-slab | v-slab | w-slab] of the reference's DMComposite (cartesianmesh.cpp:740-779).
    const int P = s->comm.nranks, rank = s->comm.rank, sd = dim - 1;
    // a periodic slab axis on several ranks: rank 0 and rank P - 1 are neighbours, every rank has both ghost pads and the
    // exchange is a ring (as in assemble_poisson)
    const bool ring = P > 1 && (per & (1 << sd));
    int64_t pk0, pk1;
    slab_range(n[sd], P, rank, &pk0, &pk1);
    int64_t rows = 0, nnz = 0, row_off[3] = {0, 0, 0}, nnz_off[3] = {0, 0, 0}, kb[3] = {0, 0, 0}, ke[3] = {0, 0, 0},
            pl[3] = {0, 0, 0}, glo[3] = {0, 0, 0}, ghi[3] = {0, 0, 0};
    int64_t ghost_lo = 0, ghost_hi = 0;
    for (int f = 0; f < dim; ++f) {
        kb[f] = pk0;
        ke[f] = std::min(pk1, fn[f][sd]);
        if (P > 1 && ke[f] - kb[f] < 1)
            return fail(PIB_ERR_SUP, "assemble_velocity: rank %d owns no plane of velocity component %d (need >= 2 pressure planes "
                                     "per rank)", rank, f);
        pl[f] = (dim == 3) ? fn[f][0] * fn[f][1] : fn[f][0];
        row_off[f] = rows;
        nnz_off[f] = nnz;
        rows += pl[f] * (ke[f] - kb[f]);
        nnz += nnz_before(pl[f] * ke[f], dim, fn[f][0], fn[f][1], fn[f][2], per) -
               nnz_before(pl[f] * kb[f], dim, fn[f][0], fn[f][1], fn[f][2], per);
        if (rank > 0 || ring) {
            glo[f] = ghost_lo;
            ghost_lo += pl[f];
        }
        if (rank < P - 1 || ring) {
            ghi[f] = ghost_hi;
            ghost_hi += pl[f];
        }
    }
    int64_t n_global = 0, row0 = 0;
    for (int q = 0; q < P; ++q) {
        int64_t b, e, nq = 0;
        slab_range(n[sd], P, q, &b, &e);
        for (int f = 0; f < dim; ++f) nq += pl[f] * (std::min(e, fn[f][sd]) - b);
        if (q < rank) row0 += nq;
        n_global += nq;
    }
    DeviceCsr &A = s->A;
    A.release();
    s->comm.ring = ring;
    vel_stencil_release(s);
    A.n = rows;
    A.row0 = row0;
    A.n_global = n_global;
    A.ghost_lo = ghost_lo;
    A.ghost_hi = ghost_hi;
    A.nnz = nnz;
    A.rp64 = nnz >= (int64_t)std::numeric_limits<int32_t>::max();
    if (ghost_lo + rows + ghost_hi >= (int64_t)std::numeric_limits<int32_t>::max())
        return fail(PIB_ERR_SUP, "assemble_velocity: more than 2^31 local columns on one GPU");
    if (P > 1) {
        // one plane of every component to each neighbour, received back to back into the ghost pads
        A.segmented = true;
        for (int f = 0; f < dim; ++f) {
            if (rank > 0 || ring) {
                A.seg_send_prev.push_back({row_off[f], pl[f]});
                A.seg_recv_lo.push_back(pl[f]);
            }
            if (rank < P - 1 || ring) {
                A.seg_send_next.push_back({row_off[f] + pl[f] * (ke[f] - kb[f] - 1), pl[f]});
                A.seg_recv_hi.push_back(pl[f]);
            }
        }
    }
    PIB_HIP(hipMalloc(&A.rowptr, (A.rp64 ? 8 : 4) * ((size_t)rows + 1)));
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(nnz + 4)));
    PIB_HIP(hipMemsetAsync(A.col + nnz, 0, sizeof(int32_t) * 4, s->stream));
    PIB_HIP(hipMemsetAsync(A.val + nnz, 0, sizeof(double) * 4, s->stream));
    const double scale = -coeff_nu, shift = 1.0 / dt;
    std::vector<double *> tofree;
    for (int f = 0; f < dim; ++f) {
        FieldDev F;
        for (int d = 0; d < 3; ++d) {
            F.n[d] = fn[f][d];
            F.dl[d] = F.co[d] = nullptr;
            if (d < dim) {
                double *p1 = nullptr, *p2 = nullptr;
                PIB_CHK(upload_vec(hdl[f][d], &p1));
                PIB_CHK(upload_vec(hco[f][d], &p2));
                F.dl[d] = p1;
                F.co[d] = p2;
                tofree.push_back(p1);
                tofree.push_back(p2);
            }
        }
        F.row_off = row_off[f];
        F.nnz_off = nnz_off[f];
        F.kb = kb[f];
        F.ke = ke[f];
        F.col_base = ghost_lo;
        F.ghost_lo_off = glo[f];
        F.ghost_hi_off = ghost_lo + rows + ghi[f];
        for (int q = 0; q < 6; ++q) F.a0[q] = a0[6 * f + q];
        F.per = per;
        F.ring = ring ? 1 : 0;
        const int64_t nf = pl[f] * (ke[f] - kb[f]);
        const int nb = (int)std::min<int64_t>(8192, (nf + 1 + 255) / 256);
        const int last = (f == dim - 1) ? 1 : 0;
        if (A.rp64)
            hipLaunchKernelGGL(k_assemble_velocity<int64_t>, dim3(nb), dim3(256), 0, s->stream, dim, F, scale, shift,
                               (int64_t *)A.rowptr, A.col, A.val, last);
        else
            hipLaunchKernelGGL(k_assemble_velocity<int32_t>, dim3(nb), dim3(256), 0, s->stream, dim, F, scale, shift,
                               (int32_t *)A.rowptr, A.col, A.val, last);
        PIB_HIP(hipGetLastError());
    }
    PIB_HIP(hipStreamSynchronize(s->stream));
    for (double *p : tofree) (void)hipFree(p);
    {
        // the operator's structure for the matrix-free product (velstencil.hip): the quotients the assembly kernel
        // evaluates per entry, once per field, direction and index (IEEE division rounds identically on the host).  On
        // several ranks: this rank's slab, the neighbours' planes in the ghost pads (the slab axis never wraps locally)
        VelStencil &V = s->vel;
        V.dim = dim;
        V.per = (P > 1) ? (per & ~(1 << sd)) : per;
        V.scale = scale;
        V.shift = shift;
        V.slab_axis = (P > 1) ? sd : -1;
        V.has_lo = P > 1 && (rank > 0 || ring);
        V.has_hi = P > 1 && (rank < P - 1 || ring);
        for (int f = 0; f < dim; ++f) {
            V.off[f] = row_off[f];
            V.pad_lo[f] = -ghost_lo + glo[f];
            V.pad_hi[f] = rows + ghi[f];
            for (int q = 0; q < 6; ++q) V.a0[f][q] = a0[6 * f + q];
            for (int d = 0; d < 3; ++d) V.n[f][d] = fn[f][d];
            if (P > 1) V.n[f][sd] = ke[f] - kb[f];
            for (int d = 0; d < dim; ++d) {
                const int64_t nfd = fn[f][d];
                std::vector<double> tn((size_t)nfd), tp((size_t)nfd);
                for (int64_t q = 0; q < nfd; ++q) {
                    const double dLSelf = hdl[f][d][(size_t)q + 1];
                    const double dLNeg = hco[f][d][(size_t)q + 1] - hco[f][d][(size_t)q];
                    const double dLPos = hco[f][d][(size_t)q + 2] - hco[f][d][(size_t)q + 1];
                    tn[(size_t)q] = 1.0 / (dLNeg * dLSelf);
                    tp[(size_t)q] = 1.0 / (dLPos * dLSelf);
                }
                double *p1 = nullptr, *p2 = nullptr;
                PIB_CHK(upload_vec(tn, &p1));
                PIB_CHK(upload_vec(tp, &p2));
                V.owned.push_back(p1);
                V.owned.push_back(p2);
                const int64_t first = (P > 1 && d == sd) ? kb[f] : 0;  // local index 0 <-> the slab's first plane
                V.lneg[f][d] = p1 + first;
                V.lpos[f][d] = p2 + first;
            }
        }
        V.valid = true;
    }
    return 0;
}

}  // namespace pib
