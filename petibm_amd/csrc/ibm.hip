// ibm.hip -- SURVEY.md 8f-3: immersed-boundary operators and the force system of the decoupled IBPM, on the device.
//
//   createDelta  (src/operators/createdelta.cpp:34-208)   Delta[row(point, dof), velocity point of component dof]
//                = prod_d kernel(X_d - x_d, h_d) over the +-window cells around the point's background cell
//                (src/body/singlebodypoints.cpp:90-113); zero entries are not stored (MAT_IGNORE_ZERO_ENTRIES)
//   kernels      (src/misc/delta.cpp:17-43)                Roma et al. 1999 (window 2), Peskin 2002 (window 3)
//   E = Delta R MHat, H = Delta^T, BNH = BN H (BN = dt I), EBNH = E BNH
//                (applications/decoupledibpm/decoupledibpm.cpp:141-216, creatediagmatrix.cpp:88-171)
//   per step     rhs1 += H f ; rhsf = -E u ; EBNH df = rhsf ; u += BNH df ; f += df   (decoupledibpm.cpp:105-131,232-313)
//
// Layouts: Delta / E share one CSR over the Lagrangian rows (columns ascend: the reference's k, j, i loops); H is
// the same entries sorted by (velocity point, row) with a compressed row index -- only the few thousand velocity
// points under the kernels' support are stored, so spreading touches nothing else of the 10^5..10^8-entry field;
// EBNH is a CSR built by intersecting the sorted column lists of row pairs whose background cells are within two
// windows (same summation order as a row-wise sparse product).  Every sum runs in a fixed order: results are
// bit-identical to the oracle (oracle/ibm.py), which follows PETSc's MatMult / MatMultAdd / MatMatMult order.
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "ns_internal.hpp"

namespace pib {

constexpr int IB_MAXW = 7;  // 2*window + 1, Peskin

struct IbDev {
    int dim, window, kernel;  // kernel: 0 Roma et al. 1999, 1 Peskin 2002
    int64_t npts, nf;         // Lagrangian points of all bodies, force unknowns = npts * dim
    const double *X;          // [npts*dim] coordinates
    const int *ijk;           // [npts*dim] background (pressure) cell
    const double *h;          // [npts*dim] kernel widths (uniform within a body: createdelta.cpp:69-76)
    // z-slabs: the engine's mesh is the rank's extended slab; a row keeps the velocity points this rank OWNS only (planes
    // [own_lo, own_hi) of component dof along direction sd), so that sums over velocity points -- E u, E BN H -- add up
    // over the ranks without counting a point twice.  sd = -1: one rank, every point.
    int sd;
    int own_lo[3], own_hi[3];
};

struct IbState {
    IbDev I;
    int nbodies = 0;
    std::vector<int64_t> npts;
    pib_solver *fsol = nullptr;
    // Delta / E
    int32_t *rowptr = nullptr, *col = nullptr;
    double *val = nullptr, *eval = nullptr;
    int64_t nnz = 0;
    // H = Delta^T, compressed rows
    int32_t *hcols = nullptr, *hptr = nullptr, *hrow = nullptr;
    double *hval = nullptr;
    int64_t hrows = 0;
    // EBNH
    int32_t *c_rowptr = nullptr, *c_col = nullptr;
    double *c_val = nullptr;
    int64_t c_nnz = 0;
    // BN order > 1: BNH = BN H assembled (UN rows, most of them empty); order 1 applies dt * H from the compressed rows
    int32_t *bnh_rowptr = nullptr, *bnh_col = nullptr;
    double *bnh_val = nullptr;
    int64_t bnh_nnz = 0;
    // ... on z-slabs BN is applied term by term with the engine's halo exchanges (navierstokes.hip: ns_bn_series_slab): bn_t is the
    // extended-slab work vector of y += BN H x, e_nf / col_nf the unit vector and the column of the dense EBNH assembly
    double *bn_t = nullptr, *e_nf = nullptr, *col_nf = nullptr;
    double *f = nullptr, *df = nullptr, *rhsf = nullptr;
    double *ub = nullptr;     // prescribed velocity of the Lagrangian points (RigidKinematicsSolver: rhsf = UB - E u)
    bool moving = false;
    double dt = 0.0;
    // coupled IBPM (applications/ibpm): work vectors of the Schur-complement operator
    bool coupled = false;
    double *t_un = nullptr, *t_un2 = nullptr;  // [UN]
    double *g_nf = nullptr, *y_nf = nullptr, *r2 = nullptr;  // [nf]
    std::vector<void *> owned;      // operators of the current body position (released by every re-assembly)
    std::vector<void *> persistent; // forces and friends: live as long as the bodies
    // A moving body re-assembles its operators every time step (rigidkinematics.cpp:75-79): some thirty device allocations of
    // nearly the same sizes as the step before.  hipFree synchronises the device and hipMalloc costs a fraction of a millisecond:
    // together they were most of the 7 ms a config-5-size plate spent "moving the body".  Blocks released by a re-assembly wait
    // in `pool` (capacity, pointer) and the next one takes the smallest block that fits (at most twice the request).
    std::vector<std::pair<size_t, void *>> pool;
    std::vector<std::pair<void *, size_t>> owned_cap;  // capacity of the blocks in `owned` that came through ib_malloc
};

// a device block of at least `bytes`: from the pool when one fits, else a new one (rounded up so that next step's slightly larger
// request still fits); tracked in ib->owned
static int ib_malloc(IbState *ib, void **p, size_t bytes)
{
    bytes = std::max<size_t>(bytes, 16);
    size_t best = (size_t)-1;
    int at = -1;
    for (int i = 0; i < (int)ib->pool.size(); ++i) {
        const size_t cap = ib->pool[(size_t)i].first;
        if (cap >= bytes && cap <= 2 * bytes + 65536 && cap < best) {
            best = cap;
            at = i;
        }
    }
    size_t cap = 0;
    if (at >= 0) {
        *p = ib->pool[(size_t)at].second;
        cap = ib->pool[(size_t)at].first;
        ib->pool.erase(ib->pool.begin() + at);
    } else {
        cap = bytes + bytes / 8 + 256;  // head room for the next position of the body
        PIB_HIP(hipMalloc(p, cap));
    }
    ib->owned.push_back(*p);
    ib->owned_cap.emplace_back(*p, cap);
    return 0;
}
// the blocks of the last assembly go back to the pool (those of unknown capacity -- handed over by other modules -- are freed)
static void ib_recycle(IbState *ib)
{
    for (void *p : ib->owned) {
        size_t cap = 0;
        for (const auto &pc : ib->owned_cap)
            if (pc.first == p) cap = pc.second;
        if (cap > 0) ib->pool.emplace_back(cap, p);
        else (void)hipFree(p);
    }
    ib->owned.clear();
    ib->owned_cap.clear();
    // blocks that found no taker for a while would pile up: keep the pool bounded
    while (ib->pool.size() > 96) {
        (void)hipFree(ib->pool.front().second);
        ib->pool.erase(ib->pool.begin());
    }
}

void ib_release(IbState *ib)
{
    if (ib == nullptr) return;
    if (ib->fsol) pib_destroy(ib->fsol);
    for (void *p : ib->owned) (void)hipFree(p);
    for (const auto &pc : ib->pool) (void)hipFree(pc.second);
    for (void *p : ib->persistent) (void)hipFree(p);
    delete ib;
}

__device__ __forceinline__ double delta_kernel(int kernel, double r, double dr)
{
    const double x = fabs(r) / dr;
    if (kernel == 0) {  // delta.cpp:17-28
        if (x > 1.5) return 0.0;
        if (x > 0.5 && x <= 1.5) return (5 - 3 * x - sqrt(-3 * (1 - x) * (1 - x) + 1)) / (6 * dr);
        return (1 + sqrt(-3 * x * x + 1)) / (3 * dr);
    }
    // delta.cpp:31-40
    if (x >= 0.0 && x <= 1.0) return (3 - 2 * x + sqrt(1 + 4 * x - 4 * x * x)) / (8 * dr);
    if (x >= 1.0 && x <= 2.0) return (5 - 2 * x - sqrt(-7 + 12 * x - 4 * x * x)) / (8 * dr);
    return 0.0;
}

// One Lagrangian row per lane.  FILL = false: count the stored entries; true: write col / val / eval at rowptr[r].
// Window points outside [0, n) are skipped, also on a periodic direction: getEulerianNeighbors (createdelta.cpp:171-208)
// does wrap their index there, but pairs it with the coordinate coord[s] -+ L, a whole domain length away from the
// Lagrangian point, where the kernels vanish -- and zero entries are not stored (MAT_IGNORE_ZERO_ENTRIES, :61-62).
template <bool FILL>
__global__ __launch_bounds__(128) void k_ib_delta(NsDev D, IbDev I, int32_t *__restrict__ count,
                                                  const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                  double *__restrict__ val, double *__restrict__ eval)
{
    const int64_t r = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (r >= I.nf) return;
    const int64_t pt = r / I.dim;
    const int dof = (int)(r - pt * I.dim);
    const NsField &F = D.f[dof];
    int ids[3][IB_MAXW], cnt[3] = {0, 1, 1};
    double phi[3][IB_MAXW];
    ids[1][0] = ids[2][0] = 0;
    for (int d = 0; d < I.dim; ++d) {
        const int c0 = I.ijk[pt * I.dim + d];
        const double x = I.X[pt * I.dim + d], h = I.h[pt * I.dim + d];
        int c = 0;
        for (int s = c0 - I.window; s <= c0 + I.window; ++s)
            if (s >= 0 && s < F.n[d] && (d != I.sd || (s >= I.own_lo[dof] && s < I.own_hi[dof]))) {
                ids[d][c] = s;
                phi[d][c] = delta_kernel(I.kernel, x - F.co[d][s + 1], h);
                ++c;
            }
        cnt[d] = c;
    }
    int64_t q = FILL ? rowptr[r] : 0;
    int n = 0;
    for (int kk = 0; kk < cnt[2]; ++kk)
        for (int jj = 0; jj < cnt[1]; ++jj)
            for (int ii = 0; ii < cnt[0]; ++ii) {
                double v = 1.0;  // delta.cpp:65-72
                v *= phi[0][ii];
                v *= phi[1][jj];
                if (I.dim == 3) v *= phi[2][kk];
                if (v == 0.0) continue;
                ++n;
                if (FILL) {
                    const int64_t i = ids[0][ii], j = ids[1][jj], k = ids[2][kk];
                    const int64_t ijk[3] = {i, j, k};
                    // R (face area) and MHat (width along the component): creatediagmatrix.cpp:88-171
                    const int a = (dof == 0) ? 1 : 0, b = (dof == 2) ? 1 : 2;
                    const double la = F.dl[a][ijk[a] + 1], lb = (b < I.dim) ? F.dl[b][ijk[b] + 1] : 1.0;
                    const double R = la * lb, M = F.dl[dof][ijk[dof] + 1];
                    col[q] = (int32_t)fidx(F, i, j, k);
                    val[q] = v;
                    double e = v * R;  // MatDiagonalScale(E, nullptr, RDiag), then MHatDiag (decoupledibpm.cpp:172-173)
                    e = e * M;
                    eval[q] = e;
                    ++q;
                }
            }
    if (!FILL) count[r] = n;
}

__global__ __launch_bounds__(256) void k_ib_keys(int64_t nf, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 uint64_t *__restrict__ keys, int32_t *__restrict__ idx)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nf; r += (int64_t)gridDim.x * 256)
        for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) {
            keys[q] = (uint64_t)col[q] * (uint64_t)nf + (uint64_t)r;
            idx[q] = q;
        }
}

// sorted entries -> H arrays and segment heads
__global__ __launch_bounds__(256) void k_ib_hfill(int64_t nnz, int64_t nf, const uint64_t *__restrict__ keys,
                                                  const int32_t *__restrict__ idx, const double *__restrict__ val,
                                                  int32_t *__restrict__ hrow, double *__restrict__ hval, int32_t *__restrict__ head)
{
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * 256) {
        const uint64_t c = keys[q] / (uint64_t)nf;
        hrow[q] = (int32_t)(keys[q] - c * (uint64_t)nf);
        hval[q] = val[idx[q]];
        head[q] = (q == 0 || keys[q - 1] / (uint64_t)nf != c) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void k_ib_hrows(int64_t nnz, int64_t nf, const uint64_t *__restrict__ keys,
                                                  const int32_t *__restrict__ head, const int32_t *__restrict__ pos,
                                                  int32_t *__restrict__ hcols, int32_t *__restrict__ hptr)
{
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * 256)
        if (head[q]) {
            hcols[pos[q] - 1] = (int32_t)(keys[q] / (uint64_t)nf);
            hptr[pos[q] - 1] = (int32_t)q;
        }
}

// EBNH row r1 = (p1, dof): one wave; lanes sweep the points p2 in ascending order, pairs whose background cells are
// within 2*window in every direction are intersected.  FILL = false: count the structurally non-empty pairs.
template <bool FILL>
__global__ __launch_bounds__(64) void k_ib_ebnh(IbDev I, double dt, const int32_t *__restrict__ rowptr,
                                                const int32_t *__restrict__ col, const double *__restrict__ val,
                                                const double *__restrict__ eval, int32_t *__restrict__ count,
                                                const int32_t *__restrict__ c_rowptr, int32_t *__restrict__ c_col,
                                                double *__restrict__ c_val)
{
    const int64_t r1 = blockIdx.x;
    const int64_t p1 = r1 / I.dim;
    const int dof = (int)(r1 - p1 * I.dim);
    const int lane = threadIdx.x;
    const int32_t a0 = rowptr[r1], a1 = rowptr[r1 + 1];
    int64_t out = FILL ? c_rowptr[r1] : 0;
    int total = 0;
    for (int64_t base = 0; base < I.npts; base += 64) {
        const int64_t p2 = base + lane;
        bool hit = false;
        double sum = 0.0;
        if (p2 < I.npts) {
            bool near = true;
            for (int d = 0; d < I.dim; ++d) {
                const int dd = I.ijk[p1 * I.dim + d] - I.ijk[p2 * I.dim + d];
                near = near && (dd <= 2 * I.window) && (dd >= -2 * I.window);
            }
            if (near) {
                const int64_t r2 = p2 * I.dim + dof;
                int32_t a = a0, b = rowptr[r2];
                const int32_t b1 = rowptr[r2 + 1];
                while (a < a1 && b < b1) {
                    const int32_t ca = col[a], cb = col[b];
                    if (ca == cb) {
                        hit = true;
                        if (FILL) {
                            const double bnh = 0.0 + dt * val[b];  // BNH = BN H, BN = dt I (one-term row product)
                            sum = sum + eval[a] * bnh;
                        }
                        ++a;
                        ++b;
                    } else if (ca < cb)
                        ++a;
                    else
                        ++b;
                }
            }
        }
        const unsigned long long m = __ballot(hit);
        if (FILL && hit) {
            const int64_t o = out + __popcll(m & ((1ull << lane) - 1ull));
            c_col[o] = (int32_t)(p2 * I.dim + dof);
            c_val[o] = sum;
        }
        const int c = __popcll(m);
        out += c;
        total += c;
    }
    if (!FILL && lane == 0) count[r1] = total;
}

// y[row] = y[row] + sum_q (scale * hval[q]) * x[hrow[q]] over the stored rows of H (MatMultAdd: the row sum starts
// from y[row]; scale = 1: H, scale = dt: BNH whose stored value is the product dt * val)
__global__ __launch_bounds__(256) void k_ib_spread(int64_t hrows, double scale, const int32_t *__restrict__ hcols,
                                                   const int32_t *__restrict__ hptr, const int32_t *__restrict__ hrow,
                                                   const double *__restrict__ hval, const double *__restrict__ x,
                                                   double *__restrict__ y)
{
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < hrows; m += (int64_t)gridDim.x * 256) {
        const int32_t c = hcols[m];
        double s = y[c];
        for (int32_t q = hptr[m]; q < hptr[m + 1]; ++q) {
            const double a = 0.0 + scale * hval[q];
            s = s + a * x[hrow[q]];
        }
        y[c] = s;
    }
}

// rhsf = -(E u)  (decoupledibpm.cpp:251-252);  with ub: rhsf = UB + 1.0 * rhsf  (rigidkinematics.cpp:155-157, VecAYPX)
__global__ __launch_bounds__(256) void k_ib_interp(int64_t nf, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ eval, const double *__restrict__ U,
                                                   const double *__restrict__ ub, double *__restrict__ rhsf)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nf; r += (int64_t)gridDim.x * 256) {
        double s = 0.0;
        for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) s = s + eval[q] * U[col[q]];
        double v = -1.0 * s;
        if (ub != nullptr) v = ub[r] + 1.0 * v;
        rhsf[r] = v;
    }
}

__global__ __launch_bounds__(256) void k_ib_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = y[i] + a * x[i];
}

static int blocks_for(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + 255) / 256)); }

template <class T>
static int dev_alloc(IbState *ib, T **p, int64_t count)
{
    void *q = nullptr;
    PIB_CHK(ib_malloc(ib, &q, sizeof(T) * (size_t)std::max<int64_t>(count, 1)));
    *p = static_cast<T *>(q);
    PIB_MEMSET(*p, 0, sizeof(T) * (size_t)std::max<int64_t>(count, 1));
    return 0;
}

// exclusive prefix sum of per-row counts on the host (a few thousand rows, set-up only)
static int scan_counts(const int32_t *d_count, int64_t n, int32_t *d_rowptr, int64_t *total, hipStream_t q)
{
    std::vector<int32_t> c((size_t)n), rp((size_t)n + 1, 0);
    PIB_HIP(hipMemcpyAsync(c.data(), d_count, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    int64_t run = 0;
    for (int64_t i = 0; i < n; ++i) {
        rp[(size_t)i] = (int32_t)run;
        run += c[(size_t)i];
    }
    if (run >= (int64_t)std::numeric_limits<int32_t>::max()) return fail(PIB_ERR_SUP, "immersed-boundary operator too large for 32-bit offsets");
    rp[(size_t)n] = (int32_t)run;
    PIB_HIP(hipMemcpyAsync(d_rowptr, rp.data(), sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, q));
    PIB_HIP(hipStreamSynchronize(q));
    *total = run;
    return 0;
}

__global__ void k_ib_eu(int64_t nf, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                        const double *__restrict__ eval, const double *__restrict__ U, double *__restrict__ out);

// y = y + A x for an assembled CSR with many empty rows (BNH with BN order > 1): MatMultAdd, the row sum starts from y[row]
__global__ __launch_bounds__(256) void k_ib_csr_mult_add(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const int32_t a = rp[r], b = rp[r + 1];
        if (a == b) continue;
        double s = y[r];
        for (int32_t q = a; q < b; ++q) s = s + val[q] * x[col[q]];
        y[r] = s;
    }
}
// rowptr counts of H as a full CSR over the velocity points, from its compressed rows
__global__ __launch_bounds__(256) void k_ib_hcounts(int64_t hrows, const int32_t *__restrict__ hcols, const int32_t *__restrict__ hptr,
                                                    int32_t *__restrict__ count)
{
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < hrows; m += (int64_t)gridDim.x * 256) count[hcols[m]] = hptr[m + 1] - hptr[m];
}

// y += BNH x  (MatMultAdd(BNH, x, y, y): decoupledibpm.cpp:283-284)
__global__ void k_ib_set1(double *x, int64_t j, double v) { x[j] = v; }
// M[i][j] = col[i] for the dense EBNH held as a CSR with nf entries per row
__global__ __launch_bounds__(256) void k_ib_put_column(int64_t nf, int64_t j, const double *__restrict__ col, double *__restrict__ val)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += (int64_t)gridDim.x * 256) val[i * nf + j] = col[i];
}
__global__ __launch_bounds__(256) void k_ib_dense_pattern(int64_t nf, int32_t *__restrict__ rp, int32_t *__restrict__ cl)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nf * nf; e += (int64_t)gridDim.x * 256) cl[e] = (int32_t)(e % nf);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= nf; i += (int64_t)gridDim.x * 256) rp[i] = (int32_t)(i * nf);
}
// z-slabs with a KRYLOV forces solver (round 4; any `forces_solver.info`, as decoupledibpm.cpp:75-80 hands whatever the file says to
// createLinSolver): every rank holds the part of each entry of E BN H that runs over its own velocity points, in its own
// sparsity pattern.  The parts are summed once, at assembly, into the dense nf x nf matrix -- the all-reduce the direct solver
// makes before it factorises -- and every rank iterates on the same replicated matrix: no communication inside the forces
// solve, the same bits on every rank (the right-hand side E u is all-reduced already).
__global__ __launch_bounds__(256) void k_ib_csr_to_dense(int64_t nf, const int32_t *__restrict__ rp, const int32_t *__restrict__ cl,
                                                         const double *__restrict__ val, double *__restrict__ dense)
{
    for (int64_t i = blockIdx.x; i < nf; i += gridDim.x)
        for (int32_t e = rp[i] + (int32_t)threadIdx.x; e < rp[i + 1]; e += 256) dense[i * nf + cl[e]] = val[e];  // (a row's columns are distinct)
}
static int ib_adopt_summed(pib_ns *ns, IbState *ib, bool dense_already)
{
    hipStream_t q = ns->stream;
    const int64_t nf = ib->I.nf;
    int err = 0;
    if (nf * nf >= (int64_t)INT32_MAX)
        return fail(PIB_ERR_SUP, "immersed bodies on several ranks with a Krylov forces solver: %lld force unknowns are too many for the "
                    "summed matrix (the direct solver, -forces_ksp_type preonly -forces_pc_type lu, has the same bound)", (long long)nf);
    if (!dense_already) {
        int32_t *rp = nullptr, *cl = nullptr;
        double *dense = nullptr;
        if ((err = dev_alloc(ib, &rp, nf + 1)) || (err = dev_alloc(ib, &cl, nf * nf)) || (err = dev_alloc(ib, &dense, nf * nf))) return err;
        PIB_HIP(hipMemsetAsync(dense, 0, sizeof(double) * (size_t)(nf * nf), q));
        hipLaunchKernelGGL(k_ib_csr_to_dense, dim3((unsigned)std::min<int64_t>(nf, 4096)), dim3(256), 0, q, nf, ib->c_rowptr, ib->c_col, ib->c_val, dense);
        hipLaunchKernelGGL(k_ib_dense_pattern, dim3(blocks_for(nf * nf)), dim3(256), 0, q, nf, rp, cl);
        PIB_HIP(hipGetLastError());
        ib->c_rowptr = rp;
        ib->c_col = cl;
        ib->c_val = dense;
        ib->c_nnz = nf * nf;
    }
    PIB_CHK(comm_allreduce_big(ns->vsol, ib->c_val, nf * nf, q));
    PIB_HIP(hipStreamSynchronize(q));
    return adopt_device_csr(ib->fsol, nf, ib->c_nnz, ib->c_rowptr, ib->c_col, ib->c_val);
}

// z-slabs, BN order > 1 (round 4): t = BN H x on this rank's velocity points = the series applied to dt H x -- the spread over the
// owned points, then (dt c nu L)^k term by term with the engine's halo exchanges, every rank in step (collective)
static int ib_bnh_slab(pib_ns *ns, IbState *ib, const double *x, double *t)
{
    hipStream_t q = ns->stream;  // (ib: the state being assembled is not the engine's yet)
    PIB_HIP(hipMemsetAsync(t, 0, sizeof(double) * (size_t)ns->D.UN, q));
    hipLaunchKernelGGL(k_ib_spread, dim3(blocks_for(ib->hrows)), dim3(256), 0, q, ib->hrows, ns->dt, ib->hcols, ib->hptr, ib->hrow, ib->hval, x, t);
    PIB_HIP(hipGetLastError());
    return ns_bn_series_slab(ns, t);
}
static int ib_bnh_mult_add(pib_ns *ns, const double *x, double *y, hipStream_t q)
{
    IbState *ib = ns->ib;
    if (ns->nranks > 1 && ns->bn_order > 1) {
        if (q != ns->stream) return fail(PIB_ERR_LIB, "immersed bodies, BN order > 1 on slabs: the BNH product runs on the engine's stream");
        PIB_CHK(ib_bnh_slab(ns, ib, x, ib->bn_t));
        hipLaunchKernelGGL(k_ib_axpy, dim3(blocks_for(ns->D.UN)), dim3(256), 0, q, ns->D.UN, 1.0, ib->bn_t, y);
        PIB_HIP(hipGetLastError());
        return 0;
    }
    if (ib->bnh_rowptr != nullptr)
        hipLaunchKernelGGL(k_ib_csr_mult_add, dim3(blocks_for(ns->D.UN)), dim3(256), 0, q, ns->D.UN, ib->bnh_rowptr, ib->bnh_col, ib->bnh_val, x, y);
    else
        hipLaunchKernelGGL(k_ib_spread, dim3(blocks_for(ib->hrows)), dim3(256), 0, q, ib->hrows, ns->dt, ib->hcols, ib->hptr, ib->hrow,
                           ib->hval, x, y);
    PIB_HIP(hipGetLastError());
    return 0;
}

int ib_spread_forces(pib_ns *ns)
{
    IbState *ib = ns->ib;
    hipLaunchKernelGGL(k_ib_spread, dim3(blocks_for(ib->hrows)), dim3(256), 0, ns->stream, ib->hrows, 1.0, ib->hcols, ib->hptr,
                       ib->hrow, ib->hval, ib->f, ns->rhs1);
    PIB_HIP(hipGetLastError());
    return 0;
}

// rhsf = -s (+ UB) from the interpolated velocity s = E u summed over the ranks
__global__ __launch_bounds__(256) void k_ib_rhsf_finish(int64_t nf, const double *__restrict__ ub, double *__restrict__ rhsf)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nf; r += (int64_t)gridDim.x * 256) {
        double v = -1.0 * rhsf[r];
        if (ub != nullptr) v = ub[r] + 1.0 * v;
        rhsf[r] = v;
    }
}

int ib_solve_forces(pib_ns *ns)
{
    IbState *ib = ns->ib;
    const int64_t nf = ib->I.nf;
    if (ns->nranks > 1) {
        // E u: this rank's velocity points, then the sum over the ranks (the same bits everywhere: the replicated
        // force solves must not drift apart)
        hipLaunchKernelGGL(k_ib_eu, dim3(blocks_for(nf)), dim3(256), 0, ns->stream, nf, ib->rowptr, ib->col, ib->eval, ns->U, ib->rhsf);
        PIB_HIP(hipGetLastError());
        PIB_CHK(comm_allreduce_big(ns->vsol, ib->rhsf, nf, ns->stream));
        hipLaunchKernelGGL(k_ib_rhsf_finish, dim3(blocks_for(nf)), dim3(256), 0, ns->stream, nf,
                           ib->moving ? ib->ub : (const double *)nullptr, ib->rhsf);
    } else
    hipLaunchKernelGGL(k_ib_interp, dim3(blocks_for(nf)), dim3(256), 0, ns->stream, nf, ib->rowptr, ib->col, ib->eval, ns->U,
                       ib->moving ? ib->ub : (const double *)nullptr, ib->rhsf);
    PIB_HIP(hipGetLastError());
    PIB_CHK(ns_before_solve(ns, ib->fsol));
    PIB_CHK(pib_solve(ib->fsol, ib->df, ib->rhsf));  // fSolver->solve(df, rhsf)  (decoupledibpm.cpp:267)
    return ib_bnh_mult_add(ns, ib->df, ns->U, ns->stream);  // MatMultAdd(BNH, df, U, U)  (:283-284)
}

// out[r] = (E u)[r]
__global__ __launch_bounds__(256) void k_ib_eu(int64_t nf, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                               const double *__restrict__ eval, const double *__restrict__ U, double *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nf; r += (int64_t)gridDim.x * 256) {
        double s = 0.0;
        for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) s = s + eval[q] * U[col[q]];
        out[r] = s;
    }
}
// y = a - b ; y = -x ; u = u - t
__global__ __launch_bounds__(256) void k_ib_sub(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a[i] - b[i];
}
__global__ __launch_bounds__(256) void k_ib_neg(int64_t n, const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = -x[i];
}

// ---- coupled IBPM (applications/ibpm/ibpm.cpp) ---------------------------------------------------------------------
// The reference stacks pressure and Lagrangian forces into one unknown, G_c = [G, -H], D_c = [D; E] (:110-183), and
// hands D_c BN G_c = [[A11, A12], [A21, A22]] to ONE linear solver (:185-194):
//     A11 = D BN G (the Poisson operator), A12 = -D BNH, A21 = E BNG, A22 = -E BNH = -EBNH.
// Here the small block is eliminated exactly -- EBNH^-1 is the explicit inverse the direct forces solver holds -- and
// the pressure solver iterates on the Schur complement
//     S phi = A11 phi - D BNH EBNH^-1 E BNG phi  =  r1 - D BNH EBNH^-1 r2 ,      df = -EBNH^-1 (r2 - E BNG phi),
// negative semi-definite like A11 (S = -k G^T BN^(1/2) (I - Pi) BN^(1/2) G with Pi the projector onto the spread
// forces), so the same PCG with the same multigrid applies; the term is added to every Krylov product by
// pib_solver::post_matmult.  The pinned pressure zeroes row and column 0 of the stacked matrix (:264-268): row 0 of the
// term is skipped, column 0 multiplies the Krylov vectors' zero entry.
static int ib_schur_term(pib_solver *ps, const double *p, double *w, bool guarded, hipStream_t q, void *ctx)
{
    (void)guarded;  // after convergence these kernels only touch work vectors and a dead w
    pib_ns *ns = (pib_ns *)ctx;
    IbState *ib = ns->ib;
    const int64_t nf = ib->I.nf, UN = ns->D.UN;
    (void)ps;
    PIB_CHK(ns_bng_apply(ns, p, ib->t_un, q));                                                      // t1 = BNG p
    hipLaunchKernelGGL(k_ib_eu, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->rowptr, ib->col, ib->eval, ib->t_un, ib->g_nf);  // E t1
    PIB_CHK(dense_apply_raw(ib->fsol, ib->g_nf, ib->y_nf, q));                                       // y = EBNH^-1 g
    PIB_HIP(hipMemsetAsync(ib->t_un2, 0, sizeof(double) * (size_t)UN, q));
    PIB_CHK(ib_bnh_mult_add(ns, ib->y_nf, ib->t_un2, q));                                            // t2 = BNH y
    return ns_div_sub(ns, ib->t_un2, w, q);                                                          // w -= D t2
}

bool ib_is_coupled(const pib_ns *ns) { return ns->ib != nullptr && ns->ib->coupled; }

int ib_coupled_solve_and_project(pib_ns *ns)
{
    IbState *ib = ns->ib;
    const int64_t nf = ib->I.nf, UN = ns->D.UN, pN = ns->D.pN;
    hipStream_t q = ns->stream;
    // r2 = E u* ; r1 (= ns->rhs2, already D u* + bc, zero at a pinned row 0) -= D BNH EBNH^-1 r2
    hipLaunchKernelGGL(k_ib_eu, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->rowptr, ib->col, ib->eval, ns->U, ib->r2);
    PIB_CHK(dense_apply_raw(ib->fsol, ib->r2, ib->y_nf, q));
    PIB_HIP(hipMemsetAsync(ib->t_un2, 0, sizeof(double) * (size_t)UN, q));
    PIB_CHK(ib_bnh_mult_add(ns, ib->y_nf, ib->t_un2, q));
    PIB_CHK(ns_div_sub(ns, ib->t_un2, ns->rhs2, q));
    PIB_HIP(hipStreamSynchronize(q));
    // S dP = r1'
    PIB_CHK(pib_solve(ns->psol, ns->dP, ns->rhs2));
    // df = -EBNH^-1 (r2 - E BNG dP)
    PIB_CHK(ns_bng_apply(ns, ns->dP, ib->t_un, q));
    hipLaunchKernelGGL(k_ib_eu, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->rowptr, ib->col, ib->eval, ib->t_un, ib->g_nf);
    hipLaunchKernelGGL(k_ib_sub, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->r2, ib->g_nf, ib->rhsf);
    PIB_CHK(dense_apply_raw(ib->fsol, ib->rhsf, ib->y_nf, q));
    hipLaunchKernelGGL(k_ib_neg, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->y_nf, ib->df);
    // u = u* - BNG dP + BNH df ; p += dP ; f += df
    hipLaunchKernelGGL(k_ib_axpy, dim3(blocks_for(UN)), dim3(256), 0, q, UN, -1.0, ib->t_un, ns->U);
    PIB_CHK(ib_bnh_mult_add(ns, ib->df, ns->U, q));
    hipLaunchKernelGGL(k_ib_axpy, dim3(blocks_for(pN)), dim3(256), 0, q, pN, 1.0, ns->dP, ns->p);
    hipLaunchKernelGGL(k_ib_axpy, dim3(blocks_for(nf)), dim3(256), 0, q, nf, 1.0, ib->df, ib->f);
    PIB_HIP(hipGetLastError());
    return 0;
}

int ib_update_forces(pib_ns *ns)
{
    IbState *ib = ns->ib;
    hipLaunchKernelGGL(k_ib_axpy, dim3(blocks_for(ib->I.nf)), dim3(256), 0, ns->stream, ib->I.nf, 1.0, ib->df, ib->f);  // :301
    PIB_HIP(hipGetLastError());
    return 0;
}

}  // namespace pib

extern "C" {

// (re-)assemble Delta, E, H, EBNH for the point coordinates `coords` and hand EBNH to the forces solver
static int ib_assemble(pib_ns *ns, pib::IbState *ib, const double *coords)
{
    using namespace pib;
    ib_recycle(ib);
    const int dim = ns->D.dim;
    const int64_t total = ib->I.npts;
    int err = 0;
    // background cells (singlebodypoints.cpp:90-113) and kernel widths (createdelta.cpp:69-76) on the host
    std::vector<int> ijk((size_t)(total * dim));
    std::vector<double> h((size_t)(total * dim));
    int64_t p0 = 0;
    for (int b = 0; b < ib->nbodies; ++b) {
        const int64_t nb = ib->npts[(size_t)b];
        for (int64_t q = p0; q < p0 + nb; ++q)
            for (int d = 0; d < dim; ++d) {
                const double x = coords[q * dim + d];
                if (ns->lo[d] >= x || ns->hi[d] <= x)
                    return fail(PIB_ERR_MAX_VALUE, "body coordinate %g is outside domain [%g, %g] !", x, ns->lo[d], ns->hi[d]);
                const std::vector<double> &v = ns->h_vtx[d];
                ijk[(size_t)(q * dim + d)] = (int)(std::upper_bound(v.begin(), v.end(), x) - v.begin()) - 1;
            }
        for (int64_t q = p0; q < p0 + nb; ++q)
            for (int d = 0; d < dim; ++d) h[(size_t)(q * dim + d)] = ns->h_dlu[d][(size_t)ijk[(size_t)(p0 * dim + d)] + 1];
        p0 += nb;
    }
    IbDev &I = ib->I;
    I.sd = -1;
    for (int f = 0; f < 3; ++f) {
        I.own_lo[f] = 0;
        I.own_hi[f] = 0x7fffffff;
    }
    if (ns->nranks > 1) {
        // background cells were found on the GLOBAL mesh; the kernels index this rank's extended slab (a cell outside it
        // simply has no window point here)
        I.sd = dim - 1;
        for (int64_t q = 0; q < total; ++q) ijk[(size_t)(q * dim + I.sd)] -= (int)ns->slab_e0;
        for (int f = 0; f < dim; ++f) {
            I.own_lo[f] = (int)ns->fld_own_lo[f];
            I.own_hi[f] = (int)(ns->fld_own_lo[f] + ns->fld_own_cnt[f]);
        }
    }
    double *dX = nullptr, *dh = nullptr;
    int *dijk = nullptr;
    if ((err = dev_alloc(ib, &dX, total * dim)) || (err = dev_alloc(ib, &dh, total * dim)) || (err = dev_alloc(ib, &dijk, total * dim)))
        return err;
    PIB_HIP(hipMemcpy(dX, coords, sizeof(double) * (size_t)(total * dim), hipMemcpyHostToDevice));
    PIB_HIP(hipMemcpy(dh, h.data(), sizeof(double) * (size_t)(total * dim), hipMemcpyHostToDevice));
    PIB_HIP(hipMemcpy(dijk, ijk.data(), sizeof(int) * (size_t)(total * dim), hipMemcpyHostToDevice));
    I.X = dX;
    I.h = dh;
    I.ijk = dijk;
    hipStream_t q = ns->stream;
    const int64_t nf = I.nf;
    // ---- Delta / E
    int32_t *count = nullptr;
    if ((err = dev_alloc(ib, &count, nf)) || (err = dev_alloc(ib, &ib->rowptr, nf + 1))) return err;
    const unsigned gb = (unsigned)((nf + 127) / 128);
    hipLaunchKernelGGL(k_ib_delta<false>, dim3(gb), dim3(128), 0, q, ns->D, I, count, (const int32_t *)nullptr, (int32_t *)nullptr,
                       (double *)nullptr, (double *)nullptr);
    PIB_HIP(hipGetLastError());
    if ((err = scan_counts(count, nf, ib->rowptr, &ib->nnz, q))) return err;
    if ((err = dev_alloc(ib, &ib->col, ib->nnz)) || (err = dev_alloc(ib, &ib->val, ib->nnz)) || (err = dev_alloc(ib, &ib->eval, ib->nnz)))
        return err;
    hipLaunchKernelGGL(k_ib_delta<true>, dim3(gb), dim3(128), 0, q, ns->D, I, count, ib->rowptr, ib->col, ib->val, ib->eval);
    PIB_HIP(hipGetLastError());
    // ---- H = Delta^T: sort the entries by (velocity point, row)
    {
        const int64_t nnz = ib->nnz;
        uint64_t *k_in = nullptr, *k_out = nullptr;
        int32_t *i_in = nullptr, *i_out = nullptr, *head = nullptr, *pos = nullptr;
        if ((err = dev_alloc(ib, &k_in, nnz)) || (err = dev_alloc(ib, &k_out, nnz)) || (err = dev_alloc(ib, &i_in, nnz)) ||
            (err = dev_alloc(ib, &i_out, nnz)) || (err = dev_alloc(ib, &head, nnz)) || (err = dev_alloc(ib, &pos, nnz)) ||
            (err = dev_alloc(ib, &ib->hrow, nnz)) || (err = dev_alloc(ib, &ib->hval, nnz)))
            return err;
        hipLaunchKernelGGL(k_ib_keys, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->rowptr, ib->col, k_in, i_in);
        PIB_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        void *tmp = nullptr;
        PIB_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, i_in, i_out, (size_t)nnz, 0, 64, q));
        PIB_CHK(ib_malloc(ib, &tmp, tmp_bytes));
        PIB_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, i_in, i_out, (size_t)nnz, 0, 64, q));
        hipLaunchKernelGGL(k_ib_hfill, dim3(blocks_for(nnz)), dim3(256), 0, q, nnz, nf, k_out, i_out, ib->val, ib->hrow, ib->hval, head);
        PIB_HIP(hipGetLastError());
        size_t tmp2_bytes = 0;
        void *tmp2 = nullptr;
        PIB_HIP(rocprim::inclusive_scan(nullptr, tmp2_bytes, head, pos, (size_t)nnz, rocprim::plus<int32_t>(), q));
        PIB_CHK(ib_malloc(ib, &tmp2, tmp2_bytes));
        PIB_HIP(rocprim::inclusive_scan(tmp2, tmp2_bytes, head, pos, (size_t)nnz, rocprim::plus<int32_t>(), q));
        int32_t m = 0;
        if (nnz > 0) PIB_HIP(hipMemcpyAsync(&m, pos + (nnz - 1), sizeof(int32_t), hipMemcpyDeviceToHost, q));
        PIB_HIP(hipStreamSynchronize(q));
        ib->hrows = m;
        if ((err = dev_alloc(ib, &ib->hcols, m)) || (err = dev_alloc(ib, &ib->hptr, (int64_t)m + 1))) return err;
        hipLaunchKernelGGL(k_ib_hrows, dim3(blocks_for(nnz)), dim3(256), 0, q, nnz, nf, k_out, head, pos, ib->hcols, ib->hptr);
        PIB_HIP(hipGetLastError());
        const int32_t last = (int32_t)nnz;
        PIB_HIP(hipMemcpyAsync(ib->hptr + m, &last, sizeof(int32_t), hipMemcpyHostToDevice, q));
        PIB_HIP(hipStreamSynchronize(q));
    }
    ib->bnh_rowptr = ib->bnh_col = nullptr;
    ib->bnh_val = nullptr;
    ib->bnh_nnz = 0;
    if (ns->bn_order > 1) {
        // ---- BN order > 1: BNH = BN H and EBNH = E BNH through the reference's MatMatMult chain (decoupledibpm.cpp:194-205) with
        // the assembled BN of pib_ns_set_bn_order (createBnHead, bn.hip); H as a full CSR over the velocity points
        if (ns->nranks > 1) {
            // z-slabs (round 4): no assembled BN here -- EBNH = E (BN H) as a DENSE nf x nf matrix, column by column through the
            // term-by-term BN of the slab engine (every rank in step; a column's entries are this rank's part of the sums over its
            // own velocity points, the direct forces solver adds the ranks' matrices up).  nf (N - 1) stencil products and
            // exchanges per assembly: meant for stationary bodies of moderate size (a moving body pays it every step).
            if (nf * nf >= (int64_t)INT32_MAX) return fail(PIB_ERR_SUP, "immersed bodies with BN order > 1 on several ranks: %lld force unknowns are too many for the dense assembly", (long long)nf);
            if ((err = dev_alloc(ib, &ib->c_rowptr, nf + 1)) || (err = dev_alloc(ib, &ib->c_col, nf * nf)) || (err = dev_alloc(ib, &ib->c_val, nf * nf))) return err;
            ib->c_nnz = nf * nf;
            hipLaunchKernelGGL(k_ib_dense_pattern, dim3(blocks_for(nf * nf)), dim3(256), 0, q, nf, ib->c_rowptr, ib->c_col);
            PIB_HIP(hipGetLastError());
            PIB_HIP(hipMemsetAsync(ib->e_nf, 0, sizeof(double) * (size_t)nf, q));
            for (int64_t j = 0; j < nf; ++j) {
                hipLaunchKernelGGL(k_ib_set1, dim3(1), dim3(1), 0, q, ib->e_nf, j, 1.0);
                if ((err = ib_bnh_slab(ns, ib, ib->e_nf, ib->bn_t))) return err;
                hipLaunchKernelGGL(k_ib_eu, dim3(blocks_for(nf)), dim3(256), 0, q, nf, ib->rowptr, ib->col, ib->eval, ib->bn_t, ib->col_nf);
                hipLaunchKernelGGL(k_ib_put_column, dim3(blocks_for(nf)), dim3(256), 0, q, nf, j, ib->col_nf, ib->c_val);
                hipLaunchKernelGGL(k_ib_set1, dim3(1), dim3(1), 0, q, ib->e_nf, j, 0.0);
                PIB_HIP(hipGetLastError());
            }
            PIB_HIP(hipStreamSynchronize(q));
            if (ib->fsol->cfg.pc != Precond::LU) return ib_adopt_summed(ns, ib, true);
            ib->fsol->reduce_via = ns->vsol;
            return adopt_device_csr(ib->fsol, nf, ib->c_nnz, ib->c_rowptr, ib->c_col, ib->c_val);
        }
        if (ns->bn_rowptr == nullptr) return fail(PIB_ERR_ORDER, "immersed bodies with BN order > 1: call pib_ns_set_bn_order first");
        const int64_t UN = ns->D.UN;
        int32_t *h_rp = nullptr;
        if ((err = dev_alloc(ib, &h_rp, UN + 1))) return err;
        PIB_HIP(hipMemsetAsync(h_rp, 0, sizeof(int32_t) * (size_t)(UN + 1), q));
        hipLaunchKernelGGL(k_ib_hcounts, dim3(blocks_for(ib->hrows)), dim3(256), 0, q, ib->hrows, ib->hcols, ib->hptr, h_rp);
        PIB_HIP(hipGetLastError());
        int64_t h_nnz = 0;
        if ((err = scan_counts(h_rp, UN, h_rp, &h_nnz, q))) return err;
        if (h_nnz != ib->nnz) return fail(PIB_ERR_LIB, "immersed bodies: H has %lld entries, Delta %lld", (long long)h_nnz, (long long)ib->nnz);
        if ((err = device_spgemm(UN, UN, nf, ns->bn_rowptr, ns->bn_col, ns->bn_val, ns->bn_nnz, h_rp, ib->hrow, ib->hval, h_nnz,
                                 &ib->bnh_rowptr, &ib->bnh_col, &ib->bnh_val, &ib->bnh_nnz, q)))
            return err;
        ib->owned.push_back(ib->bnh_rowptr);
        ib->owned.push_back(ib->bnh_col);
        ib->owned.push_back(ib->bnh_val);
        if ((err = device_spgemm(nf, UN, nf, ib->rowptr, ib->col, ib->eval, ib->nnz, ib->bnh_rowptr, ib->bnh_col, ib->bnh_val, ib->bnh_nnz,
                                 &ib->c_rowptr, &ib->c_col, &ib->c_val, &ib->c_nnz, q)))
            return err;
        ib->owned.push_back(ib->c_rowptr);
        ib->owned.push_back(ib->c_col);
        ib->owned.push_back(ib->c_val);
        return adopt_device_csr(ib->fsol, nf, ib->c_nnz, ib->c_rowptr, ib->c_col, ib->c_val);
    }
    // ---- EBNH = E (BN H)
    if ((err = dev_alloc(ib, &ib->c_rowptr, nf + 1))) return err;
    hipLaunchKernelGGL(k_ib_ebnh<false>, dim3((unsigned)nf), dim3(64), 0, q, I, ns->dt, ib->rowptr, ib->col, ib->val, ib->eval, count,
                       (const int32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr);
    PIB_HIP(hipGetLastError());
    if ((err = scan_counts(count, nf, ib->c_rowptr, &ib->c_nnz, q))) return err;
    if ((err = dev_alloc(ib, &ib->c_col, ib->c_nnz)) || (err = dev_alloc(ib, &ib->c_val, ib->c_nnz))) return err;
    hipLaunchKernelGGL(k_ib_ebnh<true>, dim3((unsigned)nf), dim3(64), 0, q, I, ns->dt, ib->rowptr, ib->col, ib->val, ib->eval, count,
                       ib->c_rowptr, ib->c_col, ib->c_val);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipStreamSynchronize(q));
    // fSolver->setMatrix(EBNH)  (decoupledibpm.cpp:80, rigidkinematics.cpp:139); on slabs every rank holds the part of
    // each entry that runs over its own velocity points: the direct solver sums the dense matrix over the ranks
    if (ns->nranks > 1) {
        if (ib->fsol->cfg.pc != Precond::LU) return ib_adopt_summed(ns, ib, false);
        ib->fsol->reduce_via = ns->vsol;
    }
    return adopt_device_csr(ib->fsol, nf, ib->c_nnz, ib->c_rowptr, ib->c_col, ib->c_val);
}

int pib_ns_set_bodies(pib_ns *ns, int nbodies, const int64_t *npts, const double *coords, const char *delta_kernel,
                      const char *forces_cfg)
try {
    using namespace pib;
    if (ns == nullptr || npts == nullptr || coords == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_ns_set_bodies: null argument");
    if (nbodies < 1) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_set_bodies: need at least one body");
    PIB_HIP(hipSetDevice(ns->device));
    if (ns->ib != nullptr && ns->psol != nullptr) {
        // the coupled scheme's Schur hook points into the state that goes away: back to the plain Poisson operator
        // (and no replay of an iteration captured with the hook); the caller asks for pib_ns_set_coupled again
        ns->psol->post_matmult = nullptr;
        ns->psol->post_ctx = nullptr;
        drop_iteration_graph(ns->psol);
        ns->psol->graph_key = 0;
    }
    ib_release(ns->ib);
    ns->ib = nullptr;
    const int dim = ns->D.dim;
    int kernel, window;
    const std::string kn = delta_kernel ? delta_kernel : "ROMA_ET_AL_1999";  // decoupledibpm.cpp:162
    if (kn == "ROMA_ET_AL_1999") { kernel = 0; window = 2; }
    else if (kn == "PESKIN_2002") { kernel = 1; window = 3; }
    else return fail(PIB_ERR_ARG_UNKNOWN_TYPE, "No support for delta kernel `%s`.", kn.c_str());  // delta.cpp:58-60
    IbState *ib = new IbState();
    auto bail = [&](int e) {
        ib_release(ib);
        return e;
    };
    int err = 0;
    ib->nbodies = nbodies;
    int64_t total = 0;
    for (int b = 0; b < nbodies; ++b) {
        if (npts[b] < 1) return bail(fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_set_bodies: body %d has no points", b));
        ib->npts.push_back(npts[b]);
        total += npts[b];
    }
    IbDev &I = ib->I;
    I.dim = dim;
    I.window = window;
    I.kernel = kernel;
    I.npts = total;
    I.nf = total * dim;
    // forces solver (decoupledibpm.cpp:75-80: createLinSolver("forces", ...)) and the force vectors
    if ((err = pib_create_from_string(&ib->fsol, "forces", forces_cfg ? forces_cfg : "", 0, 1, nullptr, ns->device))) return bail(err);
    auto palloc = [&](double **p) -> int {
        PIB_HIP(hipMalloc(p, sizeof(double) * (size_t)I.nf));
        PIB_MEMSET(*p, 0, sizeof(double) * (size_t)I.nf);
        ib->persistent.push_back(*p);
        return 0;
    };
    if ((err = palloc(&ib->f)) || (err = palloc(&ib->df)) || (err = palloc(&ib->rhsf)) || (err = palloc(&ib->ub))) return bail(err);
    if (ns->nranks > 1 && ns->bn_order > 1) {  // BN term by term on the slabs (ib_bnh_slab)
        if ((err = palloc(&ib->e_nf)) || (err = palloc(&ib->col_nf))) return bail(err);
        PIB_HIP(hipMalloc(&ib->bn_t, sizeof(double) * (size_t)ns->D.UN));
        PIB_MEMSET(ib->bn_t, 0, sizeof(double) * (size_t)ns->D.UN);
        ib->persistent.push_back(ib->bn_t);
    }
    if ((err = ib_assemble(ns, ib, coords))) return bail(err);
    ns->ib = ib;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* IBPMSolver (applications/ibpm): pressure and Lagrangian forces solved as one unknown.  Needs bodies and a direct
 * forces solver (its explicit inverse of EBNH is part of the operator). */
int pib_ns_set_coupled(pib_ns *ns, int coupled)
try {
    if (ns != nullptr && coupled && ns->bn_order > 1)
        return pib::fail(PIB_ERR_SUP, "pib_ns_set_coupled: the coupled IBPM with BN order > 1 is not provided (decoupled: yes)");
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    if (ns->ib == nullptr) return fail(PIB_ERR_ORDER, "pib_ns_set_coupled: the flow has no immersed bodies");
    IbState *ib = ns->ib;
    PIB_HIP(hipSetDevice(ns->device));
    if (!coupled) {
        ib->coupled = false;
        ns->psol->post_matmult = nullptr;
        ns->psol->post_ctx = nullptr;
        drop_iteration_graph(ns->psol);
        ns->psol->graph_key = 0;
        return 0;
    }
    if (ns->nranks > 1) return fail(PIB_ERR_SUP, "pib_ns_set_coupled: the coupled scheme runs on one rank");
    if (ib->moving) return fail(PIB_ERR_SUP, "pib_ns_set_coupled: prescribed body motion belongs to the decoupled solver");
    if (ib->fsol->dense_inv == nullptr)
        return fail(PIB_ERR_SUP, "pib_ns_set_coupled: the forces solver must be the direct one (-forces_ksp_type preonly -forces_pc_type lu)");
    if (ib->t_un == nullptr) {
        auto palloc = [&](double **p, int64_t n) -> int {
            PIB_HIP(hipMalloc(p, sizeof(double) * (size_t)n));
            PIB_MEMSET(*p, 0, sizeof(double) * (size_t)n);
            ib->persistent.push_back(*p);
            return 0;
        };
        PIB_CHK(palloc(&ib->t_un, ns->D.UN));
        PIB_CHK(palloc(&ib->t_un2, ns->D.UN));
        PIB_CHK(palloc(&ib->g_nf, ib->I.nf));
        PIB_CHK(palloc(&ib->y_nf, ib->I.nf));
        PIB_CHK(palloc(&ib->r2, ib->I.nf));
    }
    ib->coupled = true;
    ns->psol->post_matmult = ib_schur_term;
    ns->psol->post_ctx = ns;
    drop_iteration_graph(ns->psol);  // an iteration captured without the term must not be replayed
    ns->psol->graph_key = 0;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* RigidKinematicsSolver::moveBodies (applications/rigidkinematics/rigidkinematics.cpp:118-140): new coordinates of
 * every Lagrangian point and their prescribed velocities UB [nf]; the operators are re-assembled, the forces solver
 * gets the new EBNH, and the forces right-hand side becomes UB - E u (:147-160).  The accumulated forces stay. */
int pib_ns_move_bodies(pib_ns *ns, const double *coords, const double *ub)
try {
    using namespace pib;
    if (ns == nullptr || coords == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_ns_move_bodies: null argument");
    if (ns->ib == nullptr) return fail(PIB_ERR_ORDER, "pib_ns_move_bodies: the flow has no immersed bodies");
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamSynchronize(ns->stream));
    IbState *ib = ns->ib;
    PIB_CHK(ib_assemble(ns, ib, coords));
    if (ub != nullptr) {
        PIB_HIP(hipMemcpy(ib->ub, ub, sizeof(double) * (size_t)ib->I.nf, hipMemcpyHostToDevice));
        ib->moving = true;
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_num_forces(pib_ns *ns, int64_t *nf, int *nbodies)
try {
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    if (nf) *nf = ns->ib ? ns->ib->I.nf : 0;
    if (nbodies) *nbodies = ns->ib ? ns->ib->nbodies : 0;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* Lagrangian forces f [nf] and the bodies' forces [nbodies*dim] = minus the sum of the Lagrangian forces of each
 * body (singlebodypoints.cpp:228-259; one line of forces-<start>.txt, decoupledibpm.cpp:437-465) */
int pib_ns_get_forces(pib_ns *ns, double *f, double *body_forces)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    if (ns->ib == nullptr) return fail(PIB_ERR_ORDER, "pib_ns_get_forces: the flow has no immersed bodies");
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamSynchronize(ns->stream));
    IbState *ib = ns->ib;
    std::vector<double> h((size_t)ib->I.nf);
    PIB_HIP(hipMemcpy(h.data(), ib->f, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    if (f) std::memcpy(f, h.data(), sizeof(double) * h.size());
    if (body_forces) {
        const int dim = ib->I.dim;
        int64_t p0 = 0;
        for (int b = 0; b < ib->nbodies; ++b) {
            for (int d = 0; d < dim; ++d) body_forces[b * dim + d] = 0.0;
            for (int64_t q = p0; q < p0 + ib->npts[(size_t)b]; ++q)
                for (int d = 0; d < dim; ++d) body_forces[b * dim + d] -= h[(size_t)(q * dim + d)];
            p0 += ib->npts[(size_t)b];
        }
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_get_forces_solver_info(pib_ns *ns, int *f_iters, double *f_res)
try {
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    if (ns->ib == nullptr) return pib::fail(PIB_ERR_ORDER, "pib_ns_get_forces_solver_info: the flow has no immersed bodies");
    if (f_iters) pib_get_iters(ns->ib->fsol, f_iters);
    if (f_res) pib_get_residual(ns->ib->fsol, f_res);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_ns_set_forces(pib_ns *ns, const double *f)
try {
    using namespace pib;
    if (ns == nullptr || f == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_ns_set_forces: null argument");
    if (ns->ib == nullptr) return fail(PIB_ERR_ORDER, "pib_ns_set_forces: the flow has no immersed bodies");
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipMemcpy(ns->ib->f, f, sizeof(double) * (size_t)ns->ib->I.nf, hipMemcpyHostToDevice));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

/* Parity / inspection access to the operators.  which: 0 Delta, 1 E, 2 H (compressed rows, row_ids = the velocity
 * points stored), 3 EBNH.  Call with null arrays to get the sizes. */
int pib_ns_get_ib_operator(pib_ns *ns, int which, int64_t *n_rows, int64_t *nnz, int32_t *rowptr, int32_t *col, double *val,
                           int32_t *row_ids)
try {
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    if (ns->ib == nullptr) return fail(PIB_ERR_ORDER, "pib_ns_get_ib_operator: the flow has no immersed bodies");
    PIB_HIP(hipSetDevice(ns->device));
    IbState *ib = ns->ib;
    int64_t nr, nz;
    const int32_t *rp, *cl;
    const double *vl;
    switch (which) {
        case 0: nr = ib->I.nf; nz = ib->nnz; rp = ib->rowptr; cl = ib->col; vl = ib->val; break;
        case 1: nr = ib->I.nf; nz = ib->nnz; rp = ib->rowptr; cl = ib->col; vl = ib->eval; break;
        case 2: nr = ib->hrows; nz = ib->nnz; rp = ib->hptr; cl = ib->hrow; vl = ib->hval; break;
        case 3: nr = ib->I.nf; nz = ib->c_nnz; rp = ib->c_rowptr; cl = ib->c_col; vl = ib->c_val; break;
        case 4:  // BNH, assembled for BN order > 1 only
            if (ib->bnh_rowptr == nullptr) return fail(PIB_ERR_ORDER, "pib_ns_get_ib_operator: BNH is assembled for BN order > 1 only");
            nr = ns->D.UN; nz = ib->bnh_nnz; rp = ib->bnh_rowptr; cl = ib->bnh_col; vl = ib->bnh_val; break;
        default: return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_get_ib_operator: which = %d", which);
    }
    if (n_rows) *n_rows = nr;
    if (nnz) *nnz = nz;
    if (rowptr) PIB_HIP(hipMemcpy(rowptr, rp, sizeof(int32_t) * (size_t)(nr + 1), hipMemcpyDeviceToHost));
    if (col) PIB_HIP(hipMemcpy(col, cl, sizeof(int32_t) * (size_t)nz, hipMemcpyDeviceToHost));
    if (val) PIB_HIP(hipMemcpy(val, vl, sizeof(double) * (size_t)nz, hipMemcpyDeviceToHost));
    if (row_ids && which == 2) PIB_HIP(hipMemcpy(row_ids, ib->hcols, sizeof(int32_t) * (size_t)nr, hipMemcpyDeviceToHost));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

}  // extern "C"
