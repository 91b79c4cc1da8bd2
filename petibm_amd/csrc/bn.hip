// bn.hip -- the Poisson operator for BN order N > 1, built in HBM the way the reference builds
// it: assembled G, D and L, then the chain of sparse products of
//   createBnHead  (src/operators/createbn.cpp:19-95)    BN   = sum_{k=1..N} dt^k (c nu)^(k-1) L^(k-1)
//   navierstokes.cpp:349-356                             BNG  = BN * G ,  DBNG = D * BNG
// with the numerics of PETSc's SeqAIJ MatMatMult (row by row, A's row in column order, every product accumulated into
// a sparse accumulator in encounter order, result columns sorted) and MatAXPY(DIFFERENT_NONZERO_PATTERN).  The oracle
// (oracle/operators.py create_bn_head / create_poisson_operator over oracle/csrc/oracle.c) performs the same
// operations in the same order: the matrices are bit-identical.
//
// Set-up only (one thread per row, the accumulator in scratch memory); the product matrices are 13 / 25-point and
// wider, so the solve runs the CSR SpMV with the multigrid of the 7-point N = 1 operator as preconditioner (the two
// operators differ by dt c nu L + ..., a small relative perturbation at the time steps PetIBM runs).  On slabs every rank
// runs the chain on a window of the mesh around its planes (assemble_poisson_bn_slab).
#include <cmath>
#include <cstring>
#include <limits>
#include <rocprim/rocprim.hpp>

#include "pib_internal.hpp"

namespace pib {

struct Csr32 {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int32_t *rowptr = nullptr, *col = nullptr;
    double *val = nullptr;
    void release()
    {
        if (rowptr) (void)hipFree(rowptr);
        if (col) (void)hipFree(col);
        if (val) (void)hipFree(val);
        rowptr = col = nullptr;
        val = nullptr;
        nrows = ncols = nnz = 0;
    }
};

constexpr int SPGEMM_MAX_ROW = 192;  // widest product row: N = 3 in 3-D stays below 130

static unsigned row_blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 127) / 128); }

// counts -> exclusive offsets (rowptr[n] = total)
static int offsets_from_counts(int32_t *d_rowptr /* counts in [0, n), slot n unused */, int64_t n, int64_t *total, hipStream_t q)
{
    PIB_HIP(hipMemsetAsync(d_rowptr + n, 0, sizeof(int32_t), q));
    size_t bytes = 0;
    void *tmp = nullptr;
    PIB_HIP(rocprim::exclusive_scan(nullptr, bytes, d_rowptr, d_rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), q));
    PIB_HIP(hipMalloc(&tmp, std::max<size_t>(bytes, 16)));
    PIB_HIP(rocprim::exclusive_scan(tmp, bytes, d_rowptr, d_rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), q));
    int32_t last = 0;
    PIB_HIP(hipMemcpyAsync(&last, d_rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    (void)hipFree(tmp);
    *total = last;
    return 0;
}

// C = A * B, one row per lane.  FILL = false: crp[i] = number of entries of row i; true: write them at crp[i].
template <bool FILL>
__global__ __launch_bounds__(128) void k_spgemm(int64_t nrows, const int32_t *__restrict__ arp, const int32_t *__restrict__ acol,
                                                const double *__restrict__ aval, const int32_t *__restrict__ brp,
                                                const int32_t *__restrict__ bcol, const double *__restrict__ bval,
                                                int32_t *__restrict__ crp, int32_t *__restrict__ ccol, double *__restrict__ cval,
                                                int *__restrict__ overflow)
{
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= nrows) return;
    int32_t cols[SPGEMM_MAX_ROW];
    double acc[SPGEMM_MAX_ROW];
    int n = 0;
    for (int32_t p = arp[i]; p < arp[i + 1]; ++p) {
        const int32_t k = acol[p];
        const double av = aval[p];
        for (int32_t q = brp[k]; q < brp[k + 1]; ++q) {
            const int32_t j = bcol[q];
            // position of j in the sorted accumulator
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cols[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            if (lo < n && cols[lo] == j) {
                if (FILL) {
                    const double t = av * bval[q];
                    acc[lo] = acc[lo] + t;
                }
                continue;
            }
            if (n == SPGEMM_MAX_ROW) {
                *overflow = 1;
                continue;
            }
            for (int t = n; t > lo; --t) {
                cols[t] = cols[t - 1];
                if (FILL) acc[t] = acc[t - 1];
            }
            cols[lo] = j;
            if (FILL) acc[lo] = av * bval[q];
            ++n;
        }
    }
    if (!FILL) {
        crp[i] = n;
        return;
    }
    const int32_t base = crp[i];
    for (int t = 0; t < n; ++t) {
        ccol[base + t] = cols[t];
        cval[base + t] = acc[t];
    }
}

static int spgemm(const Csr32 &A, const Csr32 &B, Csr32 *C, hipStream_t q)
{
    C->release();
    C->nrows = A.nrows;
    C->ncols = B.ncols;
    int *d_over = nullptr;
    PIB_HIP(hipMalloc(&d_over, sizeof(int)));
    PIB_HIP(hipMemsetAsync(d_over, 0, sizeof(int), q));
    PIB_HIP(hipMalloc(&C->rowptr, sizeof(int32_t) * ((size_t)A.nrows + 1)));
    hipLaunchKernelGGL(k_spgemm<false>, dim3(row_blocks(A.nrows)), dim3(128), 0, q, A.nrows, A.rowptr, A.col, A.val, B.rowptr, B.col,
                       B.val, C->rowptr, (int32_t *)nullptr, (double *)nullptr, d_over);
    PIB_HIP(hipGetLastError());
    int over = 0;
    PIB_HIP(hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    (void)hipFree(d_over);
    if (over) return fail(PIB_ERR_SUP, "sparse product: a row has more than %d entries (BN order too high)", SPGEMM_MAX_ROW);
    PIB_CHK(offsets_from_counts(C->rowptr, A.nrows, &C->nnz, q));
    PIB_HIP(hipMalloc(&C->col, sizeof(int32_t) * (size_t)std::max<int64_t>(C->nnz, 1)));
    PIB_HIP(hipMalloc(&C->val, sizeof(double) * (size_t)std::max<int64_t>(C->nnz, 1)));
    hipLaunchKernelGGL(k_spgemm<true>, dim3(row_blocks(A.nrows)), dim3(128), 0, q, A.nrows, A.rowptr, A.col, A.val, B.rowptr, B.col,
                       B.val, C->rowptr, C->col, C->val, (int *)nullptr);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipStreamSynchronize(q));
    return 0;
}

// C = A B for the immersed-boundary operators (BNH = BN H, EBNH = E BNH with BN order > 1: decoupledibpm.cpp:194-205), the
// numerics of PETSc's SeqAIJ MatMatMult as above.  The caller owns the output arrays.
int device_spgemm(int64_t a_rows, int64_t a_cols, int64_t b_cols, const int32_t *arp, const int32_t *acol, const double *aval, int64_t a_nnz,
                  const int32_t *brp, const int32_t *bcol, const double *bval, int64_t b_nnz, int32_t **crp, int32_t **ccol, double **cval,
                  int64_t *c_nnz, hipStream_t q)
{
    Csr32 A, B, C;
    A.nrows = a_rows;
    A.ncols = a_cols;
    A.nnz = a_nnz;
    A.rowptr = const_cast<int32_t *>(arp);
    A.col = const_cast<int32_t *>(acol);
    A.val = const_cast<double *>(aval);
    B.nrows = a_cols;
    B.ncols = b_cols;
    B.nnz = b_nnz;
    B.rowptr = const_cast<int32_t *>(brp);
    B.col = const_cast<int32_t *>(bcol);
    B.val = const_cast<double *>(bval);
    const int err = spgemm(A, B, &C, q);
    if (err) {
        C.release();
        return err;
    }
    *crp = C.rowptr;
    *ccol = C.col;
    *cval = C.val;
    *c_nnz = C.nnz;
    return 0;
}

// Z = Y + a X on the union pattern (both inputs sorted)
template <bool FILL>
__global__ __launch_bounds__(128) void k_axpy_pattern(int64_t nrows, double a, const int32_t *__restrict__ yrp,
                                                      const int32_t *__restrict__ ycol, const double *__restrict__ yval,
                                                      const int32_t *__restrict__ xrp, const int32_t *__restrict__ xcol,
                                                      const double *__restrict__ xval, int32_t *__restrict__ zrp,
                                                      int32_t *__restrict__ zcol, double *__restrict__ zval)
{
    const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= nrows) return;
    int32_t p = yrp[i], q = xrp[i];
    const int32_t pe = yrp[i + 1], qe = xrp[i + 1];
    int32_t o = FILL ? zrp[i] : 0;
    while (p < pe || q < qe) {
        if (q >= qe || (p < pe && ycol[p] < xcol[q])) {
            if (FILL) { zcol[o] = ycol[p]; zval[o] = yval[p]; }
            ++p;
        } else if (p >= pe || xcol[q] < ycol[p]) {
            if (FILL) { zcol[o] = xcol[q]; zval[o] = a * xval[q]; }
            ++q;
        } else {
            if (FILL) { zcol[o] = ycol[p]; zval[o] = yval[p] + a * xval[q]; }
            ++p;
            ++q;
        }
        ++o;
    }
    if (!FILL) zrp[i] = o;
}

static int axpy_pattern(const Csr32 &Y, double a, const Csr32 &X, Csr32 *Z, hipStream_t q)
{
    Z->release();
    Z->nrows = Y.nrows;
    Z->ncols = Y.ncols;
    PIB_HIP(hipMalloc(&Z->rowptr, sizeof(int32_t) * ((size_t)Y.nrows + 1)));
    hipLaunchKernelGGL(k_axpy_pattern<false>, dim3(row_blocks(Y.nrows)), dim3(128), 0, q, Y.nrows, a, Y.rowptr, Y.col, Y.val, X.rowptr,
                       X.col, X.val, Z->rowptr, (int32_t *)nullptr, (double *)nullptr);
    PIB_HIP(hipGetLastError());
    PIB_CHK(offsets_from_counts(Z->rowptr, Y.nrows, &Z->nnz, q));
    PIB_HIP(hipMalloc(&Z->col, sizeof(int32_t) * (size_t)std::max<int64_t>(Z->nnz, 1)));
    PIB_HIP(hipMalloc(&Z->val, sizeof(double) * (size_t)std::max<int64_t>(Z->nnz, 1)));
    hipLaunchKernelGGL(k_axpy_pattern<true>, dim3(row_blocks(Y.nrows)), dim3(128), 0, q, Y.nrows, a, Y.rowptr, Y.col, Y.val, X.rowptr,
                       X.col, X.val, Z->rowptr, Z->col, Z->val);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipStreamSynchronize(q));
    return 0;
}

// ---- G and D as matrices ---------------------------------------------------------------------------------
struct GdMesh {
    int dim, per;
    int64_t fn[3][3], foff[3];   // points and first packed row of every velocity component
    int64_t pn[3];
    const double *dlff[3];       // dL[f][f], index s+1
    const double *pw[3];         // pressure-cell widths
    double a0n[3][2];            // a0 of component f's ghost point at its own minus / plus boundary (Neumann: 1; else 0)
};

// createGradient (creategradient.cpp:64-128, normalize = FALSE): row of velocity point (f; i,j,k) = {-1/dL at its cell,
// +1/dL at the next cell along f}, dL = dL[f][f][idx]; the next cell of the last point of a periodic direction is cell 0
// (the smaller column).
__global__ __launch_bounds__(256) void k_bn_gradient(GdMesh M, int64_t UN, int32_t *__restrict__ rp, int32_t *__restrict__ col,
                                                     double *__restrict__ val)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= UN; r += (int64_t)gridDim.x * 256) {
        rp[r] = (int32_t)(2 * r);
        if (r == UN) break;
        int f = 0;
        if (M.dim > 1 && r >= M.foff[1]) f = 1;
        if (M.dim > 2 && r >= M.foff[2]) f = 2;
        const int64_t q = r - M.foff[f];
        const int64_t ijk[3] = {q % M.fn[f][0], (q / M.fn[f][0]) % M.fn[f][1], q / (M.fn[f][0] * M.fn[f][1])};
        const double gv = 1.0 / M.dlff[f][ijk[f] + 1];
        const int64_t pst[3] = {1, M.pn[0], M.pn[0] * M.pn[1]};
        const int64_t pc = ijk[0] + M.pn[0] * (ijk[1] + M.pn[1] * ijk[2]);
        if (ijk[f] < M.pn[f] - 1) {
            col[2 * r] = (int32_t)pc;
            val[2 * r] = -gv;
            col[2 * r + 1] = (int32_t)(pc + pst[f]);
            val[2 * r + 1] = gv;
        } else {
            col[2 * r] = (int32_t)(pc - (M.pn[f] - 1) * pst[f]);
            val[2 * r] = gv;
            col[2 * r + 1] = (int32_t)pc;
            val[2 * r + 1] = -gv;
        }
    }
}

// createDivergence (createdivergence.cpp:135-223, normalize = FALSE): row of a pressure cell = -area at the minus face,
// +area at the plus face of every direction, columns = packed velocity indices in ascending order.  Ghost faces have no
// column of their own: their coefficient times a0 is ADDED to the entry of their target -- the interior face next to them, the
// cell's other face (createdivergence.cpp:231-242; a0 = 0 for the normal component with Dirichlet / convective boundaries,
// 1 with NEUMANN, singleboundaryneumann.cpp:27-28: the boundary cell's row then loses that direction altogether, the entry
// staying in the pattern as an explicit zero).
template <bool FILL>
__global__ __launch_bounds__(256) void k_bn_divergence(GdMesh M, int64_t pN, int32_t *__restrict__ rp, int32_t *__restrict__ col,
                                                       double *__restrict__ val)
{
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < pN; c += (int64_t)gridDim.x * 256) {
        const int64_t ijk[3] = {c % M.pn[0], (c / M.pn[0]) % M.pn[1], c / (M.pn[0] * M.pn[1])};
        const double wx = M.pw[0][ijk[0]], wy = M.pw[1][ijk[1]], wz = (M.dim == 3) ? M.pw[2][ijk[2]] : 1.0;
        const double area[3] = {wy * wz, wx * wz, wx * wy};
        int32_t o = FILL ? rp[c] : 0;
        for (int f = 0; f < M.dim; ++f) {
            const int64_t st[3] = {1, M.fn[f][0], M.fn[f][0] * M.fn[f][1]};
            const int64_t s = ijk[f];
            const bool wrap = (M.per >> f) & 1;
            const int64_t base = M.foff[f] + ijk[0] + M.fn[f][0] * (ijk[1] + M.fn[f][1] * ijk[2]);  // the + face (index s)
            const bool has_m = s > 0 || wrap, has_p = s < M.fn[f][f];
            const int64_t cm = (s > 0) ? base - st[f] : base + (M.fn[f][f] - 1) * st[f];
            // (ghost faces of a wall-bounded direction: the minus face of cell 0 folds onto the plus face's entry, the plus face
            // of the last cell onto the minus face's)
            const double fold_p = (!wrap && s == 0) ? (-area[f]) * M.a0n[f][0] : 0.0;
            const double fold_m = (!wrap && !has_p) ? area[f] * M.a0n[f][1] : 0.0;
            if (has_m && s > 0) {
                if (FILL) { col[o] = (int32_t)cm; val[o] = (fold_m != 0.0) ? -area[f] + fold_m : -area[f]; }
                ++o;
            }
            if (has_p) {
                if (FILL) { col[o] = (int32_t)base; val[o] = (fold_p != 0.0) ? area[f] + fold_p : area[f]; }
                ++o;
            }
            if (has_m && s == 0) {  // wrapped minus face: velocity point n-1 sorts after the plus face
                if (FILL) { col[o] = (int32_t)cm; val[o] = -area[f]; }
                ++o;
            }
        }
        if (!FILL) rp[c] = o;
    }
}

__global__ __launch_bounds__(256) void k_bn_identity(int64_t n, double v, int32_t *__restrict__ rp, int32_t *__restrict__ col,
                                                     double *__restrict__ val)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= n; r += (int64_t)gridDim.x * 256) {
        rp[r] = (int32_t)r;
        if (r < n) {
            col[r] = (int32_t)r;
            val[r] = v;
        }
    }
}

// MatZeroRowsColumns(DBNG, row 0, diag = 1) keeping the pattern (navierstokes.cpp:414-420)
// (`pin` = where row 0 of the whole matrix sits in this numbering: 0 on one rank, its place in a rank's window on slabs)
__global__ __launch_bounds__(256) void k_bn_pin(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                double *__restrict__ val, int64_t pin)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        for (int32_t p = rp[r]; p < rp[r + 1]; ++p)
            if (r == pin || col[p] == pin) val[p] = (r == pin && col[p] == pin) ? 1.0 : 0.0;
}

static int dup(const Csr32 &A, Csr32 *B, hipStream_t q)
{
    B->release();
    B->nrows = A.nrows;
    B->ncols = A.ncols;
    B->nnz = A.nnz;
    PIB_HIP(hipMalloc(&B->rowptr, sizeof(int32_t) * ((size_t)A.nrows + 1)));
    PIB_HIP(hipMalloc(&B->col, sizeof(int32_t) * (size_t)std::max<int64_t>(A.nnz, 1)));
    PIB_HIP(hipMalloc(&B->val, sizeof(double) * (size_t)std::max<int64_t>(A.nnz, 1)));
    PIB_HIP(hipMemcpyAsync(B->rowptr, A.rowptr, sizeof(int32_t) * ((size_t)A.nrows + 1), hipMemcpyDeviceToDevice, q));
    PIB_HIP(hipMemcpyAsync(B->col, A.col, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToDevice, q));
    PIB_HIP(hipMemcpyAsync(B->val, A.val, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToDevice, q));
    PIB_HIP(hipStreamSynchronize(q));
    return 0;
}

// Builds BNG (UN x pN) and DBNG (pN x pN) for BN order `order` >= 2 on the solver's device.  `s` is used as the
// workspace of the Laplacian assembly (its matrix is replaced).
static int build_bn_chain(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double mn[3],
                          const double mx[3], const double a0[18], double dt, double coeff_nu, int order, Csr32 *BNG, Csr32 *DBNG,
                          Csr32 *BNkeep = nullptr)
{
    hipStream_t q = s->stream;
    // L = createLaplacian: the velocity assembly with MatScale(1), MatShift(0)
    PIB_CHK(assemble_velocity(s, dim, n, w, mn, mx, a0, std::numeric_limits<double>::infinity(), -1.0));
    if (s->A.rp64) return fail(PIB_ERR_SUP, "BN order > 1: the Laplacian needs 64-bit offsets (too large for the product chain)");
    Csr32 L;
    L.nrows = L.ncols = s->A.n;
    L.nnz = s->A.nnz;
    L.rowptr = (int32_t *)s->A.rowptr;
    L.col = s->A.col;
    L.val = s->A.val;
    s->A.rowptr = nullptr;  // ownership moved
    s->A.col = nullptr;
    s->A.val = nullptr;
    s->A.release();
    // mesh arrays for G and D
    std::vector<double> hdl[3][3], hco[3][3];
    GdMesh M;
    std::memset(&M, 0, sizeof M);
    M.dim = dim;
    for (int f = 0; f < dim; ++f)  // the ghost fold of D (round 5): a0 of the NORMAL component at its two boundaries
        for (int e = 0; e < 2; ++e) M.a0n[f][e] = s->periodic[f] ? 0.0 : a0[6 * f + 2 * f + e];
    velocity_mesh_arrays(dim, n, w, mn, mx, s->periodic, hdl, hco, M.fn, &s->mesh_window);
    std::vector<double *> tofree;
    int64_t UN = 0, pN = 1;
    for (int d = 0; d < 3; ++d) {
        M.pn[d] = (d < dim) ? n[d] : 1;
        pN *= M.pn[d];
        M.foff[d] = UN;
        if (d < dim) {
            if (s->periodic[d]) M.per |= 1 << d;
            UN += M.fn[d][0] * M.fn[d][1] * M.fn[d][2];
            double *p1 = nullptr, *p2 = nullptr;
            std::vector<double> hw(w[d], w[d] + n[d]);
            PIB_CHK(upload_vec(hdl[d][d], &p1));
            PIB_CHK(upload_vec(hw, &p2));
            M.dlff[d] = p1;
            M.pw[d] = p2;
            tofree.push_back(p1);
            tofree.push_back(p2);
        }
    }
    if (UN != L.nrows) return fail(PIB_ERR_LIB, "BN chain: velocity sizes disagree");
    Csr32 G, D, BN, right, tmp;
    auto cleanup = [&]() {
        L.release();
        G.release();
        D.release();
        BN.release();
        right.release();
        tmp.release();
        for (double *p : tofree) (void)hipFree(p);
    };
    int err = 0;
    auto run = [&]() -> int {
        G.nrows = UN;
        G.ncols = pN;
        G.nnz = 2 * UN;
        PIB_HIP(hipMalloc(&G.rowptr, sizeof(int32_t) * ((size_t)UN + 1)));
        PIB_HIP(hipMalloc(&G.col, sizeof(int32_t) * (size_t)G.nnz));
        PIB_HIP(hipMalloc(&G.val, sizeof(double) * (size_t)G.nnz));
        const unsigned gb = (unsigned)std::min<int64_t>(8192, (UN + 256) / 256);
        hipLaunchKernelGGL(k_bn_gradient, dim3(gb), dim3(256), 0, q, M, UN, G.rowptr, G.col, G.val);
        PIB_HIP(hipGetLastError());
        D.nrows = pN;
        D.ncols = UN;
        PIB_HIP(hipMalloc(&D.rowptr, sizeof(int32_t) * ((size_t)pN + 1)));
        const unsigned db = (unsigned)std::min<int64_t>(8192, (pN + 255) / 256);
        hipLaunchKernelGGL(k_bn_divergence<false>, dim3(db), dim3(256), 0, q, M, pN, D.rowptr, (int32_t *)nullptr, (double *)nullptr);
        PIB_HIP(hipGetLastError());
        PIB_CHK(offsets_from_counts(D.rowptr, pN, &D.nnz, q));
        PIB_HIP(hipMalloc(&D.col, sizeof(int32_t) * (size_t)D.nnz));
        PIB_HIP(hipMalloc(&D.val, sizeof(double) * (size_t)D.nnz));
        hipLaunchKernelGGL(k_bn_divergence<true>, dim3(db), dim3(256), 0, q, M, pN, D.rowptr, D.col, D.val);
        PIB_HIP(hipGetLastError());
        // BN = dt I + sum_{term >= 2} dt^term (c nu)^(term-1) L^(term-1)   (createbn.cpp:47-92)
        BN.nrows = BN.ncols = BN.nnz = UN;
        PIB_HIP(hipMalloc(&BN.rowptr, sizeof(int32_t) * ((size_t)UN + 1)));
        PIB_HIP(hipMalloc(&BN.col, sizeof(int32_t) * (size_t)UN));
        PIB_HIP(hipMalloc(&BN.val, sizeof(double) * (size_t)UN));
        hipLaunchKernelGGL(k_bn_identity, dim3(gb), dim3(256), 0, q, UN, dt, BN.rowptr, BN.col, BN.val);
        PIB_HIP(hipGetLastError());
        for (int term = 2; term <= order; ++term) {
            PIB_CHK(dup(L, &right, q));
            for (int c = 2; c < term; ++c) {
                PIB_CHK(spgemm(L, right, &tmp, q));
                std::swap(right, tmp);
            }
            const double a = std::pow(dt, term) * std::pow(coeff_nu, term - 1);
            PIB_CHK(axpy_pattern(BN, a, right, &tmp, q));
            std::swap(BN, tmp);
        }
        PIB_CHK(spgemm(BN, G, BNG, q));
        PIB_CHK(spgemm(D, *BNG, DBNG, q));
        if (BNkeep != nullptr) std::swap(*BNkeep, BN);  // the immersed-boundary operators need BN itself (BNH = BN H)
        return 0;
    };
    err = run();
    cleanup();
    return err;
}

static int bn_register_preconditioner(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], double dt, int nullspace);

// Several ranks (z-slabs; y in 2-D).  A row of D BN G reaches `order` planes along the slab axis, so a rank runs the
// chain above on a WINDOW of the mesh -- its planes and order + 1 more on either side (across the seam of a periodic
// slab axis), a one-rank problem in a scratch solver -- and keeps its own rows: the operators of the chain are made of the
// cell widths and of the whole mesh's coordinates (MeshWindow), every product row is accumulated in the order of the
// whole matrix's row (the window's numbering is monotone in the global one inside a row's reach), so the kept rows carry
// the entries of the one-rank matrix bit for bit -- next to the seam of a periodic slab axis, where the wrapped
// neighbours come first instead of last, to rounding; what the window's artificial ends spoil stays within `order` planes
// of them.  The rows go through the setMatrix route (upload_csr: ghost columns `order` planes deep, ring across a seam).
static int assemble_poisson_bn_slab(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double mn[3],
                                    const double mx[3], const double a0[18], double dt, double coeff_nu, int order, int nullspace)
{
    const int P = s->comm.nranks, rank = s->comm.rank, sd = dim - 1;
    const int64_t nz = n[sd], H = order + 1;
    const bool wrap = s->periodic[sd] != 0;
    // (the same verdict on every rank: nobody is left waiting in a collective)
    if (nz / P < order) return fail(PIB_ERR_SUP, "BN order %d on %d ranks: slabs of %lld planes are thinner than the operator's reach", order, P, (long long)(nz / P));
    if (wrap && (nz + P - 1) / P + 2 * H > nz) return fail(PIB_ERR_SUP, "BN order %d on %d ranks: the periodic slab axis (%lld planes) is too short for the windows", order, P, (long long)nz);
    int64_t k0 = 0, k1 = 0;
    slab_range(nz, P, rank, &k0, &k1);
    int64_t K0 = k0 - H, K1 = k1 + H;
    if (!wrap) {
        K0 = std::max<int64_t>(K0, 0);
        K1 = std::min<int64_t>(K1, nz);
    }
    auto plane_of = [&](int64_t kl) { return ((K0 + kl) % nz + nz) % nz; };
    int64_t nwin[3] = {1, 1, 1}, pl = 1;
    for (int d = 0; d < dim; ++d) nwin[d] = n[d];
    nwin[sd] = K1 - K0;
    for (int d = 0; d < sd; ++d) pl *= n[d];
    std::vector<double> ww((size_t)(K1 - K0));
    for (int64_t q = 0; q < K1 - K0; ++q) ww[(size_t)q] = w[sd][plane_of(q)];
    const double *wwin[3] = {w[0], w[1], w[2]};
    wwin[sd] = ww.data();
    double mnw[3] = {mn[0], mn[1], mn[2]}, mxw[3] = {mx[0], mx[1], mx[2]}, a0w[18];
    if (wrap || K1 < nz) {  // (an end of the whole mesh keeps its coordinate: the wall's ghost point is made of it)
        mxw[sd] = mnw[sd];
        for (double v : ww) mxw[sd] += v;
    }
    std::memcpy(a0w, a0, sizeof a0w);
    for (int f = 0; f < 3; ++f) {
        if (wrap || K0 > 0) a0w[6 * f + 2 * sd] = 0.0;      // an artificial end: its rows are not kept
        if (wrap || K1 < nz) a0w[6 * f + 2 * sd + 1] = 0.0;
    }
    pib_solver *t = nullptr;
    PIB_CHK(pib_create_from_string(&t, "velocity", "", 0, 1, nullptr, s->device));
    for (int d = 0; d < 3; ++d) t->periodic[d] = (d == sd) ? 0 : s->periodic[d];
    t->mesh_window.active = true;
    t->mesh_window.axis = sd;
    t->mesh_window.first = K0;
    t->mesh_window.n_global = nz;
    t->mesh_window.w_global = w[sd];
    t->mesh_window.lo = mn[sd];
    t->mesh_window.hi = mx[sd];
    Csr32 BNG, DBNG;
    std::vector<int32_t> hrp, hcl;
    std::vector<double> hv;
    const int64_t r0 = (k0 - K0) * pl, nloc = (k1 - k0) * pl;
    auto run = [&]() -> int {
        PIB_CHK(build_bn_chain(t, dim, nwin, wwin, mnw, mxw, a0w, dt, coeff_nu, order, &BNG, &DBNG));
        hipStream_t q = t->stream;
        if (nullspace == PIB_NULLSPACE_PINNED) {
            int64_t pin = -1;  // row 0 of the whole matrix, if the window holds it
            for (int64_t kl = 0; kl < K1 - K0; ++kl)
                if (plane_of(kl) == 0) pin = kl * pl;
            hipLaunchKernelGGL(k_bn_pin, dim3((unsigned)std::min<int64_t>(8192, (DBNG.nrows + 255) / 256)), dim3(256), 0, q, DBNG.nrows,
                               DBNG.rowptr, DBNG.col, DBNG.val, pin);
            PIB_HIP(hipGetLastError());
        }
        hrp.resize((size_t)nloc + 1);
        PIB_HIP(hipMemcpyAsync(hrp.data(), DBNG.rowptr + r0, sizeof(int32_t) * ((size_t)nloc + 1), hipMemcpyDeviceToHost, q));
        PIB_HIP(hipStreamSynchronize(q));
        const int64_t p0 = hrp[0], nnz = hrp[(size_t)nloc] - p0;
        hcl.resize((size_t)std::max<int64_t>(nnz, 1));
        hv.resize((size_t)std::max<int64_t>(nnz, 1));
        PIB_HIP(hipMemcpyAsync(hcl.data(), DBNG.col + p0, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, q));
        PIB_HIP(hipMemcpyAsync(hv.data(), DBNG.val + p0, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, q));
        PIB_HIP(hipStreamSynchronize(q));
        return 0;
    };
    int err = run();
    BNG.release();
    DBNG.release();
    (void)pib_destroy(t);
    if (err) return err;
    int64_t pN = pl * nz;
    std::vector<int64_t> grp((size_t)nloc + 1), gcl(hcl.size());
    for (int64_t i = 0; i <= nloc; ++i) grp[(size_t)i] = (int64_t)hrp[(size_t)i] - hrp[0];
    for (int64_t p = 0; p < grp[(size_t)nloc]; ++p) {
        const int64_t c = hcl[(size_t)p];
        gcl[(size_t)p] = plane_of(c / pl) * pl + c % pl;
    }
    PIB_CHK(upload_csr(s, nloc, k0 * pl, pN, grp.data(), gcl.data(), nullptr, nullptr, hv.data()));
    PIB_CHK(after_set_matrix(s));
    return bn_register_preconditioner(s, dim, n, w, dt, nullspace);
}

int assemble_poisson_bn(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double mn[3],
                        const double mx[3], const double a0[18], double dt, double coeff_nu, int order, int nullspace,
                        int32_t **bng_rowptr, int32_t **bng_col, double **bng_val, int64_t *bng_nnz, int32_t **bn_rowptr,
                        int32_t **bn_col, double **bn_val, int64_t *bn_nnz)
{
    if (order < 1) return fail(PIB_ERR_SUP, "The order of Bn can not be smaller than 1.");  // createbn.cpp:27-29 (error 56)
    if (s->comm.nranks > 1) {
        if (bng_rowptr) return fail(PIB_ERR_SUP, "BN order > 1: the projection's BNG is assembled on one rank only");
        return assemble_poisson_bn_slab(s, dim, n, w, mn, mx, a0, dt, coeff_nu, order, nullspace);
    }
    Csr32 BNG, DBNG, BN;
    int err = build_bn_chain(s, dim, n, w, mn, mx, a0, dt, coeff_nu, order, &BNG, &DBNG, bn_rowptr ? &BN : nullptr);
    if (err) {
        BNG.release();
        DBNG.release();
        BN.release();
        return err;
    }
    if (bn_rowptr) {
        *bn_rowptr = BN.rowptr;
        *bn_col = BN.col;
        *bn_val = BN.val;
        *bn_nnz = BN.nnz;
    }
    hipStream_t q = s->stream;
    if (nullspace == PIB_NULLSPACE_PINNED) {
        hipLaunchKernelGGL(k_bn_pin, dim3((unsigned)std::min<int64_t>(8192, (DBNG.nrows + 255) / 256)), dim3(256), 0, q, DBNG.nrows,
                           DBNG.rowptr, DBNG.col, DBNG.val, (int64_t)0);
        PIB_HIP(hipGetLastError());
        PIB_HIP(hipStreamSynchronize(q));
    }
    err = adopt_device_csr(s, DBNG.nrows, DBNG.nnz, DBNG.rowptr, DBNG.col, DBNG.val);
    DBNG.release();
    if (bng_rowptr) {
        *bng_rowptr = BNG.rowptr;
        *bng_col = BNG.col;
        *bng_val = BNG.val;
        *bng_nnz = BNG.nnz;
    } else
        BNG.release();
    if (err) return err;
    return bn_register_preconditioner(s, dim, n, w, dt, nullspace);
}

// the multigrid of the N = 1 operator as preconditioner: structure from the widths, not verified against the CSR
static int bn_register_preconditioner(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], double dt, int nullspace)
{
    int err = 0;
    std::vector<double> hw[3], hg[3];
    for (int d = 0; d < 3; ++d) {
        const int64_t nd = (d < dim) ? n[d] : 1;
        hw[d].resize((size_t)nd);
        for (int64_t i = 0; i < nd; ++i) hw[d][(size_t)i] = (d < dim) ? w[d][i] : 1.0;
        const bool wrap = d < dim && s->periodic[d] != 0;
        hg[d].resize((size_t)std::max<int64_t>(nd - 1, 0) + (wrap ? 1 : 0));
        for (int64_t i = 0; i + 1 < nd; ++i) hg[d][(size_t)i] = dt * (1.0 / (0.5 * (hw[d][(size_t)i + 1] + hw[d][(size_t)i])));
        if (wrap) hg[d][(size_t)nd - 1] = dt * (1.0 / (0.5 * (hw[d][0] + hw[d][(size_t)nd - 1])));
    }
    const double *cw[3] = {hw[0].data(), hw[1].data(), hw[2].data()};
    const double *cg[3] = {hg[0].data(), hg[1].data(), hg[2].data()};
    s->hint_pc_only = true;
    err = grid_register(s, dim, n, cw, cg, nullspace, dt);
    s->hint_pc_only = false;
    return err;
}

}  // namespace pib
