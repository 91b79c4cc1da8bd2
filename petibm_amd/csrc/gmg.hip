// gmg.hip -- K2/K5/K6/K7: matrix-free stencil twin and geometric multigrid for
// the 5/7-point Poisson operator on the stretched Cartesian mesh (gfx950).
//
// Stands in for the algebraic multigrid the reference gets from third parties
// (AmgX CLASSICAL/AGGREGATION AMG: examples/navierstokes/
// liddrivencavity2dRe1000_GPU/config/poisson_solver.info:20-42; PCGAMG / hypre:
// examples/navierstokes/liddrivencavity2dRe100/config/poisson_solver.info:6-9).
// The algorithm is restated operation-for-operation on the CPU in
// oracle/csrc/gmg.c (the parity oracle of these kernels).
//
//   * grid stored (nx, ny, nz) in natural order; 2-D grids are (nx, 1, ny) so
//     the slab (decomposition) axis is always the last one;
//   * level operator = rediscretised FV operator from three 1-D width arrays
//     (16-24 B/row of HBM traffic instead of the CSR's 104 B/row);
//   * selective coarsening per direction (neighbours merge only while their
//     combined width is <= 1.5 hmin 2^(level+1): the stretched far field of a
//     PetIBM mesh waits for the refined region; plain pairing when uniform);
//   * tri-linear cell-centred prolongation with width-based weights from 1-D
//     tables (3/4, 1/4 on a uniform mesh), restriction = P^T,
//     damped-Jacobi V(nu1,nu2), pre-smoothing from a zero guess;
//   * multi-GPU: levels are z-slab distributed (one halo plane, RCCL
//     send/recv; aggregates never straddle a slab boundary) while every rank
//     keeps >= 2 planes and the level is large; below that the
//     level's right-hand side is all-gathered and the remaining levels are
//     solved redundantly on every GPU (no further communication);
//   * null space: CONSTANT leaves z un-projected (the Krylov kernels subtract
//     the mean lazily), PINNED feeds r'[0] = r[0] - sum(r) to the singular
//     operator; both keep the preconditioner symmetric.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pib_internal.hpp"

#include "gmg_level_kernels.hpp"
#include "gmg_up_kernels.hpp"
#include "gmg_down_kernels.hpp"
#include "gmg_coarse_kernels.hpp"

namespace pib {

// ------------------------------------------------------------------ host side
// halo memory / deepest exchange of a distributed level (see "halos of a distributed level" below)
constexpr int HALO_PAD_PLANES = 8;   // memory per side (the fused kernels read one plane beyond the run they process)
constexpr int HALO_MAX_DEPTH = 6;    // deepest exchange: V(2,2) needs 4 planes of the residual on level 0; below it 3, or -- when the
                                     // way up is to run without exchanges of its own (pib_deep_halo=2) -- 5 on level 1 and 6 on level 2

static LevelDev dev_of(const GridLevel &g)
{
    LevelDev L;
    L.nx = (int)g.n[0];
    L.ny = (int)g.n[1];
    L.nzg = (int)g.n[2];
    L.k0 = (int)g.k0;
    L.nk = (int)(g.k1 - g.k0);
    L.per = g.per;
    L.tper = g.tper;
    L.zring = g.zring ? 1 : 0;
    L.wx = g.w[0];
    L.wy = g.w[1];
    L.wz = g.w[2];
    L.gx = g.g[0];
    L.gy = g.g[1];
    L.gz = g.g[2];
    L.cmx = g.cm[0];
    L.cpx = g.cp[0];
    L.rwx = g.rw[0];
    L.cmy = g.cm[1];
    L.cpy = g.cp[1];
    L.rwy = g.rw[1];
    L.cmz = g.cm[2];
    L.cpz = g.cp[2];
    L.rwz = g.rw[2];
    for (int d = 0; d < 3; ++d) L.t[d] = Tr1{g.t_par[d], g.t_oth[d], g.t_fst[d], g.t_wpar[d], g.t_woth[d]};
    L.tx = TrX{g.tx_fc, g.tx_pw, g.tx_rw};
    return L;
}

static int grid_blocks(int64_t n) { return (int)std::min<int64_t>(4096, std::max<int64_t>(1, (n + 255) / 256)); }
// (x, y) = (workgroups per plane, owned planes)
static dim3 level_grid(const GridLevel &g)
{
    const int64_t plane = g.n[0] * g.n[1];
    return dim3((unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, (plane + 255) / 256)), (unsigned)std::max<int64_t>(1, g.k1 - g.k0));
}

// (+8 zero entries of padding: the level kernels read the 1-D arrays in aligned vectors of up to 4)
template <class T>
static int up(const std::vector<T> &h, T **d)
{
    PIB_HIP(hipMalloc(d, sizeof(T) * (h.size() + 8)));
    PIB_MEMSET(*d, 0, sizeof(T) * (h.size() + 8));
    if (!h.empty()) PIB_HIP(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

void gmg_release(pib_solver *s)
{
    drop_iteration_graph(s);  // (krylov.hip: the captured iteration goes before the memory it points at)
    if (s->d_tail_args) (void)hipFree(s->d_tail_args);
    s->d_tail_args = nullptr;
    if (s->d_tail_tab) (void)hipFree(s->d_tail_tab);
    s->d_tail_tab = nullptr;
    s->h_tail_args.clear();
    for (auto &L : s->levels) {
        for (int d = 0; d < 3; ++d) {
            if (L.w[d]) (void)hipFree(L.w[d]);
            if (L.g[d]) (void)hipFree(L.g[d]);
            if (L.cm[d]) (void)hipFree(L.cm[d]);
            if (L.cp[d]) (void)hipFree(L.cp[d]);
            if (L.rw[d]) (void)hipFree(L.rw[d]);
            if (L.t_par[d]) (void)hipFree(L.t_par[d]);
            if (L.t_oth[d]) (void)hipFree(L.t_oth[d]);
            if (L.t_fst[d]) (void)hipFree(L.t_fst[d]);
            if (L.t_wpar[d]) (void)hipFree(L.t_wpar[d]);
            if (L.t_woth[d]) (void)hipFree(L.t_woth[d]);
        }
        if (L.tx_fc) (void)hipFree(L.tx_fc);
        if (L.tx_pw) (void)hipFree(L.tx_pw);
        if (L.tx_rw) (void)hipFree(L.tx_rw);
        if (L.x) (void)hipFree(L.x);
        if (L.b) (void)hipFree(L.b);
        if (L.r) (void)hipFree(L.r);
        if (L.x2) (void)hipFree(L.x2);
        if (L.d) (void)hipFree(L.d);
    }
    s->levels.clear();
    s->pin_row = PinRow{};
    s->has_grid = false;
}

// halo_planes of memory below and above the owned planes (HALO_PAD_PLANES on a distributed level, 1 otherwise)
static int alloc_level_vectors(GridLevel &g, bool need_b, int halo_planes)
{
    const int64_t plane = g.n[0] * g.n[1];
    g.pad = (int64_t)halo_planes * plane;
    const size_t sz = sizeof(double) * (size_t)((g.k1 - g.k0 + 2 * halo_planes) * plane);
    PIB_HIP(hipMalloc(&g.x, sz));
    PIB_HIP(hipMalloc(&g.x2, sz));
    PIB_HIP(hipMalloc(&g.r, sz));
    PIB_MEMSET(g.x, 0, sz);
    PIB_MEMSET(g.x2, 0, sz);
    PIB_MEMSET(g.r, 0, sz);
    PIB_HIP(hipMalloc(&g.d, sz));
    PIB_MEMSET(g.d, 0, sz);
    if (need_b) {
        PIB_HIP(hipMalloc(&g.b, sz));
        PIB_MEMSET(g.b, 0, sz);
    }
    g.plane = plane;
    g.nloc = (g.k1 - g.k0) * plane;
    return 0;
}

template <int MODE>
static int launch_level(pib_solver *s, const GridLevel &g, double omega, const double *b, const double *xi, double *xo,
                        const double *pin_sum, bool guarded, hipStream_t q, double *dvec = nullptr, double a_d = 0.0);

// row-kernel geometry: 4 rows per workgroup (one per wave), row groups dealt to the 8 XCDs in contiguous ranges
struct RowGrid {
    int ngroups, per_xcd;
    dim3 grid;
};
static RowGrid row_grid(int64_t nrows, int64_t ncx, int lanes = ROW_LANES)
{
    RowGrid r;
    r.ngroups = (int)((nrows + 3) / 4);
    r.per_xcd = (r.ngroups + 7) / 8;
    r.grid = dim3((unsigned)(8 * r.per_xcd), (unsigned)((ncx + lanes - 1) / lanes));
    return r;
}
static int launch_prolong(const GridLevel &f, const GridLevel &c, const double *xc, double *xf, const Scalars *S, hipStream_t q)
{
    // the planes this launch touches, checked on the host tables (a plane outside the level, or a coarse plane outside the
    // memory the coarse vector has, is a planner error -- reported, not a memory fault)
    if (f.k1 > f.k0 && (f.k0 < 0 || f.k1 > f.n[2]))
        return fail(PIB_ERR_LIB, "gmg: prolongation onto planes [%lld, %lld) of a level with %lld", (long long)f.k0, (long long)f.k1,
                    (long long)f.n[2]);
    if (!f.hz_par.empty() && c.plane > 0) {
        const int64_t padc = c.pad / c.plane;
        for (int64_t k = f.k0; k < f.k1; ++k) {
            const int64_t lo = std::min(f.hz_par[(size_t)k], f.hz_oth[(size_t)k]), hi = std::max(f.hz_par[(size_t)k], f.hz_oth[(size_t)k]);
            if (lo < c.k0 - padc || hi >= c.k1 + padc)
                return fail(PIB_ERR_LIB, "gmg: fine plane %lld interpolates from coarse planes [%lld, %lld], the coarse slab [%lld, %lld) has %lld halo planes",
                            (long long)k, (long long)lo, (long long)hi, (long long)c.k0, (long long)c.k1, (long long)padc);
        }
    }
    const int64_t nrows = f.n[1] * (f.k1 - f.k0);
    const int vec_ok = (f.n[0] % 2 == 0 && (reinterpret_cast<uintptr_t>(xf) & 15u) == 0) ? 1 : 0;
    if (nrows >= 65536) {  // enough rows to keep the chip busy with four per wave
        const RowGrid r = row_grid((nrows + 3) / 4, c.n[0]);
        hipLaunchKernelGGL(k_prolong_rows<4>, r.grid, dim3(64, 4), 0, q, S, dev_of(f), dev_of(c), xc, xf, r.ngroups, r.per_xcd, vec_ok);
    } else {
        const RowGrid r = row_grid(nrows, c.n[0]);
        hipLaunchKernelGGL(k_prolong_rows<1>, r.grid, dim3(64, 4), 0, q, S, dev_of(f), dev_of(c), xc, xf, r.ngroups, r.per_xcd, vec_ok);
    }
    PIB_HIP(hipGetLastError());
    return 0;
}
// `c` carries the coarse planes to produce in k0 / k1 (the owned ones, which may be a part of a replicated level)
// the small-level kernels' box of coarse cells per workgroup (k_small_down: steps = pre, k_small_up: steps = post); false:
// the level does not qualify
constexpr int64_t SMALL_LEVEL_CELLS = 300000, SMALL_LEVEL_CELLS_3D = 40000;  // levels of at most this many cells (3-D: the margins cost more)
static bool small_level_boxes(const pib_solver *s, const GridLevel &f, const GridLevel &c, int steps, bool down, int bc[3], unsigned *blocks)
{
    if (!s->cfg.fuse_small_levels || steps < 1 || steps > 2) return false;
    if (f.nloc > SMALL_LEVEL_CELLS || f.zring || c.zring) return false;
    if (f.k0 != 0 || f.k1 != f.n[2] || c.k0 != 0 || c.k1 != c.n[2]) return false;  // both levels whole on this rank
    if (f.per != f.tper) return false;
    int nt = 0;
    for (int d = 0; d < 3; ++d) nt += f.n[d] > 1 ? 1 : 0;
    // a 3-D level pays 5-13 x in recomputed margins, all of it instruction issue on the workgroup's one CU
    if (nt == 3 && f.nloc > SMALL_LEVEL_CELLS_3D) return false;
    // boxes of 4^3 coarse cells on a 3-D level; 8^2 on a 2-D one while that gives at most one workgroup per CU (the kernels
    // hold one workgroup per CU: a second round of workgroups doubles the launch's time), else 16^2
    for (int side = (nt == 3 ? 4 : 8);; side *= 2) {
        int64_t cells = 1, nb = 1;
        bool ok = true;
        for (int d = 0; d < 3; ++d) {
            if (f.t_fst[d] == nullptr || f.t_par[d] == nullptr) return false;
            bc[d] = f.n[d] > 1 ? side : 1;
            int e = 2 * bc[d] + (down ? 3 + 2 * steps : 2 * steps);  // aggregates of at most two cells
            if (!((f.per >> d) & 1)) e = (int)std::min<int64_t>(e, f.n[d]);
            if (e > SM_MAXE) ok = false;
            cells *= e;
            nb *= (c.n[d] + bc[d] - 1) / bc[d];
        }
        if (!ok || cells > SM_MAXR) return false;
        if (nb <= 240 || nt == 3 || side >= SM_MAXBC) {
            if (nb > 256) return false;  // (more workgroups than CUs: the per-phase launches are faster)
            *blocks = (unsigned)nb;
            return true;
        }
    }
}

static int launch_restrict(const pib_solver *s, const GridLevel &f, const GridLevel &c, const double *rf, double *bc, const Scalars *S,
                           hipStream_t q)
{
    const int64_t nkc = c.k1 - c.k0;
    // transfers across a periodic z seam: the whole fine level is on this rank (the slab axis of a distributed level never wraps)
    const bool z_ok = !(f.tper & 4) || (f.k0 == 0 && f.k1 == f.n[2]);
    if (s->cfg.march && f.plain_pair && z_ok && f.n[0] % RX == 0 && f.n[1] % RY == 0 && nkc >= 4 &&
        nkc * c.plane * 8 >= (int64_t)s->cfg.march_min_cells) {
        // coarse planes per workgroup: 32 on a 512^3 fine level (1024 workgroups), 8 below
        const int CZ = nkc * c.plane >= ((int64_t)1 << 23) ? 32 : 8;
        hipLaunchKernelGGL(k_restrict_march, dim3((unsigned)(f.n[0] / RX), (unsigned)(f.n[1] / RY), (unsigned)((nkc + CZ - 1) / CZ)),
                           dim3(256), 0, q, S, dev_of(f), dev_of(c), rf, bc, CZ);
        PIB_HIP(hipGetLastError());
        return 0;
    }
    const int vec_ok = (f.n[0] % 2 == 0 && (reinterpret_cast<uintptr_t>(rf) & 15u) == 0) ? 1 : 0;
    // any aggregation, the level whole on this rank, no periodic z seam, 3-D, large enough to fill the chip with one wave per coarse
    // row and z-chunk: the z-marching form
    if (s->cfg.march && !(f.tper & 4) && f.k0 == 0 && f.k1 == f.n[2] && c.k0 == 0 && c.k1 == c.n[2] && f.n[1] > 1 && c.n[2] >= 4 &&
        c.n[0] * c.n[1] * c.n[2] >= std::min<int64_t>((int64_t)1 << 17, s->cfg.march_min_cells)) {  // (the tests lower the bound)
        // coarse planes per workgroup: so that there are about four workgroups per CU
        const int64_t wg_plane = ((c.n[1] + 3) / 4) * ((c.n[0] + 63) / 64);
        int CZ = (int)std::max<int64_t>(2, std::min<int64_t>(32, c.n[2] * wg_plane / 1024));
        hipLaunchKernelGGL(k_restrict_zmarch, dim3((unsigned)((c.n[1] + 3) / 4), (unsigned)((c.n[0] + 63) / 64), (unsigned)((c.n[2] + CZ - 1) / CZ)),
                           dim3(256), 0, q, S, dev_of(f), dev_of(c), rf, bc, CZ, vec_ok);
        PIB_HIP(hipGetLastError());
        return 0;
    }
    const RowGrid r = row_grid(c.n[1] * (c.k1 - c.k0), c.n[0], 64);  // aligned 64-lane chunks + edge loads (62 overlapping lanes measured slower here)
    hipLaunchKernelGGL(k_restrict_rows, r.grid, dim3(64, 4), 0, q, S, dev_of(f), dev_of(c), rf, bc, r.ngroups, r.per_xcd, vec_ok);
    PIB_HIP(hipGetLastError());
    return 0;
}

// ---- hint verification: stencil twin vs CSR SpMV on a fixed pseudo-random vector
// skip0: the pinned convention (row/column 0 of the CSR replaced by the identity) is the one place where
// the CSR and the singular stencil differ by design: x[0] = 0 removes column 0, row 0 is left out.
__global__ void k_fill_hash(int64_t n, int64_t g0, double *x, int skip0)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = (uint64_t)(g0 + p) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        x[p] = (skip0 && g0 + p == 0) ? 0.0 : (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}
__global__ void k_diff_sums(int64_t n, int64_t g0, int skip0, const double *a, const double *b, double *out /* [2] */)
{
    double d2 = 0.0, b2 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        if (skip0 && g0 + p == 0) continue;
        const double d = a[p] - b[p];
        d2 += d * d;
        b2 += b[p] * b[p];
    }
    for (int o = 32; o > 0; o >>= 1) {
        d2 += __shfl_down(d2, o, 64);
        b2 += __shfl_down(b2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], d2);
        atomicAdd(&out[1], b2);
    }
}

int gmg_verify(pib_solver *s)
{
    PIB_CHK(ensure_work(s, 4));
    hipStream_t q = s->stream;
    const int64_t n = s->A.n;
    double *X = s->vec(2), *Y1 = s->vec(3), *Y2 = s->vec(0);
    const int skip0 = (s->nullspace == PIB_NULLSPACE_PINNED) ? 1 : 0;
    hipLaunchKernelGGL(k_fill_hash, dim3(grid_blocks(n)), dim3(256), 0, q, n, s->A.row0, X, skip0);
    if (s->comm.nranks > 1) PIB_CHK(halo_exchange(s, X, q));
    PIB_CHK(spmv_rows(s, X, Y1, 0, n, nullptr, false, q));
    PIB_CHK(launch_level<0>(s, s->levels[0], 0.0, nullptr, X, Y2, nullptr, false, q));
    double *d_out = nullptr;
    PIB_HIP(hipMalloc(&d_out, 2 * sizeof(double)));
    PIB_HIP(hipMemsetAsync(d_out, 0, 2 * sizeof(double), q));
    hipLaunchKernelGGL(k_diff_sums, dim3(grid_blocks(n)), dim3(256), 0, q, n, s->A.row0, skip0, Y2, Y1, d_out);
    PIB_CHK(comm_allreduce_sum(s, d_out, 2, q));  // the same verdict on every rank
    double h[2] = {0, 0};
    PIB_HIP(hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, q));
    PIB_HIP(hipStreamSynchronize(q));
    PIB_HIP(hipFree(d_out));
    const double tol = 1e-10;
    if (!(h[0] <= tol * tol * h[1]) ) {
        gmg_release(s);
        return fail(PIB_ERR_ARG_WRONG,
                    "solver %s: the grid hint does not describe the matrix (stencil vs CSR mismatch %.3e relative)",
                    s->name.c_str(), std::sqrt(h[0] / (h[1] > 0 ? h[1] : 1.0)));
    }
    return 0;
}

// Register the structure of the matrix (pib_set_grid_hint / pib_assemble_poisson)
// and build the level hierarchy.  w[d]: n[d] widths; g[d]: n[d]-1 face factors (dt included).
int grid_register(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double *const g[3],
                  int nullspace, double dt_in)
{
    gmg_release(s);
    s->nullspace = nullspace;
    s->gmg_error.clear();
    if (dim != 2 && dim != 3) return fail(PIB_ERR_ARG_OUTOFRANGE, "grid hint: dim must be 2 or 3");
    const int P = s->comm.nranks, rank = s->comm.rank;
    // internal layout (nx, ny, nz) with 2-D -> (nx, 1, ny)
    int64_t nn[3];
    std::vector<double> hw[3], hg[3];
    bool pern[3] = {false, false, false};
    const int map[3] = {0, (dim == 3) ? 1 : -1, (dim == 3) ? 2 : 1};
    for (int d = 0; d < 3; ++d) {
        if (map[d] < 0) {
            nn[d] = 1;
            hw[d] = {1.0};
            hg[d].clear();
        } else {
            nn[d] = n[map[d]];
            if (w[map[d]] == nullptr || (nn[d] > 1 && g[map[d]] == nullptr)) return fail(PIB_ERR_ARG_NULL, "grid hint: null array");
            hw[d].assign(w[map[d]], w[map[d]] + nn[d]);
            // a periodic direction has one face more: g[n-1] couples cell n-1 and cell 0
            pern[d] = s->periodic[map[d]] != 0 && nn[d] > 1;
            hg[d].assign(g[map[d]], g[map[d]] + (nn[d] - 1) + (pern[d] ? 1 : 0));
        }
    }
    if (pern[2] && P > 1 && !s->comm.ring)
        return fail(PIB_ERR_SUP, "grid hint: a periodic slab axis on several ranks needs the on-device assembly (pib_assemble_poisson)");
    if (nn[0] * nn[1] * nn[2] != s->A.n_global)
        return fail(PIB_ERR_ARG_SIZ, "grid hint: %lld x %lld x %lld cells but the matrix has %lld rows", (long long)nn[0],
                    (long long)nn[1], (long long)nn[2], (long long)s->A.n_global);
    // slab of this rank must match the matrix rows
    int64_t k0, k1;
    slab_range(nn[2], P, rank, &k0, &k1);
    const int64_t plane0 = nn[0] * nn[1];
    if (k0 * plane0 != s->A.row0 || (k1 - k0) * plane0 != s->A.n)
        return fail(PIB_ERR_ARG_WRONG, "grid hint: the matrix rows of rank %d are not the z-slab [%lld,%lld) of the grid", rank,
                    (long long)k0, (long long)k1);

    // dt is folded in g; coarse g needs dt back: g = dt/dl  ->  dt = g[0]*0.5*(w0+w1) (any direction with >= 2 cells)
    double dt = (dt_in > 0.0) ? dt_in : 0.0;
    for (int d = 0; d < 3 && dt == 0.0; ++d)
        if (nn[d] > 1) dt = hg[d][0] * (0.5 * (hw[d][1] + hw[d][0]));

    // PINNED: row 0 of the singular operator without its diagonal (what MatZeroRowsColumns took out of the matrix): the solver's
    // recurrence for the residual's sum reads the few entries of p next to cell 0 with these coefficients (krylov.hip cg_s1)
    s->pin_row = PinRow{};
    if (nullspace == PIB_NULLSPACE_PINNED) {
        PinRow &pr = s->pin_row;
        pr.ready = true;
        if (k0 == 0) {
            const double vol = (hw[0][0] * hw[1][0]) * hw[2][0];
            const int64_t stride[3] = {1, nn[0], plane0};
            for (int d = 0; d < 3; ++d) {
                const int64_t nd = nn[d];
                if (nd <= 1) continue;
                const double w0 = hw[d][0];
                pr.off[pr.n] = stride[d];  // towards +d: face 0
                pr.coef[pr.n++] = (hg[d][0] / w0) * vol;
                if (pern[d]) {  // towards -d: the wrap face, cell nd - 1 (through the low halo plane on a ring of slabs)
                    pr.off[pr.n] = (d == 2 && P > 1) ? -plane0 : stride[d] * (nd - 1);
                    pr.coef[pr.n++] = (hg[d][(size_t)nd - 1] / w0) * vol;
                }
            }
        }
    }
    std::vector<GridLevel> &lv = s->levels;
    const int max_levels = std::max(1, s->cfg.max_levels);
    bool replicated = (P == 1);
    s->gmg_own.clear();
    {
        std::vector<std::pair<int64_t, int64_t>> o0((size_t)P);
        for (int r = 0; r < P; ++r) slab_range(nn[2], P, r, &o0[(size_t)r].first, &o0[(size_t)r].second);
        s->gmg_own.push_back(o0);
    }
    // Selective coarsening (oracle/csrc/gmg.c:orc_gmg_create): walking the cells of a direction, two neighbours
    // merge only if their combined width is <= 1.5 * hmin * 2^(level+1); cells that are already larger stay
    // alone, so the stretched far field of a PetIBM mesh catches up with the refined region and the level grids
    // become uniform (plain pairing on a uniform mesh).  On a distributed level the z aggregates never straddle
    // a slab boundary, so every coarse plane has one owner and one halo plane serves both transfers.
    double hmin = 0.0;
    for (int d = 0; d < 3; ++d)
        if (nn[d] > 1)
            for (double v : hw[d])
                if (hmin == 0.0 || v < hmin) hmin = v;
    int target_shift = 0;
    for (int l = 0;; ++l) {
        GridLevel G;
        G.dim = dim;
        for (int d = 0; d < 3; ++d) G.n[d] = nn[d];
        if (replicated) {
            G.k0 = 0;
            G.k1 = nn[2];
        } else {
            G.k0 = k0;
            G.k1 = k1;
        }
        G.replicated = replicated && P > 1;
        for (int d = 0; d < 3; ++d) {
            PIB_CHK(up(hw[d], &G.w[d]));
            const bool wrap = pern[d] && nn[d] > 1;  // a direction coarsened down to one cell has no face left
            if (wrap) G.per |= 1 << d;
            if (wrap && d == 2 && !replicated && P > 1) G.zring = true;
            if (l > 0) {
                hg[d].assign((size_t)std::max<int64_t>(nn[d] - 1, 0) + (wrap ? 1 : 0), 0.0);
                for (int64_t q = 0; q + 1 < nn[d]; ++q) {
                    const double dl = 0.5 * (hw[d][(size_t)q + 1] + hw[d][(size_t)q]);
                    const double v = 1.0 / dl;
                    hg[d][(size_t)q] = dt * v;
                }
                if (wrap) {
                    const double dl = 0.5 * (hw[d][0] + hw[d][(size_t)nn[d] - 1]);
                    const double v = 1.0 / dl;
                    hg[d][(size_t)nn[d] - 1] = dt * v;
                }
            }
            PIB_CHK(up(hg[d], &G.g[d]));
            {
                // rows divided by the cell volume: g / w per face, a function of the index along d only
                const int64_t nd = nn[d];
                std::vector<double> cm((size_t)nd, 0.0), cp((size_t)nd, 0.0), rw((size_t)nd, 1.0);
                for (int64_t q = 0; q < nd; ++q) {
                    const double wq = hw[d][(size_t)q];
                    rw[(size_t)q] = 1.0 / wq;
                    if (nd > 1) {
                        if (q > 0) cm[(size_t)q] = hg[d][(size_t)q - 1] / wq;
                        else if (wrap) cm[(size_t)q] = hg[d][(size_t)nd - 1] / wq;
                        if (q < nd - 1 || wrap) cp[(size_t)q] = hg[d][(size_t)q] / wq;
                    }
                }
                PIB_CHK(up(cm, &G.cm[d]));
                PIB_CHK(up(cp, &G.cp[d]));
                PIB_CHK(up(rw, &G.rw[d]));
            }
        }
        PIB_CHK(alloc_level_vectors(G, l > 0, (P > 1 && !replicated) ? HALO_PAD_PLANES : 1));
        const bool last = (l + 1 >= max_levels) || (nn[0] <= 2 && nn[1] <= 2 && nn[2] <= 2);
        // next level: aggregates and transfer tables
        std::vector<int32_t> par[3], oth[3], fst[3];
        std::vector<double> wpar[3], woth[3], cw[3];
        int64_t nc[3] = {nn[0], nn[1], nn[2]};
        bool twrap[3] = {false, false, false};
        bool merged_any = false;
        const auto &of = s->gmg_own.back();
        for (int tries = 0; !last && tries < 64 && !merged_any; ++tries, ++target_shift) {
            const double target = 1.5 * hmin * std::ldexp(1.0, l + 1 + target_shift);
            for (int d = 0; d < 3; ++d) {
                const int64_t n = nn[d];
                // the four fine cells a coarse cell gathers from must be distinct; across ranks the seam of the slab
                // axis stays a wall for the transfers (the halo planes serve the level operator only)
                twrap[d] = pern[d] && n >= 4 && !(d == 2 && !replicated && P > 1);
                par[d].assign((size_t)n, 0);
                oth[d].assign((size_t)n, 0);
                wpar[d].assign((size_t)n, 1.0);
                woth[d].assign((size_t)n, 0.0);
                fst[d].clear();
                cw[d].clear();
                std::vector<char> slab_start((size_t)n + 1, 0);
                if (d == 2 && !replicated)
                    for (int r = 0; r < P; ++r) slab_start[(size_t)of[(size_t)r].first] = 1;
                int64_t I = 0;
                for (int64_t q = 0; q < n; ++I) {
                    fst[d].push_back((int32_t)q);
                    if (n > 2 && q + 1 < n && !slab_start[(size_t)q + 1] && hw[d][(size_t)q] + hw[d][(size_t)q + 1] <= target) {
                        par[d][(size_t)q] = par[d][(size_t)q + 1] = (int32_t)I;
                        cw[d].push_back(hw[d][(size_t)q] + hw[d][(size_t)q + 1]);
                        q += 2;
                        merged_any = true;
                    } else {
                        par[d][(size_t)q] = (int32_t)I;
                        cw[d].push_back(hw[d][(size_t)q]);
                        q += 1;
                    }
                }
                fst[d].push_back((int32_t)n);
                nc[d] = I;
                for (int64_t q = 0; q < n; ++q) {
                    const int64_t Pq = par[d][(size_t)q];
                    const int64_t f0 = fst[d][(size_t)Pq], f1 = fst[d][(size_t)Pq + 1];
                    int64_t O = Pq;
                    double t = 0.0;
                    if (f1 - f0 == 2) {
                        const bool left = (q == f0);
                        O = left ? Pq - 1 : Pq + 1;
                        if (twrap[d]) O = (O + I) % I;  // across the periodic seam
                        if (O < 0 || O >= I)
                            O = Pq;
                        else {
                            const double sib = left ? hw[d][(size_t)q + 1] : hw[d][(size_t)q - 1];
                            t = sib / (cw[d][(size_t)Pq] + cw[d][(size_t)O]);
                        }
                    }
                    oth[d][(size_t)q] = (int32_t)O;
                    wpar[d][(size_t)q] = 1.0 - t;
                    woth[d][(size_t)q] = t;
                }
            }
        }
        if (last || !merged_any) {
            lv.push_back(G);
            break;
        }
        target_shift--;  // the loop's ++ after the successful try
        for (int d = 0; d < 3; ++d)
            if (twrap[d]) G.tper |= 1 << d;
        G.plain_pair = nc[0] * 2 == nn[0] && nc[1] * 2 == nn[1] && nc[2] * 2 == nn[2];
        G.hz_par = par[2];
        G.hz_oth = oth[2];
        for (int d = 0; d < 3; ++d) {
            PIB_CHK(up(par[d], &G.t_par[d]));
            PIB_CHK(up(oth[d], &G.t_oth[d]));
            PIB_CHK(up(fst[d], &G.t_fst[d]));
            PIB_CHK(up(wpar[d], &G.t_wpar[d]));
            PIB_CHK(up(woth[d], &G.t_woth[d]));
        }
        {
            // x tables packed per coarse cell for the row kernels
            std::vector<int2> fc((size_t)nc[0]);
            std::vector<double4> pw((size_t)nc[0]), rw((size_t)nc[0]);
            for (int64_t I = 0; I < nc[0]; ++I) {
                const int f0 = fst[0][(size_t)I], cnt = fst[0][(size_t)I + 1] - f0;
                int fl = f0 - 1, fr = f0 + cnt;
                if (twrap[0]) {
                    if (fl < 0) fl = (int)nn[0] - 1;
                    if (fr >= nn[0]) fr = 0;
                }
                fc[(size_t)I] = make_int2(f0, cnt);
                pw[(size_t)I] = make_double4(wpar[0][(size_t)f0], woth[0][(size_t)f0], cnt == 2 ? wpar[0][(size_t)f0 + 1] : 0.0,
                                             cnt == 2 ? woth[0][(size_t)f0 + 1] : 0.0);
                const double wl = (fl >= 0 && par[0][(size_t)fl] != I && oth[0][(size_t)fl] == I) ? woth[0][(size_t)fl] : 0.0;
                const double wr = (fr < nn[0] && par[0][(size_t)fr] != I && oth[0][(size_t)fr] == I) ? woth[0][(size_t)fr] : 0.0;
                rw[(size_t)I] = make_double4(wl, wpar[0][(size_t)f0], cnt == 2 ? wpar[0][(size_t)f0 + 1] : 0.0, wr);
            }
            PIB_CHK(up(fc, &G.tx_fc));
            PIB_CHK(up(pw, &G.tx_pw));
            PIB_CHK(up(rw, &G.tx_rw));
        }
        lv.push_back(G);
        {
            // ownership of the coarse planes = owner of their children (recorded also for the level at which the
            // hierarchy switches to replicated: it defines who restricts what before the all-gather)
            std::vector<std::pair<int64_t, int64_t>> oc((size_t)P);
            for (int r = 0; r < P; ++r) {
                const int64_t b = of[(size_t)r].first, e = of[(size_t)r].second;
                if (replicated || e <= b)
                    oc[(size_t)r] = {0, nc[2]};
                else
                    oc[(size_t)r] = {par[2][(size_t)b], (int64_t)par[2][(size_t)e - 1] + 1};
            }
            if (!replicated) {
                // stay distributed only while every rank keeps >= 2 coarse planes and the level is big enough
                // to amortise the halo latency
                bool ok = true;
                for (int r = 0; r < P && ok; ++r)
                    if (oc[(size_t)r].second - oc[(size_t)r].first < 2) ok = false;
                if (nc[0] * nc[1] * nc[2] <= (int64_t)s->cfg.agglomerate_below) ok = false;
                if (ok) {
                    k0 = oc[(size_t)rank].first;
                    k1 = oc[(size_t)rank].second;
                } else {
                    replicated = true;
                }
            }
            s->gmg_own.push_back(oc);
        }
        for (int d = 0; d < 3; ++d) {
            hw[d].swap(cw[d]);
            nn[d] = nc[d];
        }
    }
    // every rank derives the same aggregates and ownership from the same width arrays, so neighbours agree.
    s->has_grid = true;
    // the Krylov work vectors double as level-0 vectors: the same halo memory around their owned part
    s->work_pad = (P > 1) ? HALO_PAD_PLANES * plane0 : 0;

    // verify the hint against the CSR: stencil twin vs CSR SpMV on a fixed vector (not when the structure only describes
    // the preconditioner's operator: BN order > 1, bn.hip)
    if (s->hint_pc_only) return 0;
    return gmg_verify(s);
}

// ---- halos of a distributed level ------------------------------------------------------------------------------
// A z-slab keeps up to HALO_PAD_PLANES planes of memory below and above its owned planes.  An exchange fills the first
// `depth` of them with the neighbours' owned planes (contiguous: no pack kernel); every stencil kernel that follows may
// then run on the owned planes PLUS the ghost planes whose inputs are still valid -- each application of the 7-point
// stencil costs one plane of validity per side.  The ghost values a rank computes are the bits its neighbour computes
// for the same cells (same kernels, same expressions), so a cycle on P slabs is the cycle on one rank; what it saves
// is messages: one exchange of the right-hand side per level on the way down (deep enough for the smoothing steps, the
// residual and the restriction's reach), one of the coarse correction per level on the way up, instead of one per kernel.

static int exchange_planes(pib_solver *s, const GridLevel &g, double *x_owned, int depth, hipStream_t q)
{
    const int r = s->comm.rank, P = s->comm.nranks;
    const bool ring = s->comm.ring;
    const int64_t cnt = (int64_t)depth * g.plane;
    const int64_t lo = (r > 0 || ring) ? cnt : 0, hi = (r < P - 1 || ring) ? cnt : 0;
    return halo_exchange_planes(s, x_owned, g.nloc, lo, hi, lo, hi, q);
}

// all-gather the owned coarse planes of level `lc` (ownership = the parents of the finer level's slab planes) into the
// replicated level vector
static int gather_level(pib_solver *s, int lc, int64_t coarse_plane, const double *owned, int64_t n_owned,
                        double *full_owned_base, hipStream_t q)
{
    const int P = s->comm.nranks;
    std::vector<int64_t> cnt((size_t)P), off((size_t)P);
    for (int r = 0; r < P; ++r) {
        const auto &o = s->gmg_own[(size_t)lc][(size_t)r];
        cnt[(size_t)r] = (o.second - o.first) * coarse_plane;
        off[(size_t)r] = o.first * coarse_plane;
    }
    if (cnt[(size_t)s->comm.rank] != n_owned) return fail(PIB_ERR_LIB, "gmg gather: inconsistent slab sizes");
    return comm_allgatherv(s, owned, full_owned_base, cnt, off, q);
}

// runs of planes the LDS-tiled kernels serve (k_presmooth2, k_level_march, k_prolong_smooth): 3-D, tile-divisible, enough
// cells in the run to fill the chip; a periodic z only on the whole level (plane -1 = plane nz - 1 of the same vector)
static bool tiles_ok(const GridLevel &g) { return g.n[1] > 1 && g.n[2] > 1 && g.n[0] % FX == 0 && g.n[1] % FY == 0; }
static bool run_whole(const GridLevel &g, int64_t kb, int64_t kc) { return g.k0 + kb == 0 && kc == g.n[2]; }
static bool march_run_ok(const pib_solver *s, const GridLevel &g, int64_t kb, int64_t kc)
{
    const bool z_ok = !(g.per & 4) || (!g.zring && run_whole(g, kb, kc));
    return z_ok && tiles_ok(g) && kc >= 8 && kc * g.plane >= (int64_t)s->cfg.march_min_cells;
}
// the fused kernels (two pre-smoothing steps; prolongation + first post-smoothing step)
static bool fused_run_ok(const pib_solver *s, const GridLevel &g, int64_t kb, int64_t kc)
{
    const bool z_ok = !(g.per & 4) || (!g.zring && run_whole(g, kb, kc));
    return z_ok && tiles_ok(g) && kc >= 2 && kc * g.plane >= (int64_t)s->cfg.march_min_cells;
}
// planes per workgroup: 64 on a 512^3 run (2048 workgroups), 16 on a 256^3 one (1024)
#ifndef PIB_MARCH_PLANES_BIG
#define PIB_MARCH_PLANES_BIG 64
#endif
static int march_planes(const GridLevel &g, int64_t kc)
{
    if (kc * g.plane >= ((int64_t)1 << 26)) return PIB_MARCH_PLANES_BIG;
    return 16;
}

// The Krylov sums z.r, z.z, sum z a level-0 kernel left as per-workgroup partials: their fixed-order reduction into S->red[0..2].
// Round 5: when the solver asks for it (gmg_defer_dots) and the partials are few enough for ONE workgroup, the reduction is left
// to the solver's own closing kernel of the cycle (krylov.hip k_dots_tail: the three sums, z[0] and -- on one rank -- the scalar
// step of the iteration in a single launch instead of four).
constexpr int DEFER_DOTS_MAX = 32768;
static int reduce_dots(pib_solver *s, double *part, int part_stride, int count, hipStream_t q)
{
    if (s->gmg_defer_dots && count <= DEFER_DOTS_MAX) {
        s->gmg_pending_part = part;
        s->gmg_pending_stride = part_stride;
        s->gmg_pending_count = count;
        return 0;
    }
    double *stage = part + 3 * (int64_t)part_stride;
    hipLaunchKernelGGL(k_reduce_big, dim3(BIG_STAGE, 3), dim3(256), 0, q, s->d_s, part, part_stride, count, stage);
    hipLaunchKernelGGL(k_finalize_big, dim3(3), dim3(64), 0, q, s->d_s, stage);
    PIB_HIP(hipGetLastError());
    return 0;
}

// MODE on the planes [kb, kb + kc) relative to the first owned plane (kb < 0 / kb + kc > nk: ghost planes); the vectors
// point at the first OWNED plane.  dots: mode 8 sums over the owned planes only.
template <int MODE>
static int launch_level_planes(pib_solver *s, const GridLevel &g, int64_t kb, int64_t kc, double omega, const double *b,
                               const double *xi, double *xo, const double *pin_sum, bool guarded, hipStream_t q,
                               double *dvec = nullptr, double a_d = 0.0)
{
    if (kc <= 0) return 0;
    GridLevel sub = g;
    sub.k0 = g.k0 + kb;
    sub.k1 = sub.k0 + kc;
    const int64_t o = kb * g.plane;
    b = b ? b + o : b;
    xi = xi ? xi + o : xi;
    xo += o;
    dvec = dvec ? dvec + o : dvec;
    // owned planes inside the run (local indices of the run)
    const int dlo = (int)std::max<int64_t>(0, -kb), dhi = (int)std::min<int64_t>(kc, (g.k1 - g.k0) - kb);
    double *part = nullptr;
    int part_stride = 0;
    const int64_t nx = g.n[0], ny = g.n[1];
    const unsigned nk = (unsigned)kc;
    const bool march = (MODE == 2 || MODE == 3 || MODE == 8) && s->cfg.march && march_run_ok(s, g, kb, kc) &&
                       ((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(xi) | reinterpret_cast<uintptr_t>(xo)) & 31u) == 0;
    const int FZ = march_planes(g, kc);
    const dim3 mg((unsigned)(nx / FX), (unsigned)(ny / FY), (unsigned)((nk + FZ - 1) / FZ));
    auto aligned = [](const void *p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec_ok = aligned(b) && aligned(xi) && aligned(xo) && aligned(dvec);
    const int C = (vec_ok && nx % 4 == 0) ? 4 : ((vec_ok && nx % 2 == 0) ? 2 : 1);
    const dim3 sg((unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, (nx / C * ny + 255) / 256)), nk);
    if (MODE == 8) {
        // per-workgroup partials of the fused sums
        const int64_t cap = march ? (int64_t)mg.x * mg.y * mg.z : (int64_t)sg.x * sg.y;
        if (s->gmg_part_cap < cap) {
            if (s->d_gmg_part) PIB_HIP(hipFree(s->d_gmg_part));
            s->d_gmg_part = nullptr;
            PIB_HIP(hipMalloc(&s->d_gmg_part, sizeof(double) * (3 * (size_t)cap + 3 * BIG_STAGE)));
            s->gmg_part_cap = cap;
        }
        part = s->d_gmg_part;
        part_stride = (int)s->gmg_part_cap;
    }
    const Scalars *S = guarded ? s->d_s : nullptr;
    int nparts = 0;
    if (march) {
        constexpr int M = (MODE == 3) ? 3 : (MODE == 8 ? 8 : 2);
        hipLaunchKernelGGL((k_level_march<M>), mg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, part, part_stride, FZ, dlo, dhi);
        nparts = (int)(mg.x * mg.y * mg.z);
    } else {
        if (C == 4)
            hipLaunchKernelGGL((k_level<MODE, 4>), sg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, dvec, a_d, part, part_stride, dlo, dhi);
        else if (C == 2)
            hipLaunchKernelGGL((k_level<MODE, 2>), sg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, dvec, a_d, part, part_stride, dlo, dhi);
        else
            hipLaunchKernelGGL((k_level<MODE, 1>), sg, dim3(256), 0, q, S, dev_of(sub), omega, b, xi, xo, pin_sum, dvec, a_d, part, part_stride, dlo, dhi);
        nparts = (int)(sg.x * sg.y);
    }
    PIB_HIP(hipGetLastError());
    if (MODE == 8) PIB_CHK(reduce_dots(s, part, part_stride, nparts, q));
    return 0;
}

// the whole owned part of the level
template <int MODE>
static int launch_level(pib_solver *s, const GridLevel &g, double omega, const double *b, const double *xi, double *xo,
                        const double *pin_sum, bool guarded, hipStream_t q, double *dvec, double a_d)
{
    return launch_level_planes<MODE>(s, g, 0, g.k1 - g.k0, omega, b, xi, xo, pin_sum, guarded, q, dvec, a_d);
}

// coarse ghost planes (beyond a rank's owned coarse planes) that the interpolation of its boundary planes and of `e` fine
// ghost planes per side reads -- the maximum over ALL ranks, so that every rank asks for the same exchange depth
static int coarse_need(const pib_solver *s, int l, int e)
{
    const GridLevel &f = s->levels[(size_t)l];
    int need = 0;
    for (int r = 0; r < s->comm.nranks; ++r) {
        const auto &of = s->gmg_own[(size_t)l][(size_t)r];
        const auto &oc = s->gmg_own[(size_t)l + 1][(size_t)r];
        if (of.second <= of.first) continue;
        for (int g2 = 0; g2 <= e; ++g2) {
            const int64_t up = of.second - 1 + g2, dn = of.first - g2;
            if (up < f.n[2]) need = std::max<int>(need, (int)(std::max(f.hz_par[(size_t)up], f.hz_oth[(size_t)up]) - (oc.second - 1)));
            if (dn >= 0) need = std::max<int>(need, (int)(oc.first - std::min(f.hz_par[(size_t)dn], f.hz_oth[(size_t)dn])));
        }
    }
    return std::max(need, 0);
}

// Several ranks: the fused residual update changes WHICH vector travels (w instead of r) and where the reductions sit, so all
// ranks must take it or none.  Slabs differ by a plane (512 x 512 x 380 on 8 ranks: 47 or 48 planes, either side of the 12.58 M-cell
// threshold): the conditions are evaluated for EVERY rank's slab -- plane counts, the cell threshold, the partial-sum slots, the
// captured-graph limit -- from quantities all ranks hold (gmg_own); what is left to the launch site (pointer alignment, halo depths)
// is the same on every rank by construction (pads and strides are multiples of the level's plane and of 4 doubles).
static bool fused_update_slabs_all_ranks(const pib_solver *s, const GridLevel &g)
{
    for (int q2 = 0; q2 < s->comm.nranks; ++q2) {
        const int64_t nk = s->gmg_own[0][(size_t)q2].second - s->gmg_own[0][(size_t)q2].first;
        if (nk < 8 || nk * g.plane <= s->cfg.graph_max_rows) return false;
        const int FZ = march_planes(g, nk);
        if (!fused_run_ok(s, g, 0, nk) || (g.n[0] / FX) * (g.n[1] / FY) * ((nk + FZ - 1) / FZ + 2) > PIB_MAXPART) return false;
    }
    return true;
}

// One V-cycle: z = M^-1 r.   r, z: ghost-padded work vectors of the Krylov solver
// (their ghost planes double as the level-0 halo planes).
// PCG may leave its residual update to the level-0 pre-smoothing march (k_presmooth2<., 1>): one rank, the fused march serves
// the whole fine level, Jacobi smoothing, no pinned unknown, and few enough workgroups for the solver's partial-sum slots
bool gmg_fused_update_ok(const pib_solver *s)
{
    if (!s->has_grid || s->levels.empty() || !s->gmg_error.empty()) return false;
    if (s->cfg.smoother == Smoother::CHEBYSHEV || !s->cfg.fuse_presmooth) return false;
    // (a pinned pressure row: the compatible right-hand side needs the NEW residual's sum before the march that forms it --
    // it comes from the recurrence sum r - alpha sum w, krylov.hip cg_s1, when the row next to cell 0 is known: pin_row)
    if (s->nullspace == PIB_NULLSPACE_PINNED && (s->cfg.pin_sum_local == 0 || !s->pin_row.ready)) return false;
    if (s->levels.size() < 2) return false;
    const GridLevel &g = s->levels[0];
    const int64_t nk = g.k1 - g.k0;
    const int FZ = march_planes(g, nk);
    if (s->comm.nranks > 1) {
        // z-slabs (round 4): level 0 distributed with deep halos, every rank's slab thick enough for them, the two-step march
        // (the cycle decides again with its own predicate at the launch site and falls back to the separate pass if it must)
        if (s->cfg.fuse_residual_update < 0 || !s->cfg.deep_halo || g.replicated || g.zring || (g.per & 4)) return false;
        if (std::max(1, s->cfg.presweeps) * (s->cfg.sweep_pairs ? 2 : 1) < 2) return false;
        return fused_update_slabs_all_ranks(s, g);
    }
    if (s->A.n <= s->cfg.graph_max_rows) return false;  // (a captured iteration cannot alternate the residual's two buffers)
    if (g.k0 != 0 || nk != g.n[2] || !fused_run_ok(s, g, 0, nk)) return false;
    return (g.n[0] / FX) * (g.n[1] / FY) * ((nk + FZ - 1) / FZ) <= PIB_MAXPART;
}

int gmg_apply(pib_solver *s, const double *r, double *z, hipStream_t q)
{
    if (!s->has_grid || s->levels.empty())
        return fail(PIB_ERR_ORDER, "solver %s: multigrid preconditioner without grid structure", s->name.c_str());
    if (!s->gmg_error.empty()) return fail(PIB_ERR_SUP, "solver %s: %s", s->name.c_str(), s->gmg_error.c_str());
    const bool guarded = s->gmg_guarded;
    const Scalars *S = guarded ? s->d_s : nullptr;
    s->halo_fresh = nullptr;
    s->gmg_dots_done = false;
    s->gmg_pending_count = 0;
    s->z_halo_depth = 0;
    const double omega = s->cfg.smoother_relaxation;
    const bool cheb = (s->cfg.smoother == Smoother::CHEBYSHEV);
    const int deg = cheb ? std::max(1, s->cfg.cheby_degree) : 1;  // one Chebyshev "sweep" = a degree-`deg` polynomial
    // (damped Jacobi: a sweep of the solver file is a fused pair of steps unless pib_sweep_pairs=0 -- Config::sweep_pairs)
    const int pairs = (!cheb && s->cfg.sweep_pairs) ? 2 : 1;
    const int pre = std::max(1, s->cfg.presweeps) * deg * pairs, post = std::max(0, s->cfg.postsweeps) * deg * pairs;
    const int nl = (int)s->levels.size();
    const int P = s->comm.nranks, rank = s->comm.rank;
    // PINNED: the sum of the residual that makes the right-hand side compatible -- red[5], summed by the pass that formed the
    // residual, or pin_sigma when that pass is this cycle's own first march (krylov.hip cg_s1: sum r - alpha sum w, the second sum
    // from the few entries of p next to cell 0)
    const double *pin = (s->nullspace == PIB_NULLSPACE_PINNED) ? (s->gmg_pin_local ? &s->d_s->pin_sigma : &s->d_s->red[5]) : nullptr;
    std::vector<double *> cur((size_t)nl, nullptr);  // current iterate buffer per level (owned pointer)
    const double lmax = s->cfg.cheby_lmax, lmin = lmax / s->cfg.cheby_ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;

    // ---- per-level halo facts and the validity (in ghost planes per side) of every vector this cycle touches
    struct LI {
        bool dist, lo, hi;
        int maxd;    // deepest exchange
        int cdepth;  // ghost planes a kernel may compute on (0 on a periodic slab axis: the ghost planes of the outer ranks
                     // are the planes at the other end of the axis, the kernels index the mesh arrays by plane number)
        int64_t nk;
    };
    std::vector<LI> li((size_t)nl);
    for (int l = 0; l < nl; ++l) {
        const GridLevel &g = s->levels[(size_t)l];
        LI &I = li[(size_t)l];
        I.dist = P > 1 && !g.replicated;
        I.lo = I.dist && (rank > 0 || s->comm.ring);
        I.hi = I.dist && (rank < P - 1 || s->comm.ring);
        I.nk = g.k1 - g.k0;
        I.maxd = I.cdepth = 0;
        if (I.dist) {
            int m = (s->cfg.deep_halo && !g.zring) ? HALO_MAX_DEPTH : 1;
            for (int q2 = 0; q2 < P; ++q2)  // an exchange takes planes the NEIGHBOUR owns
                m = (int)std::min<int64_t>(m, s->gmg_own[(size_t)l][(size_t)q2].second - s->gmg_own[(size_t)l][(size_t)q2].first);
            I.maxd = std::max(1, m);
            I.cdepth = g.zring ? 0 : I.maxd;
        }
    }
    std::vector<std::pair<const double *, int>> vd;
    auto valid = [&](const double *v) -> int {
        for (auto &e : vd)
            if (e.first == v) return e.second;
        return 0;
    };
    auto set_valid = [&](const double *v, int d) {
        for (auto &e : vd)
            if (e.first == v) {
                e.second = d;
                return;
            }
        vd.push_back({v, d});
    };
    // make `vec` of level l valid on d ghost planes per side
    auto need = [&](int l, const double *vec, int d) -> int {
        const LI &I = li[(size_t)l];
        if (!I.dist || d <= 0 || valid(vec) >= d) return 0;
        if (d > I.maxd) return fail(PIB_ERR_LIB, "gmg: halo depth %d not available on level %d", d, l);
        PIB_CHK(exchange_planes(s, s->levels[(size_t)l], const_cast<double *>(vec), d, q));
        set_valid(vec, d);
        return 0;
    };
    // The same exchange on the communication stream while the main stream goes on with work that needs owned planes
    // only; wait_halo() joins.  Used for the exchange of a level's right-hand side when it is large (>= 1 MiB per
    // neighbour: at 512^3 on 8 GPUs that is 8 MiB on level 0 and 1.5 MiB on level 1): the first kernel of the way down
    // runs on the interior planes meanwhile and on the planes next to the neighbours afterwards.
    bool halo_pending = false;
    auto need_async = [&](int l, const double *vec, int d) -> int {
        const LI &I = li[(size_t)l];
        if (!I.dist || d <= 0 || valid(vec) >= d) return 0;
        if (d > I.maxd) return fail(PIB_ERR_LIB, "gmg: halo depth %d not available on level %d", d, l);
        PIB_HIP(hipEventRecord(s->ev_ready, q));
        PIB_HIP(hipStreamWaitEvent(s->stream_comm, s->ev_ready, 0));
        PIB_CHK(exchange_planes(s, s->levels[(size_t)l], const_cast<double *>(vec), d, s->stream_comm));
        PIB_HIP(hipEventRecord(s->ev_halo, s->stream_comm));
        set_valid(vec, d);
        halo_pending = true;
        return 0;
    };
    auto wait_halo = [&]() -> int {
        if (halo_pending) PIB_HIP(hipStreamWaitEvent(q, s->ev_halo, 0));
        halo_pending = false;
        return 0;
    };
    // planes [a, a + c) relative to the first owned plane for a run that reaches d ghost planes into the neighbours
    auto run = [&](int l, int d, int64_t &a, int64_t &c) {
        const LI &I = li[(size_t)l];
        a = I.lo ? -d : 0;
        c = I.nk + (I.hi ? d : 0) - a;
    };
    // Depth a stencil kernel produces: what is desired, no more than its right-hand side is valid, one plane less than its
    // input; an input without any valid ghost plane is exchanged first.  The rule depends on nothing rank-specific and
    // the fused kernels below record exactly what the steps they replace would: every rank issues the same exchanges.
    auto stencil_depth = [&](int l, int desired, const double *in, const double *b, int *out) -> int {
        const LI &I = li[(size_t)l];
        *out = 0;
        if (!I.dist) return 0;
        int o = std::max(0, std::min(std::min(desired, I.maxd - 1), valid(b)));
        if (valid(in) >= 1) o = std::min(o, valid(in) - 1);
        else PIB_CHK(need(l, in, o + 1));
        *out = o;
        return 0;
    };

    // `nsteps` smoothing steps on level l: iterate in `a` (result left in `a` after the swaps), spare buffer `c`.
    // Jacobi: x <- x + omega D^-1 (b - A x).  Chebyshev-Jacobi: three-term recurrence over [lmin, lmax] of D^-1 A,
    // restarted for every segment (oracle/csrc/gmg.c:cheby).  The last step's result is valid on d_final ghost planes,
    // the one before on d_final + 1, ... (as far as the inputs allow; a missing plane is exchanged).
    auto smooth_seq = [&](int l, const double *b, const double *pin_l, double *&a, double *&c, int nsteps, bool from_zero,
                          int d_final, bool dots_in_last = false) -> int {
        GridLevel &g = s->levels[(size_t)l];
        const LI &I = li[(size_t)l];
        double rho = 1.0 / sigma;
        double *dvec = g.d + g.pad;
        for (int sw = 0; sw < nsteps; ++sw) {
            const int desired = d_final + (nsteps - 1 - sw);
            int64_t ka, kc;
            if (from_zero && sw == 0 && nsteps >= 2 && !cheb && s->cfg.fuse_presmooth) {
                // steps 0 and 1 in one kernel; the result lands where step 1 would have put it
                int o = I.dist ? std::min(std::min(desired - 1, I.maxd - 1), valid(b) - 1) : 0;
                run(l, std::max(o, 0), ka, kc);
                if (o >= 0 && fused_run_ok(s, g, ka, kc) &&
                    ((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 31u) == 0) {
                    const int FZ = march_planes(g, kc);
                    auto launch = [&](int64_t ra, int64_t rc) {
                        if (rc <= 0) return;
                        GridLevel sub = g;
                        sub.k0 = g.k0 + ra;
                        sub.k1 = sub.k0 + rc;
                        hipLaunchKernelGGL(k_presmooth2<0>, dim3((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((rc + FZ - 1) / FZ)),
                                           dim3(256), 0, q, S, dev_of(sub), omega, b + ra * g.plane, c + ra * g.plane, pin_l, FZ, nullptr);
                    };
                    if (l == 0 && s->gmg_upd.w != nullptr && I.dist) {
                        // ... on z-slabs (round 4): w came through the exchange instead of the residual, whose ghost planes follow the
                        // recurrence -- every launch updates the planes it loads, the run's planes and the one beyond it at either end
                        // (the depth the next cycle reads r on) are written to the NEW residual's buffer, the sums cover the owned planes
                        const double *ro = s->gmg_upd.r_old, *uw = s->gmg_upd.w;
                        double *rn = const_cast<double *>(b);
                        int nblk = 0;
                        auto launch_upd = [&](int64_t ra, int64_t rc, int wext) {
                            if (rc <= 0) return;
                            GridLevel sub = g;
                            sub.k0 = g.k0 + ra;
                            sub.k1 = sub.k0 + rc;
                            const dim3 grid((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((rc + FZ - 1) / FZ));
                            hipLaunchKernelGGL((k_presmooth2<0, 1>), grid, dim3(256), 0, q, S, dev_of(sub), omega, ro + ra * g.plane, c + ra * g.plane, pin_l, FZ,
                                               (double *)nullptr, uw + ra * g.plane, rn + ra * g.plane, s->d_part + 4 * (int64_t)PIB_MAXPART, (int)PIB_MAXPART,
                                               wext, (int)g.k0, (int)g.k1, nblk);
                            nblk += (int)(grid.x * grid.y * grid.z);
                        };
                        // the run reaches o ghost planes, the residual is read (and kept) one plane deeper
                        const int wlo = (I.lo && o + 1 <= valid(uw)) ? 1 : 0, whi = (I.hi && o + 1 <= valid(uw)) ? 2 : 0;
                        if (halo_pending) {
                            const int64_t ia = I.lo ? 1 : 0, ie = I.nk - (I.hi ? 1 : 0);
                            launch_upd(ia, ie - ia, 0);
                            PIB_CHK(wait_halo());
                            launch_upd(ka, ia - ka, wlo);
                            launch_upd(ie, ka + kc - ie, whi);
                        } else
                            launch_upd(ka, kc, wlo | whi);
                        PIB_HIP(hipGetLastError());
                        if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                        s->gmg_upd.used = true;
                        PIB_CHK(s->gmg_upd.after(s, nblk, q));
                    } else if (l == 0 && s->gmg_upd.w != nullptr) {
                        // PCG's residual update folded into this march (k_presmooth2<., 1>): b is the NEW residual's buffer
                        if (I.dist || halo_pending || ka != 0 || kc != I.nk) return fail(PIB_ERR_LIB, "fused residual update: level 0 is not a whole single-rank level");
                        const dim3 grid((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ));
                        const int nblk = (int)(grid.x * grid.y * grid.z);
                        if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                        hipLaunchKernelGGL((k_presmooth2<0, 1>), grid, dim3(256), 0, q, S, dev_of(g), omega, s->gmg_upd.r_old, c, pin_l, FZ,
                                           (double *)nullptr, s->gmg_upd.w, const_cast<double *>(b), s->d_part + 4 * (int64_t)PIB_MAXPART,
                                           (int)PIB_MAXPART);
                        PIB_HIP(hipGetLastError());
                        s->gmg_upd.used = true;
                        PIB_CHK(s->gmg_upd.after(s, nblk, q));
                    } else if (halo_pending) {
                        // planes whose two steps read owned planes of b only, while the exchange is in flight
                        const int64_t ia = I.lo ? 1 : 0, ie = I.nk - (I.hi ? 1 : 0);
                        launch(ia, ie - ia);
                        PIB_CHK(wait_halo());
                        launch(ka, ia - ka);
                        launch(ie, ka + kc - ie);
                    } else
                        launch(ka, kc);
                    PIB_HIP(hipGetLastError());
                    set_valid(c, o);
                    std::swap(a, c);
                    sw = 1;
                    continue;
                }
            }
            PIB_CHK(wait_halo());
            if (from_zero && sw == 0) {
                const int o = I.dist ? std::max(0, std::min(std::min(desired, I.cdepth), valid(b))) : 0;  // pointwise: as deep as b
                run(l, o, ka, kc);
                if (cheb) PIB_CHK(launch_level_planes<6>(s, g, ka, kc, 1.0 / theta, b, nullptr, a, pin_l, guarded, q, dvec, 0.0));
                else PIB_CHK(launch_level_planes<1>(s, g, ka, kc, omega, b, nullptr, a, pin_l, guarded, q));
                set_valid(a, o);
                continue;
            }
            int o;
            PIB_CHK(stencil_depth(l, desired, a, b, &o));
            run(l, o, ka, kc);
            if (cheb) {
                double a_d = 0.0, a_z = 1.0 / theta;
                if (sw > 0) {
                    const double rho_new = 1.0 / (2.0 * sigma - rho);
                    a_d = rho_new * rho;
                    a_z = 2.0 * rho_new / delta;
                    rho = rho_new;
                }
                PIB_CHK(launch_level_planes<5>(s, g, ka, kc, a_z, b, a, c, pin_l, guarded, q, dvec, a_d));
            } else {
                const bool dots = dots_in_last && sw + 1 == nsteps;
                if (dots) {
                    PIB_CHK(launch_level_planes<8>(s, g, ka, kc, omega, b, a, c, pin_l, guarded, q));
                    s->gmg_dots_done = true;
                } else
                    PIB_CHK(launch_level_planes<2>(s, g, ka, kc, omega, b, a, c, pin_l, guarded, q));
            }
            set_valid(c, o);
            std::swap(a, c);
        }
        return 0;
    };

    // first level handled by the single-workgroup coarse tail (never level 0; Jacobi only; no communication)
    int tail0 = nl;
    // the smallest levels are launch-bound (~5 us kernels, five to seven per level): one workgroup walks those of <= 1024
    // cells.  A 2-D flow case spends most of its V-cycle there; on a 256^3 / 512^3 grid it is 0.2 ms of a time step / of
    // a solve (17.4 -> 17.2 ms, 88.2 -> 88.1 ms), and taking larger levels into the one workgroup costs more than the
    // launches it saves (4096: 90.3 ms, 32768: 20.4 ms per Taylor-Green step; docs/history/measured_and_rejected.md)
    const int tail_cells = (s->cfg.coarse_tail >= 0) ? s->cfg.coarse_tail : 1024;
    if (!cheb && tail_cells) {
        for (int l = nl - 1; l >= 1; --l) {
            const GridLevel &g = s->levels[(size_t)l];
            const bool local = (s->comm.nranks == 1) || g.replicated;
            if (local && g.nloc <= std::min<int64_t>(TAIL_MAX_CELLS, tail_cells) && nl - l < TAIL_MAX_LEVELS) tail0 = l; else break;
        }
        if (nl - tail0 < 2) tail0 = nl;  // a single level is what k_coarsest already does
    }
    // depth of the level's right-hand side on the way down: the pre-smoothing steps, the residual and the restriction's
    // reach of one plane; level 0 also carries the post-smoothing steps and one plane of the result z, so that neither
    // the corrected iterate nor p = z + beta p (the next Krylov product's input) needs an exchange of its own
    // ... and a coarser distributed level delivers ITS final iterate on as many ghost planes as the prolongation onto the finer
    // level reads (pib_deep_halo=2, round 4): the exchange of the coarse correction on the way up -- a collective with nothing to
    // hide behind -- goes, the right-hand side of that level is exchanged deeper on the way down instead (the same bytes: at
    // 512^3 / 8 five planes of level 1 instead of 3 + 2, six of level 2 instead of 3 + 3) and a few more ghost planes of two
    // latency-bound levels are relaxed redundantly.  Where the level's slabs are too thin for that depth the exchange stays.
    std::vector<int> fin_l((size_t)nl, 0);
    fin_l[0] = (li[0].dist && li[0].maxd > 1) ? 1 : 0;
    if (s->cfg.deep_halo >= 2 && !cheb && post >= 1)
        for (int l = 1; l < nl; ++l) {
            if (!li[(size_t)l].dist || !li[(size_t)l - 1].dist) continue;
            const int want = coarse_need(s, l - 1, fin_l[(size_t)l - 1] + post);
            if (want > 0 && pre + post - 1 + want <= li[(size_t)l].cdepth) fin_l[(size_t)l] = want;
        }
    auto final_depth = [&](int l) -> int { return fin_l[(size_t)l]; };
    auto down_depth = [&](int l) -> int {
        const LI &I = li[(size_t)l];
        if (!I.dist) return 0;
        return std::min(I.cdepth, std::max(pre + 1, pre + post - 1 + final_depth(l)));
    };
    // prolongation and both post-smoothing steps of level l in one march (k_prolong_smooth2): V(., 2), a level and its coarser
    // one whole on this rank, the conditions of k_prolong_smooth; decided here, before the way down, because the pair of steps
    // is one swap of the level's buffers (level 0 must end in z)
    auto post2_ok = [&](int l) -> bool {
        if (!s->cfg.fuse_post_pair || post != 2 || cheb || !s->cfg.fuse_prolong || l + 1 >= nl || l >= tail0) return false;
        const GridLevel &g = s->levels[(size_t)l];
        const GridLevel &c1 = s->levels[(size_t)l + 1];
        const LI &I = li[(size_t)l];
        const bool whole = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && c1.replicated)) && g.k0 == 0 &&
                           g.k1 == g.n[2] && c1.k0 == 0 && c1.k1 == c1.n[2];
        // ... or both levels in z-slabs of the same ranks: the result on final_depth ghost planes needs the old iterate and the
        // coarse values two planes deeper and b one (what the way down leaves valid; exchanged at the launch if not)
        const int fin0 = fin_l[(size_t)l];  // (final_depth)
        const bool slabs = I.dist && li[(size_t)l + 1].dist && !(g.per & 4) && std::min(I.maxd, I.cdepth) >= fin0 + 2 &&
                           coarse_need(s, l, fin0 + 2) <= li[(size_t)l + 1].maxd;
        if (!(whole || slabs) || g.zring) return false;
        const bool per_ok = g.per == g.tper && (!(g.per & 4) || g.n[2] >= 8);
        if (!g.plain_pair || !per_ok || !fused_run_ok(s, g, 0, I.nk) || g.n[1] % UTY != 0) return false;
        auto al32 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31u) == 0; };
        return al32(g.x + g.pad) && al32(g.x2 + g.pad) && al32(l == 0 ? (const void *)r : (const void *)(g.b + g.pad)) && (l != 0 || al32(z));
    };
    // ---- downward leg
    for (int l = 0; l < nl; ++l) {
        GridLevel &g = s->levels[(size_t)l];
        const LI &I = li[(size_t)l];
        const int64_t pl = g.plane;
        if (l == tail0) {
            TailArgs T;
            std::memset(&T, 0, sizeof T);
            T.nlev = nl - tail0;
            T.omega = omega;
            T.pre = pre;
            T.post = post;
            T.sweeps = s->cfg.coarsest_sweeps;
            for (int q2 = 0; q2 < T.nlev; ++q2) {
                GridLevel &t = s->levels[(size_t)(tail0 + q2)];
                T.lv[q2].L = dev_of(t);
                T.lv[q2].xa = t.x + t.pad;
                T.lv[q2].xb = t.x2 + t.pad;
                T.lv[q2].b = t.b + t.pad;
                T.lv[q2].r = t.r + t.pad;
            }
            const int swaps = (pre - 1) + post;
            cur[(size_t)l] = (swaps % 2 == 0) ? (g.x + g.pad) : (g.x2 + g.pad);
            int64_t need = 0;
            bool fits = s->cfg.coarse_tail_lds != 0;
            for (int q2 = 0; q2 < T.nlev; ++q2) {
                const GridLevel &t = s->levels[(size_t)(tail0 + q2)];
                T.lds_off[q2] = (int)need;
                need += 3 * t.nloc;  // iterate, spare (the residual's place on the way down), right-hand side
                fits = fits && !t.zring && t.nloc == t.n[0] * t.n[1] * t.n[2];
            }
            if (!fits || need > TAIL_POOL)
                for (int q2 = 0; q2 < TAIL_MAX_LEVELS; ++q2) T.lds_off[q2] = -1;
            // the tables of the tail's levels, packed (see TailArgs): offsets first, the block itself when the arguments change
            struct Piece { const void *src; int off, bytes; };
            std::vector<Piece> pieces;
            int used = 0;
            bool tabs_ok = true;
            auto put_d = [&](const double *src, int n) -> int {
                if (src == nullptr) tabs_ok = false;
                const int off = used;
                pieces.push_back({src, off, 8 * n});
                used += n;
                return off;
            };
            auto put_i = [&](const int32_t *src, int n) -> int {
                if (src == nullptr) tabs_ok = false;
                const int off = used;
                pieces.push_back({src, off, 4 * n});
                used += (n + 1) / 2;
                return off;
            };
            for (int q2 = 0; q2 < T.nlev; ++q2) {
                const GridLevel &t = s->levels[(size_t)(tail0 + q2)];
                TailArgs::Tabs &o = T.tt[q2];
                for (int d = 0; d < 3; ++d) {
                    const int n = (int)t.n[d];
                    o.w[d] = put_d(t.w[d], n);
                    o.rw[d] = put_d(t.rw[d], n);
                    o.cm[d] = put_d(t.cm[d], n);
                    o.cp[d] = put_d(t.cp[d], n);
                    if (q2 + 1 < T.nlev) {
                        const int nc = (int)s->levels[(size_t)(tail0 + q2 + 1)].n[d];
                        o.wpar[d] = put_d(t.t_wpar[d], n);
                        o.woth[d] = put_d(t.t_woth[d], n);
                        o.par[d] = put_i(t.t_par[d], n);
                        o.oth[d] = put_i(t.t_oth[d], n);
                        o.fst[d] = put_i(t.t_fst[d], nc);
                    }
                }
            }
            // (vectors and tables go to LDS together or not at all: the kernel has one instance for either)
            const bool tab_lds = T.lds_off[0] >= 0 && tabs_ok && used <= TAIL_TAB;
            if (!tab_lds)
                for (int q2 = 0; q2 < TAIL_MAX_LEVELS; ++q2) T.lds_off[q2] = -1;
            T.tab_used = tab_lds ? used : 0;
            if (s->d_tail_tab == nullptr) PIB_HIP(hipMalloc(&s->d_tail_tab, sizeof(double) * TAIL_TAB));
            T.tab_src = static_cast<const double *>(s->d_tail_tab);
            T.out0 = cur[(size_t)l];
            // the argument block lives in HBM, rewritten only when it changes (levels, sweeps and buffers are fixed per set-up)
            if (s->d_tail_args == nullptr) PIB_HIP(hipMalloc(&s->d_tail_args, sizeof(TailArgs)));
            if (s->h_tail_args.size() != sizeof(TailArgs) || std::memcmp(s->h_tail_args.data(), &T, sizeof(TailArgs)) != 0) {
                s->h_tail_args.assign(reinterpret_cast<const char *>(&T), reinterpret_cast<const char *>(&T) + sizeof(TailArgs));
                PIB_HIP(hipMemcpyAsync(s->d_tail_args, s->h_tail_args.data(), sizeof(TailArgs), hipMemcpyHostToDevice, q));
                if (tab_lds)
                    for (const Piece &pc : pieces)
                        PIB_HIP(hipMemcpyAsync(static_cast<double *>(s->d_tail_tab) + pc.off, pc.src, (size_t)pc.bytes, hipMemcpyDeviceToDevice, q));
            }
            if (T.lds_off[0] >= 0) hipLaunchKernelGGL(k_coarse_tail<true>, dim3(1), dim3(1024), 0, q, S, static_cast<const TailArgs *>(s->d_tail_args));
            else hipLaunchKernelGGL(k_coarse_tail<false>, dim3(1), dim3(1024), 0, q, S, static_cast<const TailArgs *>(s->d_tail_args));
            PIB_HIP(hipGetLastError());
            break;
        }
        const double *b = (l == 0) ? r : g.b + g.pad;
        const double *pin_l = (l == 0) ? pin : nullptr;
        double *xa = g.x + g.pad, *xb = g.x2 + g.pad;
        if (l == nl - 1) {
            double *out = (l == 0) ? z : xa;
            if (g.nloc <= 4096 && !I.dist) {
                hipLaunchKernelGGL(k_coarsest, dim3(1), dim3(256), 0, q, S, dev_of(g), omega, s->cfg.coarsest_sweeps, b, xa, xb,
                                   out);
                PIB_HIP(hipGetLastError());
            } else {
                PIB_CHK(launch_level<1>(s, g, omega, b, nullptr, xa, pin_l, guarded, q, nullptr, 0.0));
                double *a = xa, *c = xb;
                for (int sw = 1; sw < s->cfg.coarsest_sweeps; ++sw) {
                    set_valid(a, 0);
                    PIB_CHK(need(l, a, 1));
                    PIB_CHK(launch_level<2>(s, g, omega, b, a, c, pin_l, guarded, q, nullptr, 0.0));
                    std::swap(a, c);
                }
                if (a != out) PIB_HIP(hipMemcpyAsync(out, a, sizeof(double) * (size_t)g.nloc, hipMemcpyDeviceToDevice, q));
            }
            set_valid(out, 0);
            cur[(size_t)l] = out;
            break;
        }
        // number of ping-pong swaps left on this level: (pre-1) + post; arrange that level 0 ends in z
        double *a = xa, *c = xb;
        if (l == 0) {
            // (both post-smoothing steps in one march, k_prolong_smooth2, are ONE swap)
            const int swaps = (pre - 1) + (post2_ok(0) ? 1 : post);
            // final buffer after `swaps` swaps starting from a: a if even else c
            if (swaps % 2 == 0) { a = z; c = xa; } else { a = xa; c = z; }
        }
        double *rr = g.r + g.pad;
        if (l == 0 && s->gmg_upd.w != nullptr) {
            // PCG left r = r_old - alpha w to this cycle's first march: decided HERE, with the pointers the march would get, by the
            // predicate of its two launch sites below; if it cannot take the update, the update runs as a pass of its own first
            const int FZ0 = march_planes(g, I.nk);
            auto al32 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31u) == 0; };
            // (several ranks: the slab-dependent parts for EVERY rank's slab -- fused_update_slabs_all_ranks -- so that all ranks fall
            // back together or not at all)
            bool site = (s->cfg.fuse_residual_update == 1 || s->cfg.fuse_residual_update == -1) && !cheb && s->cfg.fuse_presmooth && al32(b) && (pre >= 2 ? al32(c) : (al32(a) && al32(rr)));
            if (I.dist) site = site && fused_update_slabs_all_ranks(s, g);
            else site = site && fused_run_ok(s, g, 0, I.nk) && (g.n[0] / FX) * (g.n[1] / FY) * ((I.nk + FZ0 - 1) / FZ0 + 2) <= PIB_MAXPART;
            // z-slabs: the two-step march only, deep halos (the residual's depth is what w is exchanged to), aligned planes
            if (I.dist)
                site = site && pre >= 2 && !g.zring && down_depth(0) >= 2 && down_depth(0) <= I.maxd && al32(s->gmg_upd.w) && al32(s->gmg_upd.r_old) &&
                       (g.plane % 4) == 0;
            if (!site) {
                if (s->gmg_upd.fallback == nullptr) return fail(PIB_ERR_LIB, "fused residual update: no fallback registered");
                PIB_CHK(s->gmg_upd.fallback(s, const_cast<double *>(b), q));
                s->gmg_upd.w = nullptr;
                s->gmg_upd.used = true;
            }
        }
        // a small level that is whole on this rank: pre-smoothing, residual and restriction in one launch
        {
            GridLevel &cg1 = s->levels[(size_t)l + 1];
            int bc3[3];
            unsigned nblk = 0;
            const bool local_pair = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && cg1.replicated));
            // (level 0 too -- a 2-D case's finest level is such a level, with the pinned row's correction of its right-hand side if
            // there is one -- unless PCG's residual update is folded in; the way up of level 0 keeps its launches: its last step
            // delivers the Krylov sums in a fixed order of workgroups)
            const bool level_ok = l >= 1 || s->gmg_upd.w == nullptr;
            if (level_ok && !cheb && local_pair && small_level_boxes(s, g, cg1, pre, true, bc3, &nblk)) {
                if (l == 0) {  // no swaps on the way down: the swaps of the way up must end in z (the fused pair of
                               // post-smoothing steps, k_prolong_smooth2, is ONE swap -- the same count as above)
                    const int up_swaps = post2_ok(0) ? 1 : post;
                    if (up_swaps % 2 == 0) { a = z; c = xa; } else { a = xa; c = z; }
                }
                hipLaunchKernelGGL(k_small_down, dim3(nblk), dim3(SM_NT), 0, q, S, dev_of(g), dev_of(cg1), omega, pre, b, a, cg1.b + cg1.pad, bc3[0],
                                   bc3[1], bc3[2], pin_l);
                PIB_HIP(hipGetLastError());
                set_valid(a, 0);
                set_valid(cg1.b + cg1.pad, 0);
                cur[(size_t)l] = a;
                s->gmg_spare[(size_t)l] = c;
                continue;
            }
        }
        // the right-hand side on as many ghost planes as the way down (and, on level 0, the way up) consumes
        const int Dd = down_depth(l);
        set_valid(b, 0);
        // (PCG's residual update inside the first march, on slabs: w = A p is what travels; the new residual comes out of the
        // march on the same planes, the neighbours' included)
        const double *xv = (l == 0 && I.dist && s->gmg_upd.w != nullptr) ? s->gmg_upd.w : b;
        set_valid(xv, 0);
        if (I.dist && s->cfg.overlap_min_bytes >= 0 && I.nk >= 4 && (int64_t)Dd * g.plane * 8 >= (int64_t)s->cfg.overlap_min_bytes)
            PIB_CHK(need_async(l, xv, Dd));
        else
            PIB_CHK(need(l, xv, Dd));
        if (xv != b) set_valid(b, valid(xv));
        if (pre == 1 && !cheb) {
            PIB_CHK(wait_halo());
            // one pre-smoothing step from zero and the residual of its result: x1 kept and r valid on o ghost planes
            const int o = I.dist ? std::max(0, std::min(std::min(Dd - 1, I.maxd - 1), valid(b) - 1)) : 0;
            int64_t ka, kc;
            run(l, o, ka, kc);
            if (s->cfg.fuse_presmooth && fused_run_ok(s, g, ka, kc) && (!I.dist || valid(b) >= o + 1) &&
                ((reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(rr)) & 31u) == 0) {
                GridLevel sub = g;  // both in one march (b read once)
                sub.k0 = g.k0 + ka;
                sub.k1 = sub.k0 + kc;
                const int FZ = march_planes(g, kc);
                if (l == 0 && s->gmg_upd.w != nullptr) {
                    if (I.dist || ka != 0 || kc != I.nk) return fail(PIB_ERR_LIB, "fused residual update: level 0 is not a whole single-rank level");
                    const dim3 grid((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ));
                    const int nblk = (int)(grid.x * grid.y * grid.z);
                    if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                    hipLaunchKernelGGL((k_presmooth2<1, 1>), grid, dim3(256), 0, q, S, dev_of(sub), omega, s->gmg_upd.r_old, a, pin_l, FZ, rr,
                                       s->gmg_upd.w, const_cast<double *>(b), s->d_part + 4 * (int64_t)PIB_MAXPART, (int)PIB_MAXPART);
                    PIB_HIP(hipGetLastError());
                    s->gmg_upd.used = true;
                    PIB_CHK(s->gmg_upd.after(s, nblk, q));
                } else {
                hipLaunchKernelGGL(k_presmooth2<1>, dim3((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ)),
                                   dim3(256), 0, q, S, dev_of(sub), omega, b + ka * pl, a + ka * pl, pin_l, FZ, rr + ka * pl);
                PIB_HIP(hipGetLastError());
                }
            } else {
                // x1 one plane deeper than it is kept (pointwise: as deep as b), then its residual
                const int dx = I.dist ? std::min(std::min(o + 1, valid(b)), I.cdepth) : 0;
                int64_t xa2, xc2;
                run(l, dx, xa2, xc2);
                PIB_CHK(launch_level_planes<1>(s, g, xa2, xc2, omega, b, nullptr, a, pin_l, guarded, q));
                set_valid(a, dx);
                if (I.dist && dx < o + 1) PIB_CHK(need(l, a, o + 1));
                PIB_CHK(launch_level_planes<3>(s, g, ka, kc, omega, b, a, rr, pin_l, guarded, q));
            }
            set_valid(a, o);
            set_valid(rr, o);
        } else {
            // the pre-smoothed iterate is kept as deep as the way up wants it (the corrected iterate starts from it)
            const int want_x = I.dist ? std::max(2, post + final_depth(l)) : 0;
            // the two steps, the residual and the restriction in ONE march (k_down_march) where both marches below would run on a
            // level that is whole on this one rank and not periodic
            {
                const GridLevel &c1 = s->levels[(size_t)l + 1];
                const int64_t nkc = c1.k1 - c1.k0;
                const int FZ = march_planes(g, I.nk);
                const bool upd = l == 0 && s->gmg_upd.w != nullptr;
                auto al32 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 31u) == 0; };
                const bool fits = s->cfg.fuse_down_march && pre == 2 && !cheb && s->cfg.fuse_presmooth && s->cfg.fuse_residual_restrict &&
                                  s->cfg.march && !I.dist && !li[(size_t)l + 1].dist && s->comm.nranks == 1 && g.k0 == 0 && g.k1 == g.n[2] &&
                                  c1.k0 == 0 && c1.k1 == c1.n[2] && !g.zring && g.per == 0 && g.tper == 0 && g.plain_pair && !halo_pending &&
                                  fused_run_ok(s, g, 0, I.nk) && g.n[0] % RX == 0 && g.n[1] % 8 == 0 && (g.n[2] & 1) == 0 && 2 * nkc == g.n[2] && nkc >= 4 &&
                                  nkc * c1.plane * 8 >= (int64_t)s->cfg.march_min_cells && (FZ & 1) == 0 && FZ / 2 <= 32 && al32(b) && al32(c) &&
                                  (!upd || (al32(s->gmg_upd.w) && al32(s->gmg_upd.r_old)));
                if (fits) {
                    const int CZ = FZ / 2;
                    // (128 x 8 tiles, 512 threads, 214 registers and no spill: 1.19 ms per 512^3 launch against 0.84 + 0.51; the 128 x 16
                    // tile -- 40 instead of 47 B per cell -- needs 768 threads, i.e. 168 registers, and spilled 65: 3.4 ms)
                    // ... 64 x 16 tiles (42 B per cell, 212 registers): 60.9-61.0 against 61.2-61.4 ms per solve on one box, and the Krylov
                    // sums no longer in k_presmooth2's grouping: not kept)
                    constexpr int TX = 128, TY = 8;
                    const dim3 grid((unsigned)(g.n[0] / TX), (unsigned)(g.n[1] / TY), (unsigned)((nkc + CZ - 1) / CZ));
                    if (upd) {
                        const int nblk = (int)(grid.x * grid.y * grid.z);  // (128 x 8 tiles: k_presmooth2's workgroups)
                        if (nblk > PIB_MAXPART) return fail(PIB_ERR_LIB, "fused residual update: too many workgroups for the partial sums");
                        hipLaunchKernelGGL((k_down_march<1, TX, TY>), grid, dim3(down_threads(TX, TY)), 0, q, S, dev_of(g), dev_of(c1), omega, s->gmg_upd.r_old, c,
                                           c1.b + c1.pad, CZ, pin_l, s->gmg_upd.w, const_cast<double *>(b), s->d_part + 4 * (int64_t)PIB_MAXPART, (int)PIB_MAXPART);
                        PIB_HIP(hipGetLastError());
                        s->gmg_upd.used = true;
                        PIB_CHK(s->gmg_upd.after(s, nblk, q));
                    } else {
                        hipLaunchKernelGGL((k_down_march<0, TX, TY>), grid, dim3(down_threads(TX, TY)), 0, q, S, dev_of(g), dev_of(c1), omega, b, c, c1.b + c1.pad, CZ,
                                           pin_l, (const double *)nullptr, (double *)nullptr, (double *)nullptr, 0);
                        PIB_HIP(hipGetLastError());
                    }
                    set_valid(c, 0);
                    std::swap(a, c);
                    set_valid(c1.b + c1.pad, 0);
                    cur[(size_t)l] = a;
                    s->gmg_spare[(size_t)l] = c;
                    continue;
                }
            }
            PIB_CHK(smooth_seq(l, b, pin_l, a, c, pre, true, want_x));
            // residual and restriction in one march where the marching restriction would run (k_resid_restrict_march): the
            // residual never goes to HBM
            {
                const GridLevel &c1 = s->levels[(size_t)l + 1];
                const int64_t nkc = c1.k1 - c1.k0;
                bool whole = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && c1.replicated)) && g.k0 == 0 &&
                             g.k1 == g.n[2] && c1.k0 == 0 && c1.k1 == c1.n[2] && !g.zring;
                // ... or both levels in z-slabs of the same ranks (an aggregate never straddles a cut): the march reads the iterate
                // two planes and b one plane beyond the slab -- what the way down left valid there -- and the residual's own
                // ghost plane needs no exchange any more
                if (!whole && I.dist && li[(size_t)l + 1].dist && !g.zring && !(g.per & 4) && valid(a) >= 2 && valid(b) >= 1 &&
                    std::min(I.maxd, I.cdepth) >= 2)
                    whole = true;
                if (s->cfg.fuse_residual_restrict && whole && s->cfg.march && g.plain_pair && g.per == g.tper &&
                    g.n[0] % RX == 0 && g.n[1] % RY == 0 && nkc >= 4 && nkc * c1.plane * 8 >= (int64_t)s->cfg.march_min_cells) {
                    const int CZ = nkc * c1.plane >= ((int64_t)1 << 23) ? 32 : 8;  // (as launch_restrict)
                    hipLaunchKernelGGL(k_resid_restrict_march, dim3((unsigned)(g.n[0] / RX), (unsigned)(g.n[1] / RY), (unsigned)((nkc + CZ - 1) / CZ)),
                                       dim3(256), 0, q, S, dev_of(g), dev_of(c1), b, a, c1.b + c1.pad, CZ, pin_l);
                    PIB_HIP(hipGetLastError());
                    set_valid(c1.b + c1.pad, 0);
                    cur[(size_t)l] = a;
                    s->gmg_spare[(size_t)l] = c;
                    continue;
                }
            }
            int o;
            PIB_CHK(stencil_depth(l, 1, a, b, &o));
            int64_t ka, kc;
            run(l, o, ka, kc);
            PIB_CHK(launch_level_planes<3>(s, g, ka, kc, omega, b, a, rr, pin_l, guarded, q));
            set_valid(rr, o);
        }
        PIB_CHK(need(l, rr, 1));  // the restriction reaches one fine plane beyond the slab
        GridLevel &cg = s->levels[(size_t)l + 1];
        if (cg.replicated && !g.replicated && s->comm.nranks > 1) {
            // restrict the owned coarse planes into a scratch (cg.r), then all-gather into cg.b
            GridLevel own = cg;
            own.k0 = s->gmg_own[(size_t)l + 1][(size_t)s->comm.rank].first;
            own.k1 = s->gmg_own[(size_t)l + 1][(size_t)s->comm.rank].second;
            own.nloc = (own.k1 - own.k0) * cg.plane;
            double *scratch = cg.r + cg.pad;
            PIB_CHK(launch_restrict(s, g, own, rr, scratch, S, q));
            PIB_CHK(gather_level(s, l + 1, cg.plane, scratch, own.nloc, cg.b + cg.pad, q));
        } else {
            PIB_CHK(launch_restrict(s, g, cg, rr, cg.b + cg.pad, S, q));
        }
        cur[(size_t)l] = a;
        // remember the spare buffer in g.scratch for the upward leg
        s->gmg_spare[(size_t)l] = c;
    }
    // ---- upward leg
    for (int l = std::min(nl - 2, tail0 - 1); l >= 0; --l) {
        GridLevel &g = s->levels[(size_t)l];
        GridLevel &cg = s->levels[(size_t)l + 1];
        const LI &I = li[(size_t)l];
        const int64_t pl = g.plane;
        const double *b = (l == 0) ? r : g.b + g.pad;
        const double *pin_l = (l == 0) ? pin : nullptr;
        double *a = cur[(size_t)l], *c = s->gmg_spare[(size_t)l];
        double *xc = cur[(size_t)l + 1];
        const int fin = final_depth(l);
        // the corrected iterate x + P e on e ghost planes: as deep as the post-smoothing consumes, no deeper than the
        // pre-smoothed iterate is valid and than the coarse correction can be had
        int e = 0;
        if (I.dist) {
            // (no deeper than a kernel may compute on: 0 on a periodic slab axis, whose outer ghost planes are the other end of the
            // axis -- V(., 2) on such a level used to take the iterate's exchanged plane for a plane it could correct: plane -1)
            e = std::min(std::min(fin + post, std::min(I.maxd, I.cdepth)), valid(a));
            if (li[(size_t)l + 1].dist)
                while (e > 0 && coarse_need(s, l, e) > li[(size_t)l + 1].maxd) --e;
            if (li[(size_t)l + 1].dist) PIB_CHK(need(l + 1, xc, coarse_need(s, l, e)));
        }
        {
            // a small level whole on this rank: prolongation and the post-smoothing in one launch
            int bc3[3];
            unsigned nblk = 0;
            const bool local_pair = !I.dist && !li[(size_t)l + 1].dist && (s->comm.nranks == 1 || (g.replicated && cg.replicated));
            if (l >= 1 && !cheb && local_pair && small_level_boxes(s, g, cg, post, false, bc3, &nblk)) {
                hipLaunchKernelGGL(k_small_up, dim3(nblk), dim3(SM_NT), 0, q, S, dev_of(g), dev_of(cg), omega, post, b, a, xc, c, bc3[0], bc3[1],
                                   bc3[2]);
                PIB_HIP(hipGetLastError());
                set_valid(c, 0);
                std::swap(a, c);
                cur[(size_t)l] = a;
                continue;
            }
        }
        const bool dots_l = l == 0 && s->gmg_want_dots && !cheb;
        if (post2_ok(l)) {
            // prolongation + both post-smoothing steps in one march; on level 0 the second one delivers the Krylov sums
            // planes per workgroup: four iterations fill the pipeline, so no fewer than 32 -- except where the Krylov sums are
            // formed, which keep the grouping (and the bits) of k_level_march<8>
            int64_t ka = 0, kc = I.nk;
            if (I.dist) {
                // the run of planes the result is wanted on, and its inputs as deep as the two steps reach
                const int cn = coarse_need(s, l, fin + 2);
                if (li[(size_t)l + 1].dist) PIB_CHK(need(l + 1, xc, cn));
                if (valid(a) < fin + 2) PIB_CHK(need(l, a, fin + 2));
                if (valid(b) < fin + 1) PIB_CHK(need(l, b, fin + 1));
                run(l, fin, ka, kc);
            }
            GridLevel sub = g;
            sub.k0 = g.k0 + ka;
            sub.k1 = sub.k0 + kc;
            const int FZ = dots_l ? march_planes(g, kc) : std::max(march_planes(g, kc), 32);
            const dim3 mg((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / UTY), (unsigned)((kc + FZ - 1) / FZ));
            const double *bq = b + ka * pl;
            double *aq = a + ka * pl, *cq = c + ka * pl;
            if (dots_l) {
                const int64_t needp = 2 * (int64_t)mg.x * mg.y * mg.z;  // (a partial per 128 x 8 tile, as k_level_march<8>)
                if (needp > s->gmg_part_cap) {
                    if (s->d_gmg_part) (void)hipFree(s->d_gmg_part);
                    s->d_gmg_part = nullptr;
                    PIB_HIP(hipMalloc(&s->d_gmg_part, sizeof(double) * (size_t)(3 * needp + 3 * BIG_STAGE)));
                    s->gmg_part_cap = needp;
                }
                double *part = s->d_gmg_part;
                const int part_stride = (int)s->gmg_part_cap;
                hipLaunchKernelGGL(k_prolong_smooth2<1>, mg, dim3(UNT), 0, q, S, dev_of(sub), dev_of(cg), omega, bq, xc, aq, cq, FZ, part, part_stride,
                                   (int)g.k0, (int)g.k1, pin_l);
                PIB_CHK(reduce_dots(s, part, part_stride, (int)needp, q));
                s->gmg_dots_done = true;
            } else
                hipLaunchKernelGGL(k_prolong_smooth2<0>, mg, dim3(UNT), 0, q, S, dev_of(sub), dev_of(cg), omega, bq, xc, aq, cq, FZ, (double *)nullptr, 0,
                                   (int)g.k0, (int)g.k1, pin_l);
            PIB_HIP(hipGetLastError());
            set_valid(c, I.dist ? fin : 0);
            std::swap(a, c);
            cur[(size_t)l] = a;
            if (l == 0 && a != z) return fail(PIB_ERR_LIB, "gmg: internal buffer parity error");
            continue;
        }
        // prolongation + first post-smoothing step in one march (the corrected iterate never goes to HBM)
        bool fused = false;
        // periodic levels: operator and transfers wrap alike, z with >= 8 planes (the ring of coarse planes counts through the seam)
        const bool per_ok = g.per == g.tper && (!(g.per & 4) || g.n[2] >= 8);
        if (s->cfg.fuse_prolong && !cheb && post >= 1 && g.plain_pair && per_ok && (!I.dist || e >= 1)) {
            const int o = I.dist ? std::min(std::min(e - 1, fin + post - 1), valid(b)) : 0;
            int64_t ka, kc;
            run(l, o, ka, kc);
            if (fused_run_ok(s, g, ka, kc) &&
                ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(b)) & 31u) == 0) {
                GridLevel sub = g;
                sub.k0 = g.k0 + ka;
                sub.k1 = sub.k0 + kc;
                const int FZ = march_planes(g, kc);
                const dim3 mg((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((kc + FZ - 1) / FZ));
                const int dlo = (int)std::max<int64_t>(0, -ka), dhi = (int)std::min<int64_t>(kc, I.nk - ka);
                if (post == 1 && dots_l) {
                    // the only post-smoothing step also delivers z.r, z.z, sum z (what mode 8 does)
                    const int64_t needp = (int64_t)mg.x * mg.y * mg.z;
                    if (needp > s->gmg_part_cap) {
                        if (s->d_gmg_part) (void)hipFree(s->d_gmg_part);
                        s->d_gmg_part = nullptr;
                        PIB_HIP(hipMalloc(&s->d_gmg_part, sizeof(double) * (size_t)(3 * needp + 3 * BIG_STAGE)));
                        s->gmg_part_cap = needp;
                    }
                    double *part = s->d_gmg_part;
                    const int part_stride = (int)s->gmg_part_cap;
                    hipLaunchKernelGGL(k_prolong_smooth<1>, mg, dim3(256), 0, q, S, dev_of(sub), dev_of(cg), omega, b + ka * pl, xc,
                                       a + ka * pl, c + ka * pl, pin_l, FZ, part, part_stride, dlo, dhi);
                    PIB_CHK(reduce_dots(s, part, part_stride, (int)needp, q));
                    s->gmg_dots_done = true;
                } else
                    hipLaunchKernelGGL(k_prolong_smooth<0>, mg, dim3(256), 0, q, S, dev_of(sub), dev_of(cg), omega, b + ka * pl, xc,
                                       a + ka * pl, c + ka * pl, pin_l, FZ, nullptr, 0, dlo, dhi);
                PIB_HIP(hipGetLastError());
                set_valid(c, o);
                std::swap(a, c);
                PIB_CHK(smooth_seq(l, b, pin_l, a, c, post - 1, false, fin, dots_l));
                fused = true;
            }
        }
        if (!fused) {
            int64_t ka, kc;
            run(l, e, ka, kc);
            GridLevel sub = g;
            sub.k0 = g.k0 + ka;
            sub.k1 = sub.k0 + kc;
            PIB_CHK(launch_prolong(sub, cg, xc, a + ka * pl, S, q));
            set_valid(a, e);
            PIB_CHK(smooth_seq(l, b, pin_l, a, c, post, false, fin, dots_l));
        }
        cur[(size_t)l] = a;
        if (l == 0 && a != z) return fail(PIB_ERR_LIB, "gmg: internal buffer parity error");
    }
    s->z_halo_depth = li[0].dist ? valid(z) : 0;
    return 0;
}

// y = A x with the matrix-free stencil (K2); x ghost-padded (halo exchanged here)
// ---- the Krylov product w = A p from the stencil twin (the device time step's Poisson solves on large grids:
// `pib_matrix_free_poisson`): 16 B/row for the product + 16 B/row for the p.w partials instead of the CSR's 104 B/row.
// Same operator to rounding (the twin sums face differences, the CSR row sums products: verified to 1e-10 at registration).
// PINNED: row 0 of the matrix is the identity; its column is zero, which the twin reproduces as long as p[0] = 0 -- the
// Krylov vectors keep that entry at zero.
__global__ __launch_bounds__(64) void k_twin_row0(const Scalars *__restrict__ S, const double *__restrict__ x, double *__restrict__ y)
{
    if (S != nullptr && S->done) return;
    if (threadIdx.x == 0) y[0] = x[0];
}
__global__ __launch_bounds__(256) void k_twin_dot(const Scalars *__restrict__ S, int64_t n, const double *__restrict__ x,
                                                  const double *__restrict__ y, double *__restrict__ part)
{
    if (S != nullptr && S->done) return;
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(lo + chunk, n);
    double v = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += x[i] * y[i];
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// x.y as spmv_launch_blocks() fixed-order partial sums (what the SpMV's fused dot delivers)
int dot_partials(pib_solver *s, const double *x, const double *y, double *part, bool guarded, hipStream_t q)
{
    hipLaunchKernelGGL(k_twin_dot, dim3((unsigned)spmv_launch_blocks()), dim3(256), 0, q, guarded ? s->d_s : nullptr, s->A.n, x, y, part);
    PIB_HIP(hipGetLastError());
    return 0;
}

bool stencil_matmult_ok(const pib_solver *s)
{
    return s->cfg.matrix_free_poisson == 1 && s->has_grid && !s->hint_pc_only && !s->levels.empty() && s->comm.nranks == 1 &&
           s->A.n >= ((int64_t)1 << 20);
}

int stencil_matmult(pib_solver *s, const double *x, double *y, double *dot_part, bool guarded, hipStream_t q)
{
    const GridLevel &g = s->levels[0];
    const Scalars *S = guarded ? s->d_s : nullptr;
    // the LDS-tiled march reads x once (the streaming kernel leans on the caches for the six neighbours) and leaves the
    // x.y partials as it goes; x[0] = 0 under a pinned pressure, so the identity row patched below adds nothing to x.y
    const int64_t nk = g.k1 - g.k0;
    const int FZ = march_planes(g, nk);
    const dim3 mg((unsigned)(g.n[0] / FX), (unsigned)(g.n[1] / FY), (unsigned)((nk + FZ - 1) / FZ));
    const bool march = s->cfg.march && g.dim == 3 && march_run_ok(s, g, 0, nk) &&
                       ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 31u) == 0 &&
                       (int64_t)mg.x * mg.y * mg.z <= spmv_launch_blocks();
    if (march) {
        hipLaunchKernelGGL((k_level_march<0>), mg, dim3(256), 0, q, S, dev_of(g), 0.0, (const double *)nullptr, x, y,
                           (const double *)nullptr, dot_part, spmv_launch_blocks(), FZ, 0, (int)nk);
        PIB_HIP(hipGetLastError());
        if (s->nullspace == PIB_NULLSPACE_PINNED) hipLaunchKernelGGL(k_twin_row0, dim3(1), dim3(64), 0, q, S, x, y);
        PIB_HIP(hipGetLastError());
        s->counters[0]++;
        return 0;
    }
    PIB_CHK(launch_level<0>(s, g, 0.0, nullptr, x, y, nullptr, guarded, q));
    if (s->nullspace == PIB_NULLSPACE_PINNED) hipLaunchKernelGGL(k_twin_row0, dim3(1), dim3(64), 0, q, S, x, y);
    if (dot_part) hipLaunchKernelGGL(k_twin_dot, dim3((unsigned)spmv_launch_blocks()), dim3(256), 0, q, S, s->A.n, x, y, dot_part);
    PIB_HIP(hipGetLastError());
    s->counters[0]++;
    return 0;
}

int stencil_apply(pib_solver *s, double *x_owned, double *y, hipStream_t q)
{
    if (!s->has_grid) return fail(PIB_ERR_ORDER, "stencil apply without grid structure");
    const GridLevel &g = s->levels[0];
    if (s->comm.nranks > 1 && !g.replicated) PIB_CHK(exchange_planes(s, g, x_owned, 1, q));
    return launch_level<0>(s, g, 0.0, nullptr, x_owned, y, nullptr, false, q, nullptr, 0.0);
}

}  // namespace pib
