// capi.cpp -- the extern "C" surface declared in include/petibm_amd.h.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>

#include <new>
#include <stdexcept>

#include "pib_internal.hpp"

using namespace pib;

static int make_solver(pib_solver **out, const char *name, const Config &cfg, const char *cfg_path, int rank,
                       int nranks, const void *uid, int device)
{
    install_crash_backtrace();  // (PIB_CRASH_BACKTRACE; once per process)
    if (out == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_create: null output handle");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_create: bad rank %d / %d", rank, nranks);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(PIB_ERR_LIB, "pib_create: no HIP device available (%s) -- this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0) device = rank % ndev;
    if (device >= ndev) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_create: device %d of %d", device, ndev);
    PIB_HIP(hipSetDevice(device));
    pib_solver *s = new (std::nothrow) pib_solver();
    if (s == nullptr) return fail(PIB_ERR_MEM, "pib_create: out of memory");
    s->name = name ? name : "";
    s->cfg_path = cfg_path ? cfg_path : "None";
    s->cfg = cfg;
    s->type_string = (cfg.flavor == Flavor::AMGX) ? "NVIDIA AmgX" : "PETSc KSP";
    s->device = device;
    PIB_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    PIB_HIP(hipStreamCreateWithFlags(&s->stream_comm, hipStreamNonBlocking));
    PIB_HIP(hipEventCreate(&s->ev_a));
    PIB_HIP(hipEventCreate(&s->ev_b));
    PIB_HIP(hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming));
    PIB_HIP(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    PIB_HIP(hipMalloc(&s->d_s, sizeof(Scalars)));
    PIB_MEMSET(s->d_s, 0, sizeof(Scalars));
    PIB_HIP(hipHostMalloc(&s->h_s, sizeof(Scalars)));
    std::memset(s->h_s, 0, sizeof(Scalars));
    PIB_HIP(hipMalloc(&s->d_part, sizeof(double) * PIB_NRED * PIB_MAXPART));
    PIB_MEMSET(s->d_part, 0, sizeof(double) * PIB_NRED * PIB_MAXPART);
    int err = comm_init(s, rank, nranks, uid);
    if (err) {
        pib_destroy(s);
        return err;
    }
    *out = s;
    return 0;
}

namespace pib {
int create_sharing_comm(pib_solver **out, const char *name, const char *cfg_text, pib_solver *other)
{
    Config cfg;
    PIB_CHK(parse_config_text(cfg_text ? cfg_text : "", name ? name : "", cfg));
    PIB_CHK(make_solver(out, name, cfg, "<string>", 0, 1, nullptr, other->device));
    (*out)->comm = other->comm;
    (*out)->comm.borrowed = true;
    (*out)->comm.ring = false;
    return 0;
}
}  // namespace pib

extern "C" {

int pib_version(void) { return 100; }

int pib_create(pib_solver **s, const char *name, const char *cfg_path, int rank, int nranks, const void *uid_or_null,
               int device)
try {
    Config cfg;
    PIB_CHK(parse_config_file(cfg_path, name ? name : "", cfg));
    return make_solver(s, name, cfg, cfg_path, rank, nranks, uid_or_null, device);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_create_from_string(pib_solver **s, const char *name, const char *cfg_text, int rank, int nranks,
                           const void *uid_or_null, int device)
try {
    Config cfg;
    PIB_CHK(parse_config_text(cfg_text ? cfg_text : "", name ? name : "", cfg));
    return make_solver(s, name, cfg, "<string>", rank, nranks, uid_or_null, device);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_slab_range(int64_t nplanes, int nranks, int rank, int64_t *begin, int64_t *end)
try {
    if (begin == nullptr || end == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_slab_range: null output");
    if (nranks < 1 || rank < 0 || rank >= nranks || nplanes < 0) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_slab_range: bad arguments");
    slab_range(nplanes, nranks, rank, begin, end);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_config_describe(const char *name, const char *cfg_text, char *buf, int buflen)
try {
    if (buf == nullptr || buflen < 1) return fail(PIB_ERR_ARG_NULL, "pib_config_describe: null buffer");
#ifdef PIB_TEST_HOOKS
    // (drill of the ABI's catch-all, tests/test_host_logic.py: an exception raised inside the library comes back as a code.
    // Compiled only into the test variant of this one translation unit -- tests build it with -DPIB_TEST_HOOKS; the product
    // library carries no hook)
    if (const char *e = std::getenv("PIB_TEST_THROW")) {
        if (e[0] == 'm') throw std::bad_alloc();
        throw std::runtime_error(e);
    }
#endif
    Config c;
    PIB_CHK(parse_config_text(cfg_text ? cfg_text : "", name ? name : "", c));
    const char *method = c.method == Method::CG ? "cg" : (c.method == Method::BICGSTAB ? "bicgstab" : (c.method == Method::CHEBYSHEV ? "chebyshev" : "preonly"));
    const char *pc = c.pc == Precond::NONE ? "none" : (c.pc == Precond::JACOBI ? "jacobi" : (c.pc == Precond::LU ? "lu" : "gmg"));
    std::snprintf(buf, (size_t)buflen,
                  "flavor=%s type=\"%s\" method=%s pc=%s norm=%s max_iters=%d rtol=%.17g atol=%.17g dtol=%.17g "
                  "monitor=%d guess_nonzero=%d error_if_not_converged=%d jacobi_relaxation=%.17g presweeps=%d "
                  "postsweeps=%d smoother=%s smoother_relaxation=%.17g coarsest_sweeps=%d max_levels=%d "
                  "cheby_degree=%d cheby_lmax=%.17g cheby_lmin=%.17g cg_single_reduction=%d sweep_pairs=%d "
                  "effective_presteps=%d effective_poststeps=%d",
                  c.flavor == Flavor::AMGX ? "amgx" : "ksp", c.flavor == Flavor::AMGX ? "NVIDIA AmgX" : "PETSc KSP",
                  method, pc, c.norm == NormType::PRECONDITIONED ? "preconditioned" : "unpreconditioned", c.max_iters,
                  c.rtol, c.atol, c.dtol, c.monitor_residual ? 1 : 0, c.initial_guess_nonzero ? 1 : 0,
                  c.error_if_not_converged ? 1 : 0, c.jacobi_relaxation, c.presweeps, c.postsweeps,
                  c.smoother == Smoother::JACOBI ? "jacobi" : "chebyshev", c.smoother_relaxation, c.coarsest_sweeps,
                  c.max_levels, c.cheby_degree, c.cheby_lmax, c.cheby_lmax / c.cheby_ratio, c.cg_single_reduction, c.sweep_pairs,
                  // what the cycle runs (gmg.hip gmg_apply): a sweep of the file is a fused PAIR of damped-Jacobi steps unless
                  // pib_sweep_pairs=0; a Chebyshev sweep is one polynomial of degree cheby_degree
                  std::max(1, c.presweeps) * (c.smoother == Smoother::CHEBYSHEV ? std::max(1, c.cheby_degree) : (c.sweep_pairs ? 2 : 1)),
                  std::max(0, c.postsweeps) * (c.smoother == Smoother::CHEBYSHEV ? std::max(1, c.cheby_degree) : (c.sweep_pairs ? 2 : 1)));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_destroy(pib_solver *s)
try {
    if (s == nullptr) return 0;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    // the captured iteration goes first: a graph exec is destroyed BEFORE the memory its kernel / copy / fill nodes point at
    // is freed (krylov.hip: drop_iteration_graph has the story)
    drop_iteration_graph(s);
    redist_release(s);
    gmg_release(s);
    dense_release(s);
    vel_stencil_release(s);
    s->A.release();
    (void)free_work(s);
    if (s->x_dev) (void)hipFree(s->x_dev);
    if (s->b_dev) (void)hipFree(s->b_dev);
    if (s->d_s) (void)hipFree(s->d_s);
    if (s->h_s) (void)hipHostFree(s->h_s);
    if (s->d_part) (void)hipFree(s->d_part);
    if (s->d_spmv_part) (void)hipFree(s->d_spmv_part);
    if (s->d_gmg_part) (void)hipFree(s->d_gmg_part);
    if (s->d_hist) (void)hipFree(s->d_hist);
    if (s->h_hist) (void)hipHostFree(s->h_hist);
    comm_release(s);
    if (s->ev_a) (void)hipEventDestroy(s->ev_a);
    if (s->ev_b) (void)hipEventDestroy(s->ev_b);
    if (s->ev_halo) (void)hipEventDestroy(s->ev_halo);
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    for (hipEvent_t e : s->ev_stage)
        if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->stream_comm) (void)hipStreamDestroy(s->stream_comm);
    delete s;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_type(pib_solver *s, char *buf, int buflen)
try {
    if (s == nullptr || buf == nullptr || buflen < 1) return fail(PIB_ERR_ARG_NULL, "pib_get_type: null argument");
    std::strncpy(buf, s->type_string.c_str(), (size_t)buflen - 1);
    buf[buflen - 1] = '\0';
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

// What this solver RUNS, as opposed to what its file says (pib_config_describe): one line of key=value pairs, then one line per
// departure from the file.  LinSolverBase::printInfo (include/petibm/linsolver.h:103) prints it under the reference's banner.
int pib_describe(pib_solver *s, char *buf, int buflen)
try {
    if (s == nullptr || buf == nullptr || buflen < 1) return fail(PIB_ERR_ARG_NULL, "pib_describe: null argument");
    const Config &c = s->cfg;
    const char *method = c.method == Method::CG ? (c.cg_single_reduction ? "cg_single_reduction" : "cg")
                                                : (c.method == Method::BICGSTAB ? "bicgstab" : (c.method == Method::CHEBYSHEV ? "chebyshev" : "preonly"));
    const char *pc = c.pc == Precond::NONE ? "none" : (c.pc == Precond::JACOBI ? "jacobi" : (c.pc == Precond::LU ? "lu" : "gmg"));
    const pib_solver *in = s->redist.active ? s->redist.inner : s;  // (rows in boxes + multigrid: the solve happens on the inner slabs)
    const char *product = "none";
    if (in->has_matrix) {
        if (in->vel.valid && c.matrix_free_velocity) product = "matrix_free_velocity";
        else if (stencil_matmult_ok(in)) product = "matrix_free_stencil";
        else if (in->A.patterned) product = "csr_row_patterns";
        else if (in->A.coded) product = "csr_column_codes";
        else product = in->A.rp64 ? "csr_int32_columns_int64_offsets" : "csr_int32_columns";
    }
    const bool cheb = c.smoother == Smoother::CHEBYSHEV;
    const int per = cheb ? std::max(1, c.cheby_degree) : (c.sweep_pairs ? 2 : 1);
    std::string out;
    char line[1024];
    std::snprintf(line, sizeof line,
                  "type=\"%s\" method=%s pc=%s product=%s partition=%s ranks=%d levels=%d smoother=%s presteps=%d poststeps=%d "
                  "nullspace=%s structure=%s residual_update=%s placement_searches=%d",
                  s->type_string.c_str(), method, pc, product,
                  s->redist.active ? "boxes_to_slabs" : (s->comm.nranks > 1 ? (s->A.general ? "general" : "slabs") : "single"), s->comm.nranks,
                  (int)in->levels.size(), cheb ? "chebyshev" : "jacobi", c.pc == Precond::GMG ? std::max(1, c.presweeps) * per : 0,
                  c.pc == Precond::GMG ? std::max(0, c.postsweeps) * per : 0,
                  in->nullspace == PIB_NULLSPACE_PINNED ? "pinned_row0" : (in->nullspace == PIB_NULLSPACE_CONSTANT ? "constant" : "none"),
                  in->has_grid ? (in->structure_detected ? "recovered" : "given") : "none",
                  // PCG + multigrid: does r -= alpha w ride in the V-cycle's first march (gmg_fused_update_ok: large systems whose level 0
                  // the fused marches serve), or run as a pass of its own (48 B/row/iteration more)
                  (c.method == Method::CG && !c.cg_single_reduction && c.pc == Precond::GMG && in->has_matrix)
                      ? ((c.fuse_residual_update != 0 && c.fuse_residual_update != 2 && gmg_fused_update_ok(in)) ? "in_vcycle" : "separate_pass")
                      : "n/a",
                  in->placements);
    out = line;
    if (c.pc == Precond::GMG && !cheb && c.sweep_pairs && (c.presweeps > 0 || c.postsweeps > 0))
        out += "\ndeparture: smoother: a sweep of the file runs as a fused pair of damped-Jacobi steps (pib_sweep_pairs=0: one step)";
    for (const std::string &d : s->departures) out += "\ndeparture: " + d;
    if (s->redist.active)
        for (const std::string &d : s->redist.inner->departures) out += "\ndeparture: " + d;
    if ((int)out.size() >= buflen) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_describe: %d bytes needed", (int)out.size() + 1);
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

extern "C++" {
namespace pib {
int after_set_matrix(pib_solver *s)
{
    PIB_CHK(comm_setup_halo(s));
    int missing = 0;
    PIB_CHK(extract_dinv(s, &missing));
    if (missing > 0 && s->cfg.pc == Precond::JACOBI)
        return fail(PIB_ERR_ARG_WRONG, "solver %s: %d rows have no (or a zero) diagonal entry: Jacobi preconditioning impossible",
                    s->name.c_str(), missing);
    // KSPReset semantics (linsolverksp.cpp:78): everything derived from the old matrix goes
    s->gersh_lo = 0.0;
    s->gersh_hi = -1.0;
    drop_iteration_graph(s);
    if (s->cfg.pc == Precond::LU) PIB_CHK(dense_setup(s));  // PCSetUp of a direct solve = the factorisation
    s->has_matrix = true;
    return 0;
}
}  // namespace pib
}  // extern "C++"

// LinSolverBase::setMatrix / AmgXSolver::setA (src/linsolver/linsolveramgx.cpp:84): this rank's rows, global columns, in
// WHATEVER partition the application's DMDA produced (partition.cpp).  Collective on several ranks.
static int set_csr_any(pib_solver *s, int64_t n_local, int64_t row0_global, int64_t n_global, const int64_t *rp64,
                       const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val)
{
    PIB_HIP(hipSetDevice(s->device));
    s->has_matrix = false;
    s->has_grid = false;
    gmg_release(s);
    redist_release(s);
    s->structure_detected = false;
    s->vel_detected = false;
    if (n_local < 0 || row0_global < 0 || n_global < n_local) return fail(PIB_ERR_ARG_OUTOFRANGE, "set_csr: bad sizes");
    if ((rp64 == nullptr && rp32 == nullptr) || (cl64 == nullptr && cl32 == nullptr && n_local > 0) || (val == nullptr && n_local > 0))
        return fail(PIB_ERR_ARG_NULL, "set_csr: null array");
    std::vector<int64_t> ranges;
    bool general = false;
    SetupTrace tr("set_csr", s->comm.rank);
    PIB_CHK(classify_partition(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, ranges, &general));
    tr.mark("classify_partition");
    if (general) {
        PIB_CHK(upload_csr_general(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val, ranges));
        tr.mark("upload_csr_general");
        PIB_CHK(after_set_matrix(s));
        tr.mark("after_set_matrix");
        if (s->cfg.pc == Precond::GMG && s->cfg.detect_structure)
            PIB_CHK(redist_setup(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val, ranges));
        tr.mark("redist_setup");
        if (s->cfg.pc == Precond::GMG && !s->redist.active)
            s->gmg_error = "the rows came in a partition that is neither z-slabs in natural ordering nor DMDA boxes of PetIBM's Poisson "
                           "operator: no mesh structure for the multigrid";
        // the velocity system in slabs of the packed ordering (an unchanged PetIBM on two ranks, a (1,1,P) process grid): the
        // matrix-free products as on one rank (structure.cpp); boxes keep their CSR products
        if (s->cfg.pc != Precond::GMG && s->cfg.detect_structure && s->cfg.matrix_free_velocity) {
            PIB_CHK(detect_velocity_structure(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val));
            // ... and the velocity system in BOXES ([u box | v box | w box] per rank, PETSC_DECIDE from 4 ranks up): moved to
            // packed slabs inside the backend, like the pressure rows for the multigrid (partition.cpp)
            if (!s->vel.valid)
                PIB_CHK(redist_velocity_setup(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val, ranges));
        }
        return 0;
    }
    PIB_CHK(upload_csr(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val));
    tr.mark("upload_csr");
    PIB_CHK(after_set_matrix(s));
    tr.mark("after_set_matrix");
    if (s->cfg.pc == Precond::GMG && s->cfg.detect_structure)
        PIB_CHK(detect_grid_structure(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val));
    tr.mark("detect_grid_structure");
    if (s->cfg.pc != Precond::GMG && s->cfg.detect_structure && s->cfg.matrix_free_velocity)
        PIB_CHK(detect_velocity_structure(s, n_local, row0_global, n_global, rp64, cl64, rp32, cl32, val));
    return 0;
}

int pib_set_csr(pib_solver *s, int64_t n_local, int64_t row0_global, int64_t n_global, const int64_t *rowptr,
                const int64_t *col_global, const double *val)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_set_csr: null solver");
    PIB_CHK(comm_usable(s));
    return set_csr_any(s, n_local, row0_global, n_global, rowptr, col_global, nullptr, nullptr, val);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_set_csr_i32(pib_solver *s, int32_t n_local, int32_t row0_global, int32_t n_global, const int32_t *rowptr,
                    const int32_t *col_global, const double *val)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_set_csr_i32: null solver");
    PIB_CHK(comm_usable(s));
    return set_csr_any(s, n_local, row0_global, n_global, nullptr, nullptr, rowptr, col_global, val);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_set_grid_hint(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                      const double *gx, const double *gy, const double *gz, int nullspace)
try {
    if (s == nullptr || n == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_set_grid_hint: null argument");
    if (!s->has_matrix) return fail(PIB_ERR_ORDER, "pib_set_grid_hint: set the matrix first");
    if (s->A.general)
        return fail(PIB_ERR_SUP, "pib_set_grid_hint: the hint describes the mesh in natural ordering and needs rows in z-slabs; these rows came "
                                 "in another partition (DMDA boxes?) -- leave the hint out: pib_set_csr recovers the structure itself");
    PIB_HIP(hipSetDevice(s->device));
    const double one = 1.0;
    const double *w[3] = {wx, wy, (dim == 3) ? wz : &one};
    const double *g[3] = {gx, gy, (dim == 3) ? gz : nullptr};
    s->structure_detected = false;
    // the hint's arrays are sized by what the CALLER declared periodic, whatever a structure recovered from the matrix
    // had found before
    for (int d = 0; d < 3; ++d) s->periodic[d] = s->periodic_user[d];
    return grid_register(s, dim, n, w, g, nullspace, -1.0);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_grid_structure(pib_solver *s, int *has, int *dim, int64_t n[3], int *nullspace, int *detected)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_grid_structure: null solver");
    if (s->redist.active) s = s->redist.inner;  // rows handed over in boxes: the structure lives in the slab solver
    const bool have = s->has_grid && !s->levels.empty();
    if (has) *has = have ? 1 : 0;
    if (detected) *detected = (have && s->structure_detected) ? 1 : 0;
    if (nullspace) *nullspace = have ? s->nullspace : PIB_NULLSPACE_NONE;
    if (have) {
        const GridLevel &g = s->levels[0];
        if (dim) *dim = g.dim;
        if (n) {  // internal layout of a 2-D grid is (nx, 1, ny)
            n[0] = g.n[0];
            n[1] = (g.dim == 3) ? g.n[1] : g.n[2];
            n[2] = (g.dim == 3) ? g.n[2] : 1;
        }
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_multigrid_levels(pib_solver *s, int *nlevels, int64_t *n3, int max_levels)
try {
    if (s == nullptr || nlevels == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_multigrid_levels: null argument");
    if (s->redist.active) s = s->redist.inner;
    *nlevels = (int)s->levels.size();
    for (int l = 0; n3 != nullptr && l < *nlevels && l < max_levels; ++l) {
        const GridLevel &g = s->levels[(size_t)l];  // internal layout of a 2-D grid is (nx, 1, ny)
        n3[3 * l] = g.n[0];
        n3[3 * l + 1] = (g.dim == 3) ? g.n[1] : g.n[2];
        n3[3 * l + 2] = (g.dim == 3) ? g.n[2] : 1;
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_velocity_structure(pib_solver *s, int *has, int *dim, int64_t n[3], int periodic[3], int *detected)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_velocity_structure: null solver");
    if (s->redist.active && s->redist.nf > 1) s = s->redist.inner;  // rows handed over in boxes: the structure lives in the slab solver
    const VelStencil &V = s->vel;
    if (has) *has = V.valid ? 1 : 0;
    if (detected) *detected = (V.valid && s->vel_detected) ? 1 : 0;
    if (dim) *dim = V.valid ? V.dim : 0;
    for (int d = 0; d < 3; ++d) {
        const bool p = V.valid && ((V.per >> d) & 1);
        if (periodic) periodic[d] = p ? 1 : 0;
        // pressure cells along d: component d has one point fewer than that unless d is periodic
        if (n) n[d] = (V.valid && d < V.dim) ? V.n[d][d] + (p ? 0 : 1) : 1;
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_set_periodic(pib_solver *s, const int periodic[3])
try {
    if (s == nullptr || periodic == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_set_periodic: null argument");
    for (int d = 0; d < 3; ++d) s->periodic[d] = s->periodic_user[d] = periodic[d] ? 1 : 0;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_assemble_poisson(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                         const double *wz, double dt, int nullspace)
try {
    if (s == nullptr || n == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_assemble_poisson: null argument");
    PIB_HIP(hipSetDevice(s->device));
    s->has_matrix = false;
    s->has_grid = false;
    gmg_release(s);
    const double *w[3] = {wx, wy, wz};
    PIB_CHK(assemble_poisson(s, dim, n, w, dt, nullspace));
    PIB_CHK(after_set_matrix(s));  // halo plan first: registering the grid verifies it with a halo exchange
    const double *cw[3] = {s->asm_w[0].data(), s->asm_w[1].data(), s->asm_w[2].data()};
    const double *cg[3] = {s->asm_g[0].data(), s->asm_g[1].data(), s->asm_g[2].data()};
    return grid_register(s, dim, n, cw, cg, nullspace, s->asm_dt);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_assemble_poisson_bn(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                            const double *wz, const double lo[3], const double hi[3], const double a0[18], double dt,
                            double coeff_nu, int bn_order, int nullspace)
try {
    if (s == nullptr || n == nullptr || lo == nullptr || hi == nullptr || a0 == nullptr)
        return fail(PIB_ERR_ARG_NULL, "pib_assemble_poisson_bn: null argument");
    bool fold = false;  // a ghost fold on a normal velocity component (NEUMANN): D changes, the symmetric assembly does not apply
    for (int f = 0; f < dim; ++f) fold = fold || a0[6 * f + 2 * f] != 0.0 || a0[6 * f + 2 * f + 1] != 0.0;
    if (bn_order == 1 && !fold) return pib_assemble_poisson(s, dim, n, wx, wy, wz, dt, nullspace);
    PIB_HIP(hipSetDevice(s->device));
    s->has_matrix = false;
    s->has_grid = false;
    gmg_release(s);
    const double *w[3] = {wx, wy, wz};
    return assemble_poisson_bn(s, dim, n, w, lo, hi, a0, dt, coeff_nu, bn_order, nullspace, nullptr, nullptr, nullptr, nullptr);
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_assemble_velocity(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                          const double *wz, const double lo[3], const double hi[3], const double a0[18], double dt,
                          double coeff_nu)
try {
    if (s == nullptr || n == nullptr || lo == nullptr || hi == nullptr || a0 == nullptr)
        return fail(PIB_ERR_ARG_NULL, "pib_assemble_velocity: null argument");
    PIB_HIP(hipSetDevice(s->device));
    s->has_matrix = false;
    s->has_grid = false;
    gmg_release(s);
    const double *w[3] = {wx, wy, wz};
    PIB_CHK(assemble_velocity(s, dim, n, w, lo, hi, a0, dt, coeff_nu));
    return after_set_matrix(s);
} catch (...) {
    return pib::fail_exception(__func__);
}

static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

static int ensure_stage(pib_solver *s)
{
    if (s->stage_n >= s->A.n && s->x_dev) return 0;
    if (s->x_dev) (void)hipFree(s->x_dev);
    if (s->b_dev) (void)hipFree(s->b_dev);
    s->x_dev = s->b_dev = nullptr;
    const size_t bytes = sizeof(double) * (size_t)(s->A.n > 0 ? s->A.n : 1);
    PIB_HIP(hipMalloc(&s->x_dev, bytes));
    PIB_HIP(hipMalloc(&s->b_dev, bytes));
    s->stage_n = s->A.n;
    return 0;
}

int pib_solve(pib_solver *s, double *x, const double *b)
try {
    if (s == nullptr || x == nullptr || b == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_solve: null argument");
    if (!s->has_matrix) return fail(PIB_ERR_ORDER, "solver %s: pib_solve called before a matrix was set", s->name.c_str());
    PIB_CHK(comm_usable(s));  // (a communicator aborted by ANY solver of the sharing group: fail at once, never a null or dangling handle)
    PIB_HIP(hipSetDevice(s->device));
    const bool xd = is_device_ptr(x), bd = is_device_ptr(b);
    double *xdev = x;
    const double *bdev = b;
    const size_t bytes = sizeof(double) * (size_t)s->A.n;
    if (!xd || !bd) PIB_CHK(ensure_stage(s));
    // (host vectors: what the copies over PCIe cost is kept for pib_get_staging_ms -- the way in between two events on the stream,
    // the way out as wall time around its enqueue + the wait the caller needs anyway)
    s->stage_ms[0] = s->stage_ms[1] = 0.0;
    if ((!xd && s->cfg.initial_guess_nonzero) || !bd) {
        if (s->ev_stage[0] == nullptr) {
            PIB_HIP(hipEventCreate(&s->ev_stage[0]));
            PIB_HIP(hipEventCreate(&s->ev_stage[1]));
        }
        PIB_HIP(hipEventRecord(s->ev_stage[0], s->stream));
    }
    if (!xd) {
        xdev = s->x_dev;
        if (s->cfg.initial_guess_nonzero) PIB_HIP(hipMemcpyAsync(xdev, x, bytes, hipMemcpyHostToDevice, s->stream));
    }
    if (!bd) {
        PIB_HIP(hipMemcpyAsync(s->b_dev, b, bytes, hipMemcpyHostToDevice, s->stream));
        bdev = s->b_dev;
    }
    // (no host round trip for the figure: two events bracket the copies on the stream, read once the solve has synchronised)
    const bool staged_in = (!xd && s->cfg.initial_guess_nonzero) || !bd;
    if (staged_in) PIB_HIP(hipEventRecord(s->ev_stage[1], s->stream));
    int err;
    if (s->redist.active) {
        // rows in DMDA boxes + multigrid: b (and the guess) go to the z-slabs of the inner solver, x comes back
        Redist &R = s->redist;
        pib_solver *in = R.inner;
        for (int k = 0; k < 8; ++k) s->counters[k] = 0;  // the moves of b / x count as this solve's exchanges and bytes
        PIB_CHK(redist_forward(s, bdev, R.b_nat, s->stream));
        if (s->cfg.initial_guess_nonzero) PIB_CHK(redist_forward(s, xdev, R.x_nat, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));
        in->cfg.error_if_not_converged = false;  // reported below, by this solver
        err = pib_solve(in, R.x_nat, R.b_nat);
        if (!err) err = hipStreamSynchronize(in->stream) == hipSuccess ? 0 : fail(PIB_ERR_LIB, "solver %s: the slab solver's stream failed", s->name.c_str());
        if (!err) err = redist_backward(s, R.x_nat, xdev, s->stream);
        // like solve_cg / solve_bicgstab (which return behind fetch_results' synchronisation), x is complete on return: the
        // scatter and its exchange were only enqueued -- none of the transports synchronises by itself
        if (hipStreamSynchronize(s->stream) != hipSuccess && !err)
            err = fail(PIB_ERR_LIB, "solver %s: moving x back to the boxes failed", s->name.c_str());
        s->iters = in->iters;
        s->reason = in->reason;
        s->residual = in->residual;
        s->history = in->history;
        for (int k = 0; k < 8; ++k) s->counters[k] += in->counters[k];
    } else if (s->cfg.method == Method::CG)
        err = s->cfg.cg_single_reduction ? solve_cg_sr(s, xdev, bdev) : solve_cg(s, xdev, bdev);
    else if (s->cfg.method == Method::BICGSTAB)
        err = solve_bicgstab(s, xdev, bdev);
    else if (s->cfg.method == Method::CHEBYSHEV)
        err = solve_chebyshev(s, xdev, bdev);
    else if (s->cfg.method == Method::PREONLY && s->cfg.pc == Precond::LU)
        err = solve_direct(s, xdev, bdev);
    else
        err = fail(PIB_ERR_SUP, "solver %s: unsupported Krylov method", s->name.c_str());
    if (err) return err;
    if (staged_in) {  // (every method returns behind a synchronisation of the stream: both events have completed)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->ev_stage[0], s->ev_stage[1]) == hipSuccess) s->stage_ms[0] = (double)ms;
        else (void)hipGetLastError();
    }
    if (!xd) {
        const auto t_out = std::chrono::steady_clock::now();
        PIB_HIP(hipMemcpyAsync(x, xdev, bytes, hipMemcpyDeviceToHost, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));
        s->stage_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_out).count();
    }
    if (s->reason < 0 && s->cfg.error_if_not_converged)
        return fail(PIB_ERR_CONV_FAILED, "PetIBM exited due to solver %s diverged with reason %d (iterations %d, residual %g).",
                    s->name.c_str(), s->reason, s->iters, s->residual);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_iters(pib_solver *s, int *iters)
try {
    if (s == nullptr || iters == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_iters: null argument");
    *iters = s->iters;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_residual(pib_solver *s, double *res)
try {
    if (s == nullptr || res == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_residual: null argument");
    *res = s->residual;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_residual_at(pib_solver *s, int iter, double *res)
try {
    if (s == nullptr || res == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_residual_at: null argument");
    if (iter < 0 || (size_t)iter >= s->history.size())
        return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_get_residual_at: iteration %d outside the stored history [0,%zu)", iter,
                    s->history.size());
    *res = s->history[(size_t)iter];
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_reason(pib_solver *s, int *reason)
try {
    if (s == nullptr || reason == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_get_reason: null argument");
    *reason = s->reason;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_mat_mult(pib_solver *s, const double *x, double *y)
try {
    if (s == nullptr || x == nullptr || y == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_mat_mult: null argument");
    if (!s->has_matrix) return fail(PIB_ERR_ORDER, "pib_mat_mult called before a matrix was set");
    PIB_CHK(comm_usable(s));
    PIB_HIP(hipSetDevice(s->device));
    PIB_CHK(ensure_work(s, 4));
    const size_t bytes = sizeof(double) * (size_t)s->A.n;
    double *P = s->vec(2), *W = s->vec(3);
    PIB_HIP(hipMemcpyAsync(P, x, bytes, hipMemcpyDefault, s->stream));
    if (s->comm.nranks > 1) PIB_CHK(halo_exchange(s, P, s->stream));
    PIB_CHK(spmv_rows(s, P, W, 0, s->A.n, nullptr, false, s->stream));
    PIB_HIP(hipMemcpyAsync(y, W, bytes, hipMemcpyDefault, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_device_alloc(pib_solver *s, int64_t nbytes, void **ptr)
try {
    if (s == nullptr || ptr == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_device_alloc: null argument");
    PIB_HIP(hipSetDevice(s->device));
    PIB_HIP(hipMalloc(ptr, (size_t)(nbytes > 0 ? nbytes : 1)));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
int pib_device_mem_info(pib_solver *s, int64_t *free_bytes, int64_t *total_bytes)
try {
    if (s == nullptr || free_bytes == nullptr || total_bytes == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_device_mem_info: null argument");
    PIB_HIP(hipSetDevice(s->device));
    size_t fr = 0, tot = 0;
    PIB_HIP(hipMemGetInfo(&fr, &tot));
    *free_bytes = (int64_t)fr;
    *total_bytes = (int64_t)tot;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
int pib_device_free(pib_solver *s, void *ptr)
try {
    if (s != nullptr) PIB_HIP(hipSetDevice(s->device));
    if (ptr) PIB_HIP(hipFree(ptr));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
int pib_memcpy_h2d(pib_solver *s, void *dst, const void *src, int64_t nbytes)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "null solver");
    PIB_HIP(hipSetDevice(s->device));
    PIB_HIP(hipMemcpy(dst, src, (size_t)nbytes, hipMemcpyHostToDevice));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
int pib_memcpy_d2h(pib_solver *s, void *dst, const void *src, int64_t nbytes)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "null solver");
    PIB_HIP(hipSetDevice(s->device));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipMemcpy(dst, src, (size_t)nbytes, hipMemcpyDeviceToHost));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
int pib_synchronize(pib_solver *s)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "null solver");
    PIB_HIP(hipSetDevice(s->device));
    PIB_HIP(hipStreamSynchronize(s->stream));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_csr(pib_solver *s, int64_t *n_local, int64_t *nnz, int64_t *rowptr, int64_t *col_global, double *val)
try {
    if (s == nullptr) return fail(PIB_ERR_ARG_NULL, "null solver");
    if (!s->has_matrix) return fail(PIB_ERR_ORDER, "pib_get_csr: no matrix");
    PIB_HIP(hipSetDevice(s->device));
    const DeviceCsr &A = s->A;
    if (n_local) *n_local = A.n;
    if (nnz) *nnz = A.nnz;
    if (rowptr) {
        if (A.rp64) {
            PIB_HIP(hipMemcpy(rowptr, A.rowptr, sizeof(int64_t) * ((size_t)A.n + 1), hipMemcpyDeviceToHost));
        } else {
            std::vector<int32_t> t((size_t)A.n + 1);
            PIB_HIP(hipMemcpy(t.data(), A.rowptr, sizeof(int32_t) * ((size_t)A.n + 1), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < t.size(); ++i) rowptr[i] = t[i];
        }
    }
    if (col_global) {
        std::vector<int32_t> t((size_t)A.nnz);
        PIB_HIP(hipMemcpy(t.data(), A.col, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToHost));
        const int64_t shift = A.row0 - A.ghost_lo;
        if (A.general) {  // [low ghosts | owned | high ghosts]: the ghosts' global columns are kept on the host
            for (size_t i = 0; i < t.size(); ++i) {
                const int64_t c = t[i];
                col_global[i] = c < A.ghost_lo ? A.ghost_cols[(size_t)c]
                                               : (c < A.ghost_lo + A.n ? A.row0 + (c - A.ghost_lo) : A.ghost_cols[(size_t)(c - A.n)]);
            }
        } else
        for (size_t i = 0; i < t.size(); ++i) col_global[i] = (int64_t)t[i] + shift;
    }
    if (val) PIB_HIP(hipMemcpy(val, A.val, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToHost));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_graph_replays(pib_solver *s, int64_t *replays)
try {
    if (s == nullptr || replays == nullptr) return fail(PIB_ERR_ARG_NULL, "null argument");
    *replays = s->graph_replays + (s->redist.active ? s->redist.inner->graph_replays : 0);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_staging_ms(pib_solver *s, double *h2d_ms, double *d2h_ms)
try {
    if (s == nullptr || h2d_ms == nullptr || d2h_ms == nullptr) return fail(PIB_ERR_ARG_NULL, "null argument");
    *h2d_ms = s->stage_ms[0];
    *d2h_ms = s->stage_ms[1];
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_product_format(pib_solver *s, int *index_bytes_per_entry)
try {
    if (s == nullptr || index_bytes_per_entry == nullptr) return fail(PIB_ERR_ARG_NULL, "null argument");
    if (!s->has_matrix) return fail(PIB_ERR_ORDER, "pib_get_product_format: no matrix");
    *index_bytes_per_entry = s->A.patterned ? 0 : (s->A.coded ? 1 : 4);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_placement(pib_solver *s, int *searches, int *candidates, double *ms_had, double *ms_kept, int64_t *held_bytes, double *search_ms)
try {
    if (s == nullptr || searches == nullptr || candidates == nullptr || ms_had == nullptr || ms_kept == nullptr) return fail(PIB_ERR_ARG_NULL, "null argument");
    *searches = s->placements;
    *candidates = s->place_tried;
    *ms_had = s->place_ms[0];
    *ms_kept = s->place_ms[1];
    if (held_bytes) *held_bytes = s->place_held_bytes;
    if (search_ms) *search_ms = s->place_search_ms;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

int pib_get_counters(pib_solver *s, int64_t counters[8])
try {
    if (s == nullptr || counters == nullptr) return fail(PIB_ERR_ARG_NULL, "null argument");
    for (int k = 0; k < 8; ++k) counters[k] = s->counters[k];
    counters[5] = 1;  // ranks of the communicator as the transport reports them
    if (s->comm.comm) {
        int cnt = 0;
        PIB_NCCL(ncclCommCount(s->comm.comm, &cnt));
        counters[5] = cnt;
    } else if (s->comm.nranks > 1)
        counters[5] = s->comm.nranks;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

}  // extern "C"
