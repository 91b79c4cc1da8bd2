// krylov.hip -- K3/K4/K5/K10: device-resident CG and BiCGStab for gfx950.
//
// Replaces what KSPSolve (src/linsolver/linsolverksp.cpp:92) and
// AmgXSolver::solve (src/linsolver/linsolveramgx.cpp:96) do for the reference:
// PETSc's KSPCG / KSPBCGS recurrences (the oracle, oracle/csrc/oracle.c,
// restates the same ones on the CPU) with none / Jacobi / multigrid
// preconditioning.
//
// MI355X design:
//   * every scalar of the recurrence (beta, dpi, a, b, norms, iteration
//     count, convergence reason) lives in one `Scalars` block in HBM and is
//     produced / consumed by kernels: no host round trip inside an iteration.
//     The host enqueues batches of iterations and polls one 200-byte struct;
//     each kernel starts with `if (S->done) return`, so over-enqueued
//     iterations cost a launch and nothing else.
//   * vector updates are fused per PETSc's dependency structure:
//       [p = z + b p]  ->  [w = A p ; p.w fused into the SpMV epilogue]
//       ->  [x += a p ; r -= a w ; z = D^-1 r ; z.r, z.z, r.r, sum z, sum r]
//     i.e. 3 passes per Jacobi-PCG iteration besides the SpMV (vs 7 BLAS-1
//     calls in KSPCG), 8 B/lane..16 B/lane coalesced, wave64 shuffle
//     reductions -> per-block partials -> single-block finalize.  The
//     partial/finalize order is fixed: results are run-to-run deterministic.
//   * multi-GPU: finalize -> ncclAllReduce on the same stream, in place on the
//     device scalars (one call for all sums of a step).
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <cstring>

#include <chrono>
#include <thread>

#include "pib_internal.hpp"

#include "krylov_ops.hpp"

namespace pib {

// ------------------------------------------------------------------ helpers
// The captured iteration body (hipGraphExec) is destroyed BEFORE any memory its kernel, copy and fill nodes point at is freed,
// and before a stream it was launched on goes: every release path (pib_destroy, a new matrix, new work vectors, the multigrid's
// and the redistribution's buffers, the immersed-boundary state behind the Schur hook) calls this first.  (Until round 4's end
// pib_destroy freed the level buffers first and the graph after them.  The order was turned round while hunting an intermittent
// crash that surfaced inside pib_destroy; that crash turned out to be the runtime's own -- docs/history/round4.md -- and the
// order stays because it is the right one.)
void drop_iteration_graph(pib_solver *s)
{
    if (s == nullptr || s->graph == nullptr) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);  // never under a launch still in flight
    (void)hipGraphExecDestroy(s->graph);
    s->graph = nullptr;
}

int free_work(pib_solver *s)
{
    if (s->work_base) PIB_HIP(hipFree(s->work_base));
    s->work_base = nullptr;
    s->work = nullptr;
    for (double *&v : s->work_split) {
        if (v) PIB_HIP(hipFree(v));
        v = nullptr;
    }
    s->placed_against = nullptr;
    s->n_work = 0;
    return 0;
}

int ensure_work(pib_solver *s, int nvec)
{
    const DeviceCsr &A = s->A;
    // [halo below | owned | halo above]: the CSR's ghost columns, or the deeper halo of the multigrid's slabs
    int64_t lo = std::max(A.ghost_lo, s->work_pad), hi = std::max(A.ghost_hi, s->work_pad);
    lo = (lo + 3) & ~int64_t(3);  // owned part 32-byte aligned
    int64_t stride = lo + A.n + hi + 4;
    stride = (stride + 3) & ~int64_t(3);
    // Large systems on one rank (pib_place_min_rows rows and more): every work vector an allocation of its own.  One pool puts all
    // the vectors of a Krylov method into ONE placement class (place_walk below), where the flat kernels that read and write
    // several of them run slowest (BiCGStab on the 400^3 velocity system: 88.2 -> 84.2 ms per solve), and only separately
    // allocated vectors can be exchanged for better placed ones.
    const bool split = s->cfg.place_min_rows >= 0 && A.n >= s->cfg.place_min_rows && A.n == A.n_global && nvec <= pib_solver::MAX_WORK;
    const bool have = split ? s->work_split[0] != nullptr : s->work != nullptr;
    if (have && s->n_work >= nvec && s->work_stride == stride && s->work_lo == lo) return 0;
    drop_iteration_graph(s);  // a captured iteration body holds the old vectors' addresses (and goes BEFORE they do)
    PIB_CHK(free_work(s));
    if (split) {
        const size_t bytes = (size_t)(stride + 4) * sizeof(double);
        for (int i = 0; i < nvec; ++i) {
            PIB_HIP(hipMalloc(&s->work_split[i], bytes));
            PIB_HIP(hipMemsetAsync(s->work_split[i], 0, bytes, s->stream));
        }
    } else {
        PIB_HIP(hipMalloc(&s->work_base, (size_t)(stride * nvec + 4) * sizeof(double)));
        PIB_HIP(hipMemsetAsync(s->work_base, 0, (size_t)(stride * nvec + 4) * sizeof(double), s->stream));
        s->work = s->work_base;  // hipMalloc is 256-byte aligned; lo and stride are multiples of 4 doubles
    }
    s->work_lo = lo;
    s->work_stride = stride;
    s->n_work = nvec;
    return 0;
}

// The host's wait for a batch of iterations.  On RCCL with several ranks it is BOUNDED (PIB_RCCL_TIMEOUT_S, default 180 s; 0: wait
// for ever): a collective that never completes -- a rank that died, a link that never came up on first contact -- becomes
// ncclCommAbort + an error the caller can act on (bench.py falls back to the peer transport) instead of a hung process.
static int poll(pib_solver *s)
{
    PIB_HIP(hipMemcpyAsync(s->h_s, s->d_s, sizeof(Scalars), hipMemcpyDeviceToHost, s->stream));
    static const double limit = [] {
        const char *e = std::getenv("PIB_RCCL_TIMEOUT_S");
        return e ? std::atof(e) : 180.0;
    }();
    if (s->comm.comm != nullptr && s->comm.nranks > 1 && limit > 0.0) {
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        for (;;) {
            const hipError_t e = hipStreamQuery(s->stream);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) {
                (void)hipGetLastError();
                return fail(PIB_ERR_LIB, "solver %s: the stream failed while waiting for a batch of iterations (%s)", s->name.c_str(), hipGetErrorString(e));
            }
            (void)hipGetLastError();
            if (++spins > 2000) {  // (the first ~2000 queries spin: a batch takes a millisecond or so; then 50 us naps)
                std::this_thread::sleep_for(std::chrono::microseconds(50));
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
                    comm_abort(s);  // (once for every solver that shares the communicator: they all see `aborted` from here on)
                    return fail(PIB_ERR_LIB, "solver %s: an RCCL collective did not complete within %.0f s (PIB_RCCL_TIMEOUT_S): communicator aborted",
                                s->name.c_str(), limit);
                }
            }
        }
    } else
        PIB_HIP(hipStreamSynchronize(s->stream));
    s->counters[4]++;
    return 0;
}

static int auto_batch(const pib_solver *s)
{
    if (s->cfg.check_every > 0) return s->cfg.check_every;
    // one iteration moves ~ (12 nnz + 100 n) bytes; aim at >= 0.5 ms per poll
    const double bytes = 12.0 * (double)s->A.nnz + 100.0 * (double)s->A.n;
    const double t_iter = std::max(bytes / 4.0e12, 15e-6);
    int b = (int)std::ceil(0.5e-3 / t_iter);
    return std::max(1, std::min(b, 64));
}
// Iterations enqueued before the first / between later host polls.  Over-enqueued iterations are no-ops but still cost
// a launch slot each (~1.6 us; rocprof on the 186^2 cylinder case: 33 iterations enqueued for 5 needed = 2.7 ms of a
// 6.4 ms time step), so inside a time loop the first batch is the iteration count of the previous solve and the later
// ones are short.
static int first_batch(const pib_solver *s)
{
    if (s->cfg.check_every > 0) return s->cfg.check_every;
    if (s->hint_iters > 0) return std::min(s->hint_iters, 256);
    return std::min(auto_batch(s), 8);
}
static int next_batch(const pib_solver *s)
{
    if (s->cfg.check_every > 0) return s->cfg.check_every;
    return std::max(1, std::min(auto_batch(s), 4));
}

int halo_exchange(pib_solver *s, double *x_owned, hipStream_t stq);
int gmg_apply(pib_solver *s, const double *r, double *z, hipStream_t stq);

// Enqueue `todo` repetitions of one Krylov iteration (`body` launches its 15..70 kernels on q).  Small systems are
// launch-bound (rocprof, 186^2 cylinder case: ~2 us kernels, the host cannot feed them faster than ~2.4 us apiece), so
// on a single GPU the body is captured ONCE per (matrix, vectors) into a hipGraph and replayed: one host call per
// iteration.  The first iteration of a solve always runs directly (lazy allocations happen there, never in a capture).
static uint64_t graph_key(int method, const void *x, const void *b)
{
    return (uint64_t)reinterpret_cast<uintptr_t>(x) * 0x9E3779B97F4A7C15ull ^ (uint64_t)reinterpret_cast<uintptr_t>(b) ^
           ((uint64_t)method << 60);
}
template <class Body>
static int run_iterations(pib_solver *s, int todo, int first_index, uint64_t key, hipStream_t q, Body body)
{
    // (several ranks: when the transport's collectives can be captured -- the device-ordered peer transport, halo.hip)
    const bool use = comm_capturable(s) && s->A.n <= s->cfg.graph_max_rows;  // (pib_graph_max_rows=0: never)
    for (int it = 0; it < todo; ++it) {
        if (!use || first_index + it == 0) {
            PIB_CHK(body());
            continue;
        }
        if (s->graph == nullptr || s->graph_key != key) {
            drop_iteration_graph(s);  // (another key: synchronises first -- the old body may still be running)
            int64_t before[8];
            for (int k = 0; k < 8; ++k) before[k] = s->counters[k];
            if (s->comm.nranks > 1) {
                PIB_HIP(hipStreamSynchronize(q));  // nothing of the collectives before is in flight on another stream
                comm_capture_boundary(s, true);
            }
            {
                const hipError_t eb = hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal);
                if (eb != hipSuccess) {
                    // the communicator group is shared with the other solvers of the step: never leave it marked as capturing
                    if (s->comm.nranks > 1) comm_capture_boundary(s, false);
                    (void)hipGetLastError();
                    return fail(PIB_ERR_LIB, "solver %s: hipStreamBeginCapture failed (%s)", s->name.c_str(), hipGetErrorString(eb));
                }
            }
            const int err = body();
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(q, &g);
            if (s->comm.nranks > 1) comm_capture_boundary(s, false);
            for (int k = 0; k < 8; ++k) {
                s->graph_counts[k] = s->counters[k] - before[k];
                s->counters[k] = before[k];
            }
            if (err || e != hipSuccess || g == nullptr) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                if (err) return err;
                return fail(PIB_ERR_LIB, "solver %s: hipGraph capture of the iteration body failed (%s)", s->name.c_str(),
                            hipGetErrorString(e));
            }
            const hipError_t ei = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) {
                s->graph = nullptr;
                return fail(PIB_ERR_LIB, "solver %s: hipGraphInstantiate failed (%s)", s->name.c_str(), hipGetErrorString(ei));
            }
            s->graph_key = key;
        }
        PIB_HIP(hipGraphLaunch(s->graph, q));
        s->graph_replays++;
        for (int k = 0; k < 8; ++k) s->counters[k] += s->graph_counts[k];
    }
    return 0;
}

// w = A p including the halo update of p (p is ghost-padded), optional fused p.w partials
static int matmult(pib_solver *s, double *p_owned, double *w, double *dot_part, bool guarded, hipStream_t stq)
{
    if (s->comm.nranks > 1 && s->halo_fresh != p_owned) PIB_CHK(halo_exchange(s, p_owned, stq));
    s->halo_fresh = nullptr;
    // the velocity operator from its mesh tables (same bits as the CSR product, 56 instead of 104 B/row)
    if (s->vel.valid && s->cfg.matrix_free_velocity && dot_part == nullptr && (s->comm.nranks == 1 || s->vel.slab_axis >= 0))
        return vel_stencil_apply(s, p_owned, w, guarded, stq);
    if (s->post_matmult != nullptr) {
        // operator = matrix + a term applied by the hook: the fused p.w of the SpMV would miss it
        if (stencil_matmult_ok(s)) PIB_CHK(stencil_matmult(s, p_owned, w, nullptr, guarded, stq));
        else PIB_CHK(spmv_rows(s, p_owned, w, 0, s->A.n, nullptr, guarded, stq));
        PIB_CHK(s->post_matmult(s, p_owned, w, guarded, stq, s->post_ctx));
        if (dot_part) PIB_CHK(dot_partials(s, p_owned, w, dot_part, guarded, stq));
        return 0;
    }
    if (stencil_matmult_ok(s)) return stencil_matmult(s, p_owned, w, dot_part, guarded, stq);
    return spmv_rows(s, p_owned, w, 0, s->A.n, dot_part, guarded, stq);
}

// p = z + beta p with the halo exchange of p overlapped (cfg.overlap_min_bytes >= 0): the entries the neighbours need are
// updated first, their exchange runs on the communication stream while the rest of p is updated; the SpMV that
// follows finds the halo fresh.
template <class Op>
static int update_p_and_exchange(pib_solver *s, int64_t n, const Op &up, double *P, hipStream_t q, bool vec2 = true)
{
    const DeviceCsr &A = s->A;
    // (a general plan's send entries are scattered over the vector: no leading / trailing part to update first)
    const bool split = s->comm.nranks > 1 && s->cfg.overlap_min_bytes >= 0 && !A.general && (A.send_prev % 2 == 0) && (A.send_next % 2 == 0) &&
                       (n % 2 == 0) && A.send_prev + A.send_next < n;
    if (!split) return launch_vec(s, n, up, vec2, 0, nullptr, true, q);
    PIB_CHK(launch_vec(s, n, up, vec2, 0, nullptr, true, q, 0, A.send_prev));
    PIB_CHK(launch_vec(s, n, up, vec2, 0, nullptr, true, q, n - A.send_next, n));
    PIB_HIP(hipEventRecord(s->ev_ready, q));
    PIB_HIP(hipStreamWaitEvent(s->stream_comm, s->ev_ready, 0));
    PIB_CHK(halo_exchange(s, P, s->stream_comm));
    PIB_HIP(hipEventRecord(s->ev_halo, s->stream_comm));
    PIB_CHK(launch_vec(s, n, up, vec2, 0, nullptr, true, q, A.send_prev, n - A.send_next));
    PIB_HIP(hipStreamWaitEvent(q, s->ev_halo, 0));
    s->halo_fresh = P;
    return 0;
}

static int init_scalars(pib_solver *s)
{
    Scalars h;
    std::memset(&h, 0, sizeof(h));
    h.atol = s->cfg.atol;
    h.rtol = s->cfg.rtol;
    h.dtol = s->cfg.dtol;
    h.maxit = s->cfg.max_iters;
    h.normtype = (s->cfg.norm == NormType::PRECONDITIONED) ? 0 : 1;
    *s->h_s = h;
    PIB_HIP(hipMemcpyAsync(s->d_s, s->h_s, sizeof(Scalars), hipMemcpyHostToDevice, s->stream));
    if (s->hist_cap < s->cfg.max_iters + 3) {
        if (s->d_hist) PIB_HIP(hipFree(s->d_hist));
        s->hist_cap = s->cfg.max_iters + 3;
        PIB_HIP(hipMalloc(&s->d_hist, (size_t)s->hist_cap * sizeof(double)));
        if (s->h_hist) (void)hipHostFree(s->h_hist);
        s->h_hist = nullptr;
        PIB_HIP(hipHostMalloc(&s->h_hist, (size_t)s->hist_cap * sizeof(double)));
    }
    return 0;
}

// Host round trips of a solve.  Each is a stream synchronisation -- 15-30 us of wake-up latency during which the GPU idles
// -- and a small system inside a time loop (the reference's 2-D cases: three solves of a few iterations per step) used to
// make four of them per solve: the state after the set-up kernels, the state after the first batch of iterations, the final
// state, the residual history.  Now two: the first poll is skipped when the previous solve of this solver iterated (the
// batch is then the previous count, its iterations guarded by the device's `done` flag, so enqueuing them blind costs
// nothing but empty launches in the rare case that the set-up already met the tolerance), and the final state comes with
// the history in one synchronisation (`enq` bounds the entries).
static bool skip_first_poll(const pib_solver *s)
{
    return s->cfg.check_every <= 0 && s->hint_iters >= 1 && s->cfg.max_iters >= 1;  // (h_s->done is 0: init_scalars)
}
static int fetch_results(pib_solver *s, int enq)
{
    const size_t entries = (size_t)std::min<int64_t>((int64_t)enq + 2, (int64_t)s->hist_cap);
    PIB_HIP(hipMemcpyAsync(s->h_hist, s->d_hist, sizeof(double) * entries, hipMemcpyDeviceToHost, s->stream));
    PIB_CHK(poll(s));
    s->iters = s->h_s->its;
    s->hint_iters = s->iters;
    s->reason = s->h_s->reason;
    s->residual = s->h_s->dp;
    if ((size_t)s->iters + 1 > entries) return fail(PIB_ERR_LIB, "solver %s: %d iterations recorded, %d enqueued", s->name.c_str(), s->iters, enq);
    s->history.assign(s->h_hist, s->h_hist + s->iters + 1);
    return 0;
}

// ------------------------------------------------------------------ CG
// Apply the multigrid preconditioner z = M^-1 r and finalize z.r, z.z, sum z (+ z[0] when pinned).
// `post` (guarded applications inside the iteration only): the scalar step that consumes the sums -- cg_s2's arguments.  On one
// rank, when the cycle deferred its sums' reduction (gmg.hip reduce_dots), that step runs inside the closing kernel and
// *post_done says so: the caller then skips its own k_cg_s2.
struct CgS2Args {
    double *hist;
    double n_global;
    int lazy_mean, do_norm, do_beta, conv_is_its;
};
static int gmg_pc_and_dots(pib_solver *s, const double *R, double *Z, bool guarded, hipStream_t q, bool reduce = true,
                           const CgS2Args *post = nullptr, bool *post_done = nullptr)
{
    int nb = 0;
    if (post_done) *post_done = false;
    s->gmg_guarded = guarded;
    s->gmg_want_dots = true;
    s->gmg_defer_dots = guarded;
    const int err = gmg_apply(s, R, Z, q);
    s->gmg_want_dots = false;
    s->gmg_defer_dots = false;
    if (err) return err;
    s->counters[1]++;
    const int owner = (s->A.row0 == 0) ? 1 : 0;
    if (s->gmg_dots_done && s->gmg_pending_count > 0) {
        // the three sums, z[0] and (one rank) the scalar step in one launch
        const bool with_post = post != nullptr && reduce && s->comm.nranks == 1;
        const CgS2Args a = post ? *post : CgS2Args{nullptr, 0.0, 0, 0, 0, 0};
        if (with_post)
            hipLaunchKernelGGL(k_dots_tail<1>, dim3(1), dim3(1024), 0, q, s->d_s, s->gmg_pending_part, s->gmg_pending_stride, s->gmg_pending_count, Z,
                               owner, a.hist, a.n_global, a.lazy_mean, a.do_norm, a.do_beta, a.conv_is_its);
        else
            hipLaunchKernelGGL(k_dots_tail<0>, dim3(1), dim3(1024), 0, q, s->d_s, s->gmg_pending_part, s->gmg_pending_stride, s->gmg_pending_count, Z,
                               owner, (double *)nullptr, 0.0, 0, 0, 0, 0);
        PIB_HIP(hipGetLastError());
        s->gmg_pending_count = 0;
        if (with_post && post_done) *post_done = true;
        if (!reduce) return 0;
        return allreduce_slots(s, 0, 4, q);
    }
    if (!s->gmg_dots_done) {  // else: z.r, z.z, sum z came out of the V-cycle's last smoothing kernel
        OpDotZR dz{Z, R};
        PIB_CHK(launch_vec(s, s->A.n, dz, true, 0, &nb, guarded, q));
        hipLaunchKernelGGL(k_finalize, dim3(3), dim3(256), 0, q, s->d_s, s->d_part, 0, nb);
    }
    hipLaunchKernelGGL(k_fetch_z0, dim3(1), dim3(1), 0, q, s->d_s, Z, owner);
    PIB_HIP(hipGetLastError());
    if (!reduce) return 0;  // (single-reduction CG: these sums travel with the product's)
    return allreduce_slots(s, 0, 4, q);
}

// ---- placing the search direction against the caller's x (cfg.place_update_vector)
// The p-update's access pattern with NEUTRAL arithmetic: z read, p and x read and written.  p is all zeros and a is -0.0, so
// x + a p gives every x back bit for bit (x + (-0.0) == x for +0.0 and -0.0 alike) and p stays zero; z only has to be loaded.
__global__ __launch_bounds__(256) void k_place_probe(int64_t n2, const double2 *__restrict__ z, double2 *__restrict__ p, double2 *__restrict__ x, double a)
{
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + 256 * u;
        if (i < n2) {
            const double2 vz = z[i], vp = p[i];
            double2 vx = x[i];
            vx.x += a * vp.x;
            vx.y += a * vp.y;
            x[i] = vx;
            double2 np = {vp.x + a * vp.x, vp.y + a * vp.y};
            if (vz.x == 1.2345678e300 && vz.y == -1.2345678e300) np.x = a;  // (keeps the load of z; p stays zero or minus zero)
            p[i] = np;
        }
    }
}

// Best of three launches, in ms (one untimed launch first).
static int place_probe_ms(pib_solver *s, int64_t n, const double *z, double *p, double *x, hipEvent_t e0, hipEvent_t e1, double *out)
{
    const int64_t n2 = n / 2;
    const unsigned nb = (unsigned)((n2 + 1023) / 1024);
    double best = 1e30;
    for (int r = 0; r < 4; ++r) {
        PIB_HIP(hipEventRecord(e0, s->stream));
        hipLaunchKernelGGL(k_place_probe, dim3(nb), dim3(256), 0, s->stream, n2, (const double2 *)z, (double2 *)p, (double2 *)x, -0.0);
        PIB_HIP(hipGetLastError());
        PIB_HIP(hipEventRecord(e1, s->stream));
        PIB_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        PIB_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = std::min(best, (double)ms);
    }
    *out = best;
    return 0;
}

// Work vector `idx` (p: dead between solves) into an allocation of its own that streams well beside x.
// What the lab found (tools/place_lab.hip pairs / mix / far, profiles/r05_vector_placement_lab.txt): device allocations fall
// into CLASSES -- runs of 4 to 64 GiB in allocation order -- and a kernel that reads and writes two vectors of one class runs
// some 12 % under the rate it has on vectors of two classes; which class an allocation is in cannot be read off its address, and
// the runs differ from box to box and from process to process.  So: time the probe on two NEIGHBOURING fresh allocations (one
// class, as good as always: the slow reference), then on p beside x; while that is not clearly under the slowest time seen, walk
// on through fresh allocations -- 2, 4, 8 GiB further each step, the gaps allocated and never touched -- and probe each beside x.
// BOUNDED (round 6): everything the walk holds at one time -- gaps, the two reference vectors, candidates not taken -- stays
// under PLACE_HOLD_CAP = min(16 GiB, a tenth of the memory free when the search starts); no search at all on a device whose
// memory is more than half taken (another solver of the process, other ranks or processes sharing the GPU: the transient
// allocations would be theirs to miss); an allocation or a probe that fails inside the walk ends the walk with what the solver
// has -- placement is optional, never a solve's error.  A search costs 10-20 ms when the pair is well placed already and
// 0.1-0.4 s when not, and runs when x is a buffer the solver has not seen (three times in a solver's life at most).
// pib_get_placement reports searches, candidates, the probe's two times, the bytes held at the walk's peak and its wall time.
static constexpr size_t PLACE_HOLD_CAP = (size_t)16 << 30;

static int place_walk(pib_solver *s, int idx, int zidx, double *x, int *tried_out, double *t_had_out, double *t_kept_out)
{
    const DeviceCsr &A = s->A;
    const size_t bytes = (size_t)(s->work_stride + 4) * sizeof(double);
    std::vector<void *> held;  // spacers and rejected candidates
    struct Held {
        std::vector<void *> &v;
        ~Held()
        {
            for (void *h : v) (void)hipFree(h);
        }
    } held_guard{held};
    size_t budget = 0, used = 0;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        if (fr < tot / 2) return 0;  // a shared or well-filled device: no search
        budget = std::min(PLACE_HOLD_CAP, fr / 10);
    }
    auto room = [&](size_t want) { return used + want <= budget; };
    auto fresh = [&](double **out) -> bool {
        *out = nullptr;
        if (!room(bytes) || hipMalloc(out, bytes) != hipSuccess) {
            (void)hipGetLastError();  // no room: the search ends with what it has
            *out = nullptr;
            return false;
        }
        if (hipMemsetAsync(*out, 0, bytes, s->stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(*out);
            *out = nullptr;
            return false;
        }
        used += bytes;
        s->place_held_bytes = std::max(s->place_held_bytes, (int64_t)used);
        return true;
    };
    double *&v = s->work_split[idx];
    if (v == nullptr) {
        if (!fresh(&v)) return 0;
        used -= bytes;  // (the solver's own vector is not a transient)
    } else
        PIB_HIP(hipMemsetAsync(v, 0, bytes, s->stream));
    hipEvent_t e0, e1;
    PIB_HIP(hipEventCreate(&e0));
    PIB_HIP(hipEventCreate(&e1));
    struct Ev {
        hipEvent_t a, b;
        ~Ev() { (void)hipEventDestroy(a), (void)hipEventDestroy(b); }
    } ev{e0, e1};
    const double *z = s->vec(zidx);
    const int64_t lo = s->work_lo;
    static const bool debug = std::getenv("PIB_PLACE_DEBUG") != nullptr;
    int tried = 0;
    // the slow reference: two neighbouring fresh allocations
    double *n0 = nullptr, *n1 = nullptr;
    double tnb = 0.0;
    if (fresh(&n0)) held.push_back(n0);
    if (fresh(&n1)) held.push_back(n1);
    if (n0 && n1) PIB_CHK(place_probe_ms(s, A.n, z, n0 + lo, n1 + lo, e0, e1, &tnb));
    if (debug) std::fprintf(stderr, "[place] neighbours %.3f ms\n", tnb);
    double t_had = 0.0;
    PIB_CHK(place_probe_ms(s, A.n, z, v + lo, x, e0, e1, &t_had));
    ++tried;
    double t_kept = t_had, tslow = std::max(tnb, t_had);
    *t_had_out = *t_kept_out = t_had;
    if (debug) std::fprintf(stderr, "[place] work vector %d (%p): %.3f ms\n", idx, (void *)v, t_had);
    // Two modes, 0.87-0.89 against 0.96-1.03 ms at 512^3 -- 0.925 of the slowest time seen separates them whatever the slow
    // sample was.
    int gap = 0;  // the next candidate comes 2^gap GiB further on
    bool first = true;
    for (int k = 0; k < 8 && t_kept > 0.925 * tslow; ++k) {
        double *c = nullptr;
        if (first && n1 != nullptr)
            c = n1;
        else {
            if (gap > 0) {
                size_t step = (size_t)1 << (30 + std::min(gap, 3));  // 2, 4, 8 GiB
                step = std::min(step, 64 * bytes);                   // (... of a vector of a GiB; small systems: in proportion)
                // (a gap that does not fit any more shrinks to what does, down to the vector's own size: then the walk ends)
                while (step > bytes && !room(step)) step >>= 1;
                void *sp = nullptr;
                if (step > bytes && hipMalloc(&sp, step - bytes) == hipSuccess) {
                    held.push_back(sp), used += step - bytes;
                    s->place_held_bytes = std::max(s->place_held_bytes, (int64_t)used);
                }
                (void)hipGetLastError();
            }
            if (!fresh(&c)) break;
            held.push_back(c);
        }
        first = false;
        double t = 0.0;
        PIB_CHK(place_probe_ms(s, A.n, z, c + lo, x, e0, e1, &t));
        ++tried;
        if (debug) std::fprintf(stderr, "[place] step %d (gap %d): %p for work vector %d: %.3f ms\n", k, gap, (void *)c, idx, t);
        tslow = std::max(tslow, t);
        if (t <= 0.925 * tslow) {
            for (void *&h : held)
                if (h == (void *)c) h = (void *)v;  // the vector the solver had goes with the rejected ones
            v = c;
            t_kept = t;
            drop_iteration_graph(s);
        }
        ++gap;
    }
    *tried_out = tried;
    *t_kept_out = t_kept;
    return 0;
}

// CG's placement: p beside x (the p-update reads and writes both).
static int place_update_vector(pib_solver *s, int idx, int zidx, double *x)
{
    const DeviceCsr &A = s->A;
    if (!s->cfg.place_update_vector || s->cfg.place_min_rows < 0 || A.n < s->cfg.place_min_rows || A.n != A.n_global || x == nullptr || !aligned16(x)) return 0;
    if (x == s->placed_against || s->placements >= 3 || idx >= pib_solver::MAX_WORK) return 0;
    int tried = 0;
    double t_had = 0.0, t_kept = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    if (place_walk(s, idx, zidx, x, &tried, &t_had, &t_kept) != 0) {
        // optional: a failure inside the walk is not the solve's failure (the vector the solver has stays)
        (void)hipGetLastError();
        s->departures.push_back(std::string("placement: a search was abandoned (") + pib_last_error() + ")");
    }
    PIB_HIP(hipStreamSynchronize(s->stream));
    s->place_search_ms += 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    s->placed_against = x;
    s->placements++;
    s->place_tried = tried;
    s->place_ms[0] = t_had;
    s->place_ms[1] = t_kept;
    return 0;
}

// x, b: device pointers, n_local entries.
template <int POST>
static int finalize_post(pib_solver *s, int slot0, int nslots, int count, double *hist, int conv_is_its, hipStream_t q,
                         const PinRowDev &pr = PinRowDev{nullptr, 0, {}, {}});

int solve_cg(pib_solver *s, double *x, const double *b)
{
    const DeviceCsr &A = s->A;
    const int64_t n = A.n;
    hipStream_t q = s->stream;
    PIB_CHK(ensure_work(s, 5));
    PIB_CHK(place_update_vector(s, 2, 1, x));
    double *R = s->vec(0), *Z = s->vec(1), *P = s->vec(2), *W = s->vec(3);
    double *R2 = s->vec(4);  // the other residual buffer of the fused update below
    const Precond pc = s->cfg.pc;
    const bool guess = s->cfg.initial_guess_nonzero;
    const double ng = (double)A.n_global;
    const bool gmg = (pc == Precond::GMG);
    int lazy = 0;
    if (s->nullspace == PIB_NULLSPACE_CONSTANT) lazy = 1;
    if (s->nullspace == PIB_NULLSPACE_PINNED && gmg) lazy = 2;
    const int monitor = s->cfg.monitor_residual ? 1 : 0;
    const int conv_is_its = monitor ? 0 : 1;
    const bool unprec = (s->cfg.norm == NormType::UNPRECONDITIONED);
    const bool v2 = aligned16(x) && aligned16(b);
    const bool xal = aligned16(x);  // the p-update carries x += a p: paired accesses need x aligned like the work vectors
    if (pc == Precond::NONE) Z = R;  // z aliases r
    if (pc == Precond::JACOBI && A.dinv == nullptr) return fail(PIB_ERR_ORDER, "Jacobi preconditioner without a diagonal");
    if (gmg && !s->has_grid)
        return fail(PIB_ERR_ORDER,
                    "solver %s: a multigrid (AMG/GMG) preconditioner needs the grid structure: call "
                    "pib_set_grid_hint or pib_assemble_poisson before pib_solve", s->name.c_str());
    const double omega = (pc == Precond::JACOBI) ? s->cfg.jacobi_relaxation : 1.0;
    for (int k = 0; k < 8; ++k) s->counters[k] = 0;
    PIB_CHK(init_scalars(s));
    int nb = 0;
    const int pin0 = (lazy == 2 && A.row0 == 0) ? 1 : 0;
    s->gmg_pin_local = false;  // (the set-up's cycle takes the sum OpInit delivers; the iterations may switch: see pin_local)
    struct PinLocalReset {
        pib_solver *s;
        ~PinLocalReset() { s->gmg_pin_local = false; }
    } pin_local_reset{s};

    // ---- initial residual, z, norms
    if (!guess) {
        OpFill z0{x, 0.0};
        PIB_CHK(launch_vec(s, n, z0, v2, 0, nullptr, false, q));
    }
    if (pin0) hipLaunchKernelGGL(k_pin_x0, dim3(1), dim3(1), 0, q, x, b);  // identity row 0: x[0] = b[0], r[0] = 0
    if (guess) {
        OpCopy cp{x, P};
        PIB_CHK(launch_vec(s, n, cp, v2, 0, nullptr, false, q));
        PIB_CHK(matmult(s, P, W, nullptr, false, q));
    }
    if (pc == Precond::JACOBI) {
        OpInit<PCM_JACOBI> op{b, W, A.dinv, R, Z, omega, guess ? 1 : 0, pin0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    } else {
        OpInit<PCM_NONE> op{b, W, nullptr, R, Z, 1.0, guess ? 1 : 0, pin0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    }
    PIB_CHK(finalize(s, 0, 6, nb, q));
    if (gmg) PIB_CHK(gmg_pc_and_dots(s, R, Z, false, q));
    hipLaunchKernelGGL(k_cg_s_init, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, lazy, monitor);
    PIB_HIP(hipGetLastError());

    // ---- iterations
    // Multigrid-preconditioned, one rank, systems too large for the captured graphs: r = r - alpha w is left to the V-cycle's
    // first kernel, which reads the residual anyway (gmg.hip k_presmooth2<., 1>: 24 B/row and a launch less per iteration).
    // That kernel recomputes halo cells, so the new residual goes to the OTHER of two buffers, iteration by iteration.
    // (several ranks, round 4: on z-slabs with deep halos too -- w is exchanged instead of the residual, gmg.hip)
    // (a pinned pressure row, round 5: the compatible right-hand side of the cycle needs the NEW residual's sum before the march
    // that forms it -- pin_sigma, from cg_s1's one-step recurrence)
    // (several ranks: gmg_fused_update_ok decides from EVERY rank's slab -- its rows against the captured-graph limit included --,
    // never from this rank's own: all ranks take the fused form or none does)
    const bool fused_upd = gmg && s->cfg.fuse_residual_update && (lazy == 1 || (lazy == 2 && s->cfg.pin_sum_local != 0 && s->pin_row.ready)) &&
                           gmg_fused_update_ok(s);
    // (several ranks: pin_sigma exists on the rank that owns cell 0 only.  With the fused update the slabs are wall-bounded and
    // thick enough that no other rank recomputes cell 0; without it -- a z-ring, where the last rank sees cell 0 as a halo cell
    // across the seam, or thin slabs -- every rank takes the all-reduced red[5] instead, whatever pib_pin_sum_local asks for)
    const bool pin_local = gmg && lazy == 2 && s->pin_row.ready && (s->cfg.pin_sum_local == 1 || (s->cfg.pin_sum_local < 0 && fused_upd)) &&
                           (s->comm.nranks <= 1 || fused_upd);
    PinRowDev pin_dev{nullptr, 0, {}, {}};
    if (pin_local && A.row0 == 0 && s->pin_row.n > 0) {
        pin_dev.p = P;
        pin_dev.n = s->pin_row.n;
        for (int f = 0; f < s->pin_row.n; ++f) {
            pin_dev.off[f] = (long long)s->pin_row.off[f];
            pin_dev.coef[f] = s->pin_row.coef[f];
        }
    }
    struct UpdCtx {
        double *hist;
        double ng;
        int conv_is_its, unprec;
        int64_t n;
        double *z;
    };
    static thread_local UpdCtx upd_ctx;
    upd_ctx = UpdCtx{s->d_hist, ng, conv_is_its, unprec ? 1 : 0, n, Z};
    // the cycle's first march cannot take the update (gmg.hip decides with its own launch-site predicate): the separate pass
    // of the general path, landing in the new residual's buffer
    auto update_fallback = +[](pib_solver *ps, double *r_new, hipStream_t st) -> int {
        int nblk = 0;
        PIB_HIP(hipMemcpyAsync(r_new, ps->gmg_upd.r_old, sizeof(double) * (size_t)upd_ctx.n, hipMemcpyDeviceToDevice, st));
        OpUpdateXR<PCM_NONE> op{ps->gmg_upd.w, nullptr, r_new, upd_ctx.z, 1.0, 0.0};
        PIB_CHK(launch_vec(ps, upd_ctx.n, op, true, 0, &nblk, true, st));
        PIB_CHK(finalize(ps, 0, 6, nblk, st));
        if (upd_ctx.unprec)
            hipLaunchKernelGGL(k_cg_s2, dim3(1), dim3(1), 0, st, ps->d_s, upd_ctx.hist, upd_ctx.ng, 0, 1, 0, upd_ctx.conv_is_its);
        PIB_HIP(hipGetLastError());
        return 0;
    };
    auto after_update = +[](pib_solver *ps, int nblocks, hipStream_t st) -> int {
        // r.r and sum r of the new residual from the march's partials (slots 4, 5), then the convergence step on |r|
        if (ps->comm.nranks == 1 && upd_ctx.unprec)  // (POST 8: the two sums, then cg_s2's norm step)
            return finalize_post<8>(ps, 4, 2, nblocks, upd_ctx.hist, upd_ctx.conv_is_its, st);
        hipLaunchKernelGGL(k_finalize, dim3(2), dim3(256), 0, st, ps->d_s, ps->d_part, 4, nblocks);
        PIB_HIP(hipGetLastError());
        PIB_CHK(allreduce_slots(ps, 4, 2, st));  // (several ranks; nothing on one)
        if (upd_ctx.unprec)
            hipLaunchKernelGGL(k_cg_s2, dim3(1), dim3(1), 0, st, ps->d_s, upd_ctx.hist, upd_ctx.ng, 0, 1, 0, upd_ctx.conv_is_its);
        PIB_HIP(hipGetLastError());
        return 0;
    };
    const int batch0 = first_batch(s), batch1 = next_batch(s);
    const int spmv_blocks = spmv_launch_blocks();
    int enq = 0;
    const int maxit = s->cfg.max_iters;
    double *part_pw = s->d_part + (int64_t)SLOT_PW * PIB_MAXPART;
    // the x update the last iteration owes
    auto flush = [&](int if_done) {
        hipLaunchKernelGGL(k_flush_x, dim3((unsigned)std::min<int64_t>(VGRID_MAX, std::max<int64_t>(1, (n + 255) / 256))), dim3(256), 0, q,
                           s->d_s, n, P, x, if_done);
        hipLaunchKernelGGL(k_flush_done, dim3(1), dim3(1), 0, q, s->d_s, if_done);
    };
    if (!skip_first_poll(s)) {
        // the state after the set-up kernels; if they already met the tolerance (a time loop near its steady state) the closing
        // kernels have gone out with the poll and this was the solve
        flush(1);
        PIB_CHK(fetch_results(s, 0));
        if (s->h_s->done) return 0;
    }
    while (!s->h_s->done && enq < maxit) {
        const int todo = std::min(enq == 0 ? batch0 : batch1, maxit - enq);
        auto body = [&]() -> int {
            const int64_t zv = gmg ? (int64_t)s->z_halo_depth * s->levels[0].plane : 0;
            if (s->comm.nranks > 1 && zv > 0 && zv >= A.ghost_lo && zv >= A.ghost_hi) {
                // the V-cycle left z valid on the ghost planes the matrix reaches: p = z + beta p there too (the ghost
                // values of p follow the same recurrence as their owners'), and the product needs no exchange
                const bool ev = (A.ghost_lo & 1) == 0 && (n & 1) == 0 && xal;
                OpUpdateP up{Z - A.ghost_lo, P - A.ghost_lo, x - A.ghost_lo, A.ghost_lo, A.ghost_lo + n, 0.0, 0.0, 0.0, 0, 0};
                PIB_CHK(launch_vec(s, n + A.ghost_lo + A.ghost_hi, up, ev, 0, nullptr, true, q));
                s->halo_fresh = P;
            } else {
                OpUpdateP up{Z, P, x, 0, n, 0.0, 0.0, 0.0, 0, 0};
                PIB_CHK(update_p_and_exchange(s, n, up, P, q, xal));
            }
            PIB_CHK(matmult(s, P, W, part_pw, true, q));
            if (s->comm.nranks == 1)
                PIB_CHK(finalize_post<4>(s, SLOT_PW, 1, spmv_blocks, nullptr, 0, q, pin_dev));
            else {
                PIB_CHK(finalize(s, SLOT_PW, 1, spmv_blocks, q));
                hipLaunchKernelGGL(k_cg_s1, dim3(1), dim3(1), 0, q, s->d_s, pin_dev);
            }
            s->gmg_pin_local = pin_local;  // (read by gmg_apply while the body is enqueued or captured; cleared behind the loop)
            if (fused_upd) {
                s->gmg_upd.w = W;
                s->gmg_upd.r_old = R;
                s->gmg_upd.after = after_update;
                s->gmg_upd.fallback = update_fallback;
                s->gmg_upd.used = false;
                const CgS2Args closing{s->d_hist, ng, lazy, unprec ? 0 : 1, 1, conv_is_its};
                bool closed = false;
                const int err = gmg_pc_and_dots(s, R2, Z, true, q, true, &closing, &closed);
                s->gmg_upd.w = nullptr;
                if (err) return err;
                if (!s->gmg_upd.used) return fail(PIB_ERR_LIB, "solver %s: the multigrid did not take the residual update", s->name.c_str());
                std::swap(R, R2);
                s->counters[6]++;
                if (!closed) hipLaunchKernelGGL(k_cg_s2, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, lazy, unprec ? 0 : 1, 1, conv_is_its);
                PIB_HIP(hipGetLastError());
                return 0;
            }
            if (pc == Precond::JACOBI) {
                OpUpdateXR<PCM_JACOBI> op{W, A.dinv, R, Z, omega, 0.0};
                PIB_CHK(launch_vec(s, n, op, true, 0, &nb, true, q));
            } else {
                OpUpdateXR<PCM_NONE> op{W, nullptr, R, Z, 1.0, 0.0};
                PIB_CHK(launch_vec(s, n, op, true, 0, &nb, true, q));
            }
            const bool merged6 = gmg && unprec && s->comm.nranks == 1;
            if (merged6) PIB_CHK(finalize_post<8>(s, 0, 6, nb, s->d_hist, conv_is_its, q));
            else PIB_CHK(finalize(s, 0, 6, nb, q));
            if (gmg) {
                if (unprec && !merged6)
                    hipLaunchKernelGGL(k_cg_s2, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, 0, 1, 0, conv_is_its);
                const CgS2Args closing{s->d_hist, ng, lazy, unprec ? 0 : 1, 1, conv_is_its};
                bool closed = false;
                PIB_CHK(gmg_pc_and_dots(s, R, Z, true, q, true, &closing, &closed));
                if (!closed)
                    hipLaunchKernelGGL(k_cg_s2, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, lazy, unprec ? 0 : 1, 1,
                                       conv_is_its);
            } else {
                hipLaunchKernelGGL(k_cg_s2, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, lazy, 1, 1, conv_is_its);
            }
            PIB_HIP(hipGetLastError());
            return 0;
        };
        PIB_CHK(run_iterations(s, todo, enq, graph_key(1, x, b), q, body));
        const bool first = enq == 0;
        enq += todo;
        if (first) {
            // the usual case inside a time loop: this batch was the whole solve.  The closing kernels go out behind it, acting
            // only if the device has set `done`, and ONE synchronisation brings the final state and the history
            flush(1);
            PIB_CHK(fetch_results(s, enq));
            if (s->h_s->done) return 0;
        } else
            PIB_CHK(poll(s));
    }
    flush(0);
    return fetch_results(s, enq);
}

// ------------------------------------------------------------------ CG, single reduction
// `cg_single_reduction` (solver file: pib_cg_single_reduction=1; PETSc options: -<name>_ksp_cg_single_reduction, the reference's
// KSP reads the options database at linsolverksp.cpp:62-66).  Per iteration: one vector pass (OpSRUpdate), the preconditioner,
// s = A z with z.s fused, ONE all-reduce of seven sums, one scalar step -- against three all-reduces of the standard recurrence.
// The price is w = s + b w: 16 B/row more vector traffic; it pays where the reductions' latency does (several ranks).
// A z needs z on the matrix's ghost entries: the V-cycle on slabs leaves it valid there (no exchange at all), any other
// preconditioner exchanges z; p never leaves the rank.  The monitored norm is tested behind the iteration's product, so the
// iteration that meets the tolerance still pays its V-cycle and product (results, counts and history are unaffected).
int solve_cg_sr(pib_solver *s, double *x, const double *b)
{
    const DeviceCsr &A = s->A;
    const int64_t n = A.n;
    hipStream_t q = s->stream;
    PIB_CHK(ensure_work(s, 5));
    PIB_CHK(place_update_vector(s, 2, 1, x));
    double *R = s->vec(0), *Z = s->vec(1), *P = s->vec(2), *W = s->vec(3), *SV = s->vec(4);
    const Precond pc = s->cfg.pc;
    const bool guess = s->cfg.initial_guess_nonzero;
    const double ng = (double)A.n_global;
    const bool gmg = (pc == Precond::GMG);
    int lazy = 0;
    if (s->nullspace == PIB_NULLSPACE_CONSTANT) lazy = 1;
    // A pinned pressure row with the multigrid (round 5): the shift by z[0] does not commute with the product (the pinned matrix
    // does not annihilate constants), and here the matrix is applied to z itself -- so it is applied for real, in a pass of its
    // own behind the cycle (OpPinShift, which also forms z.r, z.z, sum z); the cycle's compatible right-hand side needs the
    // GLOBAL sum of r ahead of it.  Costs one 24 B/row pass and, on several ranks, two small all-reduces more per iteration.
    const bool pin_gmg = gmg && s->nullspace == PIB_NULLSPACE_PINNED;
    const int pin0 = (pin_gmg && A.row0 == 0) ? 1 : 0;
    const int monitor = s->cfg.monitor_residual ? 1 : 0;
    const int conv_is_its = monitor ? 0 : 1;
    const bool v2 = aligned16(x) && aligned16(b);
    const bool xal = aligned16(x);
    if (pc == Precond::NONE) Z = R;  // z aliases r
    if (pc == Precond::JACOBI && A.dinv == nullptr) return fail(PIB_ERR_ORDER, "Jacobi preconditioner without a diagonal");
    if (gmg && !s->has_grid)
        return fail(PIB_ERR_ORDER,
                    "solver %s: a multigrid (AMG/GMG) preconditioner needs the grid structure: call "
                    "pib_set_grid_hint or pib_assemble_poisson before pib_solve", s->name.c_str());
    const double omega = (pc == Precond::JACOBI) ? s->cfg.jacobi_relaxation : 1.0;
    for (int k = 0; k < 8; ++k) s->counters[k] = 0;
    PIB_CHK(init_scalars(s));
    int nb = 0;
    const int spmv_blocks = spmv_launch_blocks();
    double *part_zs = s->d_part + (int64_t)SLOT_PW * PIB_MAXPART;

    // s = A z with the z.s partials (slot 6); the V-cycle on slabs leaves z valid on the matrix's ghost entries
    auto product = [&](bool guarded) -> int {
        const int64_t zv = gmg ? (int64_t)s->z_halo_depth * s->levels[0].plane : 0;
        if (s->comm.nranks > 1 && zv > 0 && zv >= A.ghost_lo && zv >= A.ghost_hi) s->halo_fresh = Z;
        PIB_CHK(matmult(s, Z, SV, part_zs, guarded, q));
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, q, s->d_s, s->d_part, SLOT_PW, spmv_blocks);
        PIB_HIP(hipGetLastError());
        return 0;
    };

    // the cycle under a pinned row: sum r made global first, the cycle, z[0] to every rank, the shift with the sums
    auto pinned_cycle = [&](bool guarded) -> int {
        PIB_CHK(allreduce_slots(s, 4, 2, q));  // r.r, sum r (nothing on one rank)
        s->gmg_guarded = guarded;
        s->gmg_want_dots = false;
        PIB_CHK(gmg_apply(s, R, Z, q));
        s->counters[1]++;
        hipLaunchKernelGGL(k_fetch_z0, dim3(1), dim3(1), 0, q, s->d_s, Z, (A.row0 == 0) ? 1 : 0, 3);
        PIB_HIP(hipGetLastError());
        PIB_CHK(allreduce_slots(s, 3, 1, q));
        int nbz = 0;
        OpPinShift<1> sh{Z, R, pin0, 3, 0.0};
        PIB_CHK(launch_vec(s, n, sh, true, 0, &nbz, true, q));  // (reads the shift from the scalars: always given them)
        hipLaunchKernelGGL(k_finalize, dim3(3), dim3(256), 0, q, s->d_s, s->d_part, 0, nbz);
        PIB_HIP(hipGetLastError());
        s->z_halo_depth = 0;  // (the shift ran on the owned entries: the ghost planes the cycle left behind are stale, the product exchanges z)
        return 0;
    };
    // the iteration's one big reduction: slots 0 .. 6; under a pinned row r.r / sum r (4, 5) and z[0] (3) are global already
    auto reduce_all = [&]() -> int {
        if (!pin_gmg) return allreduce_slots(s, 0, 7, q);
        PIB_CHK(allreduce_slots(s, 0, 3, q));
        return allreduce_slots(s, 6, 1, q);
    };
    // ---- set-up: r, z, their sums (as solve_cg), then the first product and its sum
    if (!guess) {
        OpFill z0{x, 0.0};
        PIB_CHK(launch_vec(s, n, z0, v2, 0, nullptr, false, q));
    }
    if (pin0) hipLaunchKernelGGL(k_pin_x0, dim3(1), dim3(1), 0, q, x, b);  // identity row 0: x[0] = b[0], r[0] = 0
    if (guess) {
        OpCopy cp{x, P};
        PIB_CHK(launch_vec(s, n, cp, v2, 0, nullptr, false, q));
        PIB_CHK(matmult(s, P, W, nullptr, false, q));
    }
    if (pc == Precond::JACOBI) {
        OpInit<PCM_JACOBI> op{b, W, A.dinv, R, Z, omega, guess ? 1 : 0, 0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    } else {
        OpInit<PCM_NONE> op{b, W, nullptr, R, Z, 1.0, guess ? 1 : 0, pin0};
        PIB_CHK(launch_vec(s, n, op, v2, 0, &nb, false, q));
    }
    hipLaunchKernelGGL(k_finalize, dim3(6), dim3(256), 0, q, s->d_s, s->d_part, 0, nb);
    PIB_HIP(hipGetLastError());
    if (pin_gmg) PIB_CHK(pinned_cycle(false));
    else if (gmg) PIB_CHK(gmg_pc_and_dots(s, R, Z, false, q, false));
    PIB_CHK(product(false));
    PIB_CHK(reduce_all());
    hipLaunchKernelGGL(k_cg_s_init, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, lazy, monitor);
    hipLaunchKernelGGL(k_cg_sr_first, dim3(1), dim3(1), 0, q, s->d_s);
    PIB_HIP(hipGetLastError());

    const int batch0 = first_batch(s), batch1 = next_batch(s);
    int enq = 0;
    const int maxit = s->cfg.max_iters;
    if (!skip_first_poll(s)) {
        PIB_CHK(fetch_results(s, 0));
        if (s->h_s->done) return 0;
    }
    while (!s->h_s->done && enq < maxit) {
        const int todo = std::min(enq == 0 ? batch0 : batch1, maxit - enq);
        auto body = [&]() -> int {
            if (pc == Precond::JACOBI) {
                OpSRUpdate<PCM_JACOBI> op{SV, A.dinv, Z, P, W, x, R, omega, 0.0, 0.0, 0.0, 0};
                PIB_CHK(launch_vec(s, n, op, xal, 0, &nb, true, q));
                hipLaunchKernelGGL(k_finalize, dim3(6), dim3(256), 0, q, s->d_s, s->d_part, 0, nb);
            } else if (pc == Precond::NONE) {
                OpSRUpdate<PCM_NONE> op{SV, nullptr, Z, P, W, x, R, 1.0, 0.0, 0.0, 0.0, 0};
                PIB_CHK(launch_vec(s, n, op, xal, 0, &nb, true, q));
                hipLaunchKernelGGL(k_finalize, dim3(6), dim3(256), 0, q, s->d_s, s->d_part, 0, nb);
            } else {
                OpSRUpdate<PCM_EXTERNAL> op{SV, nullptr, Z, P, W, x, R, 1.0, 0.0, 0.0, 0.0, 0};
                PIB_CHK(launch_vec(s, n, op, xal, 0, &nb, true, q));
                hipLaunchKernelGGL(k_finalize, dim3(2), dim3(256), 0, q, s->d_s, s->d_part, 4, nb);  // r.r, sum r
                if (pin_gmg) PIB_CHK(pinned_cycle(true));
                else PIB_CHK(gmg_pc_and_dots(s, R, Z, true, q, false));
            }
            PIB_HIP(hipGetLastError());
            PIB_CHK(product(true));
            PIB_CHK(reduce_all());
            hipLaunchKernelGGL(k_cg_sr_step, dim3(1), dim3(1), 0, q, s->d_s, s->d_hist, ng, lazy, conv_is_its);
            PIB_HIP(hipGetLastError());
            return 0;
        };
        PIB_CHK(run_iterations(s, todo, enq, graph_key(3, x, b), q, body));
        const bool first = enq == 0;
        enq += todo;
        if (first) {
            PIB_CHK(fetch_results(s, enq));
            if (s->h_s->done) return 0;
        } else
            PIB_CHK(poll(s));
    }
    return fetch_results(s, enq);
}

}  // namespace pib

#include "krylov_bicgstab.hpp"
#include "krylov_chebyshev.hpp"


// ------------------------------------------------------------ instrumentation
extern "C" int pib_time_kernel(pib_solver *s, int which, int reps, double *ms_avg)
try {
    using namespace pib;
    if (s == nullptr || ms_avg == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_time_kernel: null argument");
    if (!s->has_matrix) return fail(PIB_ERR_ORDER, "pib_time_kernel: no matrix");
    if (reps < 1) reps = 1;
    PIB_HIP(hipSetDevice(s->device));
    PIB_CHK(ensure_work(s, 4));
    const int64_t n = s->A.n;
    double *R = s->vec(0), *Z = s->vec(1), *P = s->vec(2), *W = s->vec(3);
    hipStream_t q = s->stream;
    // events are recorded on the stream the kernels are launched on
    auto run = [&](int count) -> int {
        for (int i = 0; i < count; ++i) {
            switch (which) {
                case 0: PIB_CHK(spmv_rows(s, P, W, 0, n, nullptr, false, q)); break;
                case 1: {
                    OpUpdateXR<PCM_JACOBI> op{W, s->A.dinv, R, Z, 1.0, 0.0};
                    // a = 0 from a zeroed Scalars copy is not guaranteed: use an unguarded launch with S = nullptr
                    // (prepare() needs S) -> use the guarded form; d_s->done must be 0.
                    PIB_CHK(launch_vec(s, n, op, true, 0, nullptr, true, q));
                    break;
                }
                case 2: {
                    OpDotZR op{Z, R};
                    PIB_CHK(launch_vec(s, n, op, true, 0, nullptr, false, q));
                    break;
                }
                case 3:  // the Krylov product of the stencil twin (K2, 16 B/row)
                    if (!s->has_grid || s->levels.empty() || s->comm.nranks != 1) return fail(PIB_ERR_ORDER, "pib_time_kernel: no grid structure");
                    PIB_CHK(stencil_matmult(s, P, W, nullptr, false, q));
                    break;
                case 4: PIB_CHK(gmg_apply(s, R, Z, q)); break;
                case 6: PIB_CHK(spmv_rows(s, P, W, 0, n, s->d_part, false, q)); break;  // the product with the fused p.w partials, as CG launches it
                case 7: {  // ... behind the p-update, as in a CG iteration (the pair's time is reported)
                    OpUpdateP up{Z, P, R, 0, n, 0.0, 0.0, 0.0, 0, 0};
                    PIB_CHK(launch_vec(s, n, up, true, 0, nullptr, true, q));
                    PIB_CHK(spmv_rows(s, P, W, 0, n, s->d_part, false, q));
                    break;
                }
                case 8: {  // the p-update alone
                    OpUpdateP up{Z, P, R, 0, n, 0.0, 0.0, 0.0, 0, 0};
                    PIB_CHK(launch_vec(s, n, up, true, 0, nullptr, true, q));
                    break;
                }
                case 5:  // the matrix-free product of the velocity operator (velstencil.hip, 56 B/row)
                    if (!s->vel.valid) return fail(PIB_ERR_ORDER, "pib_time_kernel: no velocity-operator structure");
                    PIB_CHK(vel_stencil_apply(s, P, W, false, q));
                    break;
                default: return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_time_kernel: unknown kernel %d", which);
            }
        }
        return 0;
    };
    if (which == 1 || which == 7 || which == 8) {
        PIB_HIP(hipMemsetAsync(s->d_s, 0, sizeof(Scalars), q));
    }
    if (which == 100) {
        // measurement only (tools/spmv_placement_scan.py): the CSR product with its INPUT, then its OUTPUT, on each of 14 fresh
        // allocations 2 GiB apart -- which vector's placement class sets the product's 2.36 / 2.50 ms modes?  Printed to stderr.
        const size_t bytes = (size_t)(s->work_stride + 4) * sizeof(double);
        std::vector<void *> held;
        std::vector<double *> cand;
        for (int k = 0; k < 14; ++k) {
            void *sp = nullptr;
            double *c = nullptr;
            if (k && bytes < ((size_t)2 << 30) && hipMalloc(&sp, ((size_t)2 << 30) - bytes) == hipSuccess) held.push_back(sp);
            if (hipMalloc(&c, bytes) != hipSuccess) break;
            PIB_HIP(hipMemsetAsync(c, 0, bytes, q));
            held.push_back(c);
            cand.push_back(c + s->work_lo);
        }
        auto time_pw = [&](double *p, double *w, double *out) -> int {
            double best = 1e30;
            for (int r = 0; r < 4; ++r) {
                PIB_HIP(hipEventRecord(s->ev_a, q));
                PIB_CHK(spmv_rows(s, p, w, 0, n, nullptr, false, q));
                PIB_HIP(hipEventRecord(s->ev_b, q));
                PIB_HIP(hipEventSynchronize(s->ev_b));
                float ms = 0.f;
                PIB_HIP(hipEventElapsedTime(&ms, s->ev_a, s->ev_b));
                if (r) best = std::min(best, (double)ms);
            }
            *out = best;
            return 0;
        };
        double t = 0.0;
        PIB_CHK(time_pw(P, W, &t));
        std::fprintf(stderr, "[spmv scan] the solver's own p %p, w %p: %.3f ms\n", (void *)P, (void *)W, t);
        for (size_t k = 0; k < cand.size(); ++k) {
            double tp = 0.0, tw = 0.0, tb = 0.0;
            PIB_CHK(time_pw(cand[k], W, &tp));
            PIB_CHK(time_pw(P, cand[k], &tw));
            PIB_CHK(time_pw(cand[k], cand[(k + 1) % cand.size()], &tb));
            std::fprintf(stderr, "[spmv scan] candidate %2zu %p: as the input %.3f ms, as the output %.3f ms, input with the next one as output %.3f ms\n", k,
                         (void *)cand[k], tp, tw, tb);
        }
        // ... then, with the best output found, the column indices and after them the coefficients COPIED to fresh allocations
        // 4 GiB apart (the arrays themselves stay where they are: the pointers are swapped for the timing only)
        double *wbest = W;
        double tbest = t;
        for (size_t k = 0; k < cand.size(); ++k) {
            double tw = 0.0;
            PIB_CHK(time_pw(P, cand[k], &tw));
            if (tw < tbest) tbest = tw, wbest = cand[k];
        }
        std::fprintf(stderr, "[spmv scan] best output %p: %.3f ms\n", (void *)wbest, tbest);
        DeviceCsr &A = s->A;
        const size_t cb = sizeof(int32_t) * (size_t)(A.nnz + 4), vb = sizeof(double) * (size_t)(A.nnz + 4);
        int32_t *const col_own = A.col;
        double *const val_own = A.val;
        int32_t *col_best = A.col;
        double *val_best = A.val;
        int rc = 0;
        for (int pass = 0; pass < 2 && rc == 0; ++pass) {
            for (int k = 0; k < 8 && rc == 0; ++k) {
                void *sp = nullptr, *c = nullptr;
                const size_t want = pass == 0 ? cb : vb;
                if (k && hipMalloc(&sp, (size_t)4 << 30) == hipSuccess) held.push_back(sp);
                if (hipMalloc(&c, want) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                held.push_back(c);
                if (hipMemcpyAsync(c, pass == 0 ? (void *)col_own : (void *)val_own, want, hipMemcpyDeviceToDevice, q) != hipSuccess) break;
                A.col = pass == 0 ? (int32_t *)c : col_best;
                A.val = pass == 0 ? val_own : (double *)c;
                double t1 = 0.0, t2 = 0.0;
                rc = time_pw(P, wbest, &t1);
                if (rc == 0) rc = time_pw(P, W, &t2);
                std::fprintf(stderr, "[spmv scan] %s copied to %p: %.3f ms with the best output, %.3f ms with the solver's own\n", pass == 0 ? "column indices" : "coefficients", c, t1,
                             t2);
                if (rc == 0 && t1 < tbest * 0.97) {
                    tbest = t1;
                    if (pass == 0) col_best = (int32_t *)c;
                    else val_best = (double *)c;
                }
            }
        }
        A.col = col_own;  // the solver's own arrays again, before the copies go
        A.val = val_own;
        PIB_HIP(hipStreamSynchronize(q));
        std::fprintf(stderr, "[spmv scan] best found %.3f ms (columns %s, coefficients %s)\n", tbest, col_best == col_own ? "own" : "moved", val_best == val_own ? "own" : "moved");
        if (rc != 0) {
            for (void *h : held) (void)hipFree(h);
            return rc;
        }
        for (void *h : held) (void)hipFree(h);
        *ms_avg = t;
        return 0;
    }
    PIB_CHK(run(2));  // warm-up
    PIB_HIP(hipEventRecord(s->ev_a, q));
    PIB_CHK(run(reps));
    PIB_HIP(hipEventRecord(s->ev_b, q));
    PIB_HIP(hipEventSynchronize(s->ev_b));
    float ms = 0.f;
    PIB_HIP(hipEventElapsedTime(&ms, s->ev_a, s->ev_b));
    *ms_avg = (double)ms / reps;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
