// pib_internal.hpp -- internal types of libpetibm_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdint>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/petibm_amd.h"

namespace pib {

int fail(int code, const char *fmt, ...);  // records the message, returns code
int fail_exception(const char *where) noexcept;  // the C ABI's catch-all: the exception in flight as an error code + message (config.cpp)
void install_crash_backtrace();  // PIB_CRASH_BACKTRACE=1: native frames on SIGSEGV / SIGABRT / SIGBUS, then the previous handler (config.cpp)
const char *last_error();

// PIB_TRACE_SETUP=1: where the time of a set-up path goes (printed by rank 0 to stderr); tools/box_route_probe.py
struct SetupTrace {
    bool on;
    const char *what;
    std::chrono::steady_clock::time_point t;
    SetupTrace(const char *w, int rank) : on(false), what(w), t(std::chrono::steady_clock::now())
    {
        const char *e = std::getenv("PIB_TRACE_SETUP");
        on = e && e[0] == '1' && rank == 0;
    }
    void mark(const char *phase)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pib setup] %s: %s %.3f s\n", what, phase, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};

// Host loops over the rows / entries of a matrix handed over through pib_set_csr (set-up only: the 16.8 M rows of a 512^3 / 8
// box go through several of them): contiguous ranges on a few threads.  f(begin, end) must touch disjoint data per range.
// PIB_HOST_THREADS overrides the count (default: the hardware's, at most 16; 1 below 2^18 items).
template <class F>
inline void par_ranges(int64_t n, F f)
{
    int T = 1;
    if (n >= ((int64_t)1 << 18)) {
        T = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = std::getenv("PIB_HOST_THREADS")) T = std::max(1, std::atoi(e));
    }
    if (T <= 1) {
        f((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + T - 1) / T;
    for (int t = 0; t < T; ++t) {
        const int64_t b = std::min<int64_t>(n, t * chunk), e = std::min<int64_t>(n, b + chunk);
        if (e <= b) continue;
        try {
            th.emplace_back([=]() { f(b, e); });
        } catch (const std::system_error &) {  // no more threads to be had (a pids limit): the range on this one
            f(b, e);
        }
    }
    for (auto &x : th) x.join();
}

#define PIB_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return pib::fail(PIB_ERR_LIB, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call,            \
                             hipGetErrorString(e_));                                                    \
    } while (0)
// hipMemset of device memory may return before it has run, and the solver streams are non-blocking (they do not wait
// for the null stream): a kernel launched next on such a stream could be overtaken by the memset.  Set-up paths only.
#define PIB_MEMSET(ptr, value, bytes)                 \
    do {                                              \
        PIB_HIP(hipMemset((ptr), (value), (bytes)));  \
        PIB_HIP(hipStreamSynchronize(nullptr));       \
    } while (0)
#define PIB_NCCL(call)                                                                                  \
    do {                                                                                                \
        ncclResult_t e_ = (call);                                                                       \
        if (e_ != ncclSuccess)                                                                          \
            return pib::fail(PIB_ERR_LIB, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call,            \
                             ncclGetErrorString(e_));                                                   \
    } while (0)
#define PIB_CHK(call)                   \
    do {                                \
        int e_ = (call);                \
        if (e_ != 0) return e_;         \
    } while (0)

// ------------------------------------------------------------------ config
enum class Flavor { AMGX, KSP };
enum class Method { CG, BICGSTAB, PREONLY, CHEBYSHEV };
enum class Precond { NONE, JACOBI, GMG, LU };
enum class Smoother { JACOBI, CHEBYSHEV };
enum class NormType { PRECONDITIONED, UNPRECONDITIONED };

struct Config {
    Flavor flavor = Flavor::AMGX;
    Method method = Method::CG;
    Precond pc = Precond::NONE;
    NormType norm = NormType::UNPRECONDITIONED;
    int max_iters = 100;       // AmgX default; KSP default 10000
    double rtol = 0.0;         // relative to the initial monitored norm
    double atol = 1e-12;       // absolute
    double dtol = 1e4;         // KSP divergence tolerance (AmgX: disabled)
    bool monitor_residual = true;
    bool store_res_history = true;
    bool error_if_not_converged = true;
    bool initial_guess_nonzero = true;  // AmgX: x is the guess; KSP: false
    double jacobi_relaxation = 1.0;     // AmgX BLOCK_JACOBI relaxation_factor (as PC)
    // geometric multigrid (stands in for AmgX AMG / PCGAMG)
    int presweeps = 1, postsweeps = 1;
    // One sweep of the solver file = one fused PAIR of damped-Jacobi steps of the geometric cycle (pib_sweep_pairs=1, the
    // default): every solver file of the reference says presweeps = postsweeps = 1 for AmgX's classical AMG, and the stand-in's
    // V(2,2) -- two steps per pass of the LDS-tiled marches -- beats its V(1,1) by 24 % in time to solution at 512^3
    // (72.3 vs 89.8 ms, 11 vs 15 iterations: INTEGRATION.md).  0: the file's counts literally.  Chebyshev smoothing: never.
    int sweep_pairs = 1;
    int fuse_chebyshev_update = 1;  // Method::CHEBYSHEV on the matrix-free velocity operator: the update inside the product's launch
    Smoother smoother = Smoother::JACOBI;
    double smoother_relaxation = 0.9;
    // Method::CHEBYSHEV: bounds of the preconditioned operator's spectrum (-<name>_ksp_chebyshev_eigenvalues emin,emax;
    // AmgX flavour: cheby_min_lambda / cheby_max_lambda of the solver's scope); both 0: the Gershgorin interval of the matrix
    double cheb_emin = 0.0, cheb_emax = 0.0;
    int cheby_degree = 2;
    double cheby_lmax = 2.0;   // eigenvalue window [lmax/ratio, lmax] of D^-1 A (Gershgorin: <= 2 for the FV Poisson operator)
    double cheby_ratio = 4.0;
    int max_levels = 100;
    int coarsest_sweeps = 32;
    // execution
    int check_every = 0;     // iterations enqueued between host convergence polls (0 = auto)
    int64_t graph_max_rows = 1 << 22;  // capture the iteration body in a hipGraph for systems of at most this many local rows (launch-bound ones); 0: never
    int overlap_min_bytes = 1 << 20;  // multigrid on slabs: a right-hand-side exchange of at least this size per neighbour runs on the communication stream behind the interior planes of its first consumer
    int coarse_tail = -1;    // > 0: multigrid levels with at most this many cells run in ONE single-workgroup kernel; -1: 1024; 0: off.
    int fuse_small_levels = 1;  // gmg.hip: a small level's way down / way up in one launch each (k_small_down / k_small_up); 0: per-phase launches
    int coarse_tail_lds = 1;  // ... with the tail levels' vectors and 1-D tables in LDS when they fit (gmg.hip: 83 -> 44 us per tail of a 448^2 mesh); 0: HBM
                             // Measured SLOWER than per-level launches at every size on MI355X (512^3: 142.5 ms off, 145 ms at
                             // 64..4096 cells, 152 ms at 32768): launches pipeline, one CU with block barriers does not. Off.
    int matrix_free_poisson = -1;  // Krylov products of a Poisson solve with the stencil twin: 1 on, 0 off, -1 = on inside the device time step only (>= 2^20 rows)
    int cg_single_reduction = 0;   // CG with PETSc's single-reduction recurrences (-<name>_ksp_cg_single_reduction; solver file: pib_cg_single_reduction=1): ONE all-reduce per iteration, 16 B/row more vector traffic (krylov.hip solve_cg_sr)
    int fuse_residual_update = 1;  // PCG + multigrid: r = r - alpha w by the V-cycle's first march instead of a pass of its own.  1 (default): on one rank and on z-slabs -- there w = A p is exchanged to the depth the residual was and the march keeps the residual's ghost planes by recurrence; -1: on one rank only; 0: the separate pass; 2 (tests): asked for, but the launch site refuses it: the fallback pass
    int pin_sum_local = -1;  // pinned pressure row + multigrid: the residual's sum that makes the cycle's right-hand side compatible from the recurrence sum r - alpha sum w, sum w = -(row 0 of the singular operator) . p (krylov.hip cg_s1) instead of the update pass's own sum -- what lets the update ride in the cycle's first march; -1: with the fused update only, 1: always, 0: never (no fused update under a pinned row)
    int bicgstab_form = 3;  // BiCGStab on the matrix-free velocity operator (krylov.hip OpBFUpdateP): 0 the general path; 1 lean -- no stored M^-1 p / M^-1 s, the x update deferred: the same iterates bit for bit; 2 + the two dot-only passes summed by the products themselves (sums grouped by tile: iterates to rounding); 3 (default) + r = s - omega t formed by the next p-update, |r|^2 and r.rp from the second product's five sums (16 B/row/iteration and one reduction less)
    int fuse_velocity_product = 1;  // 3-D: the three components' tiles and shells in one launch (velstencil.hip k_vel_product)
    int matrix_free_velocity = 1;  // Krylov products with the velocity operator from the mesh tables (velstencil.hip) instead of the CSR
    int march_min_cells = 12 << 20;  // smallest level / plane run the LDS-tiled marching kernels take: a 256^3 level, the 62 interior planes of a 512 x 512 x 64 slab (tests lower it to reach them on small grids)
    int march = 1;           // the LDS-tiled z-marching forms: Jacobi steps / residuals of the large multigrid levels (gmg_level_kernels.hpp k_level_march), the restriction of a fully paired 3-D level (gmg_down_kernels.hpp k_restrict_march), the matrix-free velocity product (velstencil.hip k_vel_march); 0: the flat streaming forms (same bits)
    int fuse_prolong = 1;  // multigrid: prolongation + first post-smoothing step of a fully paired large level in one kernel (gmg.hip k_prolong_smooth)
    int fuse_post_pair = 1;  // multigrid: prolongation + both post-smoothing steps of V(., 2) in ONE march (gmg.hip k_prolong_smooth2)
    int fuse_down_march = 1;  // multigrid V(2, .), a level whole on one rank and not periodic: the two pre-smoothing steps (with PCG's residual update), the residual and the restriction in ONE march (gmg.hip k_down_march: 40 instead of 57 B per cell); 0: k_presmooth2 + k_resid_restrict_march
    int fuse_residual_restrict = 1;  // multigrid: residual + restriction of such a level in ONE march (gmg.hip k_resid_restrict_march)
    int fuse_presmooth = 1;  // multigrid: the first two pre-smoothing steps of a level in one LDS-tiled kernel (gmg.hip k_presmooth2)
    int compress_columns = 2;  // what the CSR product streams besides the values, where the matrix allows: 2 one byte per ROW (the row's pattern of column offsets, DeviceCsr::pat_id: 73 instead of 104 B per 7-point row), 1 one byte per entry (DeviceCsr::code: 83 B), 0 the int32 columns and row offsets.  The same products in the same order, bit for bit
    int place_update_vector = 1;  // CG on one rank, systems of place_min_rows rows and more: the search direction p gets an allocation of its own, CHOSEN by timing the p-update's access pattern against the caller's x while walking through fresh allocations (krylov.hip, place_update_vector).  The flat update reads and writes both vectors, and its rate has two modes (6.3 against 5.6 TB/s at 512^3: 835 against 960 us, 8 % of the solve) set by which physical blocks the two sit in -- a property of the pair, the same for the life of the process, that no address arithmetic inside one allocation moves (profiles/r05_vector_placement_lab.txt)
    int64_t place_min_rows = (int64_t)1 << 25;  // one rank, systems of at least that many rows: every work vector of the Krylov methods an allocation of its own instead of one pool (-1: the pool always), and CG's search (measured on slabs of the 512^3 system: 2^24 rows no gain, 2^25 1.5 %, 2^26 2.6 %, 2^27 3 %)
    int detect_structure = 1;  // pib_set_csr with a multigrid preconditioner: recover the mesh structure from the matrix (structure.cpp)
    int deep_halo = 2;       // multi-GPU multigrid: exchange several ghost planes at once and recompute the ghost cells (gmg.hip); 2 (default): ... and the coarse corrections of the way up need no exchange of their own (a level's right-hand side is exchanged as deep as its final iterate is read by the finer level's prolongation: fin_l[]); 1: deep on the way down only; 0: one plane per stencil kernel
    int agglomerate_below = 300000;  // multi-GPU GMG: levels with fewer cells are solved redundantly per GPU
    std::string raw;
};

int parse_config_text(const std::string &text, const std::string &name, Config &cfg);
int parse_config_file(const char *path, const std::string &name, Config &cfg);

// ------------------------------------------------------------- device scalars
// One block of scalars in HBM drives the Krylov recurrences: no host round
// trip inside an iteration.  Every iteration kernel starts with `if (S->done)
// return;` so over-enqueued iterations are no-ops.
constexpr int PIB_MAX_RANKS = 64;  // ranks of one communicator (general halo plans, box <-> slab moves, the peer transport)
constexpr int PIB_NRED = 8;       // reduction slots
constexpr int PIB_MAXPART = 32768;  // partial sums per slot (>= the workgroups of any kernel that leaves partials: the fused marches of a 1024^3 level are 16384-18432; 4096 until round 6 sent 768^3 and up to the separate residual-update pass)
struct Scalars {
    double red[PIB_NRED];  // finalized (and all-reduced) sums
    double beta, betaold, dpi, dpiold, dp, a, b;
    double rho, rhoold, alpha, omega, omegaold;  // BiCGStab
    double rnorm0, ttol, atol, rtol, dtol;
    double mean;  // null-space projection
    int its;      // completed iterations
    int reason;   // 0 while running
    int done;     // != 0: all further kernels are no-ops
    int maxit;
    int normtype;  // 0 preconditioned 1 unpreconditioned
    // CG: x += a p of iteration k is applied by the p-update of iteration k + 1 (one pass over p less); xa_it counts the
    // updates owed, xapplied the ones the p-updates have made: they differ while one is pending (flushed after the loop)
    int xa_it, xapplied;
    // BiCGStab on the matrix-free velocity operator: x += xalpha M^-1 p + xomega M^-1 s of iteration k is applied by the
    // p-update of iteration k + 1 (xpend: one is owed; flushed after the loop)
    int xpend;
    double xalpha, xomega;
    // Chebyshev iteration (solve_chebyshev): c[km1], c[k] of the recurrence, its constants, the coefficients of the NEXT update
    // p[kp1] = a0 p[km1] + omega p[k] + cz z, and which of the two rotating vectors holds the current iterate
    double c_km1, c_k, cheb_mu, cheb_omegaprod, cheb_scale, cheb_a0, cheb_cz;
    int sol;
    // pinned pressure row + multigrid with the residual update inside the cycle: sum r of the residual the first march is about to
    // form (cg_s1: red[5] - alpha sum w)
    double pin_sigma;
};

// Row 0 of the SINGULAR level-0 operator without its diagonal (the entries MatZeroRowsColumns removed from the matrix,
// navierstokes.cpp:414-420): offsets of the neighbours of cell 0 in a ghost-padded vector and the unscaled coefficients.
// The columns of the singular operator sum to zero, so for any p with p[0] = 0:  sum_{i >= 1} (A' p)_i = -sum_f coef[f] p[off[f]].
struct PinRow {
    bool ready = false;  // the grid is registered with a pinned row (true on every rank; n > 0 on the rank that owns cell 0)
    int n = 0;
    int64_t off[6] = {0, 0, 0, 0, 0, 0};
    double coef[6] = {0, 0, 0, 0, 0, 0};
};
struct PinRowDev {  // ... as a kernel argument, with the vector
    const double *p;
    int n;
    long long off[6];
    double coef[6];
};

// ------------------------------------------------------------------ who sends how many doubles to whom
// One message per ordered pair of ranks; every rank holds the whole table, so a receiver knows where its message lies
// in the sender's stream (the sender's messages back to back in destination order) without asking.
struct ExchangePlan {
    int P = 0, me = 0;
    std::vector<int64_t> cnt;       // [P * P]: cnt[s * P + d] doubles from s to d
    std::vector<int64_t> send_off;  // [P + 1]: my message to d is stream[send_off[d] .. send_off[d + 1])
    std::vector<int64_t> src_off;   // [P]: my message from s starts at this offset of s's stream
    int64_t send_total = 0, recv_total = 0, max_stream = 0;
    void finish(int nranks, int rank)
    {
        P = nranks;
        me = rank;
        send_off.assign((size_t)P + 1, 0);
        src_off.assign((size_t)P, 0);
        recv_total = max_stream = 0;
        for (int d = 0; d < P; ++d) send_off[(size_t)d + 1] = send_off[(size_t)d] + cnt[(size_t)me * P + d];
        send_total = send_off[(size_t)P];
        for (int q = 0; q < P; ++q) {
            int64_t o = 0, tot = 0;
            for (int d = 0; d < P; ++d) {
                if (d == me) o = tot;
                tot += cnt[(size_t)q * P + d];
            }
            src_off[(size_t)q] = o;
            recv_total += cnt[(size_t)q * P + me];
            max_stream = tot > max_stream ? tot : max_stream;
        }
    }
    int64_t to(int d) const { return cnt[(size_t)me * P + d]; }
    int64_t from(int q) const { return cnt[(size_t)q * P + me]; }
};

// ------------------------------------------------------------------ matrix
struct DeviceCsr {
    int64_t n = 0;        // local rows
    int64_t nnz = 0;
    int64_t row0 = 0;     // global index of local row 0
    int64_t n_global = 0;
    int64_t ghost_lo = 0, ghost_hi = 0;  // halo widths: local col index = global col - (row0 - ghost_lo)
    bool rp64 = false;
    void *rowptr = nullptr;   // int32 or int64 [n+1], values relative to this rank's first nnz
    int32_t *col = nullptr;   // local (ghost-shifted) column index
    double *val = nullptr;
    double *dinv = nullptr;   // 1/diag
    // Column codes (kernels_spmv.hip, build_column_codes): per entry one byte, the index of (col - row) in the dictionary of the
    // entry's 256-row block (CODE_DICT entries per block) -- what the CSR product streams instead of the 4-byte column whenever
    // every block of the matrix has at most CODE_DICT distinct offsets (any stencil matrix); col stays for everybody else.
    static constexpr int CODE_DICT = 16;
    uint8_t *code = nullptr;   // [nnz + 64]
    int32_t *dict = nullptr;   // [(n + 255) / 256 + 1][CODE_DICT]
    bool coded = false;
    // Row patterns (build_row_patterns): where every row of a 256-row block has at most PAT_LEN entries and the block at most
    // PAT_N distinct rows-as-offset-lists, a row is one byte -- the number of its pattern in the block's table -- and the product
    // streams no per-entry index and no per-row offset at all: 8 B per entry + 1 B per row (+ 4 B per block).
    // Blocks with the same table share it (pat_blk: the block's table), so that the tables stay in the L2.
    static constexpr int PAT_N = 16, PAT_LEN = 8;
    uint8_t *pat_id = nullptr;    // [blocks * 256]
    int32_t *pat_blk = nullptr;   // [blocks] index of the block's table
    int32_t *pat_tab = nullptr;   // [tables][PAT_N][PAT_LEN] offsets col - row
    uint8_t *pat_len = nullptr;   // [tables][PAT_N]
    int64_t pat_tables = 0;
    bool patterned = false;
    int64_t first_boundary_lo = 0;  // rows [0, n_lo) touch the low halo
    int64_t n_lo = 0, n_hi = 0;     // rows touching the low / high halo (contiguous at both ends)
    int64_t send_prev = 0, send_next = 0;  // entries the neighbours need from this rank
    int64_t halo_group_longest = 0;        // ... and the longest such message on any rank of the group (the peer transport's protocol choice)
    int64_t max_chunk_nnz = 1 << 30;       // largest 256-row chunk (selects the SpMV's LDS capacity)
    // Segmented halo plan (packed velocity ordering: a rank's vector is [u-slab | v-slab | w-slab], so what a neighbour
    // needs is one plane out of each block): {offset in the owned vector, count} per message to the previous / next
    // rank; the ghost pads receive the neighbours' segments back to back in the same order.  Empty: the contiguous
    // plan (send_prev / send_next entries from the two ends).
    std::vector<std::pair<int64_t, int64_t>> seg_send_prev, seg_send_next;
    std::vector<int64_t> seg_recv_lo, seg_recv_hi;  // counts of the segments received into the low / high ghost pad
    bool segmented = false;
    // General plan (any row partition: the DMDA boxes PETSC_DECIDE gives an unchanged PetIBM from 4 ranks up, its
    // per-rank-packed velocity ordering; partition.cpp): the ghost columns are the sorted distinct off-rank columns --
    // those below row0 fill the low pad, the others the high pad --, every owner packs the entries its peers asked for
    // (send_idx: local rows, grouped by destination rank) and one grouped exchange delivers them (what AmgX does behind
    // AmgXSolver::setA, src/linsolver/linsolveramgx.cpp:84).
    bool general = false;
    ExchangePlan xplan;
    int32_t *send_idx = nullptr;   // device [xplan.send_total]
    double *send_buf = nullptr;    // device [xplan.send_total]
    std::vector<int64_t> ghost_cols;   // host: global column of every ghost entry, ascending
    std::vector<int64_t> ghost_off;    // [P + 1]: ghosts owned by rank q are ghost_cols[ghost_off[q] .. ghost_off[q + 1])
    void release();
};

// grid hint / structured operator of one multigrid level (device arrays)
struct GridLevel {
    int dim = 0;
    int64_t n[3] = {1, 1, 1};   // global cells
    int64_t k0 = 0, k1 = 1;     // owned slab along the last axis
    // face coefficient factors: off-diagonal toward +d of cell s is
    //   c_d[s] * (product of the two perpendicular width arrays)
    double *w[3] = {nullptr, nullptr, nullptr};  // [n[d]]
    double *g[3] = {nullptr, nullptr, nullptr};  // [n[d]-1] (g[d][s] couples s and s+1), already * dt
    // the level operator's rows divided by the cell volume have 1-D coefficients (gmg.hip): towards -d / +d of cell s
    // cm[d][s] = g[d][s-1] / w[d][s], cp[d][s] = g[d][s] / w[d][s] (0 at a wall, the wrap face when periodic); rw = 1 / w
    double *cm[3] = {nullptr, nullptr, nullptr}, *cp[3] = {nullptr, nullptr, nullptr}, *rw[3] = {nullptr, nullptr, nullptr};
    double *dinv = nullptr;                      // 1/diag per local cell (with pinned handling)
    double *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;  // level vectors (one halo plane each side)
    double *d = nullptr;  // Chebyshev direction vector
    // transfer tables towards the next coarser level, per direction (null on the coarsest level): parent /
    // other coarse index and their weights per fine cell [n[d]], first child per coarse cell [nc[d]+1]
    int32_t *t_par[3] = {nullptr, nullptr, nullptr}, *t_oth[3] = {nullptr, nullptr, nullptr},
            *t_fst[3] = {nullptr, nullptr, nullptr};
    double *t_wpar[3] = {nullptr, nullptr, nullptr}, *t_woth[3] = {nullptr, nullptr, nullptr};
    // x direction packed per coarse cell (gmg.hip: TrX)
    int2 *tx_fc = nullptr;
    double4 *tx_pw = nullptr, *tx_rw = nullptr;
    bool replicated = false;  // multi-GPU: every rank holds the whole level
    int64_t nloc = 0, plane = 0;
    int64_t pad = 0;          // entries of halo memory before the first owned plane of every level vector
    std::vector<int32_t> hz_par, hz_oth;  // host copy of the z transfer table (halo-depth planning, gmg.hip)
    int per = 0;  // bit d: direction d (internal order) is periodic and has > 1 cell: the level operator wraps
    int tper = 0; // bit d: ... and so do the transfers towards the next coarser level (>= 4 cells)
    bool plain_pair = false;  // every aggregate towards the next coarser level is a pair of cells (n = 2 nc) in all three directions
    bool zring = false;  // distributed level on a periodic slab axis: the z wrap goes through the halo planes
};

struct LoopbackGroup;  // halo.hip: test-only transport (ranks = threads of one process on one GPU)
// One RCCL communicator, shared by every solver that borrows it (one RCCL id makes one communicator: the Poisson solver of a
// flow engine, the inner slab solver of the box route).  The LAST holder to let go destroys it -- unless it was aborted
// (krylov.hip poll(): a collective that never completed): then every holder sees `aborted`, nobody destroys or uses the
// dangling handle again, and any later solve of the group fails at once with a clear error (halo.hip comm_abort / comm_usable).
struct CommShared {
    ncclComm_t comm = nullptr;
    bool aborted = false;
};
struct Comm {
    ncclComm_t comm = nullptr;
    std::shared_ptr<CommShared> shared;  // set with `comm` (null for the loopback / peer transports and for one rank)
    LoopbackGroup *loop = nullptr;
    int rank = 0, nranks = 1;
    bool borrowed = false;  // the communicator belongs to another solver of the same engine (one RCCL id makes one communicator)
    bool peer = false;      // `loop` is a peer-transport group (one process per rank, HIP IPC) that this solver attached
    bool ring = false;  // the slab axis is periodic: rank 0 and rank P-1 are neighbours (their outer ghost planes wrap)
    // a transport is attached (always and only when nranks > 1, except in pib_comm_selftest's one-rank RCCL world)
    bool active() const { return comm != nullptr || loop != nullptr; }
};

// the velocity operator's structure (velstencil.hip): per field and direction the Laplacian quotients, the ghost folds,
// MatScale / MatShift of A = I/dt - c nu L; valid after a single-rank pib_assemble_velocity
struct VelStencil {
    bool valid = false;
    int dim = 0, per = 0;
    int64_t n[3][3] = {{1, 1, 1}, {1, 1, 1}, {1, 1, 1}}, off[3] = {0, 0, 0};
    const double *lneg[3][3] = {{nullptr}}, *lpos[3][3] = {{nullptr}};
    double a0[3][6] = {{0}};
    double scale = 0.0, shift = 0.0;
    std::vector<double *> owned;
    // several ranks: this rank's slab of every field along the last axis (n[f][sd] = its planes, the tables along sd start at
    // its first plane); the neighbours' planes sit in the ghost pads of the vector, at pad_lo[f] / pad_hi[f] relative to
    // the first owned entry (0: no neighbour there -- a true boundary)
    int slab_axis = -1;
    int64_t pad_lo[3] = {0, 0, 0}, pad_hi[3] = {0, 0, 0};
    bool has_lo = false, has_hi = false;
};

}  // namespace pib

// a window of the mesh along one axis (bn.hip on slabs): the coordinates are those of the whole mesh -- running sums from
// its first cell, the Laplacian's coefficients are made of their differences -- at planes first .. first + n[axis] - 1
// (taken modulo n_global across a periodic seam, shifted by the period)
struct MeshWindow {
    bool active = false;
    int axis = 0;
    int64_t first = 0, n_global = 0;
    const double *w_global = nullptr;
    double lo = 0.0, hi = 0.0;  // ends of the whole mesh along the axis
};

// Rows handed over in DMDA boxes (PETSc ordering) and a multigrid preconditioner asked for: the rows, and every solve's
// b and x, are moved once / per solve to the z-slabs in natural ordering the geometric multigrid works on, inside an
// inner solver that shares this one's communicator (partition.cpp, redistribute.hip).
struct RedistField {
    int64_t n[3] = {1, 1, 1};       // points of the field, internal layout (a 2-D grid is (nx, 1, ny): the slab axis is always the last)
    std::vector<int64_t> box;       // [P][6]: xs, ys, zs, xm, ym, zm of every rank
    pib::ExchangePlan fwd, bwd;     // boxes -> slabs (b, the guess), slabs -> boxes (x)
    int64_t k0 = 0, k1 = 0;         // this rank's planes of the field
    int32_t *d_split = nullptr;     // device: x / y / z box boundaries (m + 1, n + 1, p + 1 entries back to back)
    int64_t *d_src = nullptr;       // device [P][2]: offset of rank q's chunk in the staging vector, first plane of the chunk
    double *stage = nullptr;        // device [n_slab]
    int64_t n_slab = 0;
    int64_t boff = 0, soff = 0;     // the field's block in this rank's box-ordered / slab-ordered vector
};
struct Redist {
    bool active = false;
    pib_solver *inner = nullptr;
    int dim = 0;
    int grid[3] = {1, 1, 1};        // process grid (m, n, p), internal layout
    int nf = 0;                     // 1: the pressure system; dim: the velocity system (per rank [u box | v box | w box] <-> [u slab | v slab | w slab])
    RedistField f[3];
    double *b_nat = nullptr, *x_nat = nullptr;  // device [slab rows]
    int64_t n_slab = 0;             // slab rows of all fields
};

struct pib_solver {
    std::string name, cfg_path, type_string;
    pib::Config cfg;
    // where the backend departs from the solver file (pib_describe, printInfo): one sentence per departure, in the order they
    // were decided -- e.g. CG of the file replaced by BiCGStab for a non-symmetric DBNG (navierstokes.hip ns_create)
    std::vector<std::string> departures;
    pib::Comm comm;
    int device = 0;
    hipStream_t stream = nullptr, stream_comm = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_halo = nullptr, ev_ready = nullptr;
    pib::DeviceCsr A;
    bool has_matrix = false;
    // grid hint
    bool has_grid = false;
    int nullspace = PIB_NULLSPACE_NONE;
    std::vector<pib::GridLevel> levels;
    std::vector<double *> gmg_spare = std::vector<double *>(64, nullptr);
    bool gmg_guarded = true;
    std::string gmg_error;  // why the hierarchy could not be built (reported when a multigrid solve is asked for)
    int periodic[3] = {0, 0, 0};             // pib_set_periodic: problem directions x, y[, z] (in effect: the structure recovery may set them for the structure it registers)
    int periodic_user[3] = {0, 0, 0};        // ... as the caller declared them: what an explicit pib_set_grid_hint describes
    pib::VelStencil vel;                     // matrix-free twin of the velocity operator (velstencil.hip)
    // w <- w + (low-rank term) after every Krylov product: the coupled immersed-boundary operator (ibm.hip) turns the
    // Poisson solve into the solve of its Schur complement
    int (*post_matmult)(pib_solver *s, const double *p, double *w, bool guarded, hipStream_t q, void *ctx) = nullptr;
    void *post_ctx = nullptr;
    bool structure_detected = false;         // the grid structure was recovered from the CSR itself (structure.cpp)
    MeshWindow mesh_window;                  // scratch solver of a slab's BN chain: the mesh is a window of the whole one
    bool hint_pc_only = false;               // the grid structure describes the preconditioner's operator only (BN order > 1)
    std::vector<double> asm_w[3], asm_g[3];  // 1-D arrays of the last on-device assembly
    double asm_dt = 0.0;
    // multi-GPU multigrid: plane ownership [b,e) of every rank on every level (the aggregates of the finer level's slab planes: gmg.hip grid_register)
    std::vector<std::vector<std::pair<int64_t, int64_t>>> gmg_own;
    // work vectors: each ghost-padded [ghost_lo + n + ghost_hi]
    double *work = nullptr, *work_base = nullptr;
    int64_t work_stride = 0;
    int n_work = 0;
    double *x_dev = nullptr, *b_dev = nullptr;  // staging for host-pointer callers
    int64_t stage_n = 0;
    hipEvent_t ev_stage[2] = {nullptr, nullptr};  // bracket the copies of a host-vector caller's b / guess (read after the solve)
    double stage_ms[2] = {0.0, 0.0};  // host-vector callers: what the last solve spent copying b (+ the guess) in / x out (pib_get_staging_ms)
    pib::Scalars *d_s = nullptr, *h_s = nullptr;
    double *d_part = nullptr;  // [PIB_NRED][PIB_MAXPART]
    double *d_spmv_part = nullptr;  // one p.Ap partial per SpMV workgroup
    int64_t spmv_part_cap = 0;
    bool vel_detected = false;     // s->vel was recovered from a matrix handed over through pib_set_csr (structure.cpp)
    double *d_vel_part = nullptr;  // per-workgroup partials of the velocity product's fused sums [2][vel_part_cap]
    int vel_part_cap = 0;
    double *d_gmg_part = nullptr;   // z.r, z.z, sum z partials of the V-cycle's last smoothing kernel (gmg.hip mode 8)
    int64_t gmg_part_cap = 0;
    bool gmg_want_dots = false, gmg_dots_done = false;
    // ... whose final reduction the cycle may leave to the solver's closing kernel (gmg.hip reduce_dots, krylov.hip k_dots_tail)
    bool gmg_defer_dots = false;
    double *gmg_pending_part = nullptr;
    int gmg_pending_stride = 0, gmg_pending_count = 0;
    pib::PinRow pin_row;          // gmg.hip grid_register (PINNED)
    bool gmg_pin_local = false;   // this cycle's compatible right-hand side takes Scalars::pin_sigma instead of red[5] (set by the solver around gmg_apply)
    // PCG's residual update r = r_old - alpha w left to the preconditioner's first kernel (gmg.hip k_presmooth2<., 1>): set by
    // the solver around gmg_apply; `after` finalizes r.r / sum r from the kernel's partials and runs the convergence step
    struct GmgUpd {
        const double *w = nullptr, *r_old = nullptr;
        int (*after)(pib_solver *, int nblocks, hipStream_t) = nullptr;
        // the march cannot take the update after all (alignment of the cycle's buffers, a level that is not tile-divisible):
        // r_new = r_old - alpha w and its sums in a pass of their own, after which the cycle runs its plain form
        int (*fallback)(pib_solver *, double *r_new, hipStream_t) = nullptr;
        bool used = false;
    } gmg_upd;
    double *d_hist = nullptr;
    double *h_hist = nullptr;  // its pinned host copy (fetched with the final scalars in one synchronisation)
    int hist_cap = 0;
    hipGraphExec_t graph = nullptr;   // one Krylov iteration (krylov.hip: run_iterations)
    uint64_t graph_key = 0;           // the (method, x, b) it was captured for
    int64_t graph_replays = 0;        // iterations launched as a graph replay since the solver was created
    int64_t graph_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // instrumentation counters one replay stands for
    pib_solver *reduce_via = nullptr;  // direct solve of a matrix whose entries every rank holds PARTIAL sums of (the force system of
                                       // immersed bodies on slabs): the dense matrix is summed over this solver's communicator first
    double *dense_inv = nullptr;  // dense.hip: explicit inverse of the direct solver [dense_n x dense_n]
    double *dense_work = nullptr; // the matrix being eliminated
    int *dense_bad = nullptr;     // zero-pivot flag
    void *d_tail_args = nullptr;        // gmg.hip: argument block of the single-workgroup coarse tail (+ its host copy)
    std::vector<char> h_tail_args;
    void *d_tail_tab = nullptr;         // ... and the 1-D tables of the tail's levels packed for its LDS copy
    double *dense_pad = nullptr;  // the matrix padded to a multiple of the block order, inverted in place by the blocked elimination (+ one block of scratch)
    hipGraphExec_t dense_graph = nullptr;  // the dense_n elimination launches
    hipStream_t dense_stream2 = nullptr;   // blocked elimination: the look-ahead's stream (the next diagonal block) and its events
    hipEvent_t dense_ev[3] = {nullptr, nullptr, nullptr};
    int64_t dense_n = 0;
    // results of the last solve
    int iters = 0, reason = 0;
    int hint_iters = 0;  // iterations of the previous solve (first enqueue batch of the next one)
    const double *vel_epi_b = nullptr, *vel_epi_dinv = nullptr;  // vel_stencil_apply_cheb's arguments on their way into the launch
    double *vel_epi_pm = nullptr;
    double vel_epi_opc = 1.0;
    double gersh_lo = 0.0, gersh_hi = -1.0;  // gershgorin_bounds' cache (hi < lo: not computed for this matrix)
    int gersh_jacobi = -1;
    int64_t work_pad = 0;     // entries of halo memory the Krylov work vectors keep below / above their owned part (>= the CSR's ghost columns)
    int z_halo_depth = 0;     // ghost planes on which the last V-cycle's result is valid (multi-GPU)
    const double *halo_fresh = nullptr;  // vector whose halo planes were exchanged by its producer (overlap path)
    double residual = 0.0;
    std::vector<double> history;
    int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t work_lo = 0;      // entries before the owned part of a work vector
    Redist redist;            // general row partition + multigrid: the solve happens on z-slabs in the inner solver
    // Work vector i: out of the one pool, or (split_work: large systems) an allocation of its own -- the rate of a flat kernel that
    // writes two vectors depends on which PHYSICAL blocks the two sit in (profiles/r05_vector_placement_lab.txt), and only
    // separately allocated vectors can be exchanged for better placed ones.
    static constexpr int MAX_WORK = 12;
    double *work_split[MAX_WORK] = {};
    const double *placed_against = nullptr;  // the x that p was last placed against (place_update_vector)
    int placements = 0, place_tried = 0;     // searches run (at most 3 in a solver's life), candidates timed by the last one
    double place_ms[2] = {0.0, 0.0};         // the probe with the vector the solver had / with the one it kept
    int64_t place_held_bytes = 0;            // the most a search held at one time (gaps, reference vectors, rejected candidates)
    double place_search_ms = 0.0;            // wall time of all searches
    double *vec(int i) const { return (work_split[i] != nullptr ? work_split[i] : work + (int64_t)i * work_stride) + work_lo; }
};

namespace pib {
// kernels_spmv.hip
int spmv_rows(pib_solver *s, const double *x_owned, double *y, int64_t r_begin, int64_t r_end, double *dot_part,
              bool guarded, hipStream_t st);
int spmv_launch_blocks();
int extract_dinv(pib_solver *s, int *n_missing);
// halo.hip
int comm_init(pib_solver *s, int rank, int nranks, const void *uid);
int comm_setup_halo(pib_solver *s);
int halo_exchange(pib_solver *s, double *x_owned, hipStream_t st);
int halo_exchange_planes(pib_solver *s, double *x_owned, int64_t n_owned, int64_t lo, int64_t hi, int64_t send_prev,
                         int64_t send_next, hipStream_t st, int64_t group_longest = 0);
int allreduce_slots(pib_solver *s, int first, int count, hipStream_t st);
int comm_allreduce_sum(pib_solver *s, double *dev, int count, hipStream_t st);
int comm_allreduce_big(pib_solver *s, double *dev, int64_t count, hipStream_t st);
int comm_allgather_host(pib_solver *s, const std::vector<double> &mine, std::vector<double> &all);
int comm_allgatherv(pib_solver *s, const double *send, double *recv_base, const std::vector<int64_t> &counts,
                    const std::vector<int64_t> &offs, hipStream_t st);
// every rank's messages lie back to back (destination order) at `stream`; the message from rank q lands at recv[q]
int comm_exchange_v(pib_solver *s, const ExchangePlan &pl, const double *stream, double *const *recv, hipStream_t st);
void comm_release(pib_solver *s);
void comm_abort(pib_solver *s);   // ncclCommAbort ONCE for the whole sharing group
int comm_usable(pib_solver *s);   // PIB_ERR_LIB when the group's communicator was aborted (also clears this solver's alias)
bool comm_capturable(const pib_solver *s);
void comm_capture_boundary(pib_solver *s, bool begin);
// partition.cpp: rows in any partition (DMDA boxes, the per-rank-packed velocity ordering)
int classify_partition(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                       const int32_t *rp32, const int32_t *cl32, std::vector<int64_t> &ranges, bool *general);
int upload_csr_general(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                       const int32_t *rp32, const int32_t *cl32, const double *val, const std::vector<int64_t> &ranges);
int redist_setup(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                 const int32_t *rp32, const int32_t *cl32, const double *val, const std::vector<int64_t> &ranges);
int redist_velocity_setup(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64, const int64_t *cl64,
                          const int32_t *rp32, const int32_t *cl32, const double *val, const std::vector<int64_t> &ranges);
void redist_release(pib_solver *s);
// redistribute.hip
int halo_exchange_general(pib_solver *s, double *x_owned, hipStream_t st);
int redist_tables(pib_solver *s);
int redist_rows_on_device(pib_solver *s, pib_solver *in, const RedistField &F, const int64_t N[3], int64_t n_local, int64_t row0, int64_t n_global,
                          const int64_t *rp64, const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val, int64_t W,
                          const std::vector<int64_t> &ranges, int32_t **h_rp, int32_t **h_cl, double **h_vl, int64_t *nnz_out);
int redist_forward(pib_solver *s, const double *v_box, double *v_nat, hipStream_t st);
int redist_backward(pib_solver *s, const double *v_nat, double *v_box, hipStream_t st);
// a solver on `other`'s device, rank and communicator (the second solver of a flow engine)
int create_sharing_comm(pib_solver **out, const char *name, const char *cfg_text, pib_solver *other);
// krylov.hip
int solve_cg(pib_solver *s, double *x, const double *b);
int solve_cg_sr(pib_solver *s, double *x, const double *b);  // KSPCGUseSingleReduction recurrences (cfg.cg_single_reduction)
int solve_bicgstab(pib_solver *s, double *x, const double *b);
int solve_chebyshev(pib_solver *s, double *x, const double *b);  // KSPCHEBYSHEV (oracle/csrc/oracle.c:orc_chebyshev)
// Gershgorin interval [lo, hi] of the (Jacobi-)preconditioned matrix, the maximum over the ranks; cached per matrix
int gershgorin_bounds(pib_solver *s, bool jacobi, double *lo, double *hi);
int ensure_work(pib_solver *s, int nvec);
int free_work(pib_solver *s);
// assemble.hip: take a copy of a CSR that already lives in HBM (single rank, 32-bit offsets) as the solver's matrix
int adopt_device_csr(pib_solver *s, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col, const double *val);
int after_set_matrix(pib_solver *s);
// dense.hip
int dense_setup(pib_solver *s);
void dense_release(pib_solver *s);
int solve_direct(pib_solver *s, double *x, const double *b);
// assemble.hip
void vel_stencil_release(pib_solver *s);
int vel_stencil_verify(pib_solver *s);  // the matrix-free product against the CSR SpMV on a pseudo-random vector; invalidates s->vel on a mismatch
int vel_stencil_apply(pib_solver *s, const double *x, double *y, bool guarded, hipStream_t q, const double *dinv = nullptr, double opc = 1.0,
                      int dot_mode = 0, const double *dot_other = nullptr, int dot_slot0 = 0);
int vel_stencil_apply_cheb(pib_solver *s, const double *x, double *pm, const double *b, const double *dinv, double opc, hipStream_t q, int slot0);
constexpr int VEL_DOT_PARTIALS = 64;  // partial sums per slot the fused sums of vel_stencil_apply leave in d_part
bool vel_stencil_fused_ok(const pib_solver *s);
int assemble_poisson(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], double dt, int nullspace);
// bn.hip: D * BN(order) * G through the reference's chain of sparse products; optionally hands out BNG (device arrays
// owned by the caller)
int assemble_poisson_bn(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double mn[3],
                        const double mx[3], const double a0[18], double dt, double coeff_nu, int order, int nullspace,
                        int32_t **bng_rowptr, int32_t **bng_col, double **bng_val, int64_t *bng_nnz, int32_t **bn_rowptr = nullptr,
                        int32_t **bn_col = nullptr, double **bn_val = nullptr, int64_t *bn_nnz = nullptr);
int device_spgemm(int64_t a_rows, int64_t a_cols, int64_t b_cols, const int32_t *arp, const int32_t *acol, const double *aval, int64_t a_nnz,
                  const int32_t *brp, const int32_t *bcol, const double *bval, int64_t b_nnz, int32_t **crp, int32_t **ccol, double **cval,
                  int64_t *c_nnz, hipStream_t q);
int assemble_velocity(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double mn[3],
                      const double mx[3], const double a0[18], double dt, double coeff_nu);
int upload_csr(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rowptr,
               const int64_t *col, const int32_t *rowptr32, const int32_t *col32, const double *val);
void slab_range(int64_t nplanes, int nranks, int rank, int64_t *b, int64_t *e);
void velocity_mesh_arrays(int dim, const int64_t n[3], const double *const w[3], const double mn[3], const double mx[3],
                          const int per[3], std::vector<double> hdl[3][3], std::vector<double> hco[3][3], int64_t fn[3][3],
                          const MeshWindow *win = nullptr);
int upload_vec(const std::vector<double> &h, double **d);
// structure.cpp
int detect_velocity_structure(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64,
                              const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val);
int detect_grid_structure(pib_solver *s, int64_t n_local, int64_t row0, int64_t n_global, const int64_t *rp64,
                          const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val);
void grid_detection_rows(const int64_t n[3], int64_t k0, int64_t k1, std::vector<int64_t> &local_rows);  // the local rows it reads
// gmg.hip
int gmg_verify(pib_solver *s);
int stencil_apply(pib_solver *s, double *x_owned, double *y, hipStream_t st);
int dense_apply_raw(const pib_solver *s, const double *b, double *y, hipStream_t q);
int dot_partials(pib_solver *s, const double *x, const double *y, double *part, bool guarded, hipStream_t q);
bool stencil_matmult_ok(const pib_solver *s);
int stencil_matmult(pib_solver *s, const double *x, double *y, double *dot_part, bool guarded, hipStream_t q);
int spmv_launch_blocks();
int gmg_apply(pib_solver *s, const double *r, double *z, hipStream_t st);
bool gmg_fused_update_ok(const pib_solver *s);
void drop_iteration_graph(pib_solver *s);  // krylov.hip: the captured iteration goes before the memory it points at
void gmg_release(pib_solver *s);
int grid_register(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double *const g[3],
                  int nullspace, double dt /* <= 0: recover from g */);
}  // namespace pib
