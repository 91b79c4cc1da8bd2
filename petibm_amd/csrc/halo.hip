// halo.cpp -- C1/C2: ghost-plane exchange, scalar all-reduce and level all-gather between ranks.
//
// Replaces the VecScatter inside PETSc's MPIAIJ MatMult and AmgX's
// communicator=MPI/MPI_DIRECT halo exchange.  One rank
// per GPU; z-slab (DMDA-style, nProc = (1,1,P)) decomposition, so in natural
// ordering a rank's ghosts are the `ghost_lo` entries just before its first
// row and the `ghost_hi` entries just after its last row: both are contiguous
// blocks of the neighbour's vector -- no pack kernel, one ncclSend/ncclRecv
// pair per neighbour grouped in a single RCCL group (point-to-point over the
// direct xGMI link between adjacent ranks; no ring involved).
//
// Three transports behind the same four operations:
//   * RCCL (production, default): ncclSend/ncclRecv, ncclAllReduce, ncclAllGather on the solver's stream.
//   * peer (production, one node; pib_comm_peer_id): one PROCESS per rank, the neighbours' vectors mapped through HIP IPC
//     (hipIpcOpenMemHandle: peer memory over xGMI) and pulled with device-to-device copies, ordered by interprocess HIP
//     events; the ranks meet in a POSIX shared-memory segment (handles, counts, a host barrier of atomics).  No RCCL kernel
//     on the critical path -- and, unlike RCCL, it accepts several ranks on one GPU, so the multi-PROCESS path (id
//     exchange, attach, exchanges, reductions) is tested end to end on the one-GPU box (tests/test_gpu_peer_transport.py).
//   * loopback (test only): P ranks = P host threads of ONE process sharing ONE GPU
//     (pib_comm_loopback_create).  RCCL refuses several ranks per device, and the test box has a single
//     GPU, so this is how the multi-rank algorithm (halo plans, distributed / replicated multigrid levels,
//     all-reduced recurrences) is exercised end to end through the C ABI.  Device-to-device copies ordered
//     with HIP events + a host barrier per collective.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "pib_internal.hpp"

namespace pib {

// ------------------------------------------------------------------ loopback / peer group
// What the ranks of the peer transport share (POSIX shared memory; plain data and lock-free atomics only)
constexpr int PEER_MAX_RANKS = 64;
struct alignas(64) PeerFlag {
    std::atomic<uint64_t> v;
};
struct PeerShm {
    std::atomic<uint32_t> state;  // 1: initialised by rank 0
    int nranks;
    std::atomic<int> attached;
    std::atomic<uint32_t> bar_count, bar_gen;
    std::atomic<int> failed;      // a rank gave up (timeout / error): everybody leaves the waits
    std::atomic<int> open_lock;   // one rank at a time inside hipIpcOpenMemHandle
    int64_t window_doubles;       // size of every rank's window (the same everywhere: rank 0's setting)
    hipIpcMemHandle_t staging;    // rank 0's [nranks][PIB_NRED] buffer of the scalar all-reduce
    hipIpcMemHandle_t window[PEER_MAX_RANKS];  // every rank's window: what it sends lies there for the others to fetch
    hipIpcMemHandle_t dwindow[PEER_MAX_RANKS]; // device-ordered flavour: every rank's RECEIVE window (peers store into it)
    // collective number `seq` (counted alike on every rank): ready[r] >= seq -- rank r's window holds its data for it;
    // done[r] >= seq -- rank r has fetched what it needs from the others' windows.  Set by host functions in stream order.
    PeerFlag ready[PEER_MAX_RANKS], done[PEER_MAX_RANKS];
    int64_t host_vals[PEER_MAX_RANKS][4];
};

// ---- device-ordered flavour of the peer transport: no host thread and no RCCL launch on the path of a collective.
// A rank STORES what it sends straight into the receiver's window (mapped through HIP IPC: peer memory over xGMI), a
// release flag per ordered pair of ranks follows the data in stream order (system-scope fence, then the store), the
// receiver's kernel spins on the flag, copies the data out of its own window and acknowledges; flags, acknowledgements
// and the payload of the scalar all-reduce live in a block of the shared segment that every process registers with HIP
// (host memory is fine-grained: a store from one GPU is visible to a polling load from another without a kernel
// boundary).  Collective numbers are counted ON THE DEVICE (ctr[]), so the kernels carry no per-call constants and a
// captured iteration graph replays them.  A window has two halves used in turn (collective k in half k & 1), laid out per
// collective; before a sender stores into a half it waits until the RECEIVER has finished collective k - 2 altogether
// (a progress counter per rank: other senders' data of that collective may lie where this one is about to write) -- which
// has practically always happened: nobody runs more than two collectives ahead of a rank it sends to.
struct PeerDevBlock {
    uint64_t ready[PEER_MAX_RANKS * PEER_MAX_RANKS];  // ready[d * P + s]: s's data of collective # has landed in d's window
    uint64_t done[PEER_MAX_RANKS];                    // done[d]: d has copied everything of collective # out of its window
    uint64_t rflag[PEER_MAX_RANKS];                   // all-reduce #: rank's values are in vals
    double vals[2 * PEER_MAX_RANKS * PIB_NRED];       // [parity][rank][slot]
    int failed;                                       // a spin timed out
};
constexpr size_t PEER_DEV_OFF = (sizeof(PeerShm) + 4095) / 4096 * 4096;
constexpr size_t PEER_DEV_BYTES = (sizeof(PeerDevBlock) + 4095) / 4096 * 4096;
constexpr int DMSG = 16;
struct DevMsgs {  // one launch: up to DMSG messages
    int n;
    int peer[DMSG];
    const double *src[DMSG];  // put: local source
    double *dst[DMSG];        // get: local destination
    int64_t off[DMSG];        // place inside the current half of the RECEIVER's window
    int64_t cnt[DMSG];
};
struct DevComm {
    int me, P;
    uint64_t *ready, *done, *rflag;
    double *vals;
    int *failed;
    uint64_t *ctr;       // device: [0] collectives done, [1] all-reduces done, [2] / [3] arrival counters of put / get
    double *const *wins; // device [P]: every rank's receive window as this process sees it
    double *win_local;
    int64_t half;        // doubles per half
    uint64_t timeout_ticks;
};

struct LoopbackGroup {
    int nranks = 0;
    // loopback flavour: threads of one process
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    // peer flavour: one process per rank
    PeerShm *shm = nullptr;
    int me = -1;
    double timeout_s = 600.0;
    uint64_t seq = 0;                 // collectives issued so far
    double *staging_local = nullptr;  // rank 0 owns the staging buffer
    // Windows.  A rank never maps another rank's vectors (mapping a 2 GB work-vector allocation in the middle of a run
    // did not come back from hipIpcOpenMemHandle on the test box; tools/ipc_probe*.hip could not reproduce it in
    // isolation): it copies what it sends into its own window -- one half for the previous rank, one for the next, the
    // whole of it for the all-to-all collectives -- and the others fetch from there.  The windows are mapped once, at
    // attach, when nothing else is going on.
    double *win_local = nullptr;
    std::vector<double *> win;        // every rank's window as this process sees it (win[me] = win_local)
    int64_t win_doubles = 0;
    // device-ordered flavour
    bool devord = false;
    DevComm dc{};
    double *dwin_local = nullptr;
    std::vector<double *> dwin;
    double **d_wins = nullptr;
    uint64_t *d_ctr = nullptr;
    PeerDevBlock *blk = nullptr;  // host view of the flag block (registered with HIP)
    // per-rank published state as this rank sees it
    std::vector<const double *> ptr;
    std::vector<int64_t> count;
    std::vector<const std::vector<std::pair<int64_t, int64_t>> *> segs_prev, segs_next;  // segmented plans of the ranks
    std::vector<hipEvent_t> ev_ready, ev_done;
    std::vector<int64_t> host_vals;  // [nranks][4]
    double *staging = nullptr;       // device, [nranks][PIB_NRED]

    int barrier()
    {
        if (shm == nullptr) {
            std::unique_lock<std::mutex> lk(mu);
            const uint64_t gen = generation;
            if (++arrived == nranks) {
                arrived = 0;
                ++generation;
                cv.notify_all();
            } else {
                cv.wait(lk, [&] { return generation != gen; });
            }
            return 0;
        }
        const uint32_t gen = shm->bar_gen.load(std::memory_order_acquire);
        if (shm->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
            shm->bar_count.store(0, std::memory_order_relaxed);
            shm->bar_gen.fetch_add(1, std::memory_order_release);
            return shm->failed.load() ? fail(PIB_ERR_LIB, "peer transport: another rank failed") : 0;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spins = 0; shm->bar_gen.load(std::memory_order_acquire) == gen; ++spins) {
            if (shm->failed.load(std::memory_order_relaxed)) return fail(PIB_ERR_LIB, "peer transport: another rank failed");
            if ((spins & 1023) == 1023) {
                sched_yield();
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                    shm->failed.store(1);
                    return fail(PIB_ERR_LIB, "peer transport: rank %d waited %.0f s for the other ranks (PIB_PEER_TIMEOUT_S)", me, timeout_s);
                }
            }
        }
        return 0;
    }

    // loopback flavour: make `p` (count entries, optionally the segment lists of a packed vector) visible to the other
    // rank threads
    int publish(int r, const double *p, int64_t cnt, const std::vector<std::pair<int64_t, int64_t>> *sp = nullptr,
                const std::vector<std::pair<int64_t, int64_t>> *sn = nullptr)
    {
        ptr[(size_t)r] = p;
        count[(size_t)r] = cnt;
        segs_prev[(size_t)r] = sp;
        segs_next[(size_t)r] = sn;
        return 0;
    }

    bool trace = std::getenv("PIB_PEER_TRACE") != nullptr;
    void note(const char *what, uint64_t k) const
    {
        if (trace) {
            std::fprintf(stderr, "[peer %d] %s %llu\n", me, what, (unsigned long long)k);
            std::fflush(stderr);
        }
    }
    // The collectives of a rank are issued one after the other by its host thread, but on whatever stream the caller works
    // on (a solver's, the engine's, the communication stream).  They are chained on the GPU as well: a collective's copies
    // start after the previous collective's fetches have finished, so `done` is never raised for collective k + 1 while
    // a fetch of collective k is still reading a neighbour's window.
    hipEvent_t chain = nullptr;
    hipStream_t chain_stream = nullptr;
    bool chained = false;
    // ranks that still have to finish fetching what this rank laid out for the previous collective: waited for at the
    // START of the next one (the window / the staging rows are rewritten only then; the host thread meanwhile goes on
    // enqueuing the kernels between the two collectives)
    std::vector<int> owed_by;
    uint64_t owed_seq = 0;
    bool capturing = false;  // an iteration is being captured into a hipGraph: only the device-ordered collectives may run
    int begin(hipStream_t st, uint64_t *out)
    {
        if (capturing) return fail(PIB_ERR_SUP, "peer transport: a host-ordered collective (a message larger than half a window?) inside a captured iteration");
        for (int q : owed_by) PIB_CHK(await(shm->done[q], owed_seq, "previous collective fetched", q));
        owed_by.clear();
        if (chained && chain_stream != st) PIB_HIP(hipStreamWaitEvent(st, chain, 0));
        *out = ++seq;
        return 0;
    }
    void owe(uint64_t k, int q)
    {
        owed_seq = k;
        owed_by.push_back(q);
    }
    int finish(hipStream_t st)
    {
        if (chain == nullptr) PIB_HIP(hipEventCreateWithFlags(&chain, hipEventDisableTiming));
        PIB_HIP(hipEventRecord(chain, st));
        chain_stream = st;
        chained = true;
        return 0;
    }

    // ---- peer flavour: ordering between the ranks.  HIP's interprocess events stop after 32 records on this runtime
    // (tools/ipc_probe2.hip), so completion travels through the shared segment: a host function enqueued behind the
    // producing work raises the rank's flag, the consumer's host waits for it before it enqueues the copies.  The GPU
    // streams never wait for another rank; the host threads do, at every collective.
    struct HostSet {
        std::atomic<uint64_t> *flag;
        uint64_t val;
        bool trace;
    };
    static void host_set(void *p)
    {
        HostSet *h = static_cast<HostSet *>(p);
        // never backwards: collectives on different streams may run their host functions out of order
        uint64_t cur = h->flag->load(std::memory_order_relaxed);
        while (cur < h->val && !h->flag->compare_exchange_weak(cur, h->val, std::memory_order_release, std::memory_order_relaxed)) {}
        if (h->trace) {
            std::fprintf(stderr, "[peer] flag raised to %llu\n", (unsigned long long)h->val);
            std::fflush(stderr);
        }
        delete h;
    }
    int raise(hipStream_t st, PeerFlag &f, uint64_t val)
    {
        HostSet *h = new HostSet{&f.v, val, trace};
        const hipError_t e = hipLaunchHostFunc(st, host_set, h);
        if (e != hipSuccess) {
            delete h;
            return fail(PIB_ERR_LIB, "peer transport: hipLaunchHostFunc: %s", hipGetErrorString(e));
        }
        return 0;
    }
    int await(const PeerFlag &f, uint64_t val, const char *what, int q)
    {
        note(what, val);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spins = 0; f.v.load(std::memory_order_acquire) < val; ++spins) {
            if (shm->failed.load(std::memory_order_relaxed)) return fail(PIB_ERR_LIB, "peer transport: another rank failed");
            if ((spins & 255) == 255) {
                sched_yield();
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                    shm->failed.store(1);
                    return fail(PIB_ERR_LIB, "peer transport: rank %d waited %.0f s for rank %d (%s, collective %llu)", me, timeout_s, q,
                                what, (unsigned long long)val);
                }
            }
        }
        return 0;
    }
};

// A rank that leaves a peer collective with an error of its own (a HIP call, a window too small for ITS message) must not
// leave the others spinning until their timeout: every error exit marks the job failed, which all the waits look at.
struct PeerFailGuard {
    LoopbackGroup *g;
    bool ok = false;
    explicit PeerFailGuard(LoopbackGroup *grp) : g(grp) {}
    ~PeerFailGuard()
    {
        if (!ok && g != nullptr && g->shm != nullptr) g->shm->failed.store(1);
    }
    int done()
    {
        ok = true;
        return 0;
    }
};

static const char LOOP_MAGIC[8] = {'P', 'I', 'B', 'L', 'O', 'O', 'P', '1'};
static const char PEER_MAGIC[8] = {'P', 'I', 'B', 'P', 'E', 'E', 'R', '1'};

__global__ void k_lb_sum(double *dst, const double *staging, int nranks, int count)
{
    const int i = threadIdx.x;
    if (i < count) {
        double s = 0.0;
        for (int r = 0; r < nranks; ++r) s += staging[r * PIB_NRED + i];
        dst[i] = s;
    }
}

// ---- device-ordered peer transport: kernels
__device__ inline uint64_t ld_sys(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void st_sys(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// spin until *p >= want; a timeout marks the job failed (every later wait returns at once: the host reports it)
__device__ inline void spin_until(const uint64_t *p, uint64_t want, const DevComm &C)
{
    if (ld_sys(p) >= want) return;
    const uint64_t t0 = wall_clock64();
    for (uint32_t it = 0;; ++it) {
        if (ld_sys(p) >= want) return;
        if ((it & 63) == 63) {
            if (__hip_atomic_load(C.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
            if (wall_clock64() - t0 > C.timeout_ticks) {
                __hip_atomic_store(C.failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
__device__ inline void copy_part(double *__restrict__ dst, const double *__restrict__ src, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const int64_t n2 = n >> 1;
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (int64_t i = i0; i < n2; i += stride) d2[i] = s2[i];
        if ((n & 1) && i0 == 0) dst[n - 1] = src[n - 1];
    } else
        for (int64_t i = i0; i < n; i += stride) dst[i] = src[i];
}
// store this rank's messages into the receivers' windows, then raise the pair flags; last & 1: the collective's last put
// launch, last & 2: ... and this rank fetches nothing in this collective (it has then finished it)
__global__ __launch_bounds__(256) void k_dput(DevMsgs M, DevComm C, int last)
{
    const uint64_t seq = *(volatile uint64_t *)&C.ctr[0] + 1;
    const int h = (int)(seq & 1);
    if ((int)threadIdx.x < M.n && seq > 2)  // the receiver is through with the collective that used this half before
        spin_until(&C.done[M.peer[threadIdx.x]], seq - 2, C);
    __syncthreads();
    for (int m = 0; m < M.n; ++m) copy_part(C.wins[M.peer[m]] + (int64_t)h * C.half + M.off[m], M.src[m], M.cnt[m]);
    __threadfence_system();  // the stores have reached the peers' memory before any flag goes up
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd((unsigned long long *)&C.ctr[2], 1ull) == (unsigned long long)gridDim.x - 1) {
        C.ctr[2] = 0;
        for (int m = 0; m < M.n; ++m) st_sys(&C.ready[(size_t)M.peer[m] * C.P + C.me], seq);
        if (last & 1) C.ctr[0] = seq;
        if (last & 2) st_sys(&C.done[C.me], seq);
    }
}
// wait for the senders' flags, copy their messages out of this rank's window; the collective's last get launch reports
// the rank's progress
__global__ __launch_bounds__(256) void k_dget(DevMsgs M, DevComm C, int last)
{
    const uint64_t seq = *(volatile uint64_t *)&C.ctr[0];
    const int h = (int)(seq & 1);
    if ((int)threadIdx.x < M.n) spin_until(&C.ready[(size_t)C.me * C.P + M.peer[threadIdx.x]], seq, C);
    __syncthreads();
    __threadfence_system();  // acquire: nothing cached of the window from before the flags went up
    for (int m = 0; m < M.n; ++m) copy_part(M.dst[m], C.win_local + (int64_t)h * C.half + M.off[m], M.cnt[m]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd((unsigned long long *)&C.ctr[3], 1ull) == (unsigned long long)gridDim.x - 1) {
        C.ctr[3] = 0;
        if (last) st_sys(&C.done[C.me], seq);
    }
}
// in-place sum over ranks of count <= PIB_NRED doubles: the values travel through the flag block, every rank sums them
// in rank order (the same bits everywhere, the order of k_lb_sum)
__global__ __launch_bounds__(64) void k_dallreduce(double *dev, int count, DevComm C)
{
    const uint64_t seq = *(volatile uint64_t *)&C.ctr[1] + 1;
    const int h = (int)(seq & 1), t = (int)threadIdx.x;
    double *mine = C.vals + ((size_t)h * C.P + C.me) * PIB_NRED;
    if (t < count) __hip_atomic_store(&mine[t], dev[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (t == 0) st_sys(&C.rflag[C.me], seq);
    if (t < C.P) spin_until(&C.rflag[t], seq, C);
    __threadfence_system();
    __syncthreads();
    if (t < count) {
        double sum = 0.0;
        for (int q = 0; q < C.P; ++q) sum += __hip_atomic_load(&C.vals[((size_t)h * C.P + q) * PIB_NRED + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dev[t] = sum;
    }
    if (t == 0) C.ctr[1] = seq;
}

struct PutMsg {
    int peer;
    const double *src;
    int64_t off, cnt;
};
struct GetMsg {
    int peer;
    double *dst;
    int64_t off, cnt;
};
// one device-ordered collective on stream st: this rank's stores, then its fetches (every rank issues one per collective,
// with or without messages of its own: the collective number is counted by the put launch)
static int dev_collective(pib_solver *s, hipStream_t st, const std::vector<PutMsg> &puts, const std::vector<GetMsg> &gets)
{
    LoopbackGroup *g = s->comm.loop;
    if (g->blk->failed || g->shm->failed.load(std::memory_order_relaxed)) {
        g->shm->failed.store(1);
        return fail(PIB_ERR_LIB, "peer transport (device-ordered): a rank timed out waiting for a flag");
    }
    if (g->chained && g->chain_stream != st) PIB_HIP(hipStreamWaitEvent(st, g->chain, 0));
    auto blocks = [](int64_t total) { return (int)std::max<int64_t>(1, std::min<int64_t>(128, (total + 4095) / 4096)); };
    int ngets = 0;
    for (const auto &m : gets) ngets += m.cnt > 0 ? 1 : 0;
    size_t a = 0;
    do {
        DevMsgs M{};
        int64_t total = 0;
        for (; a < puts.size() && M.n < DMSG; ++a) {
            if (puts[a].cnt <= 0) continue;
            M.peer[M.n] = puts[a].peer;
            M.src[M.n] = puts[a].src;
            M.off[M.n] = puts[a].off;
            M.cnt[M.n] = puts[a].cnt;
            total += puts[a].cnt;
            ++M.n;
        }
        hipLaunchKernelGGL(k_dput, dim3(blocks(total)), dim3(256), 0, st, M, g->dc, a >= puts.size() ? (ngets > 0 ? 1 : 3) : 0);
    } while (a < puts.size());
    a = 0;
    int got = 0;
    while (a < gets.size()) {
        DevMsgs M{};
        int64_t total = 0;
        for (; a < gets.size() && M.n < DMSG; ++a) {
            if (gets[a].cnt <= 0) continue;
            M.peer[M.n] = gets[a].peer;
            M.dst[M.n] = gets[a].dst;
            M.off[M.n] = gets[a].off;
            M.cnt[M.n] = gets[a].cnt;
            total += gets[a].cnt;
            ++M.n;
        }
        got += M.n;
        if (M.n > 0) hipLaunchKernelGGL(k_dget, dim3(blocks(total)), dim3(256), 0, st, M, g->dc, got == ngets ? 1 : 0);
    }
    PIB_HIP(hipGetLastError());
    return g->finish(st);
}

// ---- peer transport: the ranks (processes) meet in the shared-memory segment named in the id
static void peer_destroy(LoopbackGroup *g)
{
    if (g == nullptr) return;
    // the others may still be fetching what this rank laid out last: the window must outlive that
    if (g->shm != nullptr && !g->shm->failed.load())
        for (int q : g->owed_by) (void)g->await(g->shm->done[q], g->owed_seq, "last collective fetched", q);
    g->owed_by.clear();
    for (int q = 0; q < (int)g->win.size(); ++q)
        if (q != g->me && g->win[(size_t)q]) (void)hipIpcCloseMemHandle(g->win[(size_t)q]);
    if (g->win_local) (void)hipFree(g->win_local);
    if (g->devord) {
        // the peers may still be storing into / reading flags of this rank: leave together
        (void)hipDeviceSynchronize();
        if (g->shm != nullptr && !g->shm->failed.load()) (void)g->barrier();
        for (int q = 0; q < (int)g->dwin.size(); ++q)
            if (q != g->me && g->dwin[(size_t)q]) (void)hipIpcCloseMemHandle(g->dwin[(size_t)q]);
        if (g->dwin_local) (void)hipFree(g->dwin_local);
        if (g->d_wins) (void)hipFree(g->d_wins);
        if (g->d_ctr) (void)hipFree(g->d_ctr);
        if (g->blk) (void)hipHostUnregister(g->blk);
    }
    if (g->chain) (void)hipEventDestroy(g->chain);
    if (g->staging_local) (void)hipFree(g->staging_local);
    else if (g->staging) (void)hipIpcCloseMemHandle(g->staging);
    if (g->shm) (void)munmap(g->shm, PEER_DEV_OFF + PEER_DEV_BYTES);
    delete g;
}
static int peer_attach(pib_solver *s, int rank, int nranks, const char *name)
{
    if (nranks > PEER_MAX_RANKS) return fail(PIB_ERR_SUP, "peer transport: at most %d ranks", PEER_MAX_RANKS);
    LoopbackGroup *g = new LoopbackGroup();
    g->nranks = nranks;
    g->me = rank;
    if (const char *t = std::getenv("PIB_PEER_TIMEOUT_S")) g->timeout_s = std::max(1.0, std::atof(t));
    g->ptr.assign((size_t)nranks, nullptr);
    g->count.assign((size_t)nranks, 0);
    g->segs_prev.assign((size_t)nranks, nullptr);
    g->segs_next.assign((size_t)nranks, nullptr);
    g->win.assign((size_t)nranks, nullptr);
    g->host_vals.assign(4 * (size_t)nranks, 0);
    g->ev_ready.assign((size_t)nranks, nullptr);
    g->ev_done.assign((size_t)nranks, nullptr);
    auto bail = [&](int err) {
        if (g->shm) g->shm->failed.store(1);
        if (rank == 0) (void)shm_unlink(name);  // nobody will: the name would outlive the run
        peer_destroy(g);
        return err;
    };
    // rank 0 creates the segment, the others wait for it
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > g->timeout_s; };
    int fd = -1;
    if (rank == 0) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)(PEER_DEV_OFF + PEER_DEV_BYTES)) != 0) {
            if (fd >= 0) (void)close(fd);
            return bail(fail(PIB_ERR_LIB, "peer transport: cannot create the shared segment %s", name));
        }
    } else {
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= PEER_DEV_OFF + PEER_DEV_BYTES) break;
            if (fd >= 0) (void)close(fd);
            fd = -1;
            if (late()) return bail(fail(PIB_ERR_LIB, "peer transport: rank %d never saw rank 0's segment %s", rank, name));
            usleep(1000);
        }
    }
    void *m = mmap(nullptr, PEER_DEV_OFF + PEER_DEV_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    (void)close(fd);
    if (m == MAP_FAILED) return bail(fail(PIB_ERR_LIB, "peer transport: mmap of the shared segment failed"));
    g->shm = static_cast<PeerShm *>(m);
    if (rank == 0) {
        std::memset(m, 0, PEER_DEV_OFF + PEER_DEV_BYTES);  // a fresh segment is zero-filled already; the atomics start at 0
        g->shm->nranks = nranks;
        g->shm->state.store(1, std::memory_order_release);
    } else {
        while (g->shm->state.load(std::memory_order_acquire) != 1) {
            if (late()) return bail(fail(PIB_ERR_LIB, "peer transport: rank 0 never initialised the segment"));
            usleep(100);
        }
        if (g->shm->nranks != nranks) return bail(fail(PIB_ERR_ARG_WRONG, "peer transport: the ranks disagree about nranks"));
    }
    g->shm->attached.fetch_add(1);
    // this rank's window and (rank 0) the staging buffer of the scalar all-reduce
    if (rank == 0) {
        double mb = 128.0;
        if (const char *t = std::getenv("PIB_PEER_WINDOW_MB")) mb = std::max(0.0625, std::atof(t));  // (tests shrink it to force the chunked paths)
        g->shm->window_doubles = ((int64_t)(mb * 1048576.0 / 8.0) / 64) * 64;
    }
    int err0 = g->barrier();
    if (err0) return bail(err0);
    g->win_doubles = g->shm->window_doubles;
    hipError_t e = hipMalloc(&g->win_local, sizeof(double) * (size_t)g->win_doubles);
    if (e == hipSuccess) e = hipMemset(g->win_local, 0, sizeof(double) * (size_t)g->win_doubles);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&g->shm->window[rank], g->win_local);
    g->win[(size_t)rank] = g->win_local;
    if (e == hipSuccess && rank == 0) {
        e = hipMalloc(&g->staging_local, sizeof(double) * PIB_NRED * (size_t)nranks);
        if (e == hipSuccess) e = hipMemset(g->staging_local, 0, sizeof(double) * PIB_NRED * (size_t)nranks);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipIpcGetMemHandle(&g->shm->staging, g->staging_local);
        g->staging = g->staging_local;
    }
    if (e != hipSuccess) return bail(fail(PIB_ERR_LIB, "peer transport: staging buffer: %s", hipGetErrorString(e)));
    int err = g->barrier();
    if (err) return bail(err);
    // map the others' windows and the staging buffer, one rank at a time
    {
        const auto t1 = std::chrono::steady_clock::now();
        for (int expect = 0; !g->shm->open_lock.compare_exchange_weak(expect, 1, std::memory_order_acquire); expect = 0) {
            sched_yield();
            if (g->shm->failed.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() > g->timeout_s)
                return bail(fail(PIB_ERR_LIB, "peer transport: rank %d waited %.0f s for its turn to map the windows", rank, g->timeout_s));
        }
        for (int q = 0; q < nranks && e == hipSuccess; ++q) {
            if (q == rank) continue;
            void *base = nullptr;
            e = hipIpcOpenMemHandle(&base, g->shm->window[q], hipIpcMemLazyEnablePeerAccess);
            g->win[(size_t)q] = static_cast<double *>(base);
        }
        if (e == hipSuccess && rank != 0) {
            void *st_base = nullptr;
            e = hipIpcOpenMemHandle(&st_base, g->shm->staging, hipIpcMemLazyEnablePeerAccess);
            g->staging = static_cast<double *>(st_base);
        }
        g->shm->open_lock.store(0, std::memory_order_release);
        if (e != hipSuccess) return bail(fail(PIB_ERR_LIB, "peer transport: mapping the windows: %s", hipGetErrorString(e)));
    }
    err = g->barrier();
    if (err) return bail(err);
    g->devord = std::strncmp(name, "/pib_peerD", 10) == 0;
    if (g->devord) {
        // a second window per rank, written by the peers; the flag block of the segment registered with HIP
        g->dwin.assign((size_t)nranks, nullptr);
        e = hipMalloc(&g->dwin_local, sizeof(double) * (size_t)g->win_doubles);
        if (e == hipSuccess) e = hipMemset(g->dwin_local, 0, sizeof(double) * (size_t)g->win_doubles);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipIpcGetMemHandle(&g->shm->dwindow[rank], g->dwin_local);
        g->dwin[(size_t)rank] = g->dwin_local;
        if (e != hipSuccess) return bail(fail(PIB_ERR_LIB, "peer transport: receive window: %s", hipGetErrorString(e)));
        err = g->barrier();
        if (err) return bail(err);
        for (int expect = 0; !g->shm->open_lock.compare_exchange_weak(expect, 1, std::memory_order_acquire); expect = 0) {
            sched_yield();
            if (g->shm->failed.load() || late()) return bail(fail(PIB_ERR_LIB, "peer transport: rank %d waited for its turn to map the receive windows", rank));
        }
        for (int q = 0; q < nranks && e == hipSuccess; ++q) {
            if (q == rank) continue;
            void *base = nullptr;
            e = hipIpcOpenMemHandle(&base, g->shm->dwindow[q], hipIpcMemLazyEnablePeerAccess);
            g->dwin[(size_t)q] = static_cast<double *>(base);
        }
        g->shm->open_lock.store(0, std::memory_order_release);
        if (e != hipSuccess) return bail(fail(PIB_ERR_LIB, "peer transport: mapping the receive windows: %s", hipGetErrorString(e)));
        g->blk = reinterpret_cast<PeerDevBlock *>(static_cast<char *>(m) + PEER_DEV_OFF);
        void *dblk = nullptr;
        e = hipHostRegister(g->blk, PEER_DEV_BYTES, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&dblk, g->blk, 0);
        if (e == hipSuccess) e = hipMalloc(&g->d_wins, sizeof(double *) * (size_t)nranks);
        if (e == hipSuccess) e = hipMemcpy(g->d_wins, g->dwin.data(), sizeof(double *) * (size_t)nranks, hipMemcpyHostToDevice);
        const size_t nctr = 8;
        if (e == hipSuccess) e = hipMalloc(&g->d_ctr, sizeof(uint64_t) * nctr);
        if (e == hipSuccess) e = hipMemset(g->d_ctr, 0, sizeof(uint64_t) * nctr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            g->blk = nullptr;
            return bail(fail(PIB_ERR_LIB, "peer transport: registering the flag block: %s", hipGetErrorString(e)));
        }
        PeerDevBlock *db = static_cast<PeerDevBlock *>(dblk);
        DevComm &C = g->dc;
        C.me = rank;
        C.P = nranks;
        C.ready = db->ready;
        C.done = db->done;
        C.rflag = db->rflag;
        C.vals = db->vals;
        C.failed = &db->failed;
        C.ctr = g->d_ctr;
        C.wins = g->d_wins;
        C.win_local = g->dwin_local;
        C.half = (g->win_doubles / 2 / 64) * 64;
        C.timeout_ticks = (uint64_t)(std::min(g->timeout_s, 60.0) * 1.0e8);  // wall_clock64: 100 MHz
        err = g->barrier();
        if (err) return bail(err);
    }
    if (rank == 0) (void)shm_unlink(name);  // everybody holds a mapping: the name can go
    s->comm.loop = g;
    s->comm.peer = true;
    return 0;
}

static void adopt_nccl(pib_solver *s);
int comm_init(pib_solver *s, int rank, int nranks, const void *uid)
{
    s->comm.rank = rank;
    s->comm.nranks = nranks;
    s->comm.comm = nullptr;
    s->comm.loop = nullptr;
    if (nranks <= 1) return 0;
    if (uid == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_create: nranks > 1 needs the id from pib_comm_unique_id");
    if (std::memcmp(uid, LOOP_MAGIC, 8) == 0) {
        LoopbackGroup *g = nullptr;
        std::memcpy(&g, (const char *)uid + 8, sizeof(g));
        if (g == nullptr || g->nranks != nranks) return fail(PIB_ERR_ARG_WRONG, "loopback group does not match nranks");
        s->comm.loop = g;
        return 0;
    }
    if (std::memcmp(uid, PEER_MAGIC, 8) == 0) return peer_attach(s, rank, nranks, (const char *)uid + 8);
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) <= PIB_UID_BYTES, "unique id does not fit");
    std::memcpy(&id, uid, sizeof(id));
    PIB_NCCL(ncclCommInitRank(&s->comm.comm, nranks, id, rank));
    adopt_nccl(s);
    return 0;
}

// the communicator just created in s->comm.comm gets its shared holder (destroyed by the last solver that lets go of it)
static void adopt_nccl(pib_solver *s)
{
    if (s->comm.shared) {  // a second holder's deleter would destroy the communicator the first one still serves
        std::fprintf(stderr, "petibm_amd: adopt_nccl called twice for one communicator\n");
        std::abort();
    }
    s->comm.shared = std::shared_ptr<CommShared>(new CommShared{s->comm.comm, false}, [](CommShared *c) {
        if (c->comm && !c->aborted) (void)ncclCommDestroy(c->comm);
        delete c;
    });
}

void comm_abort(pib_solver *s)
{
    if (s->comm.shared) {
        if (!s->comm.shared->aborted && s->comm.shared->comm) (void)ncclCommAbort(s->comm.shared->comm);
        s->comm.shared->aborted = true;
        s->comm.shared->comm = nullptr;
    } else if (s->comm.comm)
        (void)ncclCommAbort(s->comm.comm);
    s->comm.comm = nullptr;
}

int comm_usable(pib_solver *s)
{
    if (s->comm.shared && s->comm.shared->aborted) {
        s->comm.comm = nullptr;  // (an alias of the aborted handle)
        return fail(PIB_ERR_LIB, "solver %s: its RCCL communicator was aborted (a collective did not complete within PIB_RCCL_TIMEOUT_S, in this "
                                 "solver or in one that shares the communicator): destroy the solvers of the group and create them again",
                    s->name.c_str());
    }
    return 0;
}

void comm_release(pib_solver *s)
{
    s->comm.shared.reset();  // (the last holder destroys the communicator, unless it was aborted)
    s->comm.comm = nullptr;
    if (s->comm.loop && s->comm.peer && !s->comm.borrowed) peer_destroy(s->comm.loop);  // this solver attached it
    s->comm.loop = nullptr;  // a loopback group is owned by whoever created it
}

// host-side all-gather of 4 int64 per rank (setup only)
static int allgather_host4(pib_solver *s, const int64_t mine[4], std::vector<int64_t> &all)
{
    const int P = s->comm.nranks, r = s->comm.rank;
    all.assign(4 * (size_t)P, 0);
    if (s->comm.loop) {
        LoopbackGroup *g = s->comm.loop;
        if (g->shm) {
            for (int k = 0; k < 4; ++k) g->shm->host_vals[r][k] = mine[k];
            PIB_CHK(g->barrier());
            for (int q = 0; q < P; ++q)
                for (int k = 0; k < 4; ++k) all[4 * (size_t)q + k] = g->shm->host_vals[q][k];
            PIB_CHK(g->barrier());
            return 0;
        }
        for (int k = 0; k < 4; ++k) g->host_vals[4 * (size_t)r + k] = mine[k];
        PIB_CHK(g->barrier());
        all = g->host_vals;
        PIB_CHK(g->barrier());
        return 0;
    }
    int64_t *d_all = nullptr;
    PIB_HIP(hipMalloc(&d_all, sizeof(int64_t) * 4 * (size_t)P));
    PIB_HIP(hipMemcpyAsync(d_all + 4 * r, mine, sizeof(int64_t) * 4, hipMemcpyHostToDevice, s->stream));
    PIB_NCCL(ncclAllGather(d_all + 4 * r, d_all, 4, ncclInt64, s->comm.comm, s->stream));
    PIB_HIP(hipMemcpyAsync(all.data(), d_all, sizeof(int64_t) * 4 * (size_t)P, hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipFree(d_all));
    return 0;
}

// host-side all-gather of `mine.size()` doubles per rank (same count everywhere; setup only)
int comm_allgather_host(pib_solver *s, const std::vector<double> &mine, std::vector<double> &all)
{
    const int P = s->comm.nranks, r = s->comm.rank;
    const size_t L = mine.size();
    if (!s->comm.active()) {
        all = mine;
        return 0;
    }
    all.assign(L * (size_t)P, 0.0);
    double *d_all = nullptr;
    PIB_HIP(hipMalloc(&d_all, sizeof(double) * L * (size_t)P));
    PIB_HIP(hipMemcpyAsync(d_all + L * (size_t)r, mine.data(), sizeof(double) * L, hipMemcpyHostToDevice, s->stream));
    std::vector<int64_t> cnt((size_t)P, (int64_t)L), off((size_t)P);
    for (int q = 0; q < P; ++q) off[(size_t)q] = (int64_t)(L * (size_t)q);
    PIB_CHK(comm_allgatherv(s, d_all + L * (size_t)r, d_all, cnt, off, s->stream));
    PIB_HIP(hipMemcpyAsync(all.data(), d_all, sizeof(double) * L * (size_t)P, hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    if (s->comm.loop) PIB_CHK(s->comm.loop->barrier());  // nobody frees while a peer still reads
    PIB_HIP(hipFree(d_all));
    return 0;
}

// After the matrix is known: tell the neighbours how many entries this rank
// needs from them (all-gather of {n_local, ghost_lo, ghost_hi, row0}).
int comm_setup_halo(pib_solver *s)
{
    DeviceCsr &A = s->A;
    A.send_prev = A.send_next = 0;
    if (s->comm.nranks <= 1) {
        if (A.ghost_lo != 0 || A.ghost_hi != 0)
            return fail(PIB_ERR_ARG_OUTOFRANGE, "single-rank matrix has columns outside [0, n)");
        return 0;
    }
    if (A.segmented || A.general) return 0;  // the assembly / upload that chose such a plan has set every list
    const int P = s->comm.nranks, r = s->comm.rank;
    const int64_t mine[4] = {A.n, A.ghost_lo, A.ghost_hi, A.row0};
    std::vector<int64_t> all;
    PIB_CHK(allgather_host4(s, mine, all));
    int err = 0;
    // (the longest message of this plan on ANY rank: what the peer transport chooses its protocol from, group-wide)
    A.halo_group_longest = 0;
    for (int q = 0; q < P; ++q) A.halo_group_longest = std::max(A.halo_group_longest, std::max(all[4 * (size_t)q + 1], all[4 * (size_t)q + 2]));
    // the setMatrix route: rank 0 reaching below its first row (or the last rank beyond its last) means the slab axis is
    // periodic -- upload_csr has placed those columns next to the rank's rows
    if (all[4 * 0 + 1] > 0 || all[4 * (size_t)(P - 1) + 2] > 0) s->comm.ring = true;
    if (s->comm.ring) {  // periodic slab axis: every rank has both neighbours
        const size_t pv = (size_t)((r + P - 1) % P), nx = (size_t)((r + 1) % P);
        if (A.ghost_lo > all[4 * pv] || A.ghost_hi > all[4 * nx]) return fail(PIB_ERR_SUP, "halo of rank %d reaches beyond its neighbour", r);
        A.send_prev = all[4 * pv + 2];
        A.send_next = all[4 * nx + 1];
        if (A.send_prev > A.n || A.send_next > A.n) return fail(PIB_ERR_SUP, "a neighbour's halo is wider than this rank's slab");
        return 0;
    }
    if (r > 0) {
        if (A.ghost_lo > all[4 * (size_t)(r - 1)])
            err = fail(PIB_ERR_SUP, "halo of rank %d reaches beyond its neighbour (needs %lld entries, neighbour owns %lld)", r,
                       (long long)A.ghost_lo, (long long)all[4 * (size_t)(r - 1)]);
        A.send_prev = all[4 * (size_t)(r - 1) + 2];  // their ghost_hi
    } else if (A.ghost_lo != 0) {
        err = fail(PIB_ERR_ARG_OUTOFRANGE, "rank 0 has columns below its first row");
    }
    if (r < P - 1) {
        if (A.ghost_hi > all[4 * (size_t)(r + 1)]) err = fail(PIB_ERR_SUP, "halo of rank %d reaches beyond its neighbour", r);
        A.send_next = all[4 * (size_t)(r + 1) + 1];  // their ghost_lo
    } else if (A.ghost_hi != 0) {
        err = fail(PIB_ERR_ARG_OUTOFRANGE, "last rank has columns above its last row");
    }
    if (!err && (A.send_prev > A.n || A.send_next > A.n)) err = fail(PIB_ERR_SUP, "a neighbour's halo is wider than this rank's slab");
    return err;
}

// loopback: publish -> barrier -> pull from the neighbours -> barrier -> order later writes after their reads
// peer: what the neighbours need goes into this rank's window (first half: for the previous rank, second half: for the
// next one), `ready` is raised behind those copies, the ghosts are fetched from the neighbours' windows once theirs is up,
// `done` is raised behind the fetches; the next collective of this rank starts by waiting for the neighbours' `done`
// (the window is free again then)
static int peer_window_exchange(pib_solver *s, hipStream_t st, int pv, int nx, bool has_pv, bool has_nx,
                                const std::vector<std::pair<const double *, int64_t>> &to_prev,
                                const std::vector<std::pair<const double *, int64_t>> &to_next, double *ghost_lo, int64_t lo,
                                double *ghost_hi, int64_t hi, const char *what, int64_t group_longest = 0)
{
    LoopbackGroup *g = s->comm.loop;
    PeerFailGuard guard(g);
    const int r = s->comm.rank;
    int64_t np = 0, nn = 0;
    for (const auto &m : to_prev) np += m.second;
    for (const auto &m : to_next) nn += m.second;
    // The two protocols below do not interoperate (puts into the neighbour's window against gets from the own one), so the choice
    // must be the GROUP's: `group_longest` is the longest message any rank of the group has in this exchange, agreed when the plan
    // was built (comm_setup_halo: DeviceCsr::halo_group_longest).  The multigrid's plane exchanges pass 0: their messages are
    // depth x plane on every rank.  (Until round 5 a rank decided from its own messages: two neighbours either side of the
    // quarter-window bound -- non-uniform ghost ranges -- would have picked different paths and hung.)
    const int64_t longest = std::max(std::max(std::max(np, nn), std::max(lo, hi)), group_longest);
    // device-ordered: a message per neighbour in one receive half (a quarter window each).  Longer messages -- up to half a
    // send window, the host-ordered limit -- take the host-ordered path below, as the general exchange and the all-gather do
    // (inside a captured iteration that path refuses: begin()).
    const bool devord = g->devord && longest <= g->dc.half / 2;
    const int64_t half = devord ? g->dc.half / 2 : g->win_doubles / 2;
    if (longest > half)
        return fail(PIB_ERR_SUP, "peer transport: a message of %lld entries does not fit half a window (%lld): raise PIB_PEER_WINDOW_MB",
                    (long long)longest, (long long)half);
    if (devord) {
        // a rank's receive half: [what the previous rank sends | what the next rank sends]
        std::vector<PutMsg> puts;
        std::vector<GetMsg> gets;
        int64_t o = half;  // this rank is the previous rank's NEXT
        if (has_pv)
            for (const auto &m : to_prev) {
                puts.push_back({pv, m.first, o, m.second});
                o += m.second;
            }
        o = 0;
        if (has_nx)
            for (const auto &m : to_next) {
                puts.push_back({nx, m.first, o, m.second});
                o += m.second;
            }
        if (has_pv && lo > 0) gets.push_back({pv, ghost_lo, 0, lo});
        if (has_nx && hi > 0) gets.push_back({nx, ghost_hi, half, hi});
        PIB_CHK(dev_collective(s, st, puts, gets));
        return guard.done();
    }
    uint64_t seq = 0;
    PIB_CHK(g->begin(st, &seq));
    double *dst = g->win_local;
    if (has_pv)
        for (const auto &m : to_prev) {
            if (m.second > 0) PIB_HIP(hipMemcpyAsync(dst, m.first, sizeof(double) * (size_t)m.second, hipMemcpyDeviceToDevice, st));
            dst += m.second;
        }
    dst = g->win_local + half;
    if (has_nx)
        for (const auto &m : to_next) {
            if (m.second > 0) PIB_HIP(hipMemcpyAsync(dst, m.first, sizeof(double) * (size_t)m.second, hipMemcpyDeviceToDevice, st));
            dst += m.second;
        }
    PIB_CHK(g->raise(st, g->shm->ready[r], seq));
    if (has_pv && lo > 0) {  // the previous rank's message for ITS next rank
        PIB_CHK(g->await(g->shm->ready[pv], seq, what, pv));
        PIB_HIP(hipMemcpyAsync(ghost_lo, g->win[(size_t)pv] + half, sizeof(double) * (size_t)lo, hipMemcpyDeviceToDevice, st));
    }
    if (has_nx && hi > 0) {
        PIB_CHK(g->await(g->shm->ready[nx], seq, what, nx));
        PIB_HIP(hipMemcpyAsync(ghost_hi, g->win[(size_t)nx], sizeof(double) * (size_t)hi, hipMemcpyDeviceToDevice, st));
    }
    PIB_CHK(g->raise(st, g->shm->done[r], seq));
    PIB_CHK(g->finish(st));
    if (has_pv) g->owe(seq, pv);
    if (has_nx && nx != pv) g->owe(seq, nx);
    return guard.done();
}

static int lb_halo(pib_solver *s, double *x_owned, int64_t n_owned, int64_t lo, int64_t hi, hipStream_t st)
{
    LoopbackGroup *g = s->comm.loop;
    const int P = s->comm.nranks, r = s->comm.rank;
    PIB_CHK(g->publish(r, x_owned, n_owned));
    PIB_HIP(hipEventRecord(g->ev_ready[(size_t)r], st));
    PIB_CHK(g->barrier());
    const bool ring = s->comm.ring;
    const size_t pv = (size_t)((r + P - 1) % P), nx = (size_t)((r + 1) % P);
    if ((r > 0 || ring) && lo > 0) {
        PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[pv], 0));
        const double *src = g->ptr[pv] + g->count[pv] - lo;
        PIB_HIP(hipMemcpyAsync(x_owned - lo, src, sizeof(double) * (size_t)lo, hipMemcpyDeviceToDevice, st));
    }
    if ((r < P - 1 || ring) && hi > 0) {
        PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[nx], 0));
        PIB_HIP(hipMemcpyAsync(x_owned + n_owned, g->ptr[nx], sizeof(double) * (size_t)hi, hipMemcpyDeviceToDevice, st));
    }
    PIB_HIP(hipEventRecord(g->ev_done[(size_t)r], st));
    PIB_CHK(g->barrier());
    if (r > 0 || ring) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[pv], 0));
    if (r < P - 1 || ring) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[nx], 0));
    PIB_CHK(g->barrier());
    return 0;
}

// Generic contiguous-plane exchange on a ghost-padded vector:
//   [lo ghosts | n_owned | hi ghosts], x_owned points at the owned part.
int halo_exchange_planes(pib_solver *s, double *x_owned, int64_t n_owned, int64_t lo, int64_t hi, int64_t send_prev,
                         int64_t send_next, hipStream_t st, int64_t group_longest)
{
    const int P = s->comm.nranks, r = s->comm.rank;
    if (!s->comm.active()) return 0;
    s->counters[3]++;
    s->counters[7] += 8 * (((r > 0 || s->comm.ring) ? send_prev : 0) + ((r < P - 1 || s->comm.ring) ? send_next : 0));  // bytes this rank sends
    if (s->comm.loop && s->comm.loop->shm) {
        const bool ring = s->comm.ring;
        return peer_window_exchange(s, st, (r + P - 1) % P, (r + 1) % P, r > 0 || ring, r < P - 1 || ring, {{x_owned, send_prev}},
                                    {{x_owned + n_owned - send_next, send_next}}, x_owned - lo, lo, x_owned + n_owned, hi, "halo planes", group_longest);
    }
    if (s->comm.loop) return lb_halo(s, x_owned, n_owned, lo, hi, st);
    if (s->comm.ring) {
        // periodic slab axis: rank 0's low ghost plane comes from rank P-1 and vice versa.  With P == 2 both messages
        // of a rank go to the same peer and are matched in issue order: bottom plane first, then top plane, so the
        // first message received is the peer's bottom plane (this rank's HIGH ghosts), the second its top plane.
        const int pv = (r + P - 1) % P, nx = (r + 1) % P;
        PIB_NCCL(ncclGroupStart());
        if (send_prev > 0) PIB_NCCL(ncclSend(x_owned, (size_t)send_prev, ncclDouble, pv, s->comm.comm, st));
        if (hi > 0) PIB_NCCL(ncclRecv(x_owned + n_owned, (size_t)hi, ncclDouble, nx, s->comm.comm, st));
        if (send_next > 0) PIB_NCCL(ncclSend(x_owned + n_owned - send_next, (size_t)send_next, ncclDouble, nx, s->comm.comm, st));
        if (lo > 0) PIB_NCCL(ncclRecv(x_owned - lo, (size_t)lo, ncclDouble, pv, s->comm.comm, st));
        PIB_NCCL(ncclGroupEnd());
        return 0;
    }
    PIB_NCCL(ncclGroupStart());
    if (r > 0) {
        if (send_prev > 0) PIB_NCCL(ncclSend(x_owned, (size_t)send_prev, ncclDouble, r - 1, s->comm.comm, st));
        if (lo > 0) PIB_NCCL(ncclRecv(x_owned - lo, (size_t)lo, ncclDouble, r - 1, s->comm.comm, st));
    }
    if (r < P - 1) {
        if (send_next > 0)
            PIB_NCCL(ncclSend(x_owned + n_owned - send_next, (size_t)send_next, ncclDouble, r + 1, s->comm.comm, st));
        if (hi > 0) PIB_NCCL(ncclRecv(x_owned + n_owned, (size_t)hi, ncclDouble, r + 1, s->comm.comm, st));
    }
    PIB_NCCL(ncclGroupEnd());
    return 0;
}

// segmented plan (DeviceCsr::seg_*): several {offset, count} pieces of the owned vector per neighbour, received back to
// back into the ghost pads
static int halo_exchange_segments(pib_solver *s, double *x_owned, hipStream_t st)
{
    const DeviceCsr &A = s->A;
    const int P = s->comm.nranks, r = s->comm.rank;
    s->counters[3]++;
    if (r > 0 || s->comm.ring)
        for (const auto &sg : A.seg_send_prev) s->counters[7] += 8 * sg.second;
    if (r < P - 1 || s->comm.ring)
        for (const auto &sg : A.seg_send_next) s->counters[7] += 8 * sg.second;
    if (s->comm.loop && s->comm.loop->shm) {
        std::vector<std::pair<const double *, int64_t>> to_prev, to_next;
        for (const auto &sg : A.seg_send_prev) to_prev.emplace_back(x_owned + sg.first, sg.second);
        for (const auto &sg : A.seg_send_next) to_next.emplace_back(x_owned + sg.first, sg.second);
        int64_t lo = 0, hi = 0;
        for (int64_t c : A.seg_recv_lo) lo += c;
        for (int64_t c : A.seg_recv_hi) hi += c;
        const bool ring = s->comm.ring;
        return peer_window_exchange(s, st, (r + P - 1) % P, (r + 1) % P, r > 0 || ring, r < P - 1 || ring, to_prev, to_next,
                                    x_owned - A.ghost_lo, lo, x_owned + A.n, hi, "halo segments", A.halo_group_longest);
    }
    if (s->comm.loop) {
        LoopbackGroup *g = s->comm.loop;
        PIB_CHK(g->publish(r, x_owned, A.n, &A.seg_send_prev, &A.seg_send_next));
        PIB_HIP(hipEventRecord(g->ev_ready[(size_t)r], st));
        PIB_CHK(g->barrier());
        const bool ring = s->comm.ring;
        const size_t pv = (size_t)((r + P - 1) % P), nx = (size_t)((r + 1) % P);
        if ((r > 0 || ring) && !A.seg_recv_lo.empty()) {
            PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[pv], 0));
            double *dst = x_owned - A.ghost_lo;
            for (const auto &sg : *g->segs_next[pv]) {  // what the previous rank sends to its next rank = my low ghosts
                PIB_HIP(hipMemcpyAsync(dst, g->ptr[pv] + sg.first, sizeof(double) * (size_t)sg.second, hipMemcpyDeviceToDevice, st));
                dst += sg.second;
            }
        }
        if ((r < P - 1 || ring) && !A.seg_recv_hi.empty()) {
            PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[nx], 0));
            double *dst = x_owned + A.n;
            for (const auto &sg : *g->segs_prev[nx]) {
                PIB_HIP(hipMemcpyAsync(dst, g->ptr[nx] + sg.first, sizeof(double) * (size_t)sg.second, hipMemcpyDeviceToDevice, st));
                dst += sg.second;
            }
        }
        PIB_HIP(hipEventRecord(g->ev_done[(size_t)r], st));
        PIB_CHK(g->barrier());
        if (r > 0 || ring) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[pv], 0));
        if (r < P - 1 || ring) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[nx], 0));
        PIB_CHK(g->barrier());
        return 0;
    }
    if (s->comm.ring) {
        // periodic slab axis: every rank has both neighbours.  With P == 2 all messages of a rank go to the one peer and are
        // matched in issue order: the pieces for the previous rank first (the peer receives them as its HIGH ghosts), then the
        // pieces for the next rank -- so the high ghosts are received first, the low ghosts second
        const int pv = (r + P - 1) % P, nx = (r + 1) % P;
        PIB_NCCL(ncclGroupStart());
        for (const auto &sg : A.seg_send_prev) PIB_NCCL(ncclSend(x_owned + sg.first, (size_t)sg.second, ncclDouble, pv, s->comm.comm, st));
        double *dst = x_owned + A.n;
        for (int64_t c : A.seg_recv_hi) {
            PIB_NCCL(ncclRecv(dst, (size_t)c, ncclDouble, nx, s->comm.comm, st));
            dst += c;
        }
        for (const auto &sg : A.seg_send_next) PIB_NCCL(ncclSend(x_owned + sg.first, (size_t)sg.second, ncclDouble, nx, s->comm.comm, st));
        dst = x_owned - A.ghost_lo;
        for (int64_t c : A.seg_recv_lo) {
            PIB_NCCL(ncclRecv(dst, (size_t)c, ncclDouble, pv, s->comm.comm, st));
            dst += c;
        }
        PIB_NCCL(ncclGroupEnd());
        return 0;
    }
    PIB_NCCL(ncclGroupStart());
    if (r > 0) {
        for (const auto &sg : A.seg_send_prev)
            PIB_NCCL(ncclSend(x_owned + sg.first, (size_t)sg.second, ncclDouble, r - 1, s->comm.comm, st));
        double *dst = x_owned - A.ghost_lo;
        for (int64_t c : A.seg_recv_lo) {
            PIB_NCCL(ncclRecv(dst, (size_t)c, ncclDouble, r - 1, s->comm.comm, st));
            dst += c;
        }
    }
    if (r < P - 1) {
        for (const auto &sg : A.seg_send_next)
            PIB_NCCL(ncclSend(x_owned + sg.first, (size_t)sg.second, ncclDouble, r + 1, s->comm.comm, st));
        double *dst = x_owned + A.n;
        for (int64_t c : A.seg_recv_hi) {
            PIB_NCCL(ncclRecv(dst, (size_t)c, ncclDouble, r + 1, s->comm.comm, st));
            dst += c;
        }
    }
    PIB_NCCL(ncclGroupEnd());
    return 0;
}

int halo_exchange(pib_solver *s, double *x_owned, hipStream_t st)
{
    const DeviceCsr &A = s->A;
    if (!s->comm.active()) return 0;
    if (A.general) return halo_exchange_general(s, x_owned, st);
    if (A.segmented) return halo_exchange_segments(s, x_owned, st);
    return halo_exchange_planes(s, x_owned, A.n, A.ghost_lo, A.ghost_hi, A.send_prev, A.send_next, st, A.halo_group_longest);
}

// in-place sum over ranks of `count` (<= PIB_NRED) doubles in device memory
int comm_allreduce_sum(pib_solver *s, double *dev, int count, hipStream_t st)
{
    PeerFailGuard guard((s->comm.loop && s->comm.loop->shm) ? s->comm.loop : nullptr);
    if (!s->comm.active()) return guard.done();
    s->counters[2]++;
    if (s->comm.loop && s->comm.loop->shm && s->comm.loop->devord) {
        LoopbackGroup *g = s->comm.loop;
        if (g->blk->failed || g->shm->failed.load(std::memory_order_relaxed)) {
            g->shm->failed.store(1);
            return fail(PIB_ERR_LIB, "peer transport (device-ordered): a rank timed out waiting for a flag");
        }
        if (g->chained && g->chain_stream != st) PIB_HIP(hipStreamWaitEvent(st, g->chain, 0));
        hipLaunchKernelGGL(k_dallreduce, dim3(1), dim3(64), 0, st, dev, count, g->dc);
        PIB_HIP(hipGetLastError());
        PIB_CHK(g->finish(st));
        return guard.done();
    }
    if (s->comm.loop && s->comm.loop->shm) {
        // every rank copies its values into its row of rank 0's staging buffer, then sums the rows in rank order
        LoopbackGroup *g = s->comm.loop;
        const int P = s->comm.nranks, r = s->comm.rank;
        uint64_t seq = 0;
        PIB_CHK(g->begin(st, &seq));
        PIB_HIP(hipMemcpyAsync(g->staging + (size_t)r * PIB_NRED, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st));
        PIB_CHK(g->raise(st, g->shm->ready[r], seq));
        for (int q = 0; q < P; ++q)
            if (q != r) PIB_CHK(g->await(g->shm->ready[q], seq, "all-reduce", q));
        hipLaunchKernelGGL(k_lb_sum, dim3(1), dim3(64), 0, st, dev, g->staging, P, count);
        PIB_HIP(hipGetLastError());
        PIB_CHK(g->raise(st, g->shm->done[r], seq));
        PIB_CHK(g->finish(st));
        for (int q = 0; q < P; ++q)
            if (q != r) g->owe(seq, q);  // the rows are free again once everybody has summed them
        return guard.done();
    }
    if (s->comm.loop) {
        LoopbackGroup *g = s->comm.loop;
        const int P = s->comm.nranks, r = s->comm.rank;
        PIB_HIP(hipMemcpyAsync(g->staging + (size_t)r * PIB_NRED, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st));
        PIB_HIP(hipEventRecord(g->ev_ready[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q)
            if (q != r) PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[(size_t)q], 0));
        hipLaunchKernelGGL(k_lb_sum, dim3(1), dim3(64), 0, st, dev, g->staging, P, count);
        PIB_HIP(hipEventRecord(g->ev_done[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q)
            if (q != r) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[(size_t)q], 0));
        PIB_CHK(g->barrier());
        return guard.done();
    }
    PIB_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, s->comm.comm, st));
    return guard.done();
}

__global__ void k_lb_add(double *__restrict__ acc, const double *__restrict__ x, int64_t n, int first)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc[i] = first ? x[i] : acc[i] + x[i];
}

// in-place sum over ranks of `count` doubles (any count): the dense force system of the immersed boundaries, whose
// entries every rank sums over its own velocity points
int comm_allreduce_big(pib_solver *s, double *dev, int64_t count, hipStream_t st)
{
    PeerFailGuard guard((s->comm.loop && s->comm.loop->shm) ? s->comm.loop : nullptr);
    if (!s->comm.active() || count <= 0) return guard.done();
    if (s->comm.loop && s->comm.loop->shm) {
        // a window's worth at a time: every rank lays its piece into its window, then sums all the windows' pieces in
        // rank order (the same bits on every rank) straight into its own buffer
        LoopbackGroup *g = s->comm.loop;
        const int P = s->comm.nranks, r = s->comm.rank;
        for (int64_t off = 0; off < count; off += g->win_doubles) {
            const int64_t len = std::min<int64_t>(g->win_doubles, count - off);
            uint64_t seq = 0;
            PIB_CHK(g->begin(st, &seq));
            PIB_HIP(hipMemcpyAsync(g->win_local, dev + off, sizeof(double) * (size_t)len, hipMemcpyDeviceToDevice, st));
            PIB_CHK(g->raise(st, g->shm->ready[r], seq));
            const int nb = (int)std::min<int64_t>(4096, (len + 255) / 256);
            for (int q = 0; q < P; ++q) {
                if (q != r) PIB_CHK(g->await(g->shm->ready[q], seq, "all-reduce (large)", q));
                hipLaunchKernelGGL(k_lb_add, dim3(nb), dim3(256), 0, st, dev + off, g->win[(size_t)q], len, q == 0 ? 1 : 0);
            }
            PIB_HIP(hipGetLastError());
            PIB_CHK(g->raise(st, g->shm->done[r], seq));
            PIB_CHK(g->finish(st));
            for (int q = 0; q < P; ++q)
                if (q != r) g->owe(seq, q);
        }
        return guard.done();
    }
    if (s->comm.loop) {
        LoopbackGroup *g = s->comm.loop;
        const int P = s->comm.nranks, r = s->comm.rank;
        double *tmp = nullptr;
        PIB_HIP(hipMalloc(&tmp, sizeof(double) * (size_t)count));
        PIB_CHK(g->publish(r, dev, count));
        PIB_HIP(hipEventRecord(g->ev_ready[(size_t)r], st));
        PIB_CHK(g->barrier());
        const int nb = (int)std::min<int64_t>(4096, (count + 255) / 256);
        for (int q = 0; q < P; ++q) {  // rank order: the same bits on every rank
            if (q != r) PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[(size_t)q], 0));
            hipLaunchKernelGGL(k_lb_add, dim3(nb), dim3(256), 0, st, tmp, g->ptr[(size_t)q], count, q == 0 ? 1 : 0);
        }
        PIB_HIP(hipEventRecord(g->ev_done[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q)
            if (q != r) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[(size_t)q], 0));  // everybody has read my buffer
        PIB_HIP(hipMemcpyAsync(dev, tmp, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st));
        PIB_HIP(hipStreamSynchronize(st));
        PIB_CHK(g->barrier());
        PIB_HIP(hipFree(tmp));
        return guard.done();
    }
    PIB_NCCL(ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, s->comm.comm, st));
    return guard.done();
}

// every rank contributes counts[rank] doubles at `send`; `recv_base + offs[q]` receives rank q's part
int comm_allgatherv(pib_solver *s, const double *send, double *recv_base, const std::vector<int64_t> &counts,
                    const std::vector<int64_t> &offs, hipStream_t st)
{
    PeerFailGuard guard((s->comm.loop && s->comm.loop->shm) ? s->comm.loop : nullptr);
    const int P = s->comm.nranks, r = s->comm.rank;
    if (!s->comm.active()) return guard.done();
    s->counters[3]++;
    if (s->comm.loop && s->comm.loop->shm && s->comm.loop->devord) {
        LoopbackGroup *g = s->comm.loop;
        std::vector<int64_t> place((size_t)P + 1, 0);  // every receive half: the ranks' parts back to back
        for (int q = 0; q < P; ++q) place[(size_t)q + 1] = place[(size_t)q] + counts[(size_t)q];
        if (place[(size_t)P] <= g->dc.half) {
            std::vector<PutMsg> puts;
            std::vector<GetMsg> gets;
            for (int q = 0; q < P; ++q) {
                if (q == r) continue;
                puts.push_back({q, send, place[(size_t)r], counts[(size_t)r]});
                gets.push_back({q, recv_base + offs[(size_t)q], place[(size_t)q], counts[(size_t)q]});
            }
            if (recv_base + offs[(size_t)r] != send && counts[(size_t)r] > 0)
                PIB_HIP(hipMemcpyAsync(recv_base + offs[(size_t)r], send, sizeof(double) * (size_t)counts[(size_t)r], hipMemcpyDeviceToDevice, st));
            PIB_CHK(dev_collective(s, st, puts, gets));
            return guard.done();
        }
    }
    if (s->comm.loop && s->comm.loop->shm) {
        LoopbackGroup *g = s->comm.loop;
        int64_t longest = 0;
        for (int q = 0; q < P; ++q) longest = std::max(longest, counts[(size_t)q]);
        for (int64_t off = 0; off < longest; off += g->win_doubles) {  // a window's worth of every rank's part at a time
            uint64_t seq = 0;
            PIB_CHK(g->begin(st, &seq));
            const int64_t mine = std::max<int64_t>(0, std::min<int64_t>(g->win_doubles, counts[(size_t)r] - off));
            if (mine > 0) PIB_HIP(hipMemcpyAsync(g->win_local, send + off, sizeof(double) * (size_t)mine, hipMemcpyDeviceToDevice, st));
            PIB_CHK(g->raise(st, g->shm->ready[r], seq));
            for (int q = 0; q < P; ++q) {
                const int64_t len = std::max<int64_t>(0, std::min<int64_t>(g->win_doubles, counts[(size_t)q] - off));
                if (q != r) PIB_CHK(g->await(g->shm->ready[q], seq, "all-gather", q));
                if (len > 0)
                    PIB_HIP(hipMemcpyAsync(recv_base + offs[(size_t)q] + off, g->win[(size_t)q], sizeof(double) * (size_t)len,
                                           hipMemcpyDeviceToDevice, st));
            }
            PIB_CHK(g->raise(st, g->shm->done[r], seq));
            PIB_CHK(g->finish(st));
            for (int q = 0; q < P; ++q)
                if (q != r) g->owe(seq, q);
        }
        return guard.done();
    }
    if (s->comm.loop) {
        LoopbackGroup *g = s->comm.loop;
        PIB_CHK(g->publish(r, send, counts[(size_t)r]));
        PIB_HIP(hipEventRecord(g->ev_ready[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q) {
            if (q != r) PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[(size_t)q], 0));
            PIB_HIP(hipMemcpyAsync(recv_base + offs[(size_t)q], g->ptr[(size_t)q], sizeof(double) * (size_t)counts[(size_t)q],
                                   hipMemcpyDeviceToDevice, st));
        }
        PIB_HIP(hipEventRecord(g->ev_done[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q)
            if (q != r) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[(size_t)q], 0));
        PIB_CHK(g->barrier());
        return guard.done();
    }
    bool equal = true;
    for (int q = 0; q < P; ++q) equal = equal && counts[(size_t)q] == counts[0] && offs[(size_t)q] == (int64_t)q * counts[0];
    if (equal) {
        PIB_NCCL(ncclAllGather(send, recv_base, (size_t)counts[0], ncclDouble, s->comm.comm, st));
    } else {
        PIB_NCCL(ncclGroupStart());
        for (int q = 0; q < P; ++q) {
            const double *src = (q == r) ? send : recv_base + offs[(size_t)q];
            PIB_NCCL(ncclBroadcast(src, recv_base + offs[(size_t)q], (size_t)counts[(size_t)q], ncclDouble, q, s->comm.comm, st));
        }
        PIB_NCCL(ncclGroupEnd());
    }
    return guard.done();
}

// One message per ordered pair of ranks (ExchangePlan): a rank's messages lie back to back, in destination order, at
// `stream`; what rank q sends to this rank lands at recv[q].  The general halo exchange of an arbitrary row partition
// and the box <-> slab moves of redistribute.hip.  RCCL: one group of ncclSend / ncclRecv (point-to-point over xGMI,
// every pair that has a message); loopback: the streams are published and pulled; peer: the streams go through the
// windows a window's worth at a time -- every rank knows the whole table, so it knows which part of which message lies
// where in a neighbour's window in every round.
int comm_exchange_v(pib_solver *s, const ExchangePlan &pl, const double *stream, double *const *recv, hipStream_t st)
{
    PeerFailGuard guard((s->comm.loop && s->comm.loop->shm) ? s->comm.loop : nullptr);
    const int P = pl.P, r = pl.me;
    if (pl.from(r) > 0)
        PIB_HIP(hipMemcpyAsync(recv[r], stream + pl.send_off[(size_t)r], sizeof(double) * (size_t)pl.from(r), hipMemcpyDeviceToDevice, st));
    if (!s->comm.active()) return guard.done();
    s->counters[3]++;
    s->counters[7] += 8 * (pl.send_total - pl.to(r));
    if (s->comm.loop && s->comm.loop->shm && s->comm.loop->devord) {
        LoopbackGroup *g = s->comm.loop;
        int64_t most = 0;  // the fullest receive half (every rank holds the whole table: the same verdict everywhere)
        for (int d = 0; d < P; ++d) {
            int64_t t = 0;
            for (int q = 0; q < P; ++q) t += pl.cnt[(size_t)q * P + d];
            most = std::max(most, t);
        }
        if (most <= g->dc.half) {
            std::vector<PutMsg> puts;
            std::vector<GetMsg> gets;
            for (int d = 0; d < P; ++d) {
                if (d == r || pl.to(d) == 0) continue;
                int64_t o = 0;
                for (int q = 0; q < r; ++q) o += pl.cnt[(size_t)q * P + d];
                puts.push_back({d, stream + pl.send_off[(size_t)d], o, pl.to(d)});
            }
            int64_t o = 0;
            for (int q = 0; q < P; ++q) {
                if (q != r && pl.from(q) > 0) gets.push_back({q, recv[q], o, pl.from(q)});
                o += pl.from(q);
            }
            PIB_CHK(dev_collective(s, st, puts, gets));
            return guard.done();
        }
    }
    if (s->comm.loop && s->comm.loop->shm) {
        LoopbackGroup *g = s->comm.loop;
        const int64_t W = g->win_doubles;
        const int64_t rounds = (pl.max_stream + W - 1) / W;
        for (int64_t k = 0; k < rounds; ++k) {
            const int64_t lo = k * W, hi = lo + W;
            uint64_t seq = 0;
            PIB_CHK(g->begin(st, &seq));
            const int64_t mine = std::max<int64_t>(0, std::min(hi, pl.send_total) - lo);
            if (mine > 0) PIB_HIP(hipMemcpyAsync(g->win_local, stream + lo, sizeof(double) * (size_t)mine, hipMemcpyDeviceToDevice, st));
            PIB_CHK(g->raise(st, g->shm->ready[r], seq));
            for (int q = 0; q < P; ++q) {
                if (q == r) continue;
                const int64_t a = std::max(lo, pl.src_off[(size_t)q]), b = std::min(hi, pl.src_off[(size_t)q] + pl.from(q));
                if (b <= a) continue;
                PIB_CHK(g->await(g->shm->ready[q], seq, "exchange", q));
                PIB_HIP(hipMemcpyAsync(recv[q] + (a - pl.src_off[(size_t)q]), g->win[(size_t)q] + (a - lo), sizeof(double) * (size_t)(b - a),
                                       hipMemcpyDeviceToDevice, st));
            }
            PIB_CHK(g->raise(st, g->shm->done[r], seq));
            PIB_CHK(g->finish(st));
            for (int q = 0; q < P; ++q)
                if (q != r) g->owe(seq, q);
        }
        return guard.done();
    }
    if (s->comm.loop) {
        LoopbackGroup *g = s->comm.loop;
        PIB_CHK(g->publish(r, stream, pl.send_total));
        PIB_HIP(hipEventRecord(g->ev_ready[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q) {
            if (q == r || pl.from(q) == 0) continue;
            PIB_HIP(hipStreamWaitEvent(st, g->ev_ready[(size_t)q], 0));
            PIB_HIP(hipMemcpyAsync(recv[q], g->ptr[(size_t)q] + pl.src_off[(size_t)q], sizeof(double) * (size_t)pl.from(q), hipMemcpyDeviceToDevice, st));
        }
        PIB_HIP(hipEventRecord(g->ev_done[(size_t)r], st));
        PIB_CHK(g->barrier());
        for (int q = 0; q < P; ++q)
            if (q != r && pl.to(q) > 0) PIB_HIP(hipStreamWaitEvent(st, g->ev_done[(size_t)q], 0));  // they have read my stream
        PIB_CHK(g->barrier());
        return guard.done();
    }
    PIB_NCCL(ncclGroupStart());
    for (int q = 0; q < P; ++q) {
        if (q == r) continue;
        if (pl.to(q) > 0) PIB_NCCL(ncclSend(stream + pl.send_off[(size_t)q], (size_t)pl.to(q), ncclDouble, q, s->comm.comm, st));
        if (pl.from(q) > 0) PIB_NCCL(ncclRecv(recv[q], (size_t)pl.from(q), ncclDouble, q, s->comm.comm, st));
    }
    PIB_NCCL(ncclGroupEnd());
    return guard.done();
}

}  // namespace pib

extern "C" int pib_comm_unique_id(void *uid_out)
try {
    if (uid_out == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "pib_comm_unique_id: null output");
    ncclUniqueId id;
    PIB_NCCL(ncclGetUniqueId(&id));
    std::memset(uid_out, 0, PIB_UID_BYTES);
    std::memcpy(uid_out, &id, sizeof(id));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

// ---- the RCCL entry points on hardware with ONE rank.  RCCL refuses several ranks per device and the test box has one
// GPU, so the multi-rank ALGORITHM is tested through the loopback transport; this runs the other half -- the RCCL calls
// themselves, exactly as the functions above issue them (grouped ncclSend / ncclRecv incl. the periodic ring whose two
// neighbours are the rank itself, in-place ncclAllReduce of the recurrence scalars and of a large buffer, in-place
// ncclAllGather, the grouped ncclBroadcast form of the all-gather-v, an exchange on the communication stream ordered
// with events against the solver's stream) -- in a one-rank world and checks what arrives.
__global__ void k_selftest_fill(double *x, int64_t n, double a, double b)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = a + b * (double)i;
}
extern "C" int pib_comm_selftest(int device, int64_t n_owned, int64_t ghost, double *max_err_out, int *comm_ranks_out)
try {
    using namespace pib;
    if (max_err_out == nullptr || n_owned < 2 * ghost || ghost < 1) return fail(PIB_ERR_ARG_WRONG, "pib_comm_selftest: bad arguments");
    int ndev = 0;
    PIB_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(PIB_ERR_LIB, "pib_comm_selftest: no GPU");
    PIB_HIP(hipSetDevice(device < 0 ? 0 : device));
    pib_solver S;
    pib_solver *s = &S;
    s->name = "selftest";
    s->device = device < 0 ? 0 : device;
    PIB_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    PIB_HIP(hipStreamCreateWithFlags(&s->stream_comm, hipStreamNonBlocking));
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    PIB_HIP(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
    PIB_HIP(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
    ncclUniqueId id;
    PIB_NCCL(ncclGetUniqueId(&id));
    s->comm.rank = 0;
    s->comm.nranks = 1;
    PIB_NCCL(ncclCommInitRank(&s->comm.comm, 1, id, 0));
    adopt_nccl(s);
    int cnt = 0;
    PIB_NCCL(ncclCommCount(s->comm.comm, &cnt));
    if (comm_ranks_out) *comm_ranks_out = cnt;
    const int64_t tot = n_owned + 2 * ghost;
    double *d = nullptr, *d2 = nullptr;
    PIB_HIP(hipMalloc(&d, sizeof(double) * (size_t)tot));
    PIB_HIP(hipMalloc(&d2, sizeof(double) * (size_t)tot));
    std::vector<double> h((size_t)tot);
    double err = 0.0;
    auto owned = [&](int64_t i) { return 1.0 + 0.5 * (double)i; };
    auto refill = [&](hipStream_t q) {
        PIB_HIP(hipMemsetAsync(d, 0xff, sizeof(double) * (size_t)tot, q));  // NaN pattern in the ghosts
        hipLaunchKernelGGL(k_selftest_fill, dim3(64), dim3(256), 0, q, d + ghost, n_owned, 1.0, 0.5);
        PIB_HIP(hipGetLastError());
        return 0;
    };
    auto fetch = [&](hipStream_t q) {
        PIB_HIP(hipMemcpyAsync(h.data(), d, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost, q));
        PIB_HIP(hipStreamSynchronize(q));
        return 0;
    };
    // 1. wall-bounded slab axis, one rank: no neighbour, an empty group; the ghosts stay untouched
    s->comm.ring = false;
    PIB_CHK(refill(s->stream));
    PIB_CHK(halo_exchange_planes(s, d + ghost, n_owned, ghost, ghost, ghost, ghost, s->stream));
    PIB_CHK(fetch(s->stream));
    for (int64_t i = 0; i < n_owned; ++i) err = std::max(err, std::fabs(h[(size_t)(ghost + i)] - owned(i)));
    if (h[0] == h[0] || h[(size_t)(tot - 1)] == h[(size_t)(tot - 1)]) err = std::max(err, 1.0);  // must still be NaN
    // 2. periodic slab axis: both neighbours are this rank -- the low ghosts are its top entries, the high ghosts its
    //    bottom entries (two messages to the same peer, matched in issue order)
    s->comm.ring = true;
    PIB_CHK(refill(s->stream));
    PIB_CHK(halo_exchange_planes(s, d + ghost, n_owned, ghost, ghost, ghost, ghost, s->stream));
    PIB_CHK(fetch(s->stream));
    for (int64_t i = 0; i < ghost; ++i) {
        err = std::max(err, std::fabs(h[(size_t)i] - owned(n_owned - ghost + i)));
        err = std::max(err, std::fabs(h[(size_t)(ghost + n_owned + i)] - owned(i)));
    }
    // 3. the same exchange on the communication stream, ordered against the solver's stream with events (gmg.hip's
    //    overlapped right-hand-side exchange)
    PIB_CHK(refill(s->stream));
    PIB_HIP(hipEventRecord(ev_a, s->stream));
    PIB_HIP(hipStreamWaitEvent(s->stream_comm, ev_a, 0));
    PIB_CHK(halo_exchange_planes(s, d + ghost, n_owned, ghost, ghost, ghost, ghost, s->stream_comm));
    PIB_HIP(hipEventRecord(ev_b, s->stream_comm));
    PIB_HIP(hipStreamWaitEvent(s->stream, ev_b, 0));
    PIB_CHK(fetch(s->stream));
    for (int64_t i = 0; i < ghost; ++i) {
        err = std::max(err, std::fabs(h[(size_t)i] - owned(n_owned - ghost + i)));
        err = std::max(err, std::fabs(h[(size_t)(ghost + n_owned + i)] - owned(i)));
    }
    // 3b. the segmented plan of the packed velocity ordering on the ring: two pieces to either neighbour (= this rank), the
    //     pieces for the previous rank arrive as the high ghosts, the pieces for the next rank as the low ghosts
    {
        DeviceCsr &A = s->A;
        const int64_t g2 = ghost / 2, g1 = ghost - g2;
        A.n = n_owned;
        A.ghost_lo = A.ghost_hi = ghost;
        A.segmented = true;
        A.seg_send_prev = {{0, g1}, {n_owned / 2, g2}};
        A.seg_send_next = {{n_owned / 2 - g1, g1}, {n_owned - g2, g2}};
        A.seg_recv_lo = {g1, g2};
        A.seg_recv_hi = {g1, g2};
        if (g2 == 0) {
            A.seg_send_prev.pop_back();
            A.seg_send_next.pop_back();
            A.seg_recv_lo.pop_back();
            A.seg_recv_hi.pop_back();
        }
        PIB_CHK(refill(s->stream));
        PIB_CHK(halo_exchange(s, d + ghost, s->stream));
        PIB_CHK(fetch(s->stream));
        for (int64_t i = 0; i < g1; ++i) {
            err = std::max(err, std::fabs(h[(size_t)i] - owned(n_owned / 2 - g1 + i)));                 // low ghosts: "next" pieces
            err = std::max(err, std::fabs(h[(size_t)(ghost + n_owned + i)] - owned(i)));               // high ghosts: "prev" pieces
        }
        for (int64_t i = 0; i < g2; ++i) {
            err = std::max(err, std::fabs(h[(size_t)(g1 + i)] - owned(n_owned - g2 + i)));
            err = std::max(err, std::fabs(h[(size_t)(ghost + n_owned + g1 + i)] - owned(n_owned / 2 + i)));
        }
        A.segmented = false;
        A.seg_send_prev.clear();
        A.seg_send_next.clear();
        A.seg_recv_lo.clear();
        A.seg_recv_hi.clear();
        A.n = 0;
        A.ghost_lo = A.ghost_hi = 0;
    }
    s->comm.ring = false;
    // 4. in-place all-reduce of the recurrence scalars and of a large buffer: one rank, the values stay
    PIB_CHK(refill(s->stream));
    PIB_CHK(comm_allreduce_sum(s, d + ghost, PIB_NRED, s->stream));
    PIB_CHK(comm_allreduce_big(s, d + ghost, n_owned, s->stream));
    PIB_CHK(fetch(s->stream));
    for (int64_t i = 0; i < n_owned; ++i) err = std::max(err, std::fabs(h[(size_t)(ghost + i)] - owned(i)));
    // 5. all-gather (in place: the send buffer is the rank's part of the receive buffer) and its grouped-broadcast form
    {
        std::vector<int64_t> counts(1, n_owned), offs(1, 0);
        PIB_CHK(comm_allgatherv(s, d + ghost, d + ghost, counts, offs, s->stream));
        offs[0] = ghost;  // not rank * count: the all-gather-v form
        PIB_HIP(hipMemsetAsync(d2, 0, sizeof(double) * (size_t)tot, s->stream));
        PIB_CHK(comm_allgatherv(s, d + ghost, d2, counts, offs, s->stream));
        PIB_CHK(fetch(s->stream));
        for (int64_t i = 0; i < n_owned; ++i) err = std::max(err, std::fabs(h[(size_t)(ghost + i)] - owned(i)));
        PIB_HIP(hipMemcpyAsync(h.data(), d2, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));
        for (int64_t i = 0; i < n_owned; ++i) err = std::max(err, std::fabs(h[(size_t)(ghost + i)] - owned(i)));
    }
    // 6. the host-side all-gather of the set-up
    {
        std::vector<double> mine = {3.0, 4.0, 5.0}, all;
        PIB_CHK(comm_allgather_host(s, mine, all));
        if (all.size() != 3) err = std::max(err, 1.0);
        else for (int k = 0; k < 3; ++k) err = std::max(err, std::fabs(all[(size_t)k] - mine[(size_t)k]));
    }
    *max_err_out = err;
    PIB_HIP(hipFree(d));
    PIB_HIP(hipFree(d2));
    comm_release(s);
    (void)hipEventDestroy(ev_a);
    (void)hipEventDestroy(ev_b);
    (void)hipStreamDestroy(s->stream);
    (void)hipStreamDestroy(s->stream_comm);
    s->stream = s->stream_comm = nullptr;
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

// Latency of the two collectives of a Krylov iteration on whatever transport `s` has attached (s == NULL: RCCL in a
// one-rank world whose ring neighbours are the rank itself, as in pib_comm_selftest): `reps` back-to-back plane exchanges
// of `count` doubles each way, then `reps` all-reduces of PIB_NRED doubles; usec[0] / usec[1] = wall time per call
// (enqueue + execution, one stream synchronisation at the end).  tools/comm_latency.py
extern "C" int pib_comm_latency(pib_solver *s, int64_t count, int reps, double usec[2])
try {
    using namespace pib;
    if (usec == nullptr || count < 1 || reps < 1) return fail(PIB_ERR_ARG_WRONG, "pib_comm_latency: bad arguments");
    pib_solver S;
    bool own = false;
    if (s == nullptr) {
        own = true;
        s = &S;
        PIB_HIP(hipSetDevice(0));
        PIB_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        ncclUniqueId id;
        PIB_NCCL(ncclGetUniqueId(&id));
        PIB_NCCL(ncclCommInitRank(&s->comm.comm, 1, id, 0));
        adopt_nccl(s);
        s->comm.ring = true;
    } else
        PIB_HIP(hipSetDevice(s->device));
    const int P = s->comm.nranks, r = s->comm.rank;
    const bool ring = s->comm.ring;
    const int64_t n = 4 * count;
    double *d = nullptr;
    PIB_HIP(hipMalloc(&d, sizeof(double) * (size_t)(n + 2 * count)));
    PIB_MEMSET(d, 0, sizeof(double) * (size_t)(n + 2 * count));
    const int64_t lo = (r > 0 || ring) ? count : 0, hi = (r < P - 1 || ring) ? count : 0;
    auto run = [&](int which) -> int {
        for (int w = 0; w < 2; ++w) {  // warm-up, then the timed pass
            const int k = w == 0 ? std::min(reps, 10) : reps;
            PIB_HIP(hipStreamSynchronize(s->stream));
            if (s->comm.loop) PIB_CHK(s->comm.loop->barrier());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < k; ++i) {
                if (which == 0) PIB_CHK(halo_exchange_planes(s, d + count, n, lo, hi, lo, hi, s->stream));
                else PIB_CHK(comm_allreduce_sum(s, d + count, PIB_NRED, s->stream));
            }
            PIB_HIP(hipStreamSynchronize(s->stream));
            usec[which] = 1.0e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / k;
        }
        return 0;
    };
    PIB_CHK(run(0));
    PIB_CHK(run(1));
    if (s->comm.loop) PIB_CHK(s->comm.loop->barrier());
    PIB_HIP(hipFree(d));
    if (own) {
        comm_release(s);
        (void)hipStreamDestroy(s->stream);
        s->stream = nullptr;
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

namespace pib {
// An iteration of a solver on several ranks can be captured into a hipGraph when its collectives are kernels and copies only
// and carry nothing per call: the device-ordered peer transport (collective numbers are counted on the device).  RCCL's
// calls capture in principle; they have never run on more than one rank here, so they are left out.
bool comm_capturable(const pib_solver *s)
{
    return s->comm.nranks == 1 || (s->comm.loop != nullptr && s->comm.loop->shm != nullptr && s->comm.loop->devord);
}
// around the capture: the event that chains a rank's collectives across streams must not cross the capture's boundary
// (an event recorded inside a capture cannot be waited for outside it and vice versa); all collectives of the iteration
// body are issued on the capturing stream or on streams forked from and joined back to it
void comm_capture_boundary(pib_solver *s, bool begin)
{
    if (s->comm.loop == nullptr || s->comm.loop->shm == nullptr) return;
    s->comm.loop->capturing = begin;
    s->comm.loop->chained = false;
}
}  // namespace pib

// the id of a peer-transport world: a fresh shared-memory name; rank 0 makes it, every rank gets it (like the RCCL id)
extern "C" int pib_comm_peer_id_ordered(void *uid_out, int device_ordered)
try {
    using namespace pib;
    if (uid_out == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_comm_peer_id: null output");
    static std::atomic<unsigned> serial{0};
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::memset(uid_out, 0, PIB_UID_BYTES);
    std::memcpy(uid_out, PEER_MAGIC, 8);
    std::snprintf((char *)uid_out + 8, PIB_UID_BYTES - 8, "/pib_peer%c_%d_%u_%llx", device_ordered ? 'D' : 'H', (int)getpid(),
                  serial.fetch_add(1), (unsigned long long)now);
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
// the ordering of the collectives -- flags written and awaited by the GPUs in stream order (default), or by host threads
// (PIB_PEER_ORDER=host: the first implementation, kept as the reference the device-ordered one is compared with)
extern "C" int pib_comm_peer_id(void *uid_out)
try {
    const char *o = std::getenv("PIB_PEER_ORDER");
    return pib_comm_peer_id_ordered(uid_out, (o != nullptr && std::strcmp(o, "host") == 0) ? 0 : 1);
} catch (...) {
    return pib::fail_exception(__func__);
}

extern "C" int pib_comm_loopback_create(int nranks, void *uid_out)
try {
    using namespace pib;
    if (uid_out == nullptr || nranks < 2) return fail(PIB_ERR_ARG_WRONG, "pib_comm_loopback_create: bad arguments");
    LoopbackGroup *g = new LoopbackGroup();
    g->nranks = nranks;
    g->ptr.assign((size_t)nranks, nullptr);
    g->segs_prev.assign((size_t)nranks, nullptr);
    g->segs_next.assign((size_t)nranks, nullptr);
    g->count.assign((size_t)nranks, 0);
    g->host_vals.assign(4 * (size_t)nranks, 0);
    g->ev_ready.resize((size_t)nranks);
    g->ev_done.resize((size_t)nranks);
    for (int r = 0; r < nranks; ++r) {
        PIB_HIP(hipEventCreateWithFlags(&g->ev_ready[(size_t)r], hipEventDisableTiming));
        PIB_HIP(hipEventCreateWithFlags(&g->ev_done[(size_t)r], hipEventDisableTiming));
    }
    PIB_HIP(hipMalloc(&g->staging, sizeof(double) * PIB_NRED * (size_t)nranks));
    PIB_MEMSET(g->staging, 0, sizeof(double) * PIB_NRED * (size_t)nranks);
    std::memset(uid_out, 0, PIB_UID_BYTES);
    std::memcpy(uid_out, LOOP_MAGIC, 8);
    std::memcpy((char *)uid_out + 8, &g, sizeof(g));
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}

extern "C" int pib_comm_loopback_destroy(const void *uid)
try {
    using namespace pib;
    if (uid == nullptr || std::memcmp(uid, LOOP_MAGIC, 8) != 0) return fail(PIB_ERR_ARG_WRONG, "not a loopback id");
    LoopbackGroup *g = nullptr;
    std::memcpy(&g, (const char *)uid + 8, sizeof(g));
    if (g) {
        for (auto e : g->ev_ready) (void)hipEventDestroy(e);
        for (auto e : g->ev_done) (void)hipEventDestroy(e);
        if (g->staging) (void)hipFree(g->staging);
        delete g;
    }
    return 0;
} catch (...) {
    return pib::fail_exception(__func__);
}
