// redistribute.hip -- the device side of partition.cpp: the packed halo exchange of a general row partition and the
// per-solve box <-> slab moves of b and x (rows handed over in DMDA boxes, PETSC_DECIDE; src/mesh/cartesianmesh.cpp:97,
// 503-519, 700-738).
//
// All HBM-bound gathers: 8 B read + 8 B written per entry moved.  The halo pack reads x[send_idx[i]] -- the faces of a
// box: runs of xm entries for the y / z faces, a stride of xm for the x faces -- into one contiguous send stream, the
// grouped exchange of halo.hip (comm_exchange_v) delivers every peer's part straight into the ghost pads of the
// receiver's vector.  The box -> slab move needs no pack at all (a box's rows are ordered by plane, so the part for slab
// d is a contiguous range); the slab side gathers from / scatters into its staging vector with the closed-form position
// of natural cell (i, j, k) inside the chunk of the box that owns it.
#include <hipcub/hipcub.hpp>

#include <cstring>

#include "pib_internal.hpp"

namespace pib {

__global__ __launch_bounds__(256) void k_pack_rows(const double *__restrict__ x, const int32_t *__restrict__ idx, double *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = x[idx[i]];
}

int halo_exchange_general(pib_solver *s, double *x_owned, hipStream_t st)
{
    const DeviceCsr &A = s->A;
    const ExchangePlan &pl = A.xplan;
    const int P = pl.P, r = pl.me;
    if (pl.send_total > 0) {
        const int nb = (int)std::min<int64_t>(2048, (pl.send_total + 255) / 256);
        hipLaunchKernelGGL(k_pack_rows, dim3(nb), dim3(256), 0, st, x_owned, A.send_idx, A.send_buf, pl.send_total);
        PIB_HIP(hipGetLastError());
    }
    double *recv[PIB_MAX_RANKS];
    if (P > PIB_MAX_RANKS) return fail(PIB_ERR_SUP, "general halo plan: at most %d ranks", PIB_MAX_RANKS);
    for (int q = 0; q < P; ++q)
        recv[q] = q < r ? x_owned - A.ghost_lo + A.ghost_off[(size_t)q] : x_owned + A.n + (A.ghost_off[(size_t)q] - A.ghost_lo);
    return comm_exchange_v(s, pl, A.send_buf, recv, st);
}

// ------------------------------------------------------------------------------------------------ boxes <-> slabs
// natural local cell g of the slab -> its position in the staging vector (the chunks of the source boxes back to back in
// rank order, every chunk in the box's own order)
template <bool TO_NATURAL>
__global__ __launch_bounds__(256) void k_redist(double *__restrict__ nat, double *__restrict__ stage, int64_t n_slab, int64_t n0, int64_t n1,
                                                int64_t k0, int gm, int gn, int gp, const int32_t *__restrict__ split,
                                                const int64_t *__restrict__ src)
{
    __shared__ int32_t sp[3 * (PIB_MAX_RANKS + 1)];
    const int nsp = (gm + 1) + (gn + 1) + (gp + 1);
    for (int t = threadIdx.x; t < nsp; t += 256) sp[t] = split[t];
    __syncthreads();
    const int32_t *sx = sp, *sy = sp + (gm + 1), *sz = sy + (gn + 1);
    const int64_t pl = n0 * n1;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n_slab; g += (int64_t)gridDim.x * 256) {
        const int64_t kk = g / pl, rem = g - kk * pl;
        const int32_t j = (int32_t)(rem / n0), i = (int32_t)(rem - (int64_t)j * n0), k = (int32_t)(k0 + kk);
        int px = 0, py = 0, pz = 0;
        while (px + 1 < gm && i >= sx[px + 1]) ++px;
        while (py + 1 < gn && j >= sy[py + 1]) ++py;
        while (pz + 1 < gp && k >= sz[pz + 1]) ++pz;
        const int q = px + gm * (py + gn * pz);
        const int64_t bx = sx[px + 1] - sx[px], by = sy[py + 1] - sy[py];
        const int64_t pos = src[2 * q] + (i - sx[px]) + bx * ((j - sy[py]) + by * (k - src[2 * q + 1]));
        if (TO_NATURAL) nat[g] = stage[pos];
        else stage[pos] = nat[g];
    }
}

int redist_tables(pib_solver *s)
{
    Redist &R = s->redist;
    const int P = s->comm.nranks;
    if (P > PIB_MAX_RANKS) return fail(PIB_ERR_SUP, "box -> slab redistribution: at most %d ranks", PIB_MAX_RANKS);
    R.n_slab = 0;
    for (int f = 0; f < R.nf; ++f) {
        RedistField &F = R.f[f];
        std::vector<int32_t> split;
        for (int d = 0; d < 3; ++d) {
            const int stride = d == 0 ? 1 : (d == 1 ? R.grid[0] : R.grid[0] * R.grid[1]);
            for (int c = 0; c < R.grid[d]; ++c) split.push_back((int32_t)F.box[6 * (size_t)(c * stride) + (size_t)d]);
            split.push_back((int32_t)F.n[d]);
        }
        std::vector<int64_t> src(2 * (size_t)P, 0);
        int64_t off = 0;
        for (int q = 0; q < P; ++q) {
            src[2 * (size_t)q] = off;
            src[2 * (size_t)q + 1] = std::max(F.box[6 * (size_t)q + 2], F.k0);
            off += F.fwd.from(q);
        }
        PIB_HIP(hipMalloc(&F.d_split, sizeof(int32_t) * split.size()));
        PIB_HIP(hipMalloc(&F.d_src, sizeof(int64_t) * src.size()));
        // (everything on the solver's own stream: a null-stream copy or fill waits for every stream of the PROCESS -- with the
        // ranks of a process as threads on one device, for all the other ranks' set-up: 1.2 of the 3.9 s of a 512^3 / 8 setMatrix)
        PIB_HIP(hipMemcpyAsync(F.d_split, split.data(), sizeof(int32_t) * split.size(), hipMemcpyHostToDevice, s->stream));
        PIB_HIP(hipMemcpyAsync(F.d_src, src.data(), sizeof(int64_t) * src.size(), hipMemcpyHostToDevice, s->stream));
        PIB_HIP(hipStreamSynchronize(s->stream));  // (split / src go out of scope)
        const size_t fb = sizeof(double) * (size_t)std::max<int64_t>(F.n_slab, 1);
        PIB_HIP(hipMalloc(&F.stage, fb));
        PIB_HIP(hipMemsetAsync(F.stage, 0, fb, s->stream));
        R.n_slab += F.n_slab;
    }
    const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(R.n_slab, 1);
    PIB_HIP(hipMalloc(&R.b_nat, bytes));
    PIB_HIP(hipMalloc(&R.x_nat, bytes));
    PIB_HIP(hipMemsetAsync(R.b_nat, 0, bytes, s->stream));
    PIB_HIP(hipMemsetAsync(R.x_nat, 0, bytes, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));  // (the inner solver reads them on a stream of its own)
    return 0;
}

// v_box: this rank's entries in box order ([u box | v box | w box] for the velocity system) -> v_nat: its slabs in natural
// order ([u slab | v slab | w slab]); one exchange per field
int redist_forward(pib_solver *s, const double *v_box, double *v_nat, hipStream_t st)
{
    Redist &R = s->redist;
    const int P = s->comm.nranks;
    for (int f = 0; f < R.nf; ++f) {
        RedistField &F = R.f[f];
        double *recv[PIB_MAX_RANKS];
        int64_t off = 0;
        for (int q = 0; q < P; ++q) {
            recv[q] = F.stage + off;
            off += F.fwd.from(q);
        }
        PIB_CHK(comm_exchange_v(s, F.fwd, v_box + F.boff, recv, st));
        if (F.n_slab > 0) {
            const int nb = (int)std::min<int64_t>(4096, (F.n_slab + 255) / 256);
            hipLaunchKernelGGL(k_redist<true>, dim3(nb), dim3(256), 0, st, v_nat + F.soff, F.stage, F.n_slab, F.n[0], F.n[1], F.k0, R.grid[0], R.grid[1],
                               R.grid[2], F.d_split, F.d_src);
            PIB_HIP(hipGetLastError());
        }
    }
    return 0;
}

int redist_backward(pib_solver *s, const double *v_nat, double *v_box, hipStream_t st)
{
    Redist &R = s->redist;
    const int P = s->comm.nranks;
    for (int f = 0; f < R.nf; ++f) {
        RedistField &F = R.f[f];
        if (F.n_slab > 0) {
            const int nb = (int)std::min<int64_t>(4096, (F.n_slab + 255) / 256);
            hipLaunchKernelGGL(k_redist<false>, dim3(nb), dim3(256), 0, st, const_cast<double *>(v_nat) + F.soff, F.stage, F.n_slab, F.n[0], F.n[1], F.k0,
                               R.grid[0], R.grid[1], R.grid[2], F.d_split, F.d_src);
            PIB_HIP(hipGetLastError());
        }
        double *recv[PIB_MAX_RANKS];
        for (int q = 0; q < P; ++q) recv[q] = v_box + F.boff + F.fwd.send_off[(size_t)q];
        PIB_CHK(comm_exchange_v(s, F.bwd, F.stage, recv, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ rows: boxes -> slabs, on the device
// Round 4: the set-up move of the matrix rows (partition.cpp: redist_setup) used to build its fixed-width records, sort every
// received row and assemble the slab's CSR in host loops -- 8 of the 9.9 s of a 512^3 / 8 `setMatrix`.  Now the raw columns
// go to the device once, three kernels do the rest, and only the finished slab CSR comes back (pinned memory) for the
// structure recovery, which reads a few thousand of its rows.
struct BoxGeom {
    int P, rank, wrap;                    // wrap: several ranks -- a column across a periodic seam is taken a vector length away (upload_csr)
    int64_t n0, n1, k0, n_global, row0s, nslab;  // grid (internal layout), first plane / first row / rows of this rank's slab
    int64_t ranges[PIB_MAX_RANKS + 1];    // first row of every rank in the callers' (box) numbering
    int64_t box[PIB_MAX_RANKS][6];        // xs, ys, zs, xm, ym, zm of every rank, internal layout
    int64_t roff[PIB_MAX_RANKS + 1];      // received records: source q's are [roff[q], roff[q + 1])
};

__device__ __forceinline__ int64_t box_natural(const BoxGeom &G, int64_t c)
{
    int q = 0;
    while (q + 1 < G.P && c >= G.ranges[q + 1]) ++q;
    const int64_t l = c - G.ranges[q], x = G.box[q][3], y = G.box[q][4];
    const int64_t i = l % x, j = (l / x) % y, k = l / (x * y);
    return (G.box[q][0] + i) + G.n0 * ((G.box[q][1] + j) + G.n1 * (G.box[q][2] + k));
}
__device__ __forceinline__ int64_t slab_near(const BoxGeom &G, int64_t c)
{
    if (!G.wrap) return c;
    const int64_t lo = G.row0s, hi = G.row0s + G.nslab - 1;
    auto dist = [&](int64_t x) { return x < lo ? lo - x : (x > hi ? x - hi : 0); };
    int64_t best = c;
    if (dist(c - G.n_global) < dist(best)) best = c - G.n_global;
    if (dist(c + G.n_global) < dist(best)) best = c + G.n_global;
    return best;
}
// one record per local row: [length | W natural columns | W values]
template <class RP, class CL>
__global__ __launch_bounds__(256) void k_box_records(BoxGeom G, int64_t n_local, const RP *__restrict__ rp, const CL *__restrict__ cl,
                                                     const double *__restrict__ val, int W, double *__restrict__ rec)
{
    const int64_t RW = 1 + 2 * (int64_t)W;
    for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < n_local; l += (int64_t)gridDim.x * 256) {
        const int64_t a = (int64_t)rp[l] - (int64_t)rp[0], len = (int64_t)rp[l + 1] - (int64_t)rp[l];
        double *rr = rec + l * RW;
        rr[0] = (double)len;
        for (int64_t t = 0; t < len; ++t) {
            rr[1 + t] = (double)box_natural(G, (int64_t)cl[a + t]);
            rr[1 + W + t] = val[a + t];
        }
    }
}
// natural local row of received record g
__device__ __forceinline__ int64_t slab_row_of_record(const BoxGeom &G, int64_t g)
{
    int q = 0;
    while (q + 1 < G.P && g >= G.roff[q + 1]) ++q;
    const int64_t t = g - G.roff[q], x = G.box[q][3], y = G.box[q][4], kf = G.box[q][2] > G.k0 ? G.box[q][2] : G.k0;
    const int64_t i = t % x, j = (t / x) % y, k = kf + t / (x * y);
    return (G.box[q][0] + i) + G.n0 * ((G.box[q][1] + j) + G.n1 * (k - G.k0));
}
__global__ __launch_bounds__(256) void k_slab_lens(BoxGeom G, const double *__restrict__ rec, int W, int64_t *__restrict__ lens,
                                                   unsigned long long *__restrict__ minmax /* [2]: ~min as max of the complement, max */)
{
    const int64_t RW = 1 + 2 * (int64_t)W;
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < G.nslab; g += (int64_t)gridDim.x * 256) {
        const double *rr = rec + g * RW;
        const int64_t len = (int64_t)rr[0];
        lens[slab_row_of_record(G, g)] = len;
        for (int64_t t = 0; t < len; ++t) {
            const long long c = (long long)slab_near(G, (int64_t)rr[1 + t]);
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
        }
    }
    // (offset by 2^62 so that the unsigned atomics order negative near-seam columns correctly)
    const long long off = 1LL << 62;
    if (lo <= hi) {
        atomicMin(&minmax[0], (unsigned long long)(lo + off));
        atomicMax(&minmax[1], (unsigned long long)(hi + off));
    }
}
// every received row sorted by ascending natural column (as MatMPIAIJGetLocalMat would deliver it on slabs) into the slab's CSR:
// local columns for the solver, natural ones for the structure recovery
template <int WMAX>
__global__ __launch_bounds__(256) void k_slab_fill(BoxGeom G, const double *__restrict__ rec, int W, const int64_t *__restrict__ rp,
                                                   int64_t shift, int32_t *__restrict__ col_local, int32_t *__restrict__ col_nat,
                                                   double *__restrict__ val)
{
    const int64_t RW = 1 + 2 * (int64_t)W;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < G.nslab; g += (int64_t)gridDim.x * 256) {
        const double *rr = rec + g * RW;
        const int len = (int)rr[0];
        int64_t c[WMAX];
        double v[WMAX];
#pragma unroll
        for (int t = 0; t < WMAX; ++t) {
            c[t] = t < len ? (int64_t)rr[1 + t] : 0x7fffffffffffffffLL;
            v[t] = t < len ? rr[1 + W + t] : 0.0;
        }
        // (stable insertion sort on registers: W <= WMAX, a handful of entries)
#pragma unroll
        for (int a = 1; a < WMAX; ++a) {
#pragma unroll
            for (int b = a; b > 0; --b) {
                const bool sw = c[b] < c[b - 1];
                const int64_t c0 = sw ? c[b] : c[b - 1], c1 = sw ? c[b - 1] : c[b];
                const double v0 = sw ? v[b] : v[b - 1], v1 = sw ? v[b - 1] : v[b];
                c[b - 1] = c0;
                c[b] = c1;
                v[b - 1] = v0;
                v[b] = v1;
            }
        }
        const int64_t o = rp[slab_row_of_record(G, g)];
#pragma unroll
        for (int t = 0; t < WMAX; ++t)
            if (t < len) {
                col_local[o + t] = (int32_t)(slab_near(G, c[t]) - shift);
                col_nat[o + t] = (int32_t)c[t];
                val[o + t] = v[t];
            }
    }
}
__global__ void k_minmax_init(unsigned long long *mm)
{
    mm[0] = ~0ULL;
    mm[1] = 0ULL;
}
__global__ __launch_bounds__(256) void k_to_i32(const int64_t *__restrict__ in, int32_t *__restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (int32_t)in[i];
}

// device scratch of the set-up move below: released on every exit
struct DevScratchSet {
    std::vector<void *> p;
    ~DevScratchSet()
    {
        for (void *q : p)
            if (q) (void)hipFree(q);
    }
    template <class T>
    int get(T **out, size_t n)
    {
        void *q = nullptr;
        PIB_HIP(hipMalloc(&q, sizeof(T) * std::max<size_t>(n, 1)));
        p.push_back(q);
        *out = static_cast<T *>(q);
        return 0;
    }
};

// The whole move.  In: this rank's rows as the caller handed them over (host), the field's geometry and plans (F), the widest
// row W.  Out: `in` holds the slab's matrix (device CSR, like upload_csr would have made it); h_rp / h_cl / h_vl: the rows the
// structure recovery reads, natural GLOBAL columns, every other row empty, in host memory (released by the caller: free).
// Returns PIB_ERR_SUP when the shape does not fit this path (rows wider than 16 entries, 64-bit sizes): the caller's host path.
int redist_rows_on_device(pib_solver *s, pib_solver *in, const RedistField &F, const int64_t N[3], int64_t n_local, int64_t row0, int64_t n_global,
                          const int64_t *rp64, const int64_t *cl64, const int32_t *rp32, const int32_t *cl32, const double *val, int64_t W,
                          const std::vector<int64_t> &ranges, int32_t **h_rp, int32_t **h_cl, double **h_vl, int64_t *nnz_out)
{
    const int P = s->comm.nranks, rank = s->comm.rank;
    constexpr int WMAX = 16;
    const int64_t base = rp64 ? rp64[0] : (int64_t)rp32[0], nnz_box = (rp64 ? rp64[n_local] : (int64_t)rp32[n_local]) - base;
    // Whether this path takes the shape is decided TOGETHER: a rank that left for the host loops on its own would issue that
    // path's exchange against the others' (sizes are per rank: boxes and slabs differ by a plane).  The slab's entries are
    // bounded by its rows times the widest row, so nothing can turn out too large after the records have travelled.
    {
        const bool fits = W <= WMAX && P <= PIB_MAX_RANKS && n_global < (int64_t)INT32_MAX && nnz_box < (int64_t)INT32_MAX &&
                          F.n_slab * W < (int64_t)INT32_MAX && s->A.val != nullptr && s->A.nnz == nnz_box;
        std::vector<double> mine(1, fits ? 1.0 : 0.0), all;
        PIB_CHK(comm_allgather_host(s, mine, all));
        for (double v : all)
            if (v == 0.0) return PIB_ERR_SUP;
    }
    hipStream_t st = s->stream;
    SetupTrace tr("rows on device", rank);
    BoxGeom G;
    std::memset(&G, 0, sizeof G);
    G.P = P;
    G.rank = rank;
    G.wrap = P > 1 ? 1 : 0;
    G.n0 = N[0];
    G.n1 = N[1];
    G.k0 = F.k0;
    G.n_global = n_global;
    G.nslab = F.n_slab;
    G.row0s = F.k0 * N[0] * N[1];
    for (int q = 0; q <= P; ++q) G.ranges[q] = ranges[(size_t)q];
    for (int q = 0; q < P; ++q)
        for (int k = 0; k < 6; ++k) G.box[q][k] = F.box[6 * (size_t)q + (size_t)k];
    G.roff[0] = 0;
    for (int q = 0; q < P; ++q) G.roff[q + 1] = G.roff[q] + F.fwd.from(q);
    if (G.roff[P] != F.n_slab) return fail(PIB_ERR_LIB, "set_csr: internal error (the boxes do not cover this rank's slab)");
    const int64_t RW = 1 + 2 * W;
    DevScratchSet sc;
    // ---- the caller's rows on the device (the values are there already: upload_csr_general's copy in s->A.val)
    char *d_rp = nullptr, *d_cl = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;
    const size_t rpb = rp64 ? sizeof(int64_t) : sizeof(int32_t), clb = cl64 ? sizeof(int64_t) : sizeof(int32_t);
    PIB_CHK(sc.get(&d_rp, rpb * (size_t)(n_local + 1)));
    PIB_CHK(sc.get(&d_cl, clb * (size_t)std::max<int64_t>(nnz_box, 1)));
    PIB_CHK(sc.get(&d_send, (size_t)(n_local * RW)));
    PIB_CHK(sc.get(&d_recv, (size_t)(F.n_slab * RW)));
    PIB_HIP(hipMemcpyAsync(d_rp, rp64 ? (const void *)rp64 : (const void *)rp32, rpb * (size_t)(n_local + 1), hipMemcpyHostToDevice, st));
    PIB_HIP(hipMemcpyAsync(d_cl, cl64 ? (const void *)(cl64 + base) : (const void *)(cl32 + base), clb * (size_t)nnz_box, hipMemcpyHostToDevice, st));
    const double *d_val = s->A.val;  // (entry order = the caller's)
    const unsigned nbl = (unsigned)std::min<int64_t>(8192, std::max<int64_t>(1, (n_local + 255) / 256));
#define PIB_REC(RPT, CLT) \
    hipLaunchKernelGGL((k_box_records<RPT, CLT>), dim3(nbl), dim3(256), 0, st, G, n_local, reinterpret_cast<const RPT *>(d_rp), reinterpret_cast<const CLT *>(d_cl), d_val, (int)W, d_send)
    if (rp64 && cl64) PIB_REC(int64_t, int64_t);
    else if (rp64) PIB_REC(int64_t, int32_t);
    else if (cl64) PIB_REC(int32_t, int64_t);
    else PIB_REC(int32_t, int32_t);
#undef PIB_REC
    PIB_HIP(hipGetLastError());
    ExchangePlan rows = F.fwd;
    for (auto &c : rows.cnt) c *= RW;
    rows.finish(P, rank);
    std::vector<double *> recv((size_t)P, nullptr);
    for (int q = 0; q < P; ++q) recv[(size_t)q] = d_recv + G.roff[q] * RW;
    PIB_HIP(hipStreamSynchronize(st));
    tr.mark("records built");
    PIB_CHK(comm_exchange_v(s, rows, d_send, recv.data(), st));
    PIB_HIP(hipStreamSynchronize(st));
    tr.mark("records exchanged");
    // ---- row lengths, their running sum, the column range
    int64_t *d_lens = nullptr, *d_rp64 = nullptr;
    unsigned long long *d_mm = nullptr;
    PIB_CHK(sc.get(&d_lens, (size_t)F.n_slab + 1));
    PIB_CHK(sc.get(&d_rp64, (size_t)F.n_slab + 1));
    PIB_CHK(sc.get(&d_mm, 2));
    PIB_HIP(hipMemsetAsync(d_lens, 0, sizeof(int64_t) * ((size_t)F.n_slab + 1), st));
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, st, d_mm);
    const unsigned nbs = (unsigned)std::min<int64_t>(8192, std::max<int64_t>(1, (F.n_slab + 255) / 256));
    hipLaunchKernelGGL(k_slab_lens, dim3(nbs), dim3(256), 0, st, G, d_recv, (int)W, d_lens, d_mm);
    PIB_HIP(hipGetLastError());
    {
        size_t tb = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_lens, d_rp64, (int)(F.n_slab + 1), st) != hipSuccess)
            return fail(PIB_ERR_LIB, "set_csr: device scan failed");
        char *tmp = nullptr;
        PIB_CHK(sc.get(&tmp, tb));
        if (hipcub::DeviceScan::ExclusiveSum(tmp, tb, d_lens, d_rp64, (int)(F.n_slab + 1), st) != hipSuccess)
            return fail(PIB_ERR_LIB, "set_csr: device scan failed");
    }
    int64_t nnz = 0;
    unsigned long long mm[2] = {0, 0};
    PIB_HIP(hipMemcpyAsync(&nnz, d_rp64 + F.n_slab, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PIB_HIP(hipMemcpyAsync(mm, d_mm, sizeof mm, hipMemcpyDeviceToHost, st));
    PIB_HIP(hipStreamSynchronize(st));
    if (nnz >= (int64_t)INT32_MAX || nnz > F.n_slab * W) return fail(PIB_ERR_LIB, "set_csr: internal error (the slab's rows hold more entries than their widest row allows)");
    const int64_t cmin = (int64_t)(mm[0] - (1ULL << 62)), cmax = (int64_t)(mm[1] - (1ULL << 62));
    // ---- the slab's matrix, as upload_csr lays it out
    DeviceCsr &A = in->A;
    A.release();
    in->comm.ring = false;
    vel_stencil_release(in);
    A.n = F.n_slab;
    A.nnz = nnz;
    A.row0 = G.row0s;
    A.n_global = n_global;
    A.ghost_lo = (nnz > 0 && cmin < G.row0s) ? G.row0s - cmin : 0;
    A.ghost_hi = (nnz > 0 && cmax > G.row0s + F.n_slab - 1) ? cmax - (G.row0s + F.n_slab - 1) : 0;
    // (behind the ranks' vote for the device route: PIB_ERR_SUP is the caller's cue to fall back to the host path, which this one rank
    // would then walk alone -- a plain error instead; unreachable while n_global < 2^31 is part of the vote)
    if (A.ghost_lo + A.n + A.ghost_hi >= (int64_t)INT32_MAX) return fail(PIB_ERR_LIB, "set_csr: local column range does not fit 32-bit indices");
    A.rp64 = false;
    PIB_HIP(hipMalloc(&A.col, sizeof(int32_t) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.val, sizeof(double) * (size_t)(nnz + 4)));
    PIB_HIP(hipMalloc(&A.rowptr, sizeof(int32_t) * ((size_t)F.n_slab + 1)));
    PIB_HIP(hipMemsetAsync(A.col, 0, sizeof(int32_t) * (size_t)(nnz + 4), st));
    PIB_HIP(hipMemsetAsync(A.val, 0, sizeof(double) * (size_t)(nnz + 4), st));
    int32_t *d_nat = nullptr;
    PIB_CHK(sc.get(&d_nat, (size_t)nnz));
    hipLaunchKernelGGL(k_slab_fill<WMAX>, dim3(nbs), dim3(256), 0, st, G, d_recv, (int)W, d_rp64, G.row0s - A.ghost_lo, A.col, d_nat, A.val);
    hipLaunchKernelGGL(k_to_i32, dim3(nbs), dim3(256), 0, st, d_rp64, static_cast<int32_t *>(A.rowptr), F.n_slab + 1);
    PIB_HIP(hipGetLastError());
    // ---- for the structure recovery on the host: ONLY the rows it reads (structure.cpp grid_detection_rows: the first rows and
    // three lines through a base cell, a thousand rows), as a CSR whose other rows are empty -- the whole slab was 1.4 GB over
    // PCIe and 1.5 GB of host memory per rank at 512^3 / 8, and 1.3 of the 3.9 s of a box-route setMatrix on the shared test GPU
    tr.mark("slab CSR built");
    std::vector<int64_t> need;
    grid_detection_rows(N, F.k0, F.k1, need);
    *h_rp = static_cast<int32_t *>(std::calloc((size_t)F.n_slab + 1, sizeof(int32_t)));
    std::vector<int32_t> drp(2 * need.size() + 2, 0);
    if (*h_rp == nullptr) return fail(PIB_ERR_MEM, "set_csr: out of host memory");
    PIB_HIP(hipStreamSynchronize(st));
    const int32_t *d_rp32 = static_cast<const int32_t *>(A.rowptr);
    for (size_t t = 0; t < need.size(); ++t)  // (begin, end) of every needed row: adjacent offsets, 8 bytes each
        PIB_HIP(hipMemcpyAsync(&drp[2 * t], d_rp32 + need[t], 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PIB_HIP(hipStreamSynchronize(st));
    int64_t keep = 0;
    for (size_t t = 0; t < need.size(); ++t) keep += drp[2 * t + 1] - drp[2 * t];
    *h_cl = static_cast<int32_t *>(std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(keep, 1)));
    *h_vl = static_cast<double *>(std::malloc(sizeof(double) * (size_t)std::max<int64_t>(keep, 1)));
    if (*h_cl == nullptr || *h_vl == nullptr) return fail(PIB_ERR_MEM, "set_csr: out of host memory");
    {
        int64_t at = 0;
        size_t t = 0;
        for (int64_t l = 0; l < F.n_slab; ++l) {  // the compact row offsets: empty rows everywhere else
            (*h_rp)[l] = (int32_t)at;
            if (t < need.size() && need[t] == l) {
                const int32_t b0 = drp[2 * t], len = drp[2 * t + 1] - drp[2 * t];
                if (len > 0) {
                    PIB_HIP(hipMemcpyAsync(*h_cl + at, d_nat + b0, sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost, st));
                    PIB_HIP(hipMemcpyAsync(*h_vl + at, A.val + b0, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, st));
                }
                at += len;
                ++t;
            }
        }
        (*h_rp)[F.n_slab] = (int32_t)at;
    }
    PIB_HIP(hipStreamSynchronize(st));  // (this stream only: a plain hipMemcpy waits for every stream of the process)
    tr.mark("the rows the structure recovery reads copied to the host");
    *nnz_out = nnz;
    return 0;
}

}  // namespace pib
