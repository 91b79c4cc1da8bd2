// kernels_spmv.hip -- K1: fp64 CSR SpMV for gfx950 (MI355X), plus diagonal
// extraction and halo-row classification.
//
// Replaces the MatMult that PETSc / AmgX run inside KSPSolve / AmgXSolver::solve
// (src/linsolver/linsolverksp.cpp:92, src/linsolver/linsolveramgx.cpp:96) and
// the applications' own MatMult calls (applications/navierstokes/navierstokes.cpp:
// 442,490,549,592).
//
// Design (HBM-bound, 0.13 flop/B; MFMA unused -- no dense contraction):
//   * default kernel k_spmv_lds ("LDS transpose"): a 256-thread workgroup owns
//     256 consecutive rows.  Phase 1 copies the rows' contiguous val/col span
//     into LDS with 16 B/8 B-per-lane fully coalesced loads (HBM sees
//     val/col/rowptr exactly once).  Phase 2: thread t owns row t, reads its
//     (val, col) pairs back from LDS (stride-7 accesses: bank-conflict free) and
//     gathers x with the ROW-PER-LANE mapping, so one gather instruction reads
//     ~64 consecutive x values (4-5 cache lines) instead of 9 rows x 7
//     diagonals (10+ lines): the vector-L1 tag rate, not HBM, was what capped
//     the entry-per-lane kernel of round 1 at 50 %.
//   * one chunk per workgroup, workgroups dispatched in sequence order: the
//     chip sweeps a moving window of addresses (measured: persistent
//     grid-stride loops stream 10-15 % slower on MI355X).
//   * chunk order: with a 3-D grid hint the chunk sequence is "plane-
//     interleaved tiles" (tile of tj grid lines, all z planes of the tile
//     before the next tile): the +-nx*ny neighbours of a chunk are the chunks
//     just before / after it in the sequence, so x is fetched from HBM ~once
//     (rocprof FETCH_SIZE: 13.8 GB vs 15.4 GB in natural order at 512^3).
//   * summation order per row is sequential in CSR order with rounded
//     products (no FMA contraction: the library is built -ffp-contract=off),
//     which the oracle reproduces -> bit-exact parity.
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include <cstring>

#include "pib_internal.hpp"

namespace pib {

constexpr int SPMV_GRID = 2048;                   // 256 CUs x 8 workgroups

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over a 256-thread block; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double *sh /* >= 4 */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

// ----------------------------------------------------------------------------
// chunk order (see header): sequence position -> chunk id
struct ChunkOrder {
    int tiled;        // 0: natural
    int64_t cpp;      // chunks per plane
    int64_t cpt;      // chunks per tile
    int64_t nplanes;
};
__device__ __forceinline__ int64_t chunk_of(const ChunkOrder &o, int64_t seq)
{
    if (!o.tiled) return seq;
    const int64_t per_tile = o.cpt * o.nplanes;
    const int64_t tile = seq / per_tile, rem = seq % per_tile;
    const int64_t k = rem / o.cpt, w = rem % o.cpt;
    return k * o.cpp + tile * o.cpt + w;
}

constexpr int LDS_ROWS = 256;

// CAP = LDS capacity in entries (a multiple of 4).  1284: 5-point rows, 1796: 7-point rows (7 workgroups per CU instead
// of 6), 2052: anything else (rows longer than a tile are walked tile by tile).
// Phase 1 is LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS at wave base + lane * 16, no staging
// registers, no ds_write pass): two values or four column indices per lane and instruction.
template <typename RP, bool DOT, int CAP>
__global__ __launch_bounds__(256) void k_spmv_lds(const Scalars *__restrict__ S, int64_t r_begin, int64_t r_end,
                                                  const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const double *__restrict__ xg,
                                                  int64_t ghost_lo, double *__restrict__ y, double *__restrict__ part,
                                                  ChunkOrder ord)
{
    if (S != nullptr && S->done) return;
    __shared__ __attribute__((aligned(16))) double vals[CAP];
    __shared__ __attribute__((aligned(16))) int32_t cols[CAP];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t nchunks = (r_end - r_begin + LDS_ROWS - 1) / LDS_ROWS;
    // workgroup b runs on XCD b%8 (observed; speed only): XCD x walks the contiguous
    // sequence range [x*cpx, (x+1)*cpx) in dispatch order
    const int64_t cpx = (nchunks + 7) >> 3;
    const int64_t sq = (int64_t)(blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    double dacc = 0.0, ysum = 0.0;
    int64_t yrow = -1;
    if ((int64_t)(blockIdx.x >> 3) < cpx && sq < nchunks) {
        const int64_t c = chunk_of(ord, sq);
        const int64_t r0 = r_begin + c * LDS_ROWS;
        const int nr = (int)((r_end - r0 < LDS_ROWS) ? (r_end - r0) : LDS_ROWS);
        // the span [p0, p1) of the chunk from two workgroup-uniform (scalar) loads: phase 1 starts at once; the
        // per-row offsets go straight to registers and are first needed in phase 2
        const RP p0 = rowptr[r0], p1 = rowptr[r0 + nr];
        RP rs = 0, re = 0;
        if (tid < nr) {
            rs = rowptr[r0 + tid];
            re = rowptr[r0 + tid + 1];
        }
        double sum = 0.0, xd = 0.0;
        bool have_diag = false;
        const int32_t dcol = (int32_t)(ghost_lo + r0 + tid);
        const RP a0 = p0 & ~(RP)3;  // val is 16-byte aligned at even entries, col at multiples of four
        for (RP t0 = a0; t0 < p1; t0 += CAP) {
#pragma unroll
            for (int u = 0; u < (CAP + 511) / 512; ++u) {
                const RP q = t0 + 2 * (tid + u * 256);
                if (q < p1 && q - t0 < CAP)
                    __builtin_amdgcn_global_load_lds((const void *)(val + q), (__attribute__((address_space(3))) void *)&vals[(int)(q - t0)],
                                                     16, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < (CAP + 1023) / 1024; ++u) {
                const RP q = t0 + 4 * (tid + u * 256);
                if (q < p1 && q - t0 < CAP)
                    __builtin_amdgcn_global_load_lds((const void *)(col + q), (__attribute__((address_space(3))) void *)&cols[(int)(q - t0)],
                                                     16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const RP lo = (rs > t0) ? rs : t0;
            const RP hi = (re < t0 + CAP) ? re : (t0 + CAP);
            for (RP p = lo; p < hi; p += 8) {
                double vv[8], xx[8];
                int32_t cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = p + u < hi;
                    cc[u] = ok ? cols[(int)(p - t0) + u] : 0;
                    vv[u] = ok ? vals[(int)(p - t0) + u] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xx[u] = (p + u < hi) ? xg[cc[u]] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p + u < hi) {
                        sum = sum + vv[u] * xx[u];
                        if (DOT && cc[u] == dcol) {
                            xd = xx[u];  // x of this row: already gathered with the diagonal entry
                            have_diag = true;
                        }
                    }
            }
            if (t0 + CAP < p1) __syncthreads();
        }
        if (DOT && tid < nr) {
            if (!have_diag) xd = xg[dcol];
            dacc = xd * sum;
        }
        ysum = sum;
        yrow = (tid < nr) ? r0 + tid : -1;
    }
    if (DOT) {
        const double s = block_sum_256(dacc, red);
        if (tid == 0) part[blockIdx.x] = s;
    }
    if (yrow >= 0) y[yrow] = ysum;  // (the rows' store last: see k_spmv_lds_pattern)
}

// The same product from COLUMN CODES (DeviceCsr::code / dict): phase 1 brings the values and one byte per entry into LDS
// (16 codes per lane and instruction), phase 2 turns a code into the column -- row + dictionary[block of the row][code] -- and
// gathers x as above.  9 instead of 12 bytes per entry from HBM (104 -> 83 B per 7-point row); the products, their order
// and the rounding are those of k_spmv_lds: bit-identical.  The dictionaries belong to the 256-row blocks of the MATRIX; a
// chunk that does not start on a block (a row range of a split product) reads two of them.
// CAP: a multiple of 16 (the span starts on a multiple of 16 entries).  1296: 5-point rows, 1808: 7-point rows.
template <typename RP, bool DOT, int CAP>
__global__ __launch_bounds__(256) void k_spmv_lds_coded(const Scalars *__restrict__ S, int64_t r_begin, int64_t r_end,
                                                        const RP *__restrict__ rowptr, const uint8_t *__restrict__ code,
                                                        const int32_t *__restrict__ dict, const double *__restrict__ val,
                                                        const double *__restrict__ xg, int64_t ghost_lo, double *__restrict__ y,
                                                        double *__restrict__ part, ChunkOrder ord)
{
    if (S != nullptr && S->done) return;
    constexpr int ND = DeviceCsr::CODE_DICT;
    __shared__ __attribute__((aligned(16))) double vals[CAP];
    __shared__ __attribute__((aligned(16))) uint8_t codes[CAP];
    __shared__ int32_t dl[2 * ND];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t nchunks = (r_end - r_begin + LDS_ROWS - 1) / LDS_ROWS;
    const int64_t cpx = (nchunks + 7) >> 3;
    const int64_t sq = (int64_t)(blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    double dacc = 0.0, ysum = 0.0;
    int64_t yrow = -1;
    if ((int64_t)(blockIdx.x >> 3) < cpx && sq < nchunks) {
        const int64_t c = chunk_of(ord, sq);
        const int64_t r0 = r_begin + c * LDS_ROWS;
        const int nr = (int)((r_end - r0 < LDS_ROWS) ? (r_end - r0) : LDS_ROWS);
        const RP p0 = rowptr[r0], p1 = rowptr[r0 + nr];
        RP rs = 0, re = 0;
        if (tid < nr) {
            rs = rowptr[r0 + tid];
            re = rowptr[r0 + tid + 1];
        }
        const int64_t blk0 = r0 >> 8;
        if (tid < 2 * ND) dl[tid] = dict[blk0 * ND + tid];  // (the array has one block of slack at its end)
        const int boff = (int)(((r0 + tid) >> 8) - blk0) * ND;
        double sum = 0.0, xd = 0.0;
        bool have_diag = false;
        const int32_t dcol = (int32_t)(ghost_lo + r0 + tid);
        const RP a0 = p0 & ~(RP)15;
        for (RP t0 = a0; t0 < p1; t0 += CAP) {
#pragma unroll
            for (int u = 0; u < (CAP + 511) / 512; ++u) {
                const RP q = t0 + 2 * (tid + u * 256);
                if (q < p1 && q - t0 < CAP)
                    __builtin_amdgcn_global_load_lds((const void *)(val + q), (__attribute__((address_space(3))) void *)&vals[(int)(q - t0)],
                                                     16, 0, 0);
            }
            {
                const RP q = t0 + 16 * tid;
                if (q < p1 && q - t0 < CAP)
                    __builtin_amdgcn_global_load_lds((const void *)(code + q), (__attribute__((address_space(3))) void *)&codes[(int)(q - t0)],
                                                     16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const RP lo = (rs > t0) ? rs : t0;
            const RP hi = (re < t0 + CAP) ? re : (t0 + CAP);
            for (RP p = lo; p < hi; p += 8) {
                double vv[8], xx[8];
                int32_t cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = p + u < hi;
                    cc[u] = ok ? dcol + dl[boff + codes[(int)(p - t0) + u]] : 0;
                    vv[u] = ok ? vals[(int)(p - t0) + u] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) xx[u] = (p + u < hi) ? xg[cc[u]] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (p + u < hi) {
                        sum = sum + vv[u] * xx[u];
                        if (DOT && cc[u] == dcol) {
                            xd = xx[u];
                            have_diag = true;
                        }
                    }
            }
            if (t0 + CAP < p1) __syncthreads();
        }
        if (DOT && tid < nr) {
            if (!have_diag) xd = xg[dcol];
            dacc = xd * sum;
        }
        ysum = sum;
        yrow = (tid < nr) ? r0 + tid : -1;
    }
    if (DOT) {
        const double s = block_sum_256(dacc, red);
        if (tid == 0) part[blockIdx.x] = s;
    }
    if (yrow >= 0) y[yrow] = ysum;  // (the rows' store last: see k_spmv_lds_pattern)
}

// The product from ROW PATTERNS (DeviceCsr::pat_id / pat_tab / pat_len): a workgroup takes one 256-row block of the matrix, brings
// the block's values into LDS (phase 1 as above), its pattern table (PAT_N lists of up to PAT_LEN offsets col - row) and one
// byte per row; a row's first entry is the exclusive prefix sum of the pattern lengths over the block's rows (wave scans), its
// columns are row + offset.  Nothing per entry but the value and nothing per row but one byte come from HBM (blocks with the same
// table share it: the few distinct tables of a stencil matrix stay in the L2): 8 nnz + 17 n + 12 B per block, 73 B per 7-point
// row against 104.  Products, order and rounding are those of k_spmv_lds.
template <typename RP, bool DOT, int CAP>
__global__ __launch_bounds__(256) void k_spmv_lds_pattern(const Scalars *__restrict__ S, int64_t r_begin, int64_t r_end, const RP *__restrict__ rowptr,
                                                          const uint8_t *__restrict__ pat_id, const int32_t *__restrict__ pat_blk,
                                                          const int32_t *__restrict__ pat_tab, const uint8_t *__restrict__ pat_len,
                                                          const double *__restrict__ val,
                                                          const double *__restrict__ xg, int64_t ghost_lo, double *__restrict__ y,
                                                          double *__restrict__ part, ChunkOrder ord)
{
    if (S != nullptr && S->done) return;
    // CAP: the block's entries + 2 (the span starts on an even entry): 1296 for 5-point rows, 1808 for 7-point rows, 2064 for anything
    // of up to PAT_LEN entries a row (less LDS: one more workgroup per CU)
    constexpr int PN = DeviceCsr::PAT_N, PL = DeviceCsr::PAT_LEN;
    __shared__ __attribute__((aligned(16))) double vals[CAP];
    __shared__ __attribute__((aligned(16))) int32_t tab[PN * PL];
    __shared__ int lens[PN];
    __shared__ int wsum[4];
    const int tid = threadIdx.x;
    const int64_t nchunks = (r_end - r_begin + LDS_ROWS - 1) / LDS_ROWS;
    const int64_t cpx = (nchunks + 7) >> 3;
    const int64_t sq = (int64_t)(blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
    double dacc = 0.0, ysum = 0.0;
    int64_t yrow = -1;
    if ((int64_t)(blockIdx.x >> 3) < cpx && sq < nchunks) {
        const int64_t c = chunk_of(ord, sq);
        const int64_t r0 = r_begin + c * LDS_ROWS;  // a multiple of 256 (the launcher sees to it)
        const int nr = (int)((r_end - r0 < LDS_ROWS) ? (r_end - r0) : LDS_ROWS);
        const int64_t blk = r0 >> 8;
        const RP p0 = rowptr[r0], p1 = rowptr[r0 + nr];
        const RP a0 = p0 & ~(RP)1;
#pragma unroll
        for (int u = 0; u < (CAP + 511) / 512; ++u) {
            const RP q = a0 + 2 * (tid + u * 256);
            if (q < p1) __builtin_amdgcn_global_load_lds((const void *)(val + q), (__attribute__((address_space(3))) void *)&vals[(int)(q - a0)], 16, 0, 0);
        }
        const int id = (tid < nr) ? (int)pat_id[r0 + tid] : 0;
        const int64_t tix = pat_blk[blk];
        if (tid < PN * PL) tab[tid] = pat_tab[tix * (PN * PL) + tid];
        if (tid < PN) lens[tid] = pat_len[tix * PN + tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int len = (tid < nr) ? lens[id] : 0;
        // exclusive prefix sum of len over the block's rows
        int incl = len;
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < w) before += wsum[k];
        const int start = (int)(p0 - a0) + before + incl - len;
        const int32_t dcol = (int32_t)(ghost_lo + r0 + tid);
        int32_t dd[PL];
#pragma unroll
        for (int u = 0; u < PL; ++u) dd[u] = tab[id * PL + u];
        double vv[PL], xx[PL];
#pragma unroll
        for (int u = 0; u < PL; ++u) vv[u] = (u < len) ? vals[start + u] : 0.0;
#pragma unroll
        for (int u = 0; u < PL; ++u) xx[u] = (u < len) ? xg[dcol + dd[u]] : 0.0;
        double sum = 0.0, xd = 0.0;
        bool have_diag = false;
        // (selects, not branches: written as nested ifs the fused-dot form compiled to eight branches behind the gathers and
        // ran 1.93 against 1.77 ms)
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const double t = sum + vv[u] * xx[u];
            sum = (u < len) ? t : sum;
            if (DOT) {
                const bool dg = (u < len) && dd[u] == 0;
                xd = dg ? xx[u] : xd;
                have_diag = have_diag || dg;
            }
        }
        if (DOT && tid < nr) {
            if (!have_diag) xd = xg[dcol];
            dacc = xd * sum;
        }
        ysum = sum;
        yrow = (tid < nr) ? r0 + tid : -1;
    }
    if (DOT) {
        // one partial per WAVE (no barrier behind the rows' work); k_reduce_partials<4> adds a workgroup's four as block_sum_256
        // did, (w0 + w1) + (w2 + w3): the same sums.  The partial is stored BEFORE the rows' results: as the wave's last
        // instruction the one-lane store cost 0.23 ms per 512^3 launch (1.98 against 1.75 ms, tools/_dot_probe.py -- a wave that
        // ends on a small store holds its slot and the workgroup's LDS for that store's round trip)
        // ... stored XCD by XCD (workgroup b runs on XCD b % 8: the partials of one XCD's workgroups are neighbours, so that the
        // 128-byte lines fill up inside ONE L2 instead of being written back in eighths by eight of them)
        const double s = wave_sum(dacc);
        if ((tid & 63) == 0) part[4 * ((int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) + (tid >> 6)] = s;
    }
    if (yrow >= 0) y[yrow] = ysum;  // (the rows' store LAST: the wave's small store of its partial is not the one it ends on)
}

// Set-up of the row patterns: one workgroup per 256-row block.  Round after round the first row without a pattern number
// publishes its list of offsets, every row with the same list takes the round's number; more than PAT_N rounds, or a row of
// more than PAT_LEN entries, and the matrix keeps its per-entry form.
template <typename RP>
__global__ __launch_bounds__(256) void k_build_patterns(int64_t n, int64_t ghost_lo, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        uint8_t *__restrict__ pat_id, int32_t *__restrict__ pat_tab, uint8_t *__restrict__ pat_len,
                                                        int *__restrict__ failed)
{
    constexpr int PN = DeviceCsr::PAT_N, PL = DeviceCsr::PAT_LEN;
    __shared__ int leader;
    __shared__ int cur[PL + 1];
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    int len = 0;
    int d[PL];
#pragma unroll
    for (int u = 0; u < PL; ++u) d[u] = 0;
    bool bad = false;
    if (r < n) {
        const RP rs = rowptr[r], re = rowptr[r + 1];
        if (re - rs > (RP)PL) bad = true;
        else {
            len = (int)(re - rs);
            for (int u = 0; u < len; ++u) {
                const int64_t v = (int64_t)col[rs + u] - (ghost_lo + r);
                if (v < -2147483647LL || v > 2147483647LL) bad = true;
                d[u] = (int)v;
            }
        }
    }
    if (tid < PN * PL) pat_tab[(int64_t)blockIdx.x * (PN * PL) + tid] = 0;
    if (tid < PN) pat_len[(int64_t)blockIdx.x * PN + tid] = 0;
    bool open = (r < n) && !bad;
    int id = 0;
    for (int round = 0; round <= PN; ++round) {
        if (tid == 0) leader = 1 << 30;
        __syncthreads();
        if (open) atomicMin(&leader, tid);
        __syncthreads();
        const int L = leader;
        if (L == (1 << 30)) break;  // every row has its number
        if (round == PN) {
            bad = true;
            break;
        }
        if (tid == L) {
            cur[PL] = len;
#pragma unroll
            for (int u = 0; u < PL; ++u) cur[u] = d[u];
            pat_len[(int64_t)blockIdx.x * PN + round] = (uint8_t)len;
#pragma unroll
            for (int u = 0; u < PL; ++u) pat_tab[(int64_t)blockIdx.x * (PN * PL) + round * PL + u] = d[u];
        }
        __syncthreads();
        if (open) {
            bool same = cur[PL] == len;
#pragma unroll
            for (int u = 0; u < PL; ++u) same = same && (u >= len || cur[u] == d[u]);
            if (same) {
                id = round;
                open = false;
            }
        }
        __syncthreads();
    }
    if (bad) atomicAdd(failed, 1);
    pat_id[r] = (uint8_t)id;  // (the array covers whole blocks)
}

// Set-up of the codes: one workgroup per 256-row block.  The block's distinct offsets col - row go through a 256-slot hash
// table in LDS (compare-and-swap insertion), the used slots are numbered in slot order (the dictionary), every entry gets
// its offset's number.  A block with more than CODE_DICT offsets raises *too_many and the matrix keeps its plain columns.
template <typename RP>
__global__ __launch_bounds__(256) void k_build_codes(int64_t n, int64_t ghost_lo, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     uint8_t *__restrict__ code, int32_t *__restrict__ dict, int *__restrict__ too_many)
{
    constexpr int ND = DeviceCsr::CODE_DICT;
    constexpr int EMPTY = INT_MIN;
    __shared__ int table[256];
    __shared__ int number[256];
    __shared__ int count;
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    table[tid] = EMPTY;
    if (tid == 0) count = 0;
    __syncthreads();
    RP rs = 0, re = 0;
    if (r < n) rs = rowptr[r], re = rowptr[r + 1];
    const int64_t base = ghost_lo + r;
    auto slot_of = [&](int d, bool insert) {
        unsigned h = ((unsigned)d * 2654435761u) >> 24;
        for (int probe = 0; probe < 256; ++probe, h = (h + 1) & 255u) {
            if (insert) {
                const int old = atomicCAS(&table[h], EMPTY, d);
                if (old == EMPTY || old == d) return (int)h;
            } else if (table[h] == d)
                return (int)h;
        }
        return -1;
    };
    bool lost = false;
    for (RP p = rs; p < re; ++p) {
        const int64_t d = (int64_t)col[p] - base;
        if (d == (int64_t)EMPTY || slot_of((int)d, true) < 0) lost = true;
    }
    if (lost) atomicAdd(&count, 1000);
    __syncthreads();
    if (tid == 0) {
        int c = count;
        for (int k = 0; k < 256; ++k) number[k] = (table[k] != EMPTY) ? c++ : -1;
        count = c;
    }
    __syncthreads();
    const int nd = count;
    if (nd > ND) {
        if (tid == 0) atomicMax(too_many, nd);
        return;
    }
    if (tid < ND) dict[(int64_t)blockIdx.x * ND + tid] = 0;
    __syncthreads();
    if (number[tid] >= 0) dict[(int64_t)blockIdx.x * ND + number[tid]] = table[tid];
    for (RP p = rs; p < re; ++p) code[p] = (uint8_t)number[slot_of((int)((int64_t)col[p] - base), false)];
}

// largest number of entries in a 256-row chunk (chooses the LDS capacity)
template <typename RP>
__global__ void k_max_chunk_nnz(int64_t n, const RP *__restrict__ rowptr, unsigned long long *__restrict__ out)
{
    const int64_t nchunks = (n + LDS_ROWS - 1) / LDS_ROWS;
    unsigned long long m = 0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r0 = c * LDS_ROWS, r1 = (r0 + LDS_ROWS < n) ? r0 + LDS_ROWS : n;
        const unsigned long long v = (unsigned long long)(rowptr[r1] - rowptr[r0]);
        m = v > m ? v : m;
    }
    atomicMax(out, m);
}

// stage 1 of the p.Ap reduction: `nin` per-workgroup partials -> <= 1024 partial sums (fixed order)
// (QUAD = 4: every input is the four per-wave partials of one workgroup, added as block_sum_256 adds them)
// (... of workgroup i, which the pattern kernel stores XCD by XCD: at (i % 8) * (nin / 8) + i / 8)
template <int QUAD = 1>
__global__ __launch_bounds__(256) void k_reduce_partials(const Scalars *__restrict__ S, const double *__restrict__ in,
                                                         int64_t nin, double *__restrict__ out)
{
    if (S != nullptr && S->done) return;
    __shared__ double red[4];
    const int64_t per = (nin + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per, e = (b + per < nin) ? b + per : nin;
    double v = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        if (QUAD == 4) {
            const int64_t at = (i & 7) * (nin >> 3) + (i >> 3);
            const double2 lo = *reinterpret_cast<const double2 *>(in + 4 * at), hi = *reinterpret_cast<const double2 *>(in + 4 * at + 2);
            v += (lo.x + lo.y) + (hi.x + hi.y);
        } else
            v += in[i];
    }
    const double s = block_sum_256(v, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

int spmv_launch_blocks() { return SPMV_GRID; }

// chunk order for rows [r_begin, r_end) given the registered grid (3-D only; the rows must start on a plane)
static ChunkOrder make_order(const pib_solver *s, int64_t r_begin, int64_t r_end)
{
    ChunkOrder o{0, 0, 0, 0};
    if (!s->has_grid || s->levels.empty()) return o;
    const GridLevel &g = s->levels[0];
    const int64_t nx = g.n[0], ny = g.n[1], plane = nx * ny;
    if (ny < 2 || plane % LDS_ROWS != 0 || r_begin % plane != 0 || (r_end - r_begin) % plane != 0) return o;
    const int64_t nplanes = (r_end - r_begin) / plane;
    if (nplanes < 3) return o;
    for (int64_t tj : {64, 32, 16, 8, 4, 2, 1}) {
        if (ny % tj == 0 && (tj * nx) % LDS_ROWS == 0) {
            o.tiled = 1;
            o.cpp = plane / LDS_ROWS;
            o.cpt = tj * nx / LDS_ROWS;
            o.nplanes = nplanes;
            break;
        }
    }
    return o;
}

// y[r_begin:r_end) = (A x)[r_begin:r_end).  x_owned points at the owned part of a ghost-padded vector.
// If dot_part != nullptr, x.y over the row range is reduced into `*n_part` partial sums written there
// (fixed order: deterministic).
int spmv_rows(pib_solver *s, const double *x_owned, double *y, int64_t r_begin, int64_t r_end, double *dot_part,
              bool guarded, hipStream_t st)
{
    const DeviceCsr &A = s->A;
    if (r_end <= r_begin) {
        if (dot_part) PIB_HIP(hipMemsetAsync(dot_part, 0, SPMV_GRID * sizeof(double), st));
        return 0;
    }
    const double *xg = x_owned - A.ghost_lo;
    const Scalars *S = guarded ? s->d_s : nullptr;
    {
        const int64_t nchunks = (r_end - r_begin + LDS_ROWS - 1) / LDS_ROWS;
        const int64_t grid = ((nchunks + 7) / 8) * 8;
        const ChunkOrder ord = make_order(s, r_begin, r_end);
        double *big = nullptr;
        if (dot_part) {
            if (s->spmv_part_cap < 4 * grid) {  // (four per workgroup for the pattern kernel's per-wave partials)
                if (s->d_spmv_part) PIB_HIP(hipFree(s->d_spmv_part));
                s->d_spmv_part = nullptr;
                PIB_HIP(hipMalloc(&s->d_spmv_part, sizeof(double) * (size_t)(4 * grid)));
                s->spmv_part_cap = 4 * grid;
            }
            big = s->d_spmv_part;
        }
#define PIB_LAUNCH_LDS(RP, CAP)                                                                                   \
    do {                                                                                                          \
        if (dot_part)                                                                                             \
            hipLaunchKernelGGL((k_spmv_lds<RP, true, CAP>), dim3((unsigned)grid), dim3(256), 0, st, S, r_begin,  \
                               r_end, (const RP *)A.rowptr, A.col, A.val, xg, A.ghost_lo, y, big, ord);           \
        else                                                                                                      \
            hipLaunchKernelGGL((k_spmv_lds<RP, false, CAP>), dim3((unsigned)grid), dim3(256), 0, st, S, r_begin, \
                               r_end, (const RP *)A.rowptr, A.col, A.val, xg, A.ghost_lo, y, (double *)nullptr, ord); \
    } while (0)
#define PIB_LAUNCH_CODED(RP, CAP)                                                                                         \
    do {                                                                                                                  \
        if (dot_part)                                                                                                     \
            hipLaunchKernelGGL((k_spmv_lds_coded<RP, true, CAP>), dim3((unsigned)grid), dim3(256), 0, st, S, r_begin, r_end, \
                               (const RP *)A.rowptr, A.code, A.dict, A.val, xg, A.ghost_lo, y, big, ord);                 \
        else                                                                                                              \
            hipLaunchKernelGGL((k_spmv_lds_coded<RP, false, CAP>), dim3((unsigned)grid), dim3(256), 0, st, S, r_begin, r_end, \
                               (const RP *)A.rowptr, A.code, A.dict, A.val, xg, A.ghost_lo, y, (double *)nullptr, ord);   \
    } while (0)
        if (A.patterned && (r_begin & 255) == 0) {
#define PIB_LAUNCH_PATTERN(RP, CAP)                                                                                                                  \
    do {                                                                                                                                             \
        if (dot_part)                                                                                                                                \
            hipLaunchKernelGGL((k_spmv_lds_pattern<RP, true, CAP>), dim3((unsigned)grid), dim3(256), 0, st, S, r_begin, r_end, (const RP *)A.rowptr, \
                               A.pat_id, A.pat_blk, A.pat_tab, A.pat_len, A.val, xg, A.ghost_lo, y, big, ord);                                     \
        else                                                                                                                                         \
            hipLaunchKernelGGL((k_spmv_lds_pattern<RP, false, CAP>), dim3((unsigned)grid), dim3(256), 0, st, S, r_begin, r_end, (const RP *)A.rowptr, \
                               A.pat_id, A.pat_blk, A.pat_tab, A.pat_len, A.val, xg, A.ghost_lo, y, (double *)nullptr, ord);                       \
    } while (0)
            const int64_t need2 = A.max_chunk_nnz + 2;
            if (need2 <= 1296) {
                if (A.rp64) PIB_LAUNCH_PATTERN(int64_t, 1296); else PIB_LAUNCH_PATTERN(int32_t, 1296);
            } else if (need2 <= 1808) {
                if (A.rp64) PIB_LAUNCH_PATTERN(int64_t, 1808); else PIB_LAUNCH_PATTERN(int32_t, 1808);
            } else {
                if (A.rp64) PIB_LAUNCH_PATTERN(int64_t, 2064); else PIB_LAUNCH_PATTERN(int32_t, 2064);
            }
#undef PIB_LAUNCH_PATTERN
            PIB_HIP(hipGetLastError());
            if (dot_part) {
                hipLaunchKernelGGL(k_reduce_partials<4>, dim3(SPMV_GRID), dim3(256), 0, st, S, big, grid, dot_part);
                PIB_HIP(hipGetLastError());
            }
            s->counters[0]++;
            return 0;
        }
        if (A.coded) {
            const int64_t need16 = A.max_chunk_nnz + 15;  // the span is widened to a start that is a multiple of 16
            if (need16 <= 1296) {
                if (A.rp64) PIB_LAUNCH_CODED(int64_t, 1296); else PIB_LAUNCH_CODED(int32_t, 1296);
            } else if (need16 <= 1808) {
                if (A.rp64) PIB_LAUNCH_CODED(int64_t, 1808); else PIB_LAUNCH_CODED(int32_t, 1808);
            } else {
                if (A.rp64) PIB_LAUNCH_CODED(int64_t, 2064); else PIB_LAUNCH_CODED(int32_t, 2064);
            }
            PIB_HIP(hipGetLastError());
            if (dot_part) {
                hipLaunchKernelGGL(k_reduce_partials<1>, dim3(SPMV_GRID), dim3(256), 0, st, S, big, grid, dot_part);
                PIB_HIP(hipGetLastError());
            }
            s->counters[0]++;
            return 0;
        }
#undef PIB_LAUNCH_CODED
        // +3: the span is widened to a start that is a multiple of four
        const int64_t need = A.max_chunk_nnz + 3;
        if (need <= 1284) {
            if (A.rp64) PIB_LAUNCH_LDS(int64_t, 1284); else PIB_LAUNCH_LDS(int32_t, 1284);
        } else if (need <= 1796) {
            if (A.rp64) PIB_LAUNCH_LDS(int64_t, 1796); else PIB_LAUNCH_LDS(int32_t, 1796);
        } else {
            if (A.rp64) PIB_LAUNCH_LDS(int64_t, 2052); else PIB_LAUNCH_LDS(int32_t, 2052);
        }
#undef PIB_LAUNCH_LDS
        PIB_HIP(hipGetLastError());
        if (dot_part) {
            // SPMV_GRID partial sums out, like the persistent kernels
            hipLaunchKernelGGL(k_reduce_partials<1>, dim3(SPMV_GRID), dim3(256), 0, st, S, big, grid, dot_part);
            PIB_HIP(hipGetLastError());
        }
        s->counters[0]++;
        return 0;
    }
}

// ---------------------------------------------------------------- 1/diag
template <typename RP>
__global__ void k_extract_dinv(int64_t n, int64_t ghost_lo, const RP *__restrict__ rowptr,
                               const int32_t *__restrict__ col, const double *__restrict__ val,
                               double *__restrict__ dinv, int *__restrict__ missing)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double d = 0.0;
    bool found = false;
    for (RP p = rowptr[r]; p < rowptr[r + 1]; ++p)
        if ((int64_t)col[p] == r + ghost_lo) {
            d = d + val[p];
            found = true;
        }
    if (!found || d == 0.0) {
        atomicAdd(missing, 1);
        dinv[r] = 0.0;
    } else {
        dinv[r] = 1.0 / d;
    }
}

// DeviceCsr::pat_* of the matrix just set (cfg.compress_columns >= 2); a matrix with a row of more than PAT_LEN entries or a
// block of more than PAT_N distinct rows keeps its per-entry codes.
// 64-bit FNV-1a over a block's table and lengths
__global__ __launch_bounds__(256) void k_hash_tables(int64_t nblk, const int32_t *__restrict__ tab, const uint8_t *__restrict__ len, unsigned long long *__restrict__ out)
{
    constexpr int PN = DeviceCsr::PAT_N, PL = DeviceCsr::PAT_LEN;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    unsigned long long h = 1469598103934665603ull;
    for (int k = 0; k < PN * PL; ++k) h = (h ^ (unsigned)tab[b * (PN * PL) + k]) * 1099511628211ull;
    for (int k = 0; k < PN; ++k) h = (h ^ len[b * PN + k]) * 1099511628211ull;
    out[b] = h;
}
// the distinct tables gathered from their first blocks
__global__ void k_gather_tables(const int32_t *__restrict__ rep, const int32_t *__restrict__ tab, const uint8_t *__restrict__ len, int32_t *__restrict__ otab,
                                uint8_t *__restrict__ olen)
{
    constexpr int PN = DeviceCsr::PAT_N, PL = DeviceCsr::PAT_LEN;
    const int64_t b = rep[blockIdx.x];
    for (int k = threadIdx.x; k < PN * PL; k += blockDim.x) otab[(int64_t)blockIdx.x * (PN * PL) + k] = tab[b * (PN * PL) + k];
    for (int k = threadIdx.x; k < PN; k += blockDim.x) olen[(int64_t)blockIdx.x * PN + k] = len[b * PN + k];
}
// ... and every block's own table compared with the shared one it was given (a hash collision would show here)
__global__ __launch_bounds__(256) void k_check_tables(int64_t nblk, const int32_t *__restrict__ blk, const int32_t *__restrict__ tab, const uint8_t *__restrict__ len,
                                                      const int32_t *__restrict__ otab, const uint8_t *__restrict__ olen, int *__restrict__ failed)
{
    constexpr int PN = DeviceCsr::PAT_N, PL = DeviceCsr::PAT_LEN;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    const int64_t t = blk[b];
    bool same = true;
    for (int k = 0; k < PN * PL; ++k) same = same && tab[b * (PN * PL) + k] == otab[t * (PN * PL) + k];
    for (int k = 0; k < PN; ++k) same = same && len[b * PN + k] == olen[t * PN + k];
    if (!same) atomicAdd(failed, 1);
}

static int build_row_patterns(pib_solver *s)
{
    DeviceCsr &A = s->A;
    const int64_t nblk = (A.n + 255) / 256;
    constexpr int PN = DeviceCsr::PAT_N, PL = DeviceCsr::PAT_LEN;
    int32_t *t_tab = nullptr, *d_rep = nullptr;
    uint8_t *t_len = nullptr;
    unsigned long long *d_hash = nullptr;
    int *d_failed = nullptr;
    struct Tmp {
        int32_t *&a, *&r;
        uint8_t *&b;
        unsigned long long *&c;
        int *&d;
        ~Tmp()
        {
            if (a) (void)hipFree(a);
            if (r) (void)hipFree(r);
            if (b) (void)hipFree(b);
            if (c) (void)hipFree(c);
            if (d) (void)hipFree(d);
        }
    } tmp{t_tab, d_rep, t_len, d_hash, d_failed};
    auto drop = [&]() {
        if (A.pat_id) (void)hipFree(A.pat_id);
        if (A.pat_blk) (void)hipFree(A.pat_blk);
        if (A.pat_tab) (void)hipFree(A.pat_tab);
        if (A.pat_len) (void)hipFree(A.pat_len);
        A.pat_id = nullptr;
        A.pat_blk = nullptr;
        A.pat_tab = nullptr;
        A.pat_len = nullptr;
        A.pat_tables = 0;
        return 0;
    };
    // (the compressed forms are optional: a device short of memory keeps the plain columns instead of failing setMatrix)
    auto opt = [](hipError_t e) {
        if (e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    };
    if (!opt(hipMalloc(&A.pat_id, (size_t)nblk * 256)) || !opt(hipMalloc(&t_tab, sizeof(int32_t) * (size_t)nblk * PN * PL)) ||
        !opt(hipMalloc(&t_len, (size_t)nblk * PN)) || !opt(hipMalloc(&d_hash, sizeof(unsigned long long) * (size_t)nblk)))
        return drop();
    PIB_HIP(hipMalloc(&d_failed, sizeof(int)));
    PIB_HIP(hipMemsetAsync(d_failed, 0, sizeof(int), s->stream));
    if (A.rp64)
        hipLaunchKernelGGL(k_build_patterns<int64_t>, dim3((unsigned)nblk), dim3(256), 0, s->stream, A.n, A.ghost_lo, (const int64_t *)A.rowptr, A.col, A.pat_id,
                           t_tab, t_len, d_failed);
    else
        hipLaunchKernelGGL(k_build_patterns<int32_t>, dim3((unsigned)nblk), dim3(256), 0, s->stream, A.n, A.ghost_lo, (const int32_t *)A.rowptr, A.col, A.pat_id,
                           t_tab, t_len, d_failed);
    PIB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_hash_tables, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s->stream, nblk, t_tab, t_len, d_hash);
    PIB_HIP(hipGetLastError());
    int h_failed = 0;
    std::vector<unsigned long long> hh((size_t)nblk);
    PIB_HIP(hipMemcpyAsync(&h_failed, d_failed, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipMemcpyAsync(hh.data(), d_hash, sizeof(unsigned long long) * (size_t)nblk, hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    if (h_failed > 0) return drop();
    // the distinct tables in the order of their first blocks
    std::unordered_map<unsigned long long, int32_t> seen;
    std::vector<int32_t> blk((size_t)nblk), rep;
    for (int64_t b = 0; b < nblk; ++b) {
        auto it = seen.find(hh[(size_t)b]);
        if (it == seen.end()) {
            it = seen.emplace(hh[(size_t)b], (int32_t)rep.size()).first;
            rep.push_back((int32_t)b);
        }
        blk[(size_t)b] = it->second;
    }
    A.pat_tables = (int64_t)rep.size();
    if (!opt(hipMalloc(&A.pat_blk, sizeof(int32_t) * (size_t)nblk)) || !opt(hipMalloc(&A.pat_tab, sizeof(int32_t) * rep.size() * PN * PL)) ||
        !opt(hipMalloc(&A.pat_len, rep.size() * PN)) || !opt(hipMalloc(&d_rep, sizeof(int32_t) * rep.size())))
        return drop();
    PIB_HIP(hipMemcpyAsync(A.pat_blk, blk.data(), sizeof(int32_t) * (size_t)nblk, hipMemcpyHostToDevice, s->stream));
    PIB_HIP(hipMemcpyAsync(d_rep, rep.data(), sizeof(int32_t) * rep.size(), hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_gather_tables, dim3((unsigned)rep.size()), dim3(64), 0, s->stream, d_rep, t_tab, t_len, A.pat_tab, A.pat_len);
    PIB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_check_tables, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s->stream, nblk, A.pat_blk, t_tab, t_len, A.pat_tab, A.pat_len, d_failed);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipMemcpyAsync(&h_failed, d_failed, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));  // (blk / rep are read by the copies until here)
    if (h_failed > 0) return drop();
    A.patterned = true;
    return 0;
}

// DeviceCsr::code / dict of the matrix just set (cfg.compress_columns); a matrix some block of which has more than
// CODE_DICT distinct offsets keeps its plain columns.
static int build_column_codes(pib_solver *s)
{
    DeviceCsr &A = s->A;
    if (A.code) PIB_HIP(hipFree(A.code));
    if (A.dict) PIB_HIP(hipFree(A.dict));
    if (A.pat_id) PIB_HIP(hipFree(A.pat_id));
    if (A.pat_tab) PIB_HIP(hipFree(A.pat_tab));
    if (A.pat_len) PIB_HIP(hipFree(A.pat_len));
    if (A.pat_blk) PIB_HIP(hipFree(A.pat_blk));
    A.pat_blk = nullptr;
    A.pat_tables = 0;
    A.code = nullptr;
    A.dict = nullptr;
    A.coded = false;
    A.pat_id = nullptr;
    A.pat_tab = nullptr;
    A.pat_len = nullptr;
    A.patterned = false;
    if (!s->cfg.compress_columns || A.n <= 0 || A.nnz <= 0 || A.col == nullptr) return 0;
    if (s->cfg.compress_columns >= 2) {  // the row patterns first: a matrix that takes them needs no per-entry codes (every
        PIB_CHK(build_row_patterns(s));  // product here covers the rows from 0, i.e. starts on a block)
        if (A.patterned) return 0;
    }
    const int64_t nblk = (A.n + 255) / 256;
    if (hipMalloc(&A.code, (size_t)A.nnz + 64) != hipSuccess || hipMalloc(&A.dict, sizeof(int32_t) * (size_t)(nblk + 1) * DeviceCsr::CODE_DICT) != hipSuccess) {
        (void)hipGetLastError();  // (optional: see build_row_patterns)
        if (A.code) (void)hipFree(A.code);
        if (A.dict) (void)hipFree(A.dict);
        A.code = nullptr;
        A.dict = nullptr;
        return 0;
    }
    PIB_HIP(hipMemsetAsync(A.code + A.nnz, 0, 64, s->stream));
    PIB_HIP(hipMemsetAsync(A.dict + nblk * DeviceCsr::CODE_DICT, 0, sizeof(int32_t) * DeviceCsr::CODE_DICT, s->stream));
    int *d_many = nullptr, h_many = 0;
    PIB_HIP(hipMalloc(&d_many, sizeof(int)));
    PIB_HIP(hipMemsetAsync(d_many, 0, sizeof(int), s->stream));
    if (A.rp64)
        hipLaunchKernelGGL(k_build_codes<int64_t>, dim3((unsigned)nblk), dim3(256), 0, s->stream, A.n, A.ghost_lo, (const int64_t *)A.rowptr, A.col, A.code,
                           A.dict, d_many);
    else
        hipLaunchKernelGGL(k_build_codes<int32_t>, dim3((unsigned)nblk), dim3(256), 0, s->stream, A.n, A.ghost_lo, (const int32_t *)A.rowptr, A.col, A.code,
                           A.dict, d_many);
    PIB_HIP(hipGetLastError());
    PIB_HIP(hipMemcpyAsync(&h_many, d_many, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipFree(d_many));
    if (h_many > 0) {
        PIB_HIP(hipFree(A.code));
        PIB_HIP(hipFree(A.dict));
        A.code = nullptr;
        A.dict = nullptr;
        return 0;
    }
    A.coded = true;
    return 0;
}

int extract_dinv(pib_solver *s, int *n_missing)
{
    DeviceCsr &A = s->A;
    if (A.dinv == nullptr) PIB_HIP(hipMalloc(&A.dinv, (size_t)(A.n > 0 ? A.n : 1) * sizeof(double)));
    int *d_missing = nullptr;
    PIB_HIP(hipMalloc(&d_missing, sizeof(int)));
    PIB_HIP(hipMemsetAsync(d_missing, 0, sizeof(int), s->stream));
    if (A.n > 0) {
        const int nb = (int)((A.n + 255) / 256);
        if (A.rp64)
            hipLaunchKernelGGL(k_extract_dinv<int64_t>, dim3(nb), dim3(256), 0, s->stream, A.n, A.ghost_lo,
                               (const int64_t *)A.rowptr, A.col, A.val, A.dinv, d_missing);
        else
            hipLaunchKernelGGL(k_extract_dinv<int32_t>, dim3(nb), dim3(256), 0, s->stream, A.n, A.ghost_lo,
                               (const int32_t *)A.rowptr, A.col, A.val, A.dinv, d_missing);
        PIB_HIP(hipGetLastError());
    }
    int h = 0;
    PIB_HIP(hipMemcpyAsync(&h, d_missing, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    // largest 256-row chunk (LDS capacity of the SpMV)
    unsigned long long *d_max = nullptr, h_max = 0;
    PIB_HIP(hipMalloc(&d_max, sizeof(unsigned long long)));
    PIB_HIP(hipMemsetAsync(d_max, 0, sizeof(unsigned long long), s->stream));
    if (A.n > 0) {
        const int nb = (int)std::min<int64_t>(1024, ((A.n + LDS_ROWS - 1) / LDS_ROWS + 255) / 256);
        if (A.rp64)
            hipLaunchKernelGGL(k_max_chunk_nnz<int64_t>, dim3(nb), dim3(256), 0, s->stream, A.n, (const int64_t *)A.rowptr, d_max);
        else
            hipLaunchKernelGGL(k_max_chunk_nnz<int32_t>, dim3(nb), dim3(256), 0, s->stream, A.n, (const int32_t *)A.rowptr, d_max);
        PIB_HIP(hipGetLastError());
    }
    PIB_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(h_max), hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipFree(d_missing));
    PIB_HIP(hipFree(d_max));
    A.max_chunk_nnz = (int64_t)h_max;
    *n_missing = h;
    PIB_CHK(build_column_codes(s));
    return 0;
}

// ---------------------------------------------------------------- Gershgorin interval (the Chebyshev solver's default bounds)
// Per row: d = a_ii, off = sum_{j != i} |a_ij|.  jacobi: the ratio off / |d| (the spectrum of D^-1 A lies in [1 - max, 1 + max]
// when it is real); otherwise the interval [min (d - off), max (d + off)] of A itself.  Minimum and maximum go through 64-bit
// atomics on an order-preserving key of the double (sign bit flipped for the non-negative ones, all bits for the negative).
__host__ __device__ inline unsigned long long dkey(double v)
{
    unsigned long long u;
    memcpy(&u, &v, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double dkey_value(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    std::memcpy(&v, &u, 8);
    return v;
}
template <typename RP>
__global__ __launch_bounds__(256) void k_gershgorin(int64_t n, int64_t ghost_lo, const RP *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const double *__restrict__ val, int jacobi, unsigned long long *__restrict__ out)
{
    double lo = 1e300, hi = -1e300;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        double d = 0.0, off = 0.0;
        for (RP p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            if ((int64_t)col[p] == r + ghost_lo) d = d + val[p];
            else off = off + fabs(val[p]);
        }
        if (jacobi) {
            const double q = d != 0.0 ? off / fabs(d) : 1e300;
            lo = fmin(lo, 1.0 - q);
            hi = fmax(hi, 1.0 + q);
        } else {
            lo = fmin(lo, d - off);
            hi = fmax(hi, d + off);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_down(lo, o, 64));
        hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], dkey(lo));
        atomicMax(&out[1], dkey(hi));
    }
}

int gershgorin_bounds(pib_solver *s, bool jacobi, double *lo, double *hi)
{
    if (s->gersh_hi >= s->gersh_lo && s->gersh_jacobi == (jacobi ? 1 : 0)) {
        *lo = s->gersh_lo;
        *hi = s->gersh_hi;
        return 0;
    }
    DeviceCsr &A = s->A;
    unsigned long long *d_out = nullptr, h[2] = {dkey(1e300), dkey(-1e300)};
    PIB_HIP(hipMalloc(&d_out, 2 * sizeof(unsigned long long)));
    PIB_HIP(hipMemcpyAsync(d_out, h, sizeof h, hipMemcpyHostToDevice, s->stream));
    if (A.n > 0) {
        const int nb = (int)std::min<int64_t>(4096, (A.n + 255) / 256);
        if (A.rp64)
            hipLaunchKernelGGL(k_gershgorin<int64_t>, dim3(nb), dim3(256), 0, s->stream, A.n, A.ghost_lo, (const int64_t *)A.rowptr, A.col, A.val,
                               jacobi ? 1 : 0, d_out);
        else
            hipLaunchKernelGGL(k_gershgorin<int32_t>, dim3(nb), dim3(256), 0, s->stream, A.n, A.ghost_lo, (const int32_t *)A.rowptr, A.col, A.val,
                               jacobi ? 1 : 0, d_out);
    }
    const hipError_t e = hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, s->stream);
    const hipError_t e2 = hipStreamSynchronize(s->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess || e2 != hipSuccess) return fail(PIB_ERR_LIB, "solver %s: the Gershgorin bounds' kernel failed", s->name.c_str());
    double m0 = dkey_value(h[0]), m1 = dkey_value(h[1]);
    // over the ranks (collective: every rank of a solver gets here in the same solve)
    std::vector<double> mine = {m0, m1}, all;
    PIB_CHK(comm_allgather_host(s, mine, all));
    for (size_t q = 0; q + 1 < all.size(); q += 2) {
        m0 = std::min(m0, all[q]);
        m1 = std::max(m1, all[q + 1]);
    }
    *lo = m0;
    *hi = m1;
    s->gersh_lo = m0;
    s->gersh_hi = m1;
    s->gersh_jacobi = jacobi ? 1 : 0;
    return 0;
}

}  // namespace pib
