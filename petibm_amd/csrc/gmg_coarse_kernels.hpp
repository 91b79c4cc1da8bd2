// gmg_coarse_kernels.hpp -- geometric multigrid (gmg.hip), device side: the bottom of the cycle: the coarsest level, the single-workgroup coarse tail, the fused small levels.
// Included by gmg.hip only (one translation unit: the launches there instantiate these templates).
#pragma once
#include "pib_internal.hpp"

namespace pib {
// coarsest level in ONE workgroup: `sweeps` damped-Jacobi sweeps from zero,
// ping-pong between xa / xb (global, L2-resident), block barrier between sweeps.
__global__ __launch_bounds__(256) void k_coarsest(const Scalars *__restrict__ S, LevelDev L, double omega, int sweeps,
                                                  const double *__restrict__ b, double *__restrict__ xa,
                                                  double *__restrict__ xb, double *__restrict__ xout)
{
    if (S != nullptr && S->done) return;
    const int plane = L.nx * L.ny, n = plane * L.nk;
    double *cur = xa, *nxt = xb;
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int p = threadIdx.x; p < n; p += blockDim.x) {
            const int i = p % L.nx, j = (p / L.nx) % L.ny, k = L.k0 + p / plane;
            double d;
            if (sw == 0) {
                double c[6];
                face_coefs(L, i, j, k, c);
                d = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
                nxt[p] = jweight(omega, d) * scale_b(L, i, j, k, b[p]);
            } else {
                const double t = relax_cell(L, cur, p, i, j, k, scale_b(L, i, j, k, b[p]), &d);
                nxt[p] = jrelax(cur[p], 1.0 - omega, jweight(omega, d), t);
            }
        }
        __threadfence_block();
        __syncthreads();
        double *t = cur;
        cur = nxt;
        nxt = t;
    }
    if (cur != xout) {
        for (int p = threadIdx.x; p < n; p += blockDim.x) xout[p] = cur[p];
    }
}


// ---- coarse tail: every level with <= TAIL_MAX_CELLS cells (replicated / single rank) in ONE workgroup ----
// The levels below ~32^3 are launch-latency bound: five 4.4-us launches per level and V-cycle (rocprof: 40 of
// the 53 kernels of a 512^3 iteration are such launches).  One 1024-thread workgroup walks the whole remaining
// V-cycle (pre-smooth, residual, restriction, ..., coarsest sweeps, ..., prolongation, post-smooth) with block
// barriers between the phases; the per-cell arithmetic is the same as in k_level / k_restrict_rows / k_prolong_rows,
// so results are bit-identical to the per-level launches.  Damped Jacobi only.
//
// What a phase costs decides whether the tail pays (tools/tail_probe.py with the kernel's phase stamps,
// profiles/r03_coarse_tail_phases.txt).  With vectors and tables in HBM a phase is a chain of L2 round trips (83 us per V-cycle
// on a 448^2 mesh: five levels); as first written for LDS it was no better -- table-by-table staging, per-level pointer
// arrays in scratch memory, and every access through a pointer that may be LDS or HBM, i.e. a FLAT instruction at the latency
// of a vector-cache hit.  Now: one coalesced copy of the packed tables, the level descriptors in LDS, the kernel compiled
// twice so that the LDS instance addresses pool and tables with LDS instructions, a level of at most one cell per thread
// keeps the cell's row in registers for its whole visit, a level of at most 64 cells is walked by one wave without block
// barriers, and the visit / restriction / prolongation code exists once, in a loop over the V-cycle's legs (the kernel runs
// once per V-cycle from a cold instruction cache): 46 us on that mesh.
constexpr int TAIL_MAX_CELLS = 32768;
constexpr int TAIL_MAX_LEVELS = 12;
struct TailLevel {
    LevelDev L;
    double *xa, *xb, *b, *r;
};
constexpr int TAIL_POOL = 16384;  // doubles of LDS for the tail's vectors (128 KB of the CU's 160)
constexpr int TAIL_GUARD = 512;   // ... and of margin around them
constexpr int TAIL_TAB = 2048;    // doubles of LDS for the 1-D tables of the tail's levels
struct TailArgs {
    int nlev;
    TailLevel lv[TAIL_MAX_LEVELS];
    double omega;
    int pre, post, sweeps;
    // LDS instance: place of level l's three vectors (iterate, spare, right-hand side; the residual takes the spare) in the pool
    int lds_off[TAIL_MAX_LEVELS];
    // ... and the levels' 1-D coefficient and transfer tables, which the set-up packs into ONE block of HBM in the layout they
    // have in LDS (tab_src, tab_used doubles; tt[l]: where level l's tables start inside it)
    int tab_used;
    const double *tab_src;
    struct Tabs { int w[3], rw[3], cm[3], cp[3], wpar[3], woth[3], par[3], oth[3], fst[3]; } tt[TAIL_MAX_LEVELS];
    double *out0;  // where the tail's first level leaves its result (global memory)
};
// (4 KB: more than a kernel's argument segment takes beside the hidden arguments -- the kernel reads it from HBM)

// Reads of the tail's 1-D tables.  LDS: the descriptors' table pointers were redirected into the kernel's LDS block, but a
// pointer loaded from a descriptor is a generic one (a FLAT load); the reader turns it back into an index of the block,
// which the compiler addresses with LDS instructions.
template <bool LDS>
struct TailTab {
    const double *g;  // the block's generic address
    double *s;        // the block
    __device__ __forceinline__ double operator()(const double *p, int i) const
    {
        if constexpr (LDS) return s[(p - g) + i];
        else return p[i];
    }
    __device__ __forceinline__ int operator()(const int *p, int i) const
    {
        if constexpr (LDS) return reinterpret_cast<const int *>(s)[(p - reinterpret_cast<const int *>(g)) + i];
        else return p[i];
    }
};

// rs1d4 for the tail's lanes (every lane its own coarse cell, nothing wave-uniform): the four slots' table entries are loaded
// unconditionally at clamped indices and selected afterwards -- sixteen independent loads instead of twelve dependent
// little chains; the same weights
template <class RD>
__device__ __forceinline__ void rs1d4_lane(const RD &rd, const Tr1 &t, int I, int nf, bool wrap, double w[4], int f[4])
{
    const int f0 = rd(t.fst, I) - 1;
    bool in[4];
    int par[4], oth[4];
    double wp[4], wo[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        int ff = f0 + o;
        if (wrap) ff = ff < 0 ? ff + nf : (ff >= nf ? ff - nf : ff);
        in[o] = ff >= 0 && ff < nf;
        f[o] = ff < 0 ? 0 : (ff >= nf ? nf - 1 : ff);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        par[o] = rd(t.par, f[o]);
        oth[o] = rd(t.oth, f[o]);
        wp[o] = rd(t.wpar, f[o]);
        wo[o] = rd(t.woth, f[o]);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) w[o] = in[o] ? (par[o] == I ? wp[o] : (oth[o] == I ? wo[o] : 0.0)) : 0.0;
}

// One cell's row of the level operator -- face coefficients, diagonal, volume factors, where its six neighbours sit -- and
// its scaled right-hand side.  Same expressions in the same order as face_coefs / scale_b / unscale / apply_cell.
struct TailCell {
    double c[6], d, rxy, rz, wxy, wz, bs;
    int off[6];
    int has;  // bit q: neighbour q exists (or is reached across a periodic seam); bit 8: there is a cell at all
};
template <class RD>
__device__ __forceinline__ void tail_cell(const RD &rd, const LevelDev &L, int p, int n, const double *b, TailCell &t)
{
    t.has = 0;
    if (p >= n) return;
    const int plane = L.nx * L.ny;
    const int i = p % L.nx, j = (p / L.nx) % L.ny, k = L.k0 + p / plane;
    t.c[0] = rd(L.cmx, i);
    t.c[1] = rd(L.cpx, i);
    t.c[2] = rd(L.cmy, j);
    t.c[3] = rd(L.cpy, j);
    t.c[4] = rd(L.cmz, k);
    t.c[5] = rd(L.cpz, k);
    t.d = -(((((t.c[0] + t.c[1]) + t.c[2]) + t.c[3]) + t.c[4]) + t.c[5]);
    t.rxy = rd(L.rwx, i) * rd(L.rwy, j);
    t.rz = rd(L.rwz, k);
    t.wxy = rd(L.wx, i) * rd(L.wy, j);
    t.wz = rd(L.wz, k);
    t.bs = (b[p] * t.rxy) * t.rz;
    const int sy = L.nx, sz = plane;
    const bool px = L.per & 1, py = L.per & 2, pz = L.per & 4;
    int has = 256;
    t.off[0] = i > 0 ? -1 : L.nx - 1;
    if (i > 0 || px) has |= 1;
    t.off[1] = i < L.nx - 1 ? 1 : -(L.nx - 1);
    if (i < L.nx - 1 || px) has |= 2;
    t.off[2] = j > 0 ? -sy : (L.ny - 1) * sy;
    if (j > 0 || py) has |= 4;
    t.off[3] = j < L.ny - 1 ? sy : -(L.ny - 1) * sy;
    if (j < L.ny - 1 || py) has |= 8;
    t.off[4] = k > 0 ? -sz : (L.zring ? -sz : (L.nzg - 1) * sz);
    if (k > 0 || pz) has |= 16;
    t.off[5] = k < L.nzg - 1 ? sz : (L.zring ? sz : -(L.nzg - 1) * sz);
    if (k < L.nzg - 1 || pz) has |= 32;
    t.has = has;
}
__device__ __forceinline__ double tail_row(const TailCell &t, const double *x, int p)
{
    const double xc = x[p];
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if ((t.has >> q) & 1) s = facc(s, t.c[q], x[p + t.off[q]], xc);
    return s;
}
// one phase of one cell: the step from a zero guess, a damped-Jacobi step x -> out, or the residual of x
__device__ __forceinline__ void tail_cell_phase(const TailCell &t, bool zero, bool res, double omega, const double *b, const double *x,
                                                double *out, int p)
{
    if (!t.has) return;
    if (zero) {
        out[p] = jweight(omega, t.d) * t.bs;
        return;
    }
    if (res) {
        out[p] = resid(b[p], tail_row(t, x, p) * t.wxy, t.wz);
        return;
    }
    double s = t.bs;
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if ((t.has >> q) & 1) s = nacc(s, t.c[q], x[p + t.off[q]]);
    out[p] = jrelax(x[p], 1.0 - omega, jweight(omega, t.d), s);
}

// One visit of a level: `steps` smoothing steps (the first from a zero guess on the way down) and, on the way down, the
// residual; every step swaps the level's two vectors a / c, the residual goes to r.  Returns where the iterate is.
// KIND 0: at most 64 cells -- the first wave alone, a wavefront-scope fence between the steps (a wave's memory operations are
// issued and performed in order), the row in registers; 1: at most one cell per thread, block barriers, the row in
// registers; 2: several cells per thread, the rows rebuilt from the tables in every phase.
template <int KIND, class RD>
__device__ __forceinline__ double *tail_visit(const RD &rd, const LevelDev &F, int n, double omega, int steps, bool zero_first, bool resid,
                                              const double *b, double *a, double *c, double *r)
{
    const int phases = steps + (resid ? 1 : 0);
    const int p0 = threadIdx.x;
    const bool mine = KIND != 0 || p0 < 64;
    TailCell tc;
    tc.has = 0;
    if (KIND != 2 && mine) tail_cell(rd, F, p0, n, b, tc);
    for (int k = 0; k < phases; ++k) {
        const bool zero = zero_first && k == 0, res = k == steps;
        double *out = zero ? a : (res ? r : c);
        if (KIND == 2) {
            for (int p = p0; p < n; p += blockDim.x) {
                tail_cell(rd, F, p, n, b, tc);
                tail_cell_phase(tc, zero, res, omega, b, a, out, p);
            }
        } else if (mine)
            tail_cell_phase(tc, zero, res, omega, b, a, out, p0);
        if (KIND == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __threadfence_block();
            __syncthreads();
        }
        if (!zero && !res) {
            double *t = a; a = c; c = t;
        }
    }
    if (KIND == 0) {
        __threadfence_block();
        __syncthreads();
    }
    return a;
}

// LDS: the levels' vectors and tables live in LDS -- a compile-time fact, so that they are reached with LDS instructions
template <bool LDS>
__global__ __launch_bounds__(1024) void k_coarse_tail(const Scalars *__restrict__ S, const TailArgs *__restrict__ Tp)
{
    if (S != nullptr && S->done) return;
    const TailArgs &T = *Tp;
#ifdef PIB_TAIL_STAMPS
    __shared__ unsigned long long st_[256];
    __shared__ int sg_[256];
    int nst_ = 0;
#define STAMP(tag) do { if (threadIdx.x == 0 && nst_ < 256) { st_[nst_] = wall_clock64(); sg_[nst_] = (tag); } ++nst_; } while (0)
#else
#define STAMP(tag) do { } while (0)
#endif
    STAMP(0);
    // (a margin on either side of the pool: a neighbour of zero weight may be read before its weight is known; an access
    // below the LDS aperture is a fault, not a zero)
    __shared__ double pool_[LDS ? TAIL_GUARD + TAIL_POOL + TAIL_GUARD : 1];
    __shared__ double tab_[LDS ? TAIL_TAB : 1];
    __shared__ LevelDev Ls_[TAIL_MAX_LEVELS];
    double *const pool = pool_ + (LDS ? TAIL_GUARD : 0);
    const int nl = T.nlev;
    const TailTab<LDS> rd = {tab_, tab_};
    // the levels' descriptors into LDS (word by word, all threads), the tables after them; then one thread per level redirects
    // its descriptor's table pointers to the LDS copies
    {
        constexpr int words = (int)(sizeof(LevelDev) / sizeof(int));
        static_assert(sizeof(LevelDev) % sizeof(int) == 0, "LevelDev is copied in 4-byte words");
        for (int e = threadIdx.x; e < nl * words; e += blockDim.x) {
            const int l = e / words, w = e - l * words;
            reinterpret_cast<int *>(&Ls_[l])[w] = reinterpret_cast<const int *>(&T.lv[l].L)[w];
        }
        if (LDS) {
            for (int p = threadIdx.x; p < T.tab_used; p += blockDim.x) tab_[p] = T.tab_src[p];
            // (the margins finite; inside the pool every entry is written before it is read)
            for (int p = threadIdx.x; p < 2 * TAIL_GUARD; p += blockDim.x) pool_[p < TAIL_GUARD ? p : TAIL_POOL + p] = 0.0;
        }
        __syncthreads();
        if (LDS && (int)threadIdx.x < nl) {
            LevelDev &L = Ls_[threadIdx.x];
            const TailArgs::Tabs &o = T.tt[threadIdx.x];
            auto dbl = [&](int off) -> const double * { return tab_ + off; };
            auto i32 = [&](int off) -> const int * { return reinterpret_cast<const int *>(tab_ + off); };
            L.wx = dbl(o.w[0]), L.wy = dbl(o.w[1]), L.wz = dbl(o.w[2]);
            L.rwx = dbl(o.rw[0]), L.rwy = dbl(o.rw[1]), L.rwz = dbl(o.rw[2]);
            L.cmx = dbl(o.cm[0]), L.cmy = dbl(o.cm[1]), L.cmz = dbl(o.cm[2]);
            L.cpx = dbl(o.cp[0]), L.cpy = dbl(o.cp[1]), L.cpz = dbl(o.cp[2]);
            if ((int)threadIdx.x + 1 < nl) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    L.t[d].wpar = dbl(o.wpar[d]);
                    L.t[d].woth = dbl(o.woth[d]);
                    L.t[d].par = i32(o.par[d]);
                    L.t[d].oth = i32(o.oth[d]);
                    L.t[d].fst = i32(o.fst[d]);
                }
            }
        }
        if (LDS) {  // the first level's right-hand side
            const int n0 = T.lv[0].L.nx * T.lv[0].L.ny * T.lv[0].L.nk;
            double *b0 = pool + T.lds_off[0] + 2 * n0;
            for (int p = threadIdx.x; p < n0; p += blockDim.x) b0[p] = T.lv[0].b[p];
        }
        __syncthreads();
    }
    STAMP(1);
    // level l's iterate / spare / right-hand side (which = 0, 1, 2) and where its residual goes: the spare in LDS
    auto vec = [&](int l, int which) -> double * {
        const TailLevel &V = T.lv[l];
        if constexpr (LDS) return pool + T.lds_off[l] + which * (V.L.nx * V.L.ny * V.L.nk);
        else return which == 0 ? V.xa : (which == 1 ? V.xb : V.b);
    };
    // a level's two vectors swap with every step but the one from a zero guess: which of them holds the iterate after the
    // way down (no per-level pointer arrays: indexed by a runtime level they would live in scratch memory)
    const int dsteps = T.pre > 1 ? T.pre : 1;
    int l = 0;
    bool down = true;
    double *a = nullptr;  // the iterate of the level just visited
    for (;;) {
        const LevelDev &F = Ls_[l];
        const int fplane = F.nx * F.ny, nf = fplane * F.nk;
        const bool coarsest = l == nl - 1;
        const int steps = down ? (coarsest ? (T.sweeps > 1 ? T.sweeps : 1) : dsteps) : T.post;
        const bool resid = down && !coarsest;
        double *b = vec(l, 2);
        const int at = down ? 0 : ((dsteps - 1) & 1);
        double *xa = vec(l, at), *xc = vec(l, 1 - at);
        double *r = xc;
        if (resid) {
            // (the spare at the time of the residual: the vector the last step did NOT write)
            r = LDS ? vec(l, 1 - ((dsteps - 1) & 1)) : T.lv[l].r;
        }
        if (nf <= 64) a = tail_visit<0>(rd, F, nf, T.omega, steps, down, resid, b, xa, xc, r);
        else if (nf <= (int)blockDim.x) a = tail_visit<1>(rd, F, nf, T.omega, steps, down, resid, b, xa, xc, r);
        else a = tail_visit<2>(rd, F, nf, T.omega, steps, down, resid, b, xa, xc, r);
        STAMP((down ? 10 : 90) + l);
        if (resid) {
            // ---- restriction: right-hand side of level l + 1 = P^T r
            const LevelDev &C = Ls_[l + 1];
            const int cplane = C.nx * C.ny, nc = cplane * C.nk;
            double *bc = vec(l + 1, 2);
            for (int q = threadIdx.x; q < nc; q += blockDim.x) {
                const int I = q % C.nx, J = (q / C.nx) % C.ny, K = C.k0 + q / cplane;
                double wi[4], wj[4], wk[4];
                int si[4], sj[4], sk[4];
                rs1d4_lane(rd, F.t[0], I, F.nx, F.tper & 1, wi, si);
                __builtin_amdgcn_sched_barrier(0);  // (one direction's sixteen loads at a time: all three at once spill)
                rs1d4_lane(rd, F.t[1], J, F.ny, F.tper & 2, wj, sj);
                __builtin_amdgcn_sched_barrier(0);
                rs1d4_lane(rd, F.t[2], K, F.nzg, F.tper & 4, wk, sk);
                __builtin_amdgcn_sched_barrier(0);
                double sum = 0.0;
                for (int c2 = 0; c2 < 4; ++c2) {
                    if (wk[c2] == 0.0) continue;
                    const double *pk = r + fplane * (sk[c2] - F.k0);
                    double u = 0.0;
                    for (int b2 = 0; b2 < 4; ++b2) {
                        if (wj[b2] == 0.0) continue;  // (a 2-D level: one row of the four; the terms left out are +-0)
                        const double *pj = pk + F.nx * sj[b2];
                        double t = 0.0;
#pragma unroll
                        for (int a2 = 0; a2 < 4; ++a2) t = tacc(t, wi[a2], pj[si[a2]]);
                        u = tacc(u, wj[b2], t);
                    }
                    sum = tacc(sum, wk[c2], u);
                }
                bc[q] = sum;
            }
            __threadfence_block();
            __syncthreads();
            STAMP(50 + l);
            ++l;
            continue;
        }
        if (l == 0) break;
        // ---- prolongation: the iterate of level l - 1 += P a
        {
            const LevelDev &C = F;
            const LevelDev &G = Ls_[l - 1];
            const int gplane = G.nx * G.ny, ng = gplane * G.nk;
            const int cplane = C.nx * C.ny;
            double *xf = vec(l - 1, (dsteps - 1) & 1);
            for (int p = threadIdx.x; p < ng; p += blockDim.x) {
                const int i = p % G.nx, j = (p / G.nx) % G.ny, k = G.k0 + p / gplane;
                int I[2], J[2], K[2];
                double wi[2], wj[2], wk[2];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const Tr1 &t = G.t[d];
                    const int sd = d == 0 ? i : (d == 1 ? j : k);
                    int *Id = d == 0 ? I : (d == 1 ? J : K);
                    double *wd = d == 0 ? wi : (d == 1 ? wj : wk);
                    Id[0] = rd(t.par, sd);
                    Id[1] = rd(t.oth, sd);
                    wd[0] = rd(t.wpar, sd);
                    wd[1] = rd(t.woth, sd);
                }
                double sum = 0.0;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                        for (int a2 = 0; a2 < 2; ++a2) {
                            const double wgt = (wk[c2] * wj[b2]) * wi[a2];
                            if (wgt != 0.0) sum = tacc(sum, wgt, a[I[a2] + C.nx * J[b2] + cplane * (K[c2] - C.k0)]);
                        }
                xf[p] += sum;
            }
            __threadfence_block();
            __syncthreads();
            STAMP(70 + l - 1);
        }
        --l;
        down = false;
    }
    if (LDS) {
        const int n0 = T.lv[0].L.nx * T.lv[0].L.ny * T.lv[0].L.nk;
        for (int p = threadIdx.x; p < n0; p += blockDim.x) T.out0[p] = a[p];
    }
    STAMP(4);
#ifdef PIB_TAIL_STAMPS
    // tools: phase times of some launches (tag: 1 staged, 10+l way-down visit of level l -- the coarsest's sweeps for the last
    // --, 50+l restriction, 70+l prolongation onto l, 90+l post-smoothing, 4 result written); 10 ns units
    if (threadIdx.x == 0) {
        static __device__ int launches_ = 0;
        if (atomicAdd(&launches_, 1) % 40 == 20)
            for (int q = 1; q < nst_ && q < 256; ++q) printf("tail-stamp %d %llu\n", sg_[q], st_[q] - st_[q - 1]);
    }
#endif
#undef STAMP
}

// ---- small levels: a level's whole way down, and its whole way up, in ONE launch each -------------------------------
// Between the marching kernels of the large levels and the single-workgroup tail sit three or four levels (64^3 ... 16^3
// under a 512^3 grid, 224^2 ... 56^2 under a 448^2 one) that are launch-bound: seven kernels per level and V-cycle (step
// from zero, step, residual, restriction | prolongation, two steps).  Here a workgroup owns the children of a box of coarse
// cells and evaluates everything it needs on that box plus a margin in LDS -- the first step on the box grown by pre + 1
// cells, every further step and the residual one cell less, then the restriction for its own coarse cells (way down); the
// corrected iterate on the box grown by post cells, every post-smoothing step one cell less (way up).  The margins are
// recomputed by the neighbouring workgroups; no field crosses HBM between the phases.
// What such a kernel costs is the number of DEPENDENT trips to memory, ~2 us each when the lines are cold (phase stamps,
// profiles/r03_small_level_kernels.txt: a first version that fetched tables and right-hand side where it used them took 19
// us per launch, a third of it the restriction's little pointer chases).  So: ONE round of loads -- the 1-D coefficient and
// transfer tables of the region into LDS, the thread's right-hand-side values (and iterate values, way up) into registers,
// a thread keeping the same region cells through all phases -- and after it LDS only (way up: plus the gather of the coarse
// values).  Same per-cell expressions in the same order as k_level / the transfers: the same bits.
// Levels whole on this rank, Jacobi, 1-2 pre- / post-smoothing steps, operator and transfers wrapping alike.
constexpr int SM_MAXE = 40;    // largest extent of a workgroup's region per direction
constexpr int SM_MAXR = 3456;  // ... and its cells (two LDS buffers of that many doubles)
constexpr int SM_NT = 512;     // threads per workgroup
constexpr int SM_NB = (SM_MAXR + SM_NT - 1) / SM_NT;  // region cells per thread
constexpr int SM_MAXBC = 16;   // coarse cells per direction and workgroup
struct SmTabs {
    double cm[3][SM_MAXE], cp[3][SM_MAXE], rw[3][SM_MAXE], w[3][SM_MAXE];  // coefficients, by region coordinate
    double wpar[3][SM_MAXE], woth[3][SM_MAXE];                            // transfers of the region's fine cells
    int par[3][SM_MAXE], oth[3][SM_MAXE];
    int fst[3][SM_MAXBC + 1];                                              // first child of the owned coarse cells
};
struct SmGeom {
    int n[3];           // the level's cells
    int per[3];         // periodic directions
    int I0[3], I1[3];   // owned coarse cells [I0, I1)
    int F0[3], F1[3];   // their children: the owned fine cells [F0, F1)
    int lo[3], ext[3];  // the largest region, in unwrapped level coordinates (clipped to the domain where it does not wrap)
};
__device__ __forceinline__ int sm_wrap(int g, int n) { return g < 0 ? g + n : (g >= n ? g - n : g); }

// geometry of workgroup `blk`: its box of bc[] coarse cells, their children, the region grown by `grow_lo` / `grow_hi`.
// Aggregates that are all pairs (or all single cells) need no table for the children's range.
__device__ __forceinline__ void sm_geometry(const LevelDev &F, const LevelDev &C, int blk, int bcx, int bcy, int bcz, int grow_lo, int grow_hi,
                                            SmGeom &G)
{
    const int bc[3] = {bcx, bcy, bcz};
    const int nc[3] = {C.nx, C.ny, C.nzg};
    G.n[0] = F.nx, G.n[1] = F.ny, G.n[2] = F.nzg;
    int nb[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) nb[d] = (nc[d] + bc[d] - 1) / bc[d];
    const int b3[3] = {blk % nb[0], (blk / nb[0]) % nb[1], blk / (nb[0] * nb[1])};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        G.per[d] = (F.per >> d) & 1;
        G.I0[d] = b3[d] * bc[d];
        G.I1[d] = min(G.I0[d] + bc[d], nc[d]);
        if (G.n[d] == 2 * nc[d]) G.F0[d] = 2 * G.I0[d], G.F1[d] = 2 * G.I1[d];
        else if (G.n[d] == nc[d]) G.F0[d] = G.I0[d], G.F1[d] = G.I1[d];
        else {
            G.F0[d] = F.t[d].fst[G.I0[d]];
            G.F1[d] = G.I1[d] < nc[d] ? F.t[d].fst[G.I1[d]] : G.n[d];
        }
        int lo = G.F0[d] - grow_lo, hi = G.F1[d] + grow_hi;
        if (!G.per[d]) lo = max(lo, 0), hi = min(hi, G.n[d]);
        G.lo[d] = lo;
        G.ext[d] = hi - lo;
    }
}
// the one round of table loads
__device__ __forceinline__ void sm_stage_tabs(const LevelDev &F, const LevelDev &C, const SmGeom &G, SmTabs &T)
{
    const double *cm[3] = {F.cmx, F.cmy, F.cmz}, *cp[3] = {F.cpx, F.cpy, F.cpz}, *rw[3] = {F.rwx, F.rwy, F.rwz}, *w[3] = {F.wx, F.wy, F.wz};
    const int nc[3] = {C.nx, C.ny, C.nzg};
    // (thread t: direction t / 64, entry t % 64 -- all of a table's loads in one wave's single pass; the direction as a
    // compile-time constant of an unrolled loop: indexed by a runtime one the geometry would live in scratch memory)
    const int dw = threadIdx.x >> 6, r = threadIdx.x & 63;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (dw == d && r < G.ext[d]) {
            const int s = sm_wrap(G.lo[d] + r, G.n[d]);
            const Tr1 &tt = F.t[d];
            const double a0 = cm[d][s], a1 = cp[d][s], a2 = rw[d][s], a3 = w[d][s], a4 = tt.wpar[s], a5 = tt.woth[s];
            const int i0 = tt.par[s], i1 = tt.oth[s];
            T.cm[d][r] = a0, T.cp[d][r] = a1, T.rw[d][r] = a2, T.w[d][r] = a3, T.wpar[d][r] = a4, T.woth[d][r] = a5;
            T.par[d][r] = i0, T.oth[d][r] = i1;
        }
        if (dw == d + 3 && r <= G.I1[d] - G.I0[d] && G.I0[d] + r < nc[d]) T.fst[d][r] = F.t[d].fst[G.I0[d] + r];
    }
}
// the thread's cells of the largest region (the same in every phase): coordinates packed as rx | ry << 8 | rz << 16, -1: none
__device__ __forceinline__ void sm_cells(const SmGeom &G, int cell[SM_NB])
{
    const int cells = G.ext[0] * G.ext[1] * G.ext[2], e01 = G.ext[0] * G.ext[1];
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) {
        const int t = threadIdx.x + u * SM_NT;
        if (t < cells) {
            const int rz = t / e01, tr = t - rz * e01;
            const int ry = tr / G.ext[0], rx = tr - ry * G.ext[0];
            cell[u] = rx | (ry << 8) | (rz << 16);
        } else
            cell[u] = -1;
    }
}
__device__ __forceinline__ int64_t sm_global(const LevelDev &F, const SmGeom &G, int c)
{
    const int i = sm_wrap(G.lo[0] + (c & 255), G.n[0]), j = sm_wrap(G.lo[1] + ((c >> 8) & 255), G.n[1]), k = sm_wrap(G.lo[2] + (c >> 16), G.n[2]);
    return (int64_t)i + (int64_t)F.nx * (j + (int64_t)F.ny * (k - F.k0));
}
// One phase over the region's cells that lie `m` cells inside its unclipped faces (a clipped face is the domain's: no margin
// there).  MODE 1: x = omega bs / d; 2: x' = x + omega (bs - t) / d; 3: r = b - (t wx wy) wz.  src / dst: LDS fields indexed
// like the largest region; gdst (MODE 2 only): the owned cells' values go to global memory as well.
template <int MODE>
__device__ __forceinline__ void sm_phase(const LevelDev &F, const SmGeom &G, const SmTabs &T, const int cell[SM_NB], const double bq[SM_NB],
                                         int grow_lo, int grow_hi, int m, double omega, const double *src, double *dst,
                                         double *__restrict__ gdst)
{
    int a[3], e[3];  // the phase's box, relative to the region
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int lo = G.F0[d] - grow_lo + m, hi = G.F1[d] + grow_hi - m;
        if (!G.per[d]) lo = max(lo, 0), hi = min(hi, G.n[d]);
        a[d] = lo - G.lo[d];
        e[d] = hi - lo;
    }
    const int sy = G.ext[0], sz = G.ext[0] * G.ext[1];
    // Branch-free up to the store: the thread's cells are independent chains of LDS reads, ~20 dependent fp64 operations and a
    // division -- as separate basic blocks (a `continue` per cell) they ran one after the other, 0.3 us each.  A cell outside
    // the phase's box is evaluated all the same (at cell 0 if the thread has none: any finite or non-finite value will do)
    // and not stored.
    constexpr int GR = 3;  // cells evaluated together (all of them at once: 256 registers and a hundred spilled)
#pragma unroll
    for (int u0 = 0; u0 < SM_NB; u0 += GR) {
    double v[GR];
    int qv[GR];
    bool st[GR];
#pragma unroll
    for (int uu = 0; uu < GR; ++uu) {
        const int u = u0 + uu < SM_NB ? u0 + uu : SM_NB - 1;
        const int c = cell[u] < 0 ? 0 : cell[u];
        const int rx = c & 255, ry = (c >> 8) & 255, rz = c >> 16;
        st[uu] = u0 + uu < SM_NB && cell[u] >= 0 && (unsigned)(rx - a[0]) < (unsigned)e[0] && (unsigned)(ry - a[1]) < (unsigned)e[1] &&
                 (unsigned)(rz - a[2]) < (unsigned)e[2];
        const int q = rx + sy * ry + sz * rz;
        qv[uu] = q;
        const double cxm = T.cm[0][rx], cxp = T.cp[0][rx], cym = T.cm[1][ry], cyp = T.cp[1][ry], czm = T.cm[2][rz], czp = T.cp[2][rz];
        const double d = -(((((cxm + cxp) + cym) + cyp) + czm) + czp);
        const double bv = bq[u];
        const double bs = (bv * (T.rw[0][rx] * T.rw[1][ry])) * T.rw[2][rz];
        if (MODE == 1) {
            v[uu] = jweight(omega, d) * bs;
        } else if (MODE == 2) {
            // a neighbour beyond the region: only at a clipped face, i.e. a wall -- zero coefficient, the centre's own value
            const double xc = src[q];
            double t = bs;
            t = nacc(t, cxm, src[rx > 0 ? q - 1 : q]);
            t = nacc(t, cxp, src[rx < G.ext[0] - 1 ? q + 1 : q]);
            t = nacc(t, cym, src[ry > 0 ? q - sy : q]);
            t = nacc(t, cyp, src[ry < G.ext[1] - 1 ? q + sy : q]);
            t = nacc(t, czm, src[rz > 0 ? q - sz : q]);
            t = nacc(t, czp, src[rz < G.ext[2] - 1 ? q + sz : q]);
            v[uu] = jrelax(xc, 1.0 - omega, jweight(omega, d), t);
        } else {
            const double xc = src[q];
            double s = 0.0;
            s = facc(s, cxm, src[rx > 0 ? q - 1 : q], xc);
            s = facc(s, cxp, src[rx < G.ext[0] - 1 ? q + 1 : q], xc);
            s = facc(s, cym, src[ry > 0 ? q - sy : q], xc);
            s = facc(s, cyp, src[ry < G.ext[1] - 1 ? q + sy : q], xc);
            s = facc(s, czm, src[rz > 0 ? q - sz : q], xc);
            s = facc(s, czp, src[rz < G.ext[2] - 1 ? q + sz : q], xc);
            v[uu] = resid(bv, s * (T.w[0][rx] * T.w[1][ry]), T.w[2][rz]);
        }
    }
#pragma unroll
    for (int uu = 0; uu < GR; ++uu) {
        if (!st[uu]) continue;
        dst[qv[uu]] = v[uu];
        if (MODE == 2 && gdst != nullptr) {
            const int c = cell[u0 + uu < SM_NB ? u0 + uu : SM_NB - 1];
            const int gx = G.lo[0] + (c & 255), gy = G.lo[1] + ((c >> 8) & 255), gz = G.lo[2] + (c >> 16);
            if (gx >= G.F0[0] && gx < G.F1[0] && gy >= G.F0[1] && gy < G.F1[1] && gz >= G.F0[2] && gz < G.F1[2]) gdst[sm_global(F, G, c)] = v[uu];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    }
}

// way down: b -> the pre-smoothed iterate x (owned cells) and the next level's right-hand side bc = P^T (b - A x)
__global__ __launch_bounds__(SM_NT) void k_small_down(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega, int pre,
                                                      const double *__restrict__ b, double *__restrict__ x, double *__restrict__ bc, int bcx,
                                                      int bcy, int bcz, const double *__restrict__ pin_sum = nullptr)
{
    __shared__ double A_[SM_MAXR], B_[SM_MAXR];
    __shared__ SmTabs T;
#ifdef PIB_SMALL_STAMPS
    unsigned long long st_[12]; int ns_ = 0;
#define SST() do { st_[ns_++] = wall_clock64(); } while (0)
#else
#define SST() do { } while (0)
#endif
    SST();
    const int done = (S != nullptr) ? S->done : 0;
    // the residual on the owned children and one cell around them (two above: the fourth slot of the last coarse cell's
    // restriction stencil), the iterate one cell beyond that, every earlier step one more
    const int glo = 1 + pre, ghi = 2 + pre;
    SmGeom G;
    sm_geometry(F, C, blockIdx.x, bcx, bcy, bcz, glo, ghi, G);
    if (done) return;
    SST();
    int cell[SM_NB];
    double bq[SM_NB];
    sm_cells(G, cell);
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) bq[u] = cell[u] >= 0 ? b[sm_global(F, G, cell[u])] : 0.0;
    if (pin_sum != nullptr && F.k0 == 0) {  // PINNED (level 0): effective b at global cell 0, in every region that holds it
#pragma unroll
        for (int u = 0; u < SM_NB; ++u)
            if (cell[u] >= 0 && sm_global(F, G, cell[u]) == 0) bq[u] = bq[u] - *pin_sum;
    }
    sm_stage_tabs(F, C, G, T);
    __syncthreads();
    SST();
    sm_phase<1>(F, G, T, cell, bq, glo, ghi, 0, omega, nullptr, A_, nullptr);
    __syncthreads();
    SST();
    double *cur = A_, *oth = B_;
    for (int sw = 1; sw < pre; ++sw) {
        sm_phase<2>(F, G, T, cell, bq, glo, ghi, sw, omega, cur, oth, nullptr);
        __syncthreads();
        double *t = cur; cur = oth; oth = t;
    }
    SST();
    sm_phase<3>(F, G, T, cell, bq, glo, ghi, pre, omega, cur, oth, nullptr);
    // the owned cells of the iterate
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) {
        const int c = cell[u];
        if (c < 0) continue;
        const int gx = G.lo[0] + (c & 255), gy = G.lo[1] + ((c >> 8) & 255), gz = G.lo[2] + (c >> 16);
        if (gx >= G.F0[0] && gx < G.F1[0] && gy >= G.F0[1] && gy < G.F1[1] && gz >= G.F0[2] && gz < G.F1[2])
            x[sm_global(F, G, c)] = cur[(c & 255) + G.ext[0] * ((c >> 8) & 255) + G.ext[0] * G.ext[1] * (c >> 16)];
    }
    __syncthreads();
    SST();
    // restriction: one thread per owned coarse cell, the order of the sum as in k_restrict_rows / the tail
    {
        const double *r = oth;
        const int e0 = G.I1[0] - G.I0[0], e1 = G.I1[1] - G.I0[1], e2 = G.I1[2] - G.I0[2];
        const int sy = G.ext[0], sz = G.ext[0] * G.ext[1];
        for (int t = threadIdx.x; t < e0 * e1 * e2; t += blockDim.x) {
            const int tz = t / (e0 * e1), tr = t - tz * (e0 * e1);
            const int ty = tr / e0, tx = tr - ty * e0;
            const int Ic[3] = {G.I0[0] + tx, G.I0[1] + ty, G.I0[2] + tz};
            const int Tc[3] = {tx, ty, tz};
            double w[3][4];
            int pos[3][4];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int f0 = T.fst[d][Tc[d]] - 1;
                const bool wrap = (F.tper >> d) & 1;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int fu = f0 + o;  // unwrapped: the region's coordinates
                    const int ps = min(max(fu - G.lo[d], 0), G.ext[d] - 1);
                    double wt = 0.0;
                    if (wrap || (fu >= 0 && fu < G.n[d])) {
                        if (T.par[d][ps] == Ic[d]) wt = T.wpar[d][ps];
                        else if (T.oth[d][ps] == Ic[d]) wt = T.woth[d][ps];
                    }
                    w[d][o] = wt;
                    pos[d][o] = ps;
                }
            }
            // (no skipping of zero weights: a term of zero weight adds +-0 to a sum that is never -0, the residual is finite
            // on the whole region -- and sixteen loads at a time are in flight instead of four behind a branch)
            double sum = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                double rv[4][4];
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                    for (int a2 = 0; a2 < 4; ++a2) rv[b2][a2] = r[sz * pos[2][c2] + sy * pos[1][b2] + pos[0][a2]];
                double u = 0.0;
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) {
                    double t = 0.0;
#pragma unroll
                    for (int a2 = 0; a2 < 4; ++a2) t = tacc(t, w[0][a2], rv[b2][a2]);
                    u = tacc(u, w[1][b2], t);
                }
                sum = tacc(sum, w[2][c2], u);
            }
            bc[(int64_t)Ic[0] + (int64_t)C.nx * (Ic[1] + (int64_t)C.ny * (Ic[2] - C.k0))] = sum;
        }
    }
    SST();
#ifdef PIB_SMALL_STAMPS
    // tools: geometry | one round of loads | first step | further steps | residual + iterate out | restriction; 10 ns units
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        static __device__ int launches_ = 0;
        if (atomicAdd(&launches_, 1) % 60 == 30) {
            printf("small-down n=%d,%d,%d blocks=%d blk=%d:", F.nx, F.ny, F.nzg, (int)gridDim.x, (int)blockIdx.x);
            for (int q = 1; q < ns_; ++q) printf(" %llu", st_[q] - st_[q - 1]);
            printf("\n");
        }
    }
#endif
#undef SST
}

// way up: out = `post` smoothing steps on x + P xc (owned cells)
__global__ __launch_bounds__(SM_NT) void k_small_up(const Scalars *__restrict__ S, LevelDev F, LevelDev C, double omega, int post,
                                                    const double *__restrict__ b, const double *__restrict__ x, const double *__restrict__ xc,
                                                    double *__restrict__ out, int bcx, int bcy, int bcz)
{
    __shared__ double A_[SM_MAXR], B_[SM_MAXR];
    __shared__ SmTabs T;
    const int done = (S != nullptr) ? S->done : 0;
    SmGeom G;
    sm_geometry(F, C, blockIdx.x, bcx, bcy, bcz, post, post, G);
    if (done) return;
    int cell[SM_NB];
    double bq[SM_NB], xq[SM_NB];
    sm_cells(G, cell);
#pragma unroll
    for (int u = 0; u < SM_NB; ++u) {
        const int64_t p = cell[u] >= 0 ? sm_global(F, G, cell[u]) : 0;
        bq[u] = cell[u] >= 0 ? b[p] : 0.0;
        xq[u] = cell[u] >= 0 ? x[p] : 0.0;
    }
    sm_stage_tabs(F, C, G, T);
    __syncthreads();
    // the corrected iterate on the whole region
    {
        const int64_t cplane = (int64_t)C.nx * C.ny;
#pragma unroll
        for (int u = 0; u < SM_NB; ++u) {
            const int c = cell[u];
            if (c < 0) continue;
            const int r3[3] = {c & 255, (c >> 8) & 255, c >> 16};
            double sum = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        const double wgt = ((c2 ? T.woth[2][r3[2]] : T.wpar[2][r3[2]]) * (b2 ? T.woth[1][r3[1]] : T.wpar[1][r3[1]])) *
                                           (a2 ? T.woth[0][r3[0]] : T.wpar[0][r3[0]]);
                        const int I = a2 ? T.oth[0][r3[0]] : T.par[0][r3[0]], J = b2 ? T.oth[1][r3[1]] : T.par[1][r3[1]],
                                  K = c2 ? T.oth[2][r3[2]] : T.par[2][r3[2]];
                        if (wgt != 0.0) sum = tacc(sum, wgt, xc[I + (int64_t)C.nx * J + cplane * (K - C.k0)]);
                    }
            A_[r3[0] + G.ext[0] * r3[1] + G.ext[0] * G.ext[1] * r3[2]] = xq[u] + sum;
        }
    }
    __syncthreads();
    double *cur = A_, *oth = B_;
    for (int sw = 1; sw <= post; ++sw) {
        sm_phase<2>(F, G, T, cell, bq, post, post, sw, omega, cur, oth, sw == post ? out : nullptr);
        if (sw < post) __syncthreads();
        double *t = cur; cur = oth; oth = t;
    }
}
}  // namespace pib
