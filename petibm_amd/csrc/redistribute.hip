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
        PIB_HIP(hipMemcpy(F.d_split, split.data(), sizeof(int32_t) * split.size(), hipMemcpyHostToDevice));
        PIB_HIP(hipMemcpy(F.d_src, src.data(), sizeof(int64_t) * src.size(), hipMemcpyHostToDevice));
        const size_t fb = sizeof(double) * (size_t)std::max<int64_t>(F.n_slab, 1);
        PIB_HIP(hipMalloc(&F.stage, fb));
        PIB_MEMSET(F.stage, 0, fb);
        R.n_slab += F.n_slab;
    }
    const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(R.n_slab, 1);
    PIB_HIP(hipMalloc(&R.b_nat, bytes));
    PIB_HIP(hipMalloc(&R.x_nat, bytes));
    PIB_MEMSET(R.b_nat, 0, bytes);
    PIB_MEMSET(R.x_nat, 0, bytes);
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

}  // namespace pib
