// halo.cpp -- C1/C2: RCCL ghost-plane exchange and scalar all-reduce over xGMI.
//
// Replaces the VecScatter inside PETSc's MPIAIJ MatMult and AmgX's
// communicator=MPI/MPI_DIRECT halo exchange.  One rank
// per GPU; z-slab (DMDA-style, nProc = (1,1,P)) decomposition, so in natural
// ordering a rank's ghosts are the `ghost_lo` entries just before its first
// row and the `ghost_hi` entries just after its last row: both are contiguous
// blocks of the neighbour's vector -- no pack kernel, one ncclSend/ncclRecv
// pair per neighbour grouped in a single RCCL group (point-to-point over the
// direct xGMI link between adjacent ranks; no ring involved).
#include <cstring>

#include "pib_internal.hpp"

namespace pib {

int comm_init(pib_solver *s, int rank, int nranks, const void *uid)
{
    s->comm.rank = rank;
    s->comm.nranks = nranks;
    s->comm.comm = nullptr;
    if (nranks <= 1) return 0;
    if (uid == nullptr) return fail(PIB_ERR_ARG_NULL, "pib_create: nranks > 1 needs the id from pib_comm_unique_id");
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) <= PIB_UID_BYTES, "unique id does not fit");
    std::memcpy(&id, uid, sizeof(id));
    PIB_NCCL(ncclCommInitRank(&s->comm.comm, nranks, id, rank));
    return 0;
}

// After the matrix is known: tell the neighbours how many entries this rank
// needs from them (all-gather of {n_local, ghost_lo, ghost_hi}).
int comm_setup_halo(pib_solver *s)
{
    DeviceCsr &A = s->A;
    A.send_prev = A.send_next = 0;
    if (s->comm.nranks <= 1) {
        if (A.ghost_lo != 0 || A.ghost_hi != 0)
            return fail(PIB_ERR_ARG_OUTOFRANGE, "single-rank matrix has columns outside [0, n)");
        return 0;
    }
    const int P = s->comm.nranks, r = s->comm.rank;
    int64_t mine[4] = {A.n, A.ghost_lo, A.ghost_hi, A.row0};
    int64_t *d_all = nullptr;
    PIB_HIP(hipMalloc(&d_all, sizeof(int64_t) * 4 * (size_t)P));
    PIB_HIP(hipMemcpyAsync(d_all + 4 * r, mine, sizeof(mine), hipMemcpyHostToDevice, s->stream));
    PIB_NCCL(ncclAllGather(d_all + 4 * r, d_all, 4, ncclInt64, s->comm.comm, s->stream));
    std::vector<int64_t> all(4 * (size_t)P);
    PIB_HIP(hipMemcpyAsync(all.data(), d_all, sizeof(int64_t) * 4 * (size_t)P, hipMemcpyDeviceToHost, s->stream));
    PIB_HIP(hipStreamSynchronize(s->stream));
    PIB_HIP(hipFree(d_all));
    if (r > 0) {
        if (A.ghost_lo > all[4 * (r - 1)])
            return fail(PIB_ERR_SUP, "halo of rank %d reaches beyond its neighbour (needs %lld entries, neighbour owns %lld)", r,
                        (long long)A.ghost_lo, (long long)all[4 * (r - 1)]);
        A.send_prev = all[4 * (r - 1) + 2];  // their ghost_hi
    } else if (A.ghost_lo != 0) {
        return fail(PIB_ERR_ARG_OUTOFRANGE, "rank 0 has columns below its first row");
    }
    if (r < P - 1) {
        if (A.ghost_hi > all[4 * (r + 1)])
            return fail(PIB_ERR_SUP, "halo of rank %d reaches beyond its neighbour", r);
        A.send_next = all[4 * (r + 1) + 1];  // their ghost_lo
    } else if (A.ghost_hi != 0) {
        return fail(PIB_ERR_ARG_OUTOFRANGE, "last rank has columns above its last row");
    }
    if (A.send_prev > A.n || A.send_next > A.n) return fail(PIB_ERR_SUP, "a neighbour's halo is wider than this rank's slab");
    return 0;
}

// Generic contiguous-plane exchange on a ghost-padded vector:
//   [lo ghosts | n_owned | hi ghosts], x_owned points at the owned part.
int halo_exchange_planes(pib_solver *s, double *x_owned, int64_t n_owned, int64_t lo, int64_t hi, int64_t send_prev,
                         int64_t send_next, hipStream_t st)
{
    const int P = s->comm.nranks, r = s->comm.rank;
    if (P <= 1) return 0;
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
    s->counters[3]++;
    return 0;
}

int halo_exchange(pib_solver *s, double *x_owned, hipStream_t st)
{
    const DeviceCsr &A = s->A;
    return halo_exchange_planes(s, x_owned, A.n, A.ghost_lo, A.ghost_hi, A.send_prev, A.send_next, st);
}

}  // namespace pib

extern "C" int pib_comm_unique_id(void *uid_out)
{
    if (uid_out == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "pib_comm_unique_id: null output");
    ncclUniqueId id;
    PIB_NCCL(ncclGetUniqueId(&id));
    std::memset(uid_out, 0, PIB_UID_BYTES);
    std::memcpy(uid_out, &id, sizeof(id));
    return 0;
}
