// poisson_boxes_mpi.cpp -- the launch model of an unchanged PetIBM, without PETSc: `mpiexec -n P` processes, one per GPU,
// each handing the rows of ITS DMDA box to the backend through the C ABI.
//
// What PetIBM + AmgXWrapper do around src/linsolver/linsolveramgx.cpp:69 (`amgx.initialize(PETSC_COMM_WORLD, "dDDI", config)`),
// :84 (`setA`) and :96 (`solve`) is done here with plain MPI calls, i.e. what include/petibm_amd/petsc_adapter.hpp does with
// PETSc's communicator (broadcastUniqueId, localDevice):
//   * rank 0 draws the communicator id (RCCL, or the HIP-IPC peer windows with PIB_TRANSPORT=peer), MPI_Bcast hands it round;
//   * MPI_Comm_split_type(MPI_COMM_TYPE_SHARED) gives the node-local rank = the HIP device (a test box with one GPU carries every
//     rank on device 0 -- third argument --, which RCCL refuses and the peer transport accepts);
//   * the mesh is cut into the (m, n, p) boxes DMDACreate3d picks with PETSC_DECIDE for a cube on a power-of-two communicator
//     (src/mesh/cartesianmesh.cpp:97,503-519: (1,1,2), (1,2,2), (2,2,2)), rows numbered rank by rank, inside a box x fastest
//     (the "PETSc ordering" of a DMDA) -- int32 indices as AmgX's mode dDDI, global columns;
//   * pib_create / pib_set_csr_i32 / pib_solve on host arrays / pib_get_iters / pib_get_residual / pib_destroy.
// The matrix is the pressure Poisson operator of the N^3 unit cavity (D (dt I) G, uniform mesh), the right-hand side A x* for a
// cosine field; the program checks the true residual and the error against x* (constants removed) and fails loudly.
//
//   g++ -std=c++14 -I include -I /opt/conda/include examples/mpi/poisson_boxes_mpi.cpp -L petibm_amd/lib -lpetibm_amd
//       -L /opt/conda/lib -lmpi -Wl,-rpath,/opt/conda/lib -o examples/mpi/poisson_boxes_mpi
//   mpiexec -n 8 examples/mpi/poisson_boxes_mpi 64                              (8 GPUs, RCCL)
//   PIB_TRANSPORT=peer mpiexec -n 8 examples/mpi/poisson_boxes_mpi 64 1e-10 1   (8 ranks sharing ONE GPU: the test box)
#include <mpi.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "petibm_amd.h"

namespace
{
struct Box {
    int lo[3], n[3];
    int64_t first;  // global number of the box's first row
};

void die(int rank, const char *what, int code)
{
    std::fprintf(stderr, "[rank %d] %s failed with %d: %s\n", rank, what, code, pib_last_error());
    MPI_Abort(MPI_COMM_WORLD, 1);
}

// DMDA's split of n cells over p processes along one direction: the first n % p processes get one cell more
void split(int n, int p, int q, int *lo, int *cnt)
{
    *cnt = n / p + (q < n % p ? 1 : 0);
    *lo = q * (n / p) + (q < n % p ? q : n % p);
}
}  // namespace

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, size = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    const int N = argc > 1 ? std::atoi(argv[1]) : 48;
    const double tol = argc > 2 ? std::atof(argv[2]) : 1e-10;
    // (a test box with fewer GPUs than ranks: the node-local rank modulo this many devices -- the peer transport only; 0: one each)
    const int devices = argc > 3 ? std::atoi(argv[3]) : 0;

    // ---- the communicator id: rank 0 draws it, everybody receives it
    char uid[PIB_UID_BYTES];
    std::memset(uid, 0, sizeof uid);
    if (size > 1) {
        if (rank == 0) {
            const char *t = std::getenv("PIB_TRANSPORT");
            const int e = (t != nullptr && std::string(t) == "peer") ? pib_comm_peer_id(uid) : pib_comm_unique_id(uid);
            if (e) die(rank, "drawing the communicator id", e);
        }
        MPI_Bcast(uid, PIB_UID_BYTES, MPI_BYTE, 0, MPI_COMM_WORLD);
    }
    // ---- this rank's device: its number among the ranks of the node (pib_create takes it modulo the device count)
    MPI_Comm node;
    int lrank = 0;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node);
    MPI_Comm_rank(node, &lrank);
    MPI_Comm_free(&node);

    // ---- the process grid PETSC_DECIDE picks for a cube: powers of two, z first
    int grid[3] = {1, 1, 1};
    for (int left = size, d = 2; left > 1; d = (d + 2) % 3) {
        if (left % 2) {
            if (rank == 0) std::fprintf(stderr, "this example takes a power-of-two number of ranks\n");
            MPI_Abort(MPI_COMM_WORLD, 2);
        }
        grid[d] *= 2;
        left /= 2;
    }
    // every rank's box, in rank order (rank = px + m (py + n pz)): the rows are numbered box by box
    std::vector<Box> boxes((size_t)size);
    int64_t first = 0;
    for (int r = 0; r < size; ++r) {
        const int q[3] = {r % grid[0], (r / grid[0]) % grid[1], r / (grid[0] * grid[1])};
        Box &b = boxes[(size_t)r];
        for (int d = 0; d < 3; ++d) split(N, grid[d], q[d], &b.lo[d], &b.n[d]);
        b.first = first;
        first += (int64_t)b.n[0] * b.n[1] * b.n[2];
    }
    const int64_t n_global = first;
    if (n_global >= INT32_MAX) {
        if (rank == 0) std::fprintf(stderr, "N too large for int32 indices (AmgX mode dDDI)\n");
        MPI_Abort(MPI_COMM_WORLD, 2);
    }
    // global row of cell (i, j, k): found through the box that owns it
    auto owner_1d = [&](int c, int d) {
        for (int q = 0; q < grid[d]; ++q) {
            int lo, cnt;
            split(N, grid[d], q, &lo, &cnt);
            if (c >= lo && c < lo + cnt) return q;
        }
        return -1;
    };
    auto row_of = [&](int i, int j, int k) -> int32_t {
        const int r = owner_1d(i, 0) + grid[0] * (owner_1d(j, 1) + grid[1] * owner_1d(k, 2));
        const Box &b = boxes[(size_t)r];
        return (int32_t)(b.first + (i - b.lo[0]) + (int64_t)b.n[0] * ((j - b.lo[1]) + (int64_t)b.n[1] * (k - b.lo[2])));
    };

    // ---- this rank's rows of D (dt I) G on the uniform mesh: off-diagonal dt * (face area) / (centre distance) = dt h, the
    // diagonal minus their sum, a missing neighbour at a wall left out (navierstokes.cpp:349-356); columns ascending
    const Box &me = boxes[(size_t)rank];
    const double h = 1.0 / N, dt = 0.01, off = dt * h;
    const double pi = std::acos(-1.0);
    auto exact = [&](int i, int j, int k) {
        return std::cos(pi * (i + 0.5) * h) * std::cos(pi * (j + 0.5) * h) * std::cos(pi * (k + 0.5) * h);
    };
    const int32_t n_local = (int32_t)((int64_t)me.n[0] * me.n[1] * me.n[2]);
    std::vector<int32_t> rowptr(1, 0), col;
    std::vector<double> val, b((size_t)n_local), xs((size_t)n_local), x((size_t)n_local, 0.0);
    col.reserve((size_t)n_local * 7);
    val.reserve((size_t)n_local * 7);
    int32_t row = 0;
    for (int k = me.lo[2]; k < me.lo[2] + me.n[2]; ++k)
        for (int j = me.lo[1]; j < me.lo[1] + me.n[1]; ++j)
            for (int i = me.lo[0]; i < me.lo[0] + me.n[0]; ++i, ++row) {
                struct Entry {
                    int32_t c;
                    double v, x;
                } e[7];
                int ne = 0;
                double diag = 0.0;
                const int nb[6][3] = {{i - 1, j, k}, {i + 1, j, k}, {i, j - 1, k}, {i, j + 1, k}, {i, j, k - 1}, {i, j, k + 1}};
                for (const auto &q : nb) {
                    if (q[0] < 0 || q[0] >= N || q[1] < 0 || q[1] >= N || q[2] < 0 || q[2] >= N) continue;
                    e[ne++] = Entry{row_of(q[0], q[1], q[2]), off, exact(q[0], q[1], q[2])};
                    diag -= off;
                }
                e[ne++] = Entry{row_of(i, j, k), diag, exact(i, j, k)};
                for (int a = 1; a < ne; ++a)  // ascending columns, as MatGetRowIJ delivers them
                    for (int c = a; c > 0 && e[c - 1].c > e[c].c; --c) std::swap(e[c - 1], e[c]);
                double s = 0.0;
                for (int a = 0; a < ne; ++a) {
                    col.push_back(e[a].c);
                    val.push_back(e[a].v);
                    s += e[a].v * e[a].x;
                }
                rowptr.push_back((int32_t)col.size());
                b[(size_t)row] = s;
                xs[(size_t)row] = exact(i, j, k);
            }

    // ---- the solver: the reference's AmgX-style file (PCG + AMG, V(1,1) in AmgX's terms), relative tolerance
    char cfg[1024];
    std::snprintf(cfg, sizeof cfg,
                  "config_version=2\nsolver(pcgf)=PCG\npcgf:max_iters=500\npcgf:tolerance=%.3e\npcgf:convergence=RELATIVE_INI\n"
                  "pcgf:norm=L2\npcgf:monitor_residual=1\npcgf:store_res_history=1\npcgf:preconditioner(prec)=AMG\nprec:cycle=V\n"
                  "prec:presweeps=1\nprec:postsweeps=1\nprec:smoother=BLOCK_JACOBI\nprec:relaxation_factor=0.9\n",
                  tol);
    pib_solver *s = nullptr;
    int e = pib_create_from_string(&s, "poisson", cfg, rank, size, size > 1 ? uid : nullptr, devices > 0 ? lrank % devices : lrank);
    if (e) die(rank, "pib_create", e);
    e = pib_set_csr_i32(s, n_local, (int32_t)me.first, (int32_t)n_global, rowptr.data(), col.data(), val.data());
    if (e) die(rank, "pib_set_csr_i32", e);
    e = pib_solve(s, x.data(), b.data());
    if (e) die(rank, "pib_solve", e);
    int iters = 0;
    double res = 0.0;
    pib_get_iters(s, &iters);
    pib_get_residual(s, &res);

    // ---- checks: the error against x* with the constants removed (the operator's null space), the true residual
    double loc[1] = {0.0}, glob[1];
    for (int32_t r = 0; r < n_local; ++r) loc[0] += x[(size_t)r] - xs[(size_t)r];
    MPI_Allreduce(loc, glob, 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    const double shift = glob[0] / (double)n_global;
    std::vector<double> ax((size_t)n_local);
    e = pib_mat_mult(s, x.data(), ax.data());
    if (e) die(rank, "pib_mat_mult", e);
    double emax_loc = 0.0, sums[2] = {0.0, 0.0}, gs[2], emax = 0.0;
    for (int32_t r = 0; r < n_local; ++r) {
        const double d = x[(size_t)r] - xs[(size_t)r] - shift, rr = b[(size_t)r] - ax[(size_t)r];
        emax_loc = std::fmax(emax_loc, std::fabs(d));
        sums[0] += rr * rr;
        sums[1] += b[(size_t)r] * b[(size_t)r];
    }
    MPI_Allreduce(&emax_loc, &emax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    MPI_Allreduce(sums, gs, 2, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    const double rel = std::sqrt(gs[0] / gs[1]);
    char runs[2048] = "";
    pib_describe(s, runs, (int)sizeof runs);
    pib_destroy(s);
    int bad = (rel > 10.0 * tol || emax > 1e-6 || iters < 1 || iters > 100) ? 1 : 0, anybad = 0;
    MPI_Allreduce(&bad, &anybad, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (rank == 0) {
        std::printf("{\"ranks\": %d, \"grid\": [%d, %d, %d], \"n\": %d, \"iters\": %d, \"residual\": %.6e, \"true_rel_residual\": %.6e, "
                    "\"max_error\": %.6e, \"ok\": %s}\n",
                    size, grid[0], grid[1], grid[2], N, iters, res, rel, emax, anybad ? "false" : "true");
        std::printf("runs: %s\n", std::strtok(runs, "\n"));
    }
    MPI_Finalize();
    return anybad;
}
