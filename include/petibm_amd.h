/*
 * petibm_amd.h -- C ABI of the MI355X-native linear-solve backend for PetIBM.
 *
 * This is the drop-in boundary: a PETSc-free, torch-free
 * shared library (libpetibm_amd.so, hand-written HIP for gfx950 + RCCL) whose
 * entry points are exactly what PetIBM's linear-solver plugin would bind.
 * Each entry point cites the reference interface it replaces
 * (paths relative to the PetIBM source tree).
 *
 * Conventions mirrored from the reference:
 *   - every function returns an int error code, 0 = success, propagated by the
 *     caller with CHKERRQ (all reference methods return PetscErrorCode);
 *     the non-zero values are PETSc's own PETSC_ERR_* integers so a PETSc
 *     caller can hand them straight to CHKERRQ;
 *   - non-convergence of a "CPU"/KSP-flavoured solver is an ERROR
 *     (PETSC_ERR_CONV_FAILED, src/linsolver/linsolverksp.cpp:96-104); an
 *     AmgX-flavoured solver does not check (src/linsolver/linsolveramgx.cpp:90-99);
 *     both are selectable with the config key `error_if_not_converged`;
 *   - the caller owns matrices and vectors; the solver copies the matrix at
 *     set time and may be given a new one any number of times
 *     (applications/rigidkinematics/rigidkinematics.cpp:135);
 *   - all calls are collective over the ranks given at pib_create, one rank
 *     per GPU (the reference is collective over PETSC_COMM_WORLD,
 *     src/linsolver/linsolverksp.cpp:62, src/linsolver/linsolveramgx.cpp:69);
 *   - single-threaded per rank.
 */
#ifndef PETIBM_AMD_H
#define PETIBM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: the PETSc integers the reference's CHKERRQ propagates ---- */
#define PIB_SUCCESS 0
#define PIB_ERR_MEM 55             /* PETSC_ERR_MEM */
#define PIB_ERR_SUP 56             /* PETSC_ERR_SUP: createbn.cpp:27 uses 56 for order < 1 */
#define PIB_ERR_ORDER 58           /* PETSC_ERR_ORDER: solve before setMatrix */
#define PIB_ERR_ARG_SIZ 60         /* PETSC_ERR_ARG_SIZ */
#define PIB_ERR_ARG_WRONG 62       /* PETSC_ERR_ARG_WRONG: src/linsolver/linsolver.cpp:85-88 */
#define PIB_ERR_ARG_OUTOFRANGE 63  /* PETSC_ERR_ARG_OUTOFRANGE */
#define PIB_ERR_FILE_OPEN 65       /* PETSC_ERR_FILE_OPEN */
#define PIB_ERR_MAT_LU_ZRPVT 71    /* PETSC_ERR_MAT_LU_ZRPVT: zero pivot in the direct solver */
#define PIB_ERR_LIB 76             /* PETSC_ERR_LIB: a HIP / RCCL call failed */
#define PIB_ERR_CONV_FAILED 82     /* PETSC_ERR_CONV_FAILED: linsolverksp.cpp:100 */
#define PIB_ERR_ARG_NULL 85        /* PETSC_ERR_ARG_NULL */
#define PIB_ERR_ARG_UNKNOWN_TYPE 86 /* PETSC_ERR_ARG_UNKNOWN_TYPE: src/misc/delta.cpp:58-60 */
#define PIB_ERR_MAX_VALUE 99       /* PETSC_ERR_MAX_VALUE: body outside the domain, singlebodypoints.cpp:99-104 */

/* ---- convergence reasons (PETSc KSPConvergedReason numbering) ---- */
#define PIB_CONVERGED_RTOL 2
#define PIB_CONVERGED_ATOL 3
#define PIB_CONVERGED_ITS 4
#define PIB_DIVERGED_ITS (-3)
#define PIB_DIVERGED_DTOL (-4)
#define PIB_DIVERGED_BREAKDOWN (-5)
#define PIB_DIVERGED_INDEFINITE_PC (-8)
#define PIB_DIVERGED_NANORINF (-9)
#define PIB_DIVERGED_INDEFINITE_MAT (-10)

/* ---- null-space convention of the Poisson system
 *      (applications/navierstokes/navierstokes.cpp:395-429) ---- */
#define PIB_NULLSPACE_NONE 0
#define PIB_NULLSPACE_CONSTANT 1 /* "PETSc KSP": MatSetNullSpace(constant) (:404-412) */
#define PIB_NULLSPACE_PINNED 2   /* "NVIDIA AmgX": MatZeroRowsColumns(row 0, diag 1) (:414-420) */

typedef struct pib_solver pib_solver;

/* Message of the last error raised on this thread (never NULL). */
const char *pib_last_error(void);

/* Library version (major*10000 + minor*100 + patch). */
int pib_version(void);

/* ---- multi-GPU bootstrap ------------------------------------------------
 * One rank per GPU.  Rank 0 calls pib_comm_unique_id(), the host program
 * broadcasts the 128 bytes (MPI_Bcast in PetIBM, torch.distributed in
 * bench.py) and every rank passes them to pib_create.  Replaces the
 * communicator argument of AmgXSolver::initialize(PETSC_COMM_WORLD, ...)
 * (src/linsolver/linsolveramgx.cpp:69). */
#define PIB_UID_BYTES 128
int pib_comm_unique_id(void *uid_out /* PIB_UID_BYTES */);

/* PEER transport (one node): one process per rank, every rank's windows mapped by all the others through HIP IPC (peer
 * memory over xGMI); the ranks meet in a POSIX shared-memory segment whose name this id carries.  Rank 0 calls
 * pib_comm_peer_id, the id travels to the other ranks like the RCCL one, every rank passes it to pib_create.  Two
 * orderings of the collectives:
 *   device-ordered (default): a rank STORES its boundary planes straight into the receiver's window, a release flag
 *     follows in stream order, the receiver's kernel spins on it -- no host thread and no RCCL launch on the path of a
 *     collective, and nothing per call that a captured iteration graph could not replay;
 *   host-ordered (pib_comm_peer_id_ordered(uid, 0) or PIB_PEER_ORDER=host): device-to-device copies ordered by host
 *     functions and host threads waiting on shared counters -- the first implementation, the reference the other is
 *     compared with (same bits).
 * Several ranks may share a GPU (RCCL refuses that), so this is also how the multi-process path is tested on a one-GPU
 * box.  PIB_PEER_TIMEOUT_S (default 600; device-side spins: at most 60) bounds every wait for another rank. */
int pib_comm_peer_id(void *uid_out /* PIB_UID_BYTES */);
int pib_comm_peer_id_ordered(void *uid_out /* PIB_UID_BYTES */, int device_ordered);
/* Wall time per plane exchange (`count` doubles each way) and per scalar all-reduce on the transport `s` has attached,
 * `reps` back-to-back calls each; s == NULL: RCCL in a one-rank world (ring neighbours = the rank itself).  Collective.
 * tools/comm_latency.py */
int pib_comm_latency(pib_solver *s, int64_t count, int reps, double usec[2]);

/* TEST transport: `nranks` ranks = host threads of ONE process sharing ONE GPU (RCCL refuses several
 * ranks per device).  Fills a PIB_UID_BYTES id to pass to pib_create from every thread.  Used by the
 * parity tests to run the multi-rank algorithm on a single-GPU box; never by an application. */
int pib_comm_loopback_create(int nranks, void *uid_out /* PIB_UID_BYTES */);
int pib_comm_loopback_destroy(const void *uid);

/* TEST: the RCCL calls of the production transport (grouped ncclSend / ncclRecv incl. the periodic ring of plane and of
 * segmented -- packed velocity -- exchanges, in-place
 * ncclAllReduce, ncclAllGather / grouped ncclBroadcast, an exchange on the communication stream ordered with events) in a
 * ONE-rank RCCL world on `device`, on a vector of n_owned entries with `ghost` ghost entries at either end; *max_err_out
 * is the largest deviation of what arrived from what must arrive (0), *comm_ranks_out ncclCommCount (1).  The multi-rank
 * algorithm itself runs through the loopback transport; a one-GPU box can do no more with RCCL than this. */
int pib_comm_selftest(int device, int64_t n_owned, int64_t ghost, double *max_err_out, int *comm_ranks_out);

/* ---- life cycle ---------------------------------------------------------
 * pib_create replaces LinSolverAmgX::LinSolverAmgX + init
 * (src/linsolver/linsolveramgx.cpp:20-75; AmgXSolver::initialize(comm,"dDDI",cfg))
 * and LinSolverKSP::init (src/linsolver/linsolverksp.cpp:48-69).
 *   name      solver name ("velocity", "poisson", "forces"): also the option
 *             prefix `-<name>_` of a PETSc-style options file
 *   cfg_path  solver configuration file, or NULL / "None" for defaults.  Two
 *             syntaxes are auto-detected: the AmgX key=value subset used by
 *             every reference example (the *.info files under examples/<case>_GPU/config) and
 *             the PETSc options subset (the <name>_solver.info files under examples/<case>/config)
 *   device    HIP device ordinal (-1: rank % device count)
 */
int pib_create(pib_solver **s, const char *name, const char *cfg_path, int rank, int nranks,
               const void *uid_or_null, int device);
/* Same, configuration given as text (tests; no temp files). */
int pib_create_from_string(pib_solver **s, const char *name, const char *cfg_text, int rank, int nranks,
                           const void *uid_or_null, int device);
/* Parse a configuration exactly as pib_create would and write a one-line
 * normalised description ("flavor=amgx method=cg pc=gmg ...") into buf.  Pure
 * host code (no GPU needed): lets a caller validate a *_solver.info file. */
int pib_config_describe(const char *name, const char *cfg_text, char *buf, int buflen);
/* LinSolverBase::destroy / AmgXSolver::finalize (linsolveramgx.cpp:37,47). */
int pib_destroy(pib_solver *s);

/* LinSolverBase::getType (include/petibm/linsolver.h:97): "NVIDIA AmgX" for a
 * solver created from an AmgX-style config or `type: GPU`, "PETSc KSP" for a
 * PETSc-options config -- unchanged applications select the null-space
 * convention from this string (navierstokes.cpp:402-426). */
int pib_get_type(pib_solver *s, char *buf, int buflen);

/* What this solver RUNS (pib_config_describe says what its file asks for): first line key=value pairs -- type, method, pc,
 * product (csr_int32_columns | csr_column_codes | csr_row_patterns | matrix_free_stencil | matrix_free_velocity), partition,
 * ranks, levels, smoother, presteps / poststeps (smoothing STEPS per level: a sweep of the file is a pair unless
 * pib_sweep_pairs=0), nullspace, structure, residual_update (PCG + multigrid: in_vcycle when r -= alpha w rides in the cycle's first
 * march, separate_pass when the system is too small or not tile-divisible for that), placement_searches --, then one "departure: ..." line for every place where the
 * backend departs from the file (CG -> BiCGStab for a non-symmetric DBNG, ...).  getType keeps the reference's two strings
 * (navierstokes.cpp:402-426 compares them); LinSolverBase::printInfo (include/petibm/linsolver.h:103) prints this text under
 * the reference's banner.  Valid at any time after pib_create; the product is "none" before setMatrix. */
int pib_describe(pib_solver *s, char *buf, int buflen);

/* ---- setMatrix ------------------------------------------------------------
 * LinSolverBase::setMatrix(const Mat&) / AmgXSolver::setA(A)
 * (src/linsolver/linsolveramgx.cpp:84, src/linsolver/linsolverksp.cpp:78-79).
 * Local rows [row0, row0+n_local) of an assembled AIJ matrix, GLOBAL column
 * indices, host pointers (what MatMPIAIJGetLocalMat/MatGetRowIJ give).
 * The matrix is copied to HBM; the arrays may be freed on return. */
int pib_set_csr(pib_solver *s, int64_t n_local, int64_t row0_global, int64_t n_global, const int64_t *rowptr,
                const int64_t *col_global, const double *val);
/* PetscInt is 32-bit by default (AmgX mode dDDI): same with int32 indices. */
int pib_set_csr_i32(pib_solver *s, int32_t n_local, int32_t row0_global, int32_t n_global, const int32_t *rowptr,
                    const int32_t *col_global, const double *val);

/* Optional structure hint that enables the matrix-free stencil twin and the
 * geometric-multigrid preconditioner: the matrix set by pib_set_csr is the
 * (5/7-point) operator
 *     (A x)_ijk = sum_d  cx_d[s] * w_d(perp) * (x_{s+1} - x_s) - (same at s-1)
 * of a tensor-product grid in natural ordering i + nx*(j + ny*k), i.e. the
 * DBNG of applications/navierstokes/navierstokes.cpp:349-356 for BN order 1:
 *   n[d]   global cell counts (n[2] = 1 in 2D)
 *   w[d]   cell widths dL[3][d][0..n[d])             (createdivergence.cpp:140-151)
 *   g[d]   dt / dL[d][d][s], s in [0, n[d]-1)          (creategradient.cpp:70-86, createbn.cpp:49)
 * The hint is verified against the CSR on the device; a mismatch is an error. */
int pib_set_grid_hint(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                      const double *wz, const double *gx, const double *gy, const double *gz, int nullspace);

/* The grid structure the solver holds (from pib_set_grid_hint, an on-device assembly, or recovered from the matrix):
 * *has = 0 none; dim, n[3] (problem order), nullspace as registered; *detected != 0 when pib_set_csr recovered it from
 * the CSR itself.  With a multigrid (AMG) preconditioner pib_set_csr[_i32] inspects the matrix: the 5/7-point DBNG of a
 * tensor-product mesh (periodic directions included: they show as wrapped neighbours and are set as pib_set_periodic would)
 * in natural ordering on z-slabs (y-slabs in 2-D) factorises into 1-D arrays
 * (csrc/structure.cpp), the recovered operator is verified against the CSR on the device, and an application that only
 * ever calls setMatrix -- AmgXSolver::setA, src/linsolver/linsolveramgx.cpp:84 -- gets the geometric multigrid without
 * registering anything.  Any other matrix is left without structure (`pib_detect_structure=0` switches the search off).
 * Any output pointer may be NULL. */
int pib_get_grid_structure(pib_solver *s, int *has, int *dim, int64_t n[3], int *nullspace, int *detected);

/* The hierarchy of the geometric multigrid this solver holds (diagnostics; what AmgX prints with print_grid_stats=1,
 * examples/navierstokes/liddrivencavity2dRe1000_GPU/config/poisson_solver.info): *nlevels = number of levels (0: none),
 * n3[3 l .. 3 l + 2] = cells per direction of level l for l < min(*nlevels, max_levels); n3 may be NULL.  The selective
 * coarsening merges the finest cells of a stretched mesh first, so the sizes need not halve. */
int pib_get_multigrid_levels(pib_solver *s, int *nlevels, int64_t *n3, int max_levels);

/* The structure of the velocity operator A = I/dt - c nu L the solver holds for its matrix-free products (16 B/row instead
 * of the CSR's 104): *has = 0 none; dim, n[3] = pressure cells per direction, periodic[3]; *detected != 0 when
 * pib_set_csr[_i32] recovered it from the matrix -- vSolver->setMatrix(A) of an unchanged PetIBM, navierstokes.cpp:345:
 * in the packed [u | v | w] ordering one line of entries per field and direction is the coefficient table, a wall's ghost
 * fold is read off a boundary point's diagonal (csrc/structure.cpp), and the recovered product is verified against the CSR
 * SpMV on the device (1e-12) before it is used.  The recovery runs on one rank (pib_assemble_velocity holds the structure
 * on slabs too: the neighbour rank's planes are read from the ghost pads); a preconditioner other than AMG; `pib_detect_structure=0` or
 * `pib_matrix_free_velocity=0` switch it off.  Any output pointer may be NULL. */
int pib_get_velocity_structure(pib_solver *s, int *has, int *dim, int64_t n[3], int periodic[3], int *detected);

/* Periodic directions of the mesh (flow.boundaryConditions type PERIODIC at both ends of a direction for every
 * component: src/misc/misc.cpp checkPeriodicBC, cartesianmesh.cpp:595-681 wraps the neighbour indices).  Call BEFORE
 * pib_assemble_poisson / pib_assemble_velocity / pib_set_grid_hint: the assembled operators then carry the wrapped
 * columns (component d has n[d] points along a periodic d instead of n[d]-1, cartesianmesh.cpp:259-266) and the grid
 * hint takes one more face factor per periodic direction, g[d][n[d]-1] = dt / (0.5*(w[0] + w[n[d]-1])).  A periodic
 * direction needs >= 3 cells.  A periodic SLAB axis on several ranks makes rank 0 and rank P-1 neighbours (ring halo):
 * pib_assemble_poisson, pib_assemble_velocity, the pib_set_csr route (columns across the seam are recognised) and the
 * time step on slabs all provide it. */
int pib_set_periodic(pib_solver *s, const int periodic[3]);

/* Assemble the Poisson operator DBNG = D * (dt*I) * G directly in HBM from the
 * mesh widths (the product of createDivergence(normalize=FALSE),
 * createBnHead(N=1) and createGradient(normalize=FALSE) --
 * navierstokes.cpp:326-356 -- evaluated in the same floating-point order),
 * for this rank's z-slab [k0,k1) (y-slab in 2D), set it as the solver's matrix
 * and register the grid hint.  pinned != 0 applies MatZeroRowsColumns(row 0,
 * diag = 1) (navierstokes.cpp:414-420).  w[d] has n[d] entries, dt scalar. */
int pib_assemble_poisson(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                         const double *wz, double dt, int nullspace);

/* The Poisson operator for `parameters.BN` > 1: DBNG = D * BN * G with
 *   BN = sum_{k=1..order} dt^k (coeff_nu)^(k-1) L^(k-1)      createBnHead, src/operators/createbn.cpp:19-95
 * built in HBM by the reference's own chain -- assembled G (creategradient.cpp:64-128), D (createdivergence.cpp:135-223)
 * and L (createlaplacian.cpp:108-263), MatMatMult / MatAXPY(DIFFERENT_NONZERO_PATTERN) in PETSc's summation order
 * (navierstokes.cpp:349-356) -- so the matrix is bit-identical to what the application would hand to setMatrix.
 * lo / hi / a0 as in pib_assemble_velocity; coeff_nu = implicit diffusion coefficient * nu (0.5 nu for Crank-Nicolson).
 * order 1 is pib_assemble_poisson.  The 13 / 25-point operator is solved with the CSR SpMV; an AMG entry in the solver
 * file is served by the multigrid of the 7-point order-1 operator (same mesh) as preconditioner.  On several ranks
 * (n, the widths, lo / hi of the WHOLE mesh, as pib_assemble_poisson takes them) a rank runs the chain on a window of the
 * mesh around its slab and keeps its rows -- the entries of the one-rank matrix, ghost columns `order` planes deep; a slab
 * must hold at least `order` planes.  order < 1 -> PIB_ERR_SUP like createBnHead; honours pib_set_periodic. */
int pib_assemble_poisson_bn(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                            const double *wz, const double lo[3], const double hi[3], const double a0[18], double dt,
                            double coeff_nu, int bn_order, int nullspace);

/* Assemble the velocity operator A = I/dt - c*nu*L directly in HBM and set it as the solver's matrix:
 * createLaplacian (src/operators/createlaplacian.cpp:108-263, incl. the ghost-point a0 fold :232-243) on the
 * packed (u,v[,w]) ordering, then MatScale(-c*nu) + MatShift(1/dt) (navierstokes.cpp:342-344), evaluated in
 * the reference's floating-point order.  w[d]: pressure-cell widths; lo/hi: domain start / end per direction
 * (mesh->min / mesh->max); a0[6*f + loc]: ghost coefficient of field f at boundary loc (xMinus,xPlus,yMinus,
 * yPlus,zMinus,zPlus): Dirichlet/convective 0 when loc's normal is f else -1, Neumann 1
 * (src/boundary/singleboundary{dirichlet,neumann,convective}.cpp).  Honours pib_set_periodic.  On several ranks every
 * rank assembles the rows of its slab of the packed [u | v | w] ordering with a segmented halo plan; with a
 * periodic slab axis rank 0 and rank P - 1 are neighbours (every rank has both ghost pads, the exchange is a ring). */
int pib_assemble_velocity(pib_solver *s, int dim, const int64_t n[3], const double *wx, const double *wy,
                          const double *wz, const double lo[3], const double hi[3], const double a0[18], double dt,
                          double coeff_nu);

/* The z-slab (y-slab in 2D) of planes [*begin, *end) that rank `rank` of `nranks`
 * owns: the DMDA default split m = N/P + ((N % P) > rank) the reference gets from
 * DMDACreate3d (src/mesh/cartesianmesh.cpp:492-538).  Pure host code. */
int pib_slab_range(int64_t nplanes, int nranks, int rank, int64_t *begin, int64_t *end);

/* ---- solve ------------------------------------------------------------------
 * LinSolverBase::solve(Vec &x, Vec &b) / AmgXSolver::solve(x, b)
 * (src/linsolver/linsolveramgx.cpp:96, src/linsolver/linsolverksp.cpp:92).
 * x, b: this rank's n_local entries; each pointer may be host or device
 * memory (detected).  b is read-only.  x is in/out: an AmgX-flavoured solver
 * uses it as the initial guess, a KSP-flavoured one zeroes it first unless
 * `ksp_initial_guess_nonzero` is set.
 * Returns PIB_ERR_CONV_FAILED when error_if_not_converged is on and the
 * iteration diverged or hit max_iters. */
int pib_solve(pib_solver *s, double *x_inout, const double *b);

/* LinSolverBase::getIters (linsolveramgx.cpp:108, linsolverksp.cpp:116). */
int pib_get_iters(pib_solver *s, int *iters);
/* LinSolverBase::getResidual: final residual norm of the last solve
 * (linsolveramgx.cpp:120-123 -> AmgXSolver::getResidual(iters, res);
 *  linsolverksp.cpp:128 -> KSPGetResidualNorm). */
int pib_get_residual(pib_solver *s, double *res);
/* AmgXSolver::getResidual(iter, res): entry `iter` of the stored history. */
int pib_get_residual_at(pib_solver *s, int iter, double *res);
/* KSPGetConvergedReason (linsolverksp.cpp:94). */
int pib_get_reason(pib_solver *s, int *reason);

/* ---- MatMult on the solver's matrix (K1; also what the applications call in
 * RHS assembly, navierstokes.cpp:442,490,549,592).  host or device pointers. */
int pib_mat_mult(pib_solver *s, const double *x, double *y);

/* ---- device-side helpers for callers that keep vectors in HBM (bench.py) ---- */
int pib_device_alloc(pib_solver *s, int64_t nbytes, void **ptr);
/* free / total bytes of the solver's device (hipMemGetInfo): what the bounded placement search budgets against */
int pib_device_mem_info(pib_solver *s, int64_t *free_bytes, int64_t *total_bytes);
int pib_device_free(pib_solver *s, void *ptr);
int pib_memcpy_h2d(pib_solver *s, void *dst, const void *src, int64_t nbytes);
int pib_memcpy_d2h(pib_solver *s, void *dst, const void *src, int64_t nbytes);
int pib_synchronize(pib_solver *s);
/* Download the solver's local CSR (tests: device assembly vs oracle).
 * Pass NULL arrays to query sizes only. */
int pib_get_csr(pib_solver *s, int64_t *n_local, int64_t *nnz, int64_t *rowptr, int64_t *col_global, double *val);

/* ---- device-resident time step --------------------------------------------------
 * One NavierStokesSolver::advance (applications/navierstokes/navierstokes.cpp:240-266) with all vectors in HBM:
 * assembleRHSVelocity (:432-521), solveVelocity (:524-537), assembleRHSPoisson (:540-563), solvePoisson
 * (:566-580), applyDivergenceFreeVelocity (:583-598), updatePressure (:601-615).  G, D, L, BNG and the
 * convective term N(u) (src/operators/createconvection.cpp) are applied matrix-free in the summation order of
 * the reference's assembled matrices; AB2 convection + Crank-Nicolson diffusion and BN order 1 unless
 * pib_ns_set_time_integration / pib_ns_set_bn_order say otherwise.
 *   bc_type[6*f+loc]: 0 DIRICHLET, 1 NEUMANN, 2 CONVECTIVE; bc_value[6*f+loc] (flow.boundaryConditions of the YAML
 *   file; ghost points keep the reference's per-point state: src/boundary/singleboundary*.cpp);
 *   velocity_cfg / poisson_cfg: solver configuration TEXT (same syntaxes as pib_create).  The Poisson
 *   solver's flavour selects the null-space convention exactly like NavierStokesSolver::setNullSpace
 *   (:395-429): "NVIDIA AmgX" -> pinned row 0 and rhs2[0] = 0, "PETSc KSP" -> constant null space.
 * One GPU; pib_ns_create_slab is the same engine on z-slabs. */
typedef struct pib_ns pib_ns;
int pib_ns_create(pib_ns **ns, int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                  const double lo[3], const double hi[3], const int bc_type[18], const double bc_value[18], double dt,
                  double nu, const char *velocity_cfg, const char *poisson_cfg, int device);
/* The same engine on this rank's z-slab (y-slab in 2-D) of the mesh: NavierStokesSolver on the DMDA decomposition of
 * src/mesh/cartesianmesh.cpp:492-538 with nProc = (1, 1, P) (BASELINE configs 3 and 5 are stated on 8
 * GPUs).  Arguments as pib_ns_create -- the GLOBAL mesh and boundary conditions -- plus rank / nranks / uid as
 * pib_create.  Collective.  A rank keeps its planes of every field plus one plane of each neighbour; the two solvers
 * share one communicator; per step: one exchange of u* after the velocity solve, one of dP after the Poisson solve, one
 * of the projected velocity.  pib_ns_sizes / pib_ns_set_state / pib_ns_get_state / the history terms then speak of this
 * rank's part of the distributed vectors: the packed [u-slab | v-slab | w-slab] of the reference's DMComposite
 * (cartesianmesh.cpp:740-779; the component along the slab axis has one plane fewer, on the last rank) and the owned
 * pressure cells.  Every rank needs >= 2 planes.  Immersed bodies (pib_ns_set_bodies / pib_ns_move_bodies, collective,
 * the same bodies on every rank; direct forces solver): every rank assembles the operators on the velocity points it
 * owns, E u and the force system E BN H are summed over the ranks, the forces are replicated.  A periodic slab axis (the
 * Taylor-Green box, the cylinder array of multicylinders2dRe100 on several GPUs) makes both ends of every rank's slab a
 * cut and the plane exchanges a ring.  BN order > 1 (pib_ns_set_bn_order) works on slabs too: the operator from per-rank windows of the product chain, the
 * projection by applying BN term by term with the velocity solver's matrix-free L (one exchange per extra term).  Immersed
 * bodies with BN order > 1 on slabs: E BN H is assembled densely, one column per force unknown through the same term-by-term
 * BN, for the direct forces solver.  Not on slabs: the coupled IBPM, the vorticity utility (PIB_ERR_SUP). */
int pib_ns_create_slab(pib_ns **ns, int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                       const double lo[3], const double hi[3], const int bc_type[18], const double bc_value[18], double dt,
                       double nu, const char *velocity_cfg, const char *poisson_cfg, int rank, int nranks,
                       const void *uid_or_null, int device);
/* parameters.BN of config.yaml (default 1): order of BN in the Poisson operator D*BN*G and in the projection
 * u = u* - BN G dP (navierstokes.cpp:349-356,583-598).  Call after pib_ns_create, before the first step.  N > 1 builds
 * the operator through pib_assemble_poisson_bn's product chain and keeps the assembled BN: immersed bodies set afterwards
 * (pib_ns_set_bodies / pib_ns_move_bodies) build BNH = BN H and EBNH = E BNH from it through the same chain
 * (applications/decoupledibpm/decoupledibpm.cpp:194-205) -- so call this BEFORE pib_ns_set_bodies (PIB_ERR_ORDER
 * otherwise).  Not with the coupled IBPM (PIB_ERR_SUP). */
int pib_ns_set_bn_order(pib_ns *ns, int order);
/* parameters.convection / parameters.diffusion of config.yaml (createTimeIntegration, src/timeintegration/
 * timeintegration.cpp:41-80): "EULER_EXPLICIT" | "EULER_IMPLICIT" | "ADAMS_BASHFORTH_2" | "CRANK_NICOLSON" for either
 * term (include/petibm/timeintegration.h:107-166 for the coefficients); default ADAMS_BASHFORTH_2 + CRANK_NICOLSON.
 * Call after pib_ns_create, before pib_ns_set_bodies and the first step; the velocity operator is re-assembled when
 * the implicit coefficient changes.  Unknown name -> PIB_ERR_ARG_OUTOFRANGE like the reference. */
int pib_ns_set_time_integration(pib_ns *ns, const char *convection, const char *diffusion);
/* One explicit term kept between steps, for restart files with any scheme (/convection/<index>, /diffusion/<index>):
 * kind 0 convection, 1 diffusion; set != 0 uploads `host`, 0 downloads into it (UN entries). */
int pib_ns_history_term(pib_ns *ns, int kind, int index, int set, double *host);
/* The vorticity of the reference's post-processing utility petibm-vorticity (applications/vorticity/main.cpp:185-372,
 * fields and their point sets :384-470), from the current velocity and ghost values: comp 0 wx, 1 wy, 2 wz (2-D: wz at
 * the vertices).  n_out[3] receives the point counts; out == NULL only queries them. */
int pib_ns_get_vorticity(pib_ns *ns, int comp, int64_t n_out[3], double *out);
int pib_ns_sizes(pib_ns *ns, int64_t *UN, int64_t *pN);
int pib_ns_set_state(pib_ns *ns, const double *U_packed_or_null, const double *p_or_null);        /* host arrays */
int pib_ns_get_state(pib_ns *ns, double *U, double *p, double *rhs1, double *rhs2);               /* any may be NULL */
int pib_ns_advance(pib_ns *ns, int nsteps);
/* explicit terms kept between steps, for restart files (navierstokes.cpp:637-686,689-746: /convection/0,
 * /convection/1, /diffusion/0); host arrays of UN entries, any may be NULL */
int pib_ns_get_history(pib_ns *ns, double *conv0, double *conv1, double *diff0);
int pib_ns_set_history(pib_ns *ns, const double *conv0, const double *conv1);
/* Stage timers under the reference's PetscLogStage names -- "rhsVelocity", "solveVelocity", "rhsPoisson", "solvePoisson",
 * "update" (applications/navierstokes/navierstokes.cpp:186-199; pushed / popped at :436-530, :532, :540-572, :575, :583-615)
 * and, with immersed bodies, "rhsForces" + "solveForces" as ONE entry "solveForces" (applications/decoupledibpm/
 * decoupledibpm.cpp:93-97, 259-282).  pib_ns_stage_timers(ns, 1) switches them on and zeroes the sums (off by default: a step
 * then records seven events on the engine's stream and waits for the last one); pib_ns_get_stage_times returns the
 * milliseconds accumulated per stage in the order of pib_ns_stage_name(0..5) = rhsVelocity, solveVelocity, solveForces,
 * rhsPoisson, solvePoisson, update, and the number of steps they cover.  What PETSc's -log_view prints per stage. */
int pib_ns_stage_timers(pib_ns *ns, int enable);
int pib_ns_get_stage_times(pib_ns *ns, double ms[6], int64_t *steps);
const char *pib_ns_stage_name(int stage);
/* the columns of iterations-<start>.txt (navierstokes.cpp:766-794) for the last step */
int pib_ns_get_solver_info(pib_ns *ns, int *v_iters, double *v_res, int *p_iters, double *p_res);
/* writeLinSolversInfo (navierstokes.cpp:780-787): pib_describe of the engine's velocity (0) or Poisson (1) solver */
int pib_ns_describe_solver(pib_ns *ns, int which, char *buf, int buflen);
int pib_ns_destroy(pib_ns *ns);

/* ---- immersed bodies: DecoupledIBPMSolver (applications/decoupledibpm/decoupledibpm.cpp) ----------------------
 * pib_ns_set_bodies turns the engine into the decoupled IBPM: it assembles on the device Delta
 * (src/operators/createdelta.cpp:34-208, kernels src/misc/delta.cpp:17-43: "ROMA_ET_AL_1999" | "PESKIN_2002"),
 * E = Delta R MHat, H = Delta^T, BNH = dt H and EBNH = E BNH (decoupledibpm.cpp:141-216), creates the forces solver
 * from `forces_cfg` (the text of forces_solver.info; the examples use -forces_ksp_type preonly -forces_pc_type lu)
 * and hands it EBNH; pib_ns_advance then runs DecoupledIBPMSolver::advance (:105-131).
 *   npts[b]: Lagrangian points of body b; coords: all points, body after body, dim values per point (the body
 *   files of the reference, src/io/io.cpp:23-118); force unknown of (point q, direction d) = q*dim + d over the
 *   concatenated points (src/body/bodypack.cpp:261-283 on one rank).
 * Errors: unknown kernel -> PIB_ERR_ARG_UNKNOWN_TYPE, a point outside the domain -> PIB_ERR_MAX_VALUE. */
int pib_ns_set_bodies(pib_ns *ns, int nbodies, const int64_t *npts, const double *coords, const char *delta_kernel,
                      const char *forces_cfg);
/* IBPMSolver (applications/ibpm/ibpm.cpp): the coupled immersed-boundary projection method -- pressure and Lagrangian
 * forces are ONE unknown of the modified Poisson system D_c BN G_c with G_c = [G, -H], D_c = [D; E] (:110-194), so
 * no-slip and continuity hold together at the end of a step.  The engine eliminates the small forces block exactly
 * (EBNH^-1 from the direct forces solver) and runs the Poisson solver on the Schur complement; the step then is
 * IBPMSolver's: u = u* - BN G_c [dP; df], p += dP, f += df.  Call after pib_ns_set_bodies (forces solver: preonly + lu);
 * coupled = 0 goes back to the decoupled scheme. */
int pib_ns_set_coupled(pib_ns *ns, int coupled);
/* RigidKinematicsSolver::moveBodies (applications/rigidkinematics/rigidkinematics.cpp:118-140): new coordinates of all
 * points (layout of pib_ns_set_bodies) and, if not NULL, their prescribed velocities UB [nf]: the operators are
 * re-assembled on the device, the forces solver receives the new EBNH and the forces right-hand side becomes
 * UB - E u (:147-160).  Call before pib_ns_advance(ns, 1) of the step that ends at the new position. */
int pib_ns_move_bodies(pib_ns *ns, const double *coords, const double *body_velocity);
int pib_ns_num_forces(pib_ns *ns, int64_t *nf, int *nbodies);
/* Lagrangian forces f [nf] and/or the bodies' forces [nbodies*dim] = minus the sum over the body's points
 * (src/body/singlebodypoints.cpp:228-259; one line of forces-<start>.txt, decoupledibpm.cpp:437-465) */
int pib_ns_get_forces(pib_ns *ns, double *f, double *body_forces);
int pib_ns_set_forces(pib_ns *ns, const double *f);   /* restart (decoupledibpm.cpp:372-375) */
/* iterations / residual of the forces solver in the last step (third column pair of iterations-<start>.txt) */
int pib_ns_get_forces_solver_info(pib_ns *ns, int *f_iters, double *f_res);
/* Inspection of the operators (host arrays, 32-bit CSR).  which: 0 Delta, 1 E, 2 H (only the velocity points under
 * the kernels' support are stored: row_ids[n_rows] names them), 3 EBNH.  Null arrays: sizes only. */
int pib_ns_get_ib_operator(pib_ns *ns, int which, int64_t *n_rows, int64_t *nnz, int32_t *rowptr, int32_t *col,
                           double *val, int32_t *row_ids);

/* ---- instrumentation (bench.py roofline leg) --------------------------------
 * Time `reps` launches of kernel `which` on the solver's stream with HIP events
 * (events recorded on that stream); *ms_avg = average launch duration.
 * which: 0 = CSR SpMV (K1), 1 = fused CG vector update, 2 = dot,
 *        3 = matrix-free stencil product (K2), 4 = GMG V-cycle,
 *        5 = matrix-free product of the velocity operator (the kernels BiCGStab runs after pib_assemble_velocity). */
int pib_time_kernel(pib_solver *s, int which, int reps, double *ms_avg);
/* Counters of the last solve: [0]=spmv launches, [1]=pc applies, [2]=reductions,
 * [3]=halo exchanges (all-gathers included), [4]=host syncs, [5]=ranks of the communicator (ncclCommCount),
 * [6]=PCG iterations whose residual update ran inside the V-cycle's first kernel (pib_fuse_residual_update). */
int pib_get_counters(pib_solver *s, int64_t counters[8]);
/* Host-vector callers (the Vecs of linsolverksp.cpp:97-116 / AmgXSolver::solve, linsolveramgx.cpp:96-105, live in host memory):
 * milliseconds the last pib_solve spent staging b (and x, when it is the guess) to HBM and x back over PCIe; 0 / 0 when both
 * pointers were device pointers. */
int pib_get_staging_ms(pib_solver *s, double *h2d_ms, double *d2h_ms);
/* Krylov iterations launched as replays of the captured iteration graph since the solver was created (launch-bound systems:
 * the solver-file key pib_graph_max_rows, 0 = never; on several ranks with the device-ordered peer transport only). */
int pib_get_graph_replays(pib_solver *s, int64_t *replays);
/* The placement of the search direction against the caller's x (pib_place_update_vector: CG on one rank, 2^25 rows and more):
 * searches run since the solver was created (a device whose memory is more than half taken gets a search that times nothing),
 * allocations the last search timed, the probe's time (ms: the p-update's access pattern over both vectors) with the vector the
 * solver had and with the one it kept, the most bytes any search held at one time beyond the solver's own vectors (gaps,
 * reference vectors, rejected candidates: never more than min(16 GiB, a tenth of the free memory)) and the wall time of all
 * searches (ms).  All zero when no search ran; held_bytes / search_ms may be NULL. */
int pib_get_placement(pib_solver *s, int *searches, int *candidates, double *ms_had, double *ms_kept, int64_t *held_bytes, double *search_ms);
/* What the CSR product (the MatMult inside KSPSolve / AmgXSolver::solve) streams per matrix entry besides the 8-byte value: 4 (the
 * int32 column), 1 (pib_compress_columns=1: a one-byte code into the dictionary of column offsets of the entry's 256-row block,
 * built at setMatrix when no block has more than 16 distinct offsets) or 0 (pib_compress_columns=2, the default: one byte per
 * ROW, the number of the row's list of offsets in the block's table of up to 8 patterns of up to 8 entries -- every 5- / 7-point
 * stencil matrix).  Same products, same order, same bits. */
int pib_get_product_format(pib_solver *s, int *index_bytes_per_entry);

#ifdef __cplusplus
}
#endif
#endif /* PETIBM_AMD_H */
