// petsc_adapter.hpp -- the PETSc-facing half of the drop-in boundary: Mat / Vec -> the plain arrays of the C ABI.
//
// Shared by the two plug points of SURVEY.md 8b:
//   * petibm_amd::linsolver::LinSolverHIP   (linsolver.hpp, -DPIB_WITH_PETSC): the LinSolverBase subclass
//   * AmgXSolver                            (AmgXSolver.hpp): the six AmgXWrapper members linsolveramgx.cpp uses
//
// What it does is what AmgXWrapper's setA does before it hands the matrix to AmgX (src/linsolver/linsolveramgx.cpp:84):
// take the LOCAL rows of the assembled AIJ matrix with GLOBAL column indices.  PetIBM creates its operators with
// DMCreateMatrix / MatCreate + MATAIJ, which is MATSEQAIJ on one process and MATMPIAIJ on several, so the type is
// checked first: MatMPIAIJGetLocalMat on a SeqAIJ matrix is PETSC_ERR_SUP in PETSc 3.16.
//
// Needs <petscmat.h>; it has never been compiled against a real PETSc in this repository's image (none installed) --
// tests/test_boundary_headers.py runs a SYNTAX check of it against a declarations-only stub (tests/stubs/petsc).
#pragma once
#include <cstdlib>
#include <string>
#include <petscmat.h>
#include <petscvec.h>

#include <climits>
#include <cstdint>

#include "../petibm_amd.h"

namespace petibm_amd
{
namespace petsc
{
/** rank 0 draws the RCCL id, everybody receives it (the communicator argument of AmgXSolver::initialize,
 *  src/linsolver/linsolveramgx.cpp:69).  uid: PIB_UID_BYTES bytes. */
inline PetscErrorCode broadcastUniqueId(MPI_Comm comm, PetscMPIInt *rank, PetscMPIInt *size, char *uid)
{
    PetscErrorCode ierr;
    ierr = MPI_Comm_rank(comm, rank); CHKERRQ(ierr);
    ierr = MPI_Comm_size(comm, size); CHKERRQ(ierr);
    for (int i = 0; i < PIB_UID_BYTES; ++i) uid[i] = 0;
    if (*size > 1) {
        // RCCL unless PIB_TRANSPORT=peer asks for the HIP-IPC window transport (one node; petibm_amd.h, pib_comm_peer_id)
        if (*rank == 0) {
            const char *t = std::getenv("PIB_TRANSPORT");
            ierr = (t != nullptr && std::string(t) == "peer") ? pib_comm_peer_id(uid) : pib_comm_unique_id(uid); CHKERRQ(ierr);
        }
        ierr = MPI_Bcast(uid, PIB_UID_BYTES, MPI_BYTE, 0, comm); CHKERRQ(ierr);
    }
    return 0;
}

/** HIP device of this rank: its index among the ranks of the same node (one rank per GPU). */
inline PetscErrorCode localDevice(MPI_Comm comm, int *device)
{
    PetscErrorCode ierr;
    MPI_Comm node;
    PetscMPIInt lrank;
    ierr = MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node); CHKERRQ(ierr);
    ierr = MPI_Comm_rank(node, &lrank); CHKERRQ(ierr);
    ierr = MPI_Comm_free(&node); CHKERRQ(ierr);
    *device = (int)lrank;
    return 0;
}

/** setMatrix: the local rows of A (SeqAIJ or MPIAIJ), global columns, copied to HBM by pib_set_csr[_i32]. */
inline PetscErrorCode setMatrix(pib_solver *h, const Mat &A)
{
    PetscErrorCode ierr;
    PetscBool isSeq = PETSC_FALSE, isMpi = PETSC_FALSE, done = PETSC_FALSE;
    PetscInt n = 0, rstart = 0, rend = 0, N = 0;
    const PetscInt *ia = nullptr, *ja = nullptr;
    PetscScalar *va = nullptr;
    Mat lA = nullptr;  // the matrix whose CSR is read: A itself (SeqAIJ) or the merged local rows (MPIAIJ)

    ierr = PetscObjectTypeCompare((PetscObject)A, MATSEQAIJ, &isSeq); CHKERRQ(ierr);
    ierr = PetscObjectTypeCompare((PetscObject)A, MATMPIAIJ, &isMpi); CHKERRQ(ierr);
    if (!isSeq && !isMpi) SETERRQ(PetscObjectComm((PetscObject)A), PETSC_ERR_SUP, "petibm_amd: the matrix must be MATSEQAIJ or MATMPIAIJ");
    ierr = MatGetOwnershipRange(A, &rstart, &rend); CHKERRQ(ierr);
    ierr = MatGetSize(A, &N, nullptr); CHKERRQ(ierr);
    if (isMpi) {
        ierr = MatMPIAIJGetLocalMat(A, MAT_INITIAL_MATRIX, &lA); CHKERRQ(ierr);  // n_local x N, global columns
    } else {
        lA = A;
    }
    ierr = MatGetRowIJ(lA, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done); CHKERRQ(ierr);
    if (!done || n != rend - rstart) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "petibm_amd: MatGetRowIJ did not deliver the local CSR");
    ierr = MatSeqAIJGetArray(lA, &va); CHKERRQ(ierr);
    int e;
    if (sizeof(PetscInt) == 4) {
        // AmgX mode dDDI: 32-bit indices; PetscInt is 32-bit unless PETSc was configured --with-64-bit-indices
        e = pib_set_csr_i32(h, (int32_t)n, (int32_t)rstart, (int32_t)N, (const int32_t *)(const void *)ia,
                            (const int32_t *)(const void *)ja, va);
    } else {
        if ((int64_t)N > (int64_t)INT32_MAX * 1024) e = PIB_ERR_ARG_OUTOFRANGE;  // absurd size: corrupted Mat
        else e = pib_set_csr(h, (int64_t)n, (int64_t)rstart, (int64_t)N, (const int64_t *)(const void *)ia,
                             (const int64_t *)(const void *)ja, va);
    }
    ierr = MatSeqAIJRestoreArray(lA, &va); CHKERRQ(ierr);
    ierr = MatRestoreRowIJ(lA, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done); CHKERRQ(ierr);
    if (isMpi) { ierr = MatDestroy(&lA); CHKERRQ(ierr); }
    if (e) SETERRQ1(PETSC_COMM_SELF, e, "%s", pib_last_error());
    return 0;
}

/** solve: the local arrays of x (in/out) and b (read-only); host pointers, staged by pib_solve. */
inline PetscErrorCode solve(pib_solver *h, Vec &x, Vec &b)
{
    PetscErrorCode ierr;
    PetscScalar *xa = nullptr;
    const PetscScalar *ba = nullptr;
    ierr = VecGetArray(x, &xa); CHKERRQ(ierr);
    ierr = VecGetArrayRead(b, &ba); CHKERRQ(ierr);
    const int e = pib_solve(h, xa, ba);
    ierr = VecRestoreArrayRead(b, &ba); CHKERRQ(ierr);
    ierr = VecRestoreArray(x, &xa); CHKERRQ(ierr);
    if (e) SETERRQ1(PETSC_COMM_SELF, e, "%s", pib_last_error());
    return 0;
}

/** true once PetscFinalize has run: destructors of static / late objects must not touch the device then
 *  (src/linsolver/linsolveramgx.cpp:28-38). */
inline bool finalized()
{
    PetscBool f = PETSC_FALSE;
    if (PetscFinalized(&f)) return true;
    return f == PETSC_TRUE;
}
}  // namespace petsc
}  // namespace petibm_amd
