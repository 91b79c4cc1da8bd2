/* petsc_ksp_driver.c -- the reference's OWN CPU path on the bench's system: KSPSolve of PETSc on the int32 CSR the GPU
 * solves (src/linsolver/linsolverksp.cpp:48-107: KSPCreate, KSPSetOperators, KSPSetFromOptions with the solver's prefix,
 * zeroed initial guess, KSPSolve, KSPGetIterationNumber / KSPGetResidualNorm; the constant null space of
 * applications/navierstokes/navierstokes.cpp:395-408).  SURVEY.md 8d / BASELINE.md 4: built and run by bench.py ONLY when
 * a PETSc installation is found on the host (PETSC_DIR or pkg-config petsc), reported as a second cpu_baseline entry of
 * kind "petsc".  No PETSc exists in the build image: this file is syntax-checked against the declarations-only stub
 * (tests/stubs/petsc, tests/test_boundary_headers.py) and has never been linked.
 *
 *   petsc_ksp_driver <system.bin> [-poisson_ksp_type cg -poisson_pc_type gamg -poisson_ksp_rtol 1e-10 ...]
 *   system.bin: int64 n, int64 nnz, int32 rowptr[n+1], int32 col[nnz], double val[nnz], double b[n]   (bench.py writes it)
 * prints one JSON line: {"iters": .., "seconds": .., "residual": .., "reason": .., "ranks": ..}
 */
#include <petscksp.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    PetscErrorCode ierr;
    Mat A;
    Vec x, b;
    KSP ksp;
    MatNullSpace nsp;
    PetscInt its = 0, n32;
    PetscReal res = 0.0;
    KSPConvergedReason reason;
    PetscMPIInt size = 1;
    long long n = 0, nnz = 0;
    int *rp = NULL, *cl = NULL;
    double *vl = NULL, *bv = NULL, t0, t1;
    FILE *f;

    ierr = PetscInitialize(&argc, &argv, NULL, NULL); if (ierr) return ierr;
    ierr = MPI_Comm_size(PETSC_COMM_WORLD, &size); CHKERRQ(ierr);
    if (argc < 2 || size != 1) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "usage: petsc_ksp_driver <system.bin> [options], one MPI process");
    if (sizeof(PetscInt) != 4) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "built for 32-bit PetscInt (the bench's int32 CSR)");
    f = fopen(argv[1], "rb");
    if (!f) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "cannot open the system file");
    if (fread(&n, 8, 1, f) != 1 || fread(&nnz, 8, 1, f) != 1) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "short header");
    rp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    cl = (int *)malloc(sizeof(int) * (size_t)nnz);
    vl = (double *)malloc(sizeof(double) * (size_t)nnz);
    bv = (double *)malloc(sizeof(double) * (size_t)n);
    if (!rp || !cl || !vl || !bv) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "out of memory");
    if (fread(rp, sizeof(int), (size_t)(n + 1), f) != (size_t)(n + 1) || fread(cl, sizeof(int), (size_t)nnz, f) != (size_t)nnz ||
        fread(vl, sizeof(double), (size_t)nnz, f) != (size_t)nnz || fread(bv, sizeof(double), (size_t)n, f) != (size_t)n)
        SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "short system file");
    fclose(f);
    n32 = (PetscInt)n;
    ierr = MatCreateSeqAIJWithArrays(PETSC_COMM_SELF, n32, n32, rp, cl, vl, &A); CHKERRQ(ierr);
    ierr = VecCreateSeqWithArray(PETSC_COMM_SELF, 1, n32, bv, &b); CHKERRQ(ierr);
    ierr = VecDuplicate(b, &x); CHKERRQ(ierr);
    /* setNullSpace, "PETSc KSP" branch (navierstokes.cpp:401-408) */
    ierr = MatNullSpaceCreate(PETSC_COMM_SELF, PETSC_TRUE, 0, NULL, &nsp); CHKERRQ(ierr);
    ierr = MatSetNullSpace(A, nsp); CHKERRQ(ierr);
    ierr = MatSetNearNullSpace(A, nsp); CHKERRQ(ierr);
    ierr = MatNullSpaceDestroy(&nsp); CHKERRQ(ierr);
    /* LinSolverKSP::init + setMatrix (linsolverksp.cpp:48-91) */
    ierr = KSPCreate(PETSC_COMM_SELF, &ksp); CHKERRQ(ierr);
    ierr = KSPSetOptionsPrefix(ksp, "poisson_"); CHKERRQ(ierr);
    ierr = KSPSetInitialGuessNonzero(ksp, PETSC_FALSE); CHKERRQ(ierr);
    ierr = KSPSetOperators(ksp, A, A); CHKERRQ(ierr);
    ierr = KSPSetFromOptions(ksp); CHKERRQ(ierr);
    ierr = KSPSetUp(ksp); CHKERRQ(ierr);            /* the AMG set-up is not part of the timed solve (the GPU's is not either) */
    t0 = MPI_Wtime();
    ierr = KSPSolve(ksp, b, x); CHKERRQ(ierr);     /* LinSolverKSP::solve (linsolverksp.cpp:94-107) */
    t1 = MPI_Wtime();
    ierr = KSPGetIterationNumber(ksp, &its); CHKERRQ(ierr);
    ierr = KSPGetResidualNorm(ksp, &res); CHKERRQ(ierr);
    ierr = KSPGetConvergedReason(ksp, &reason); CHKERRQ(ierr);
    printf("{\"iters\": %d, \"seconds\": %.6f, \"residual\": %.6e, \"reason\": %d, \"ranks\": %d}\n", (int)its, t1 - t0, (double)res,
           (int)reason, (int)size);
    ierr = KSPDestroy(&ksp); CHKERRQ(ierr);
    ierr = VecDestroy(&x); CHKERRQ(ierr);
    ierr = VecDestroy(&b); CHKERRQ(ierr);
    ierr = MatDestroy(&A); CHKERRQ(ierr);
    free(rp); free(cl); free(vl); free(bv);
    ierr = PetscFinalize();
    return ierr;
}
