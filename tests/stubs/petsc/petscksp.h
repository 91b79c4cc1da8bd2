/* declarations-only PETSc stub (3.16 signatures): see README.md -- what tools/petsc_ksp_driver.c uses beyond petscmat.h */
#ifndef PIB_STUB_PETSCKSP_H
#define PIB_STUB_PETSCKSP_H
#include <petscmat.h>
typedef struct _p_KSP *KSP;
typedef struct _p_MatNullSpace *MatNullSpace;
typedef enum { KSP_CONVERGED_RTOL = 2, KSP_CONVERGED_ITERATING = 0, KSP_DIVERGED_ITS = -3 } KSPConvergedReason;
#ifdef __cplusplus
extern "C" {
#endif
PetscErrorCode PetscInitialize(int *argc, char ***args, const char file[], const char help[]);
PetscErrorCode PetscFinalize(void);
double MPI_Wtime(void);
PetscErrorCode MatCreateSeqAIJWithArrays(MPI_Comm comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat);
PetscErrorCode VecCreateSeqWithArray(MPI_Comm comm, PetscInt bs, PetscInt n, const PetscScalar array[], Vec *V);
PetscErrorCode VecDuplicate(Vec v, Vec *newv);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode MatNullSpaceCreate(MPI_Comm comm, PetscBool has_cnst, PetscInt n, const Vec vecs[], MatNullSpace *SP);
PetscErrorCode MatSetNullSpace(Mat mat, MatNullSpace nullsp);
PetscErrorCode MatSetNearNullSpace(Mat mat, MatNullSpace nullsp);
PetscErrorCode MatNullSpaceDestroy(MatNullSpace *sp);
PetscErrorCode KSPCreate(MPI_Comm comm, KSP *ksp);
PetscErrorCode KSPSetOptionsPrefix(KSP ksp, const char prefix[]);
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg);
PetscErrorCode KSPSetOperators(KSP ksp, Mat Amat, Mat Pmat);
PetscErrorCode KSPSetFromOptions(KSP ksp);
PetscErrorCode KSPSetUp(KSP ksp);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm);
PetscErrorCode KSPGetConvergedReason(KSP ksp, KSPConvergedReason *reason);
PetscErrorCode KSPDestroy(KSP *ksp);
#ifdef __cplusplus
}
#endif
#endif
