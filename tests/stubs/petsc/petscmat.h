/* declarations-only PETSc stub (3.16 signatures): see README.md */
#ifndef PIB_STUB_PETSCMAT_H
#define PIB_STUB_PETSCMAT_H
#include <petscvec.h>
typedef struct _p_Mat *Mat;
typedef const char *MatType;
typedef struct { PetscInt k, j, i, c; } MatStencil;
#define MATSEQAIJ "seqaij"
#define MATMPIAIJ "mpiaij"
typedef enum { MAT_INITIAL_MATRIX, MAT_REUSE_MATRIX, MAT_IGNORE_MATRIX, MAT_INPLACE_MATRIX } MatReuse;
#ifdef __cplusplus
extern "C" {
#endif
PetscErrorCode MatGetOwnershipRange(Mat mat, PetscInt *m, PetscInt *n);
PetscErrorCode MatGetSize(Mat mat, PetscInt *m, PetscInt *n);
PetscErrorCode MatMPIAIJGetLocalMat(Mat A, MatReuse scall, Mat *A_loc);
PetscErrorCode MatGetRowIJ(Mat mat, PetscInt shift, PetscBool symmetric, PetscBool inodecompressed, PetscInt *n,
                           const PetscInt *ia[], const PetscInt *ja[], PetscBool *done);
PetscErrorCode MatRestoreRowIJ(Mat mat, PetscInt shift, PetscBool symmetric, PetscBool inodecompressed, PetscInt *n,
                               const PetscInt *ia[], const PetscInt *ja[], PetscBool *done);
PetscErrorCode MatSeqAIJGetArray(Mat A, PetscScalar **array);
PetscErrorCode MatSeqAIJRestoreArray(Mat A, PetscScalar **array);
PetscErrorCode MatDestroy(Mat *A);
#ifdef __cplusplus
}
#endif
#endif
