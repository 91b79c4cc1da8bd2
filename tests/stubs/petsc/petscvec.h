/* declarations-only PETSc stub (3.16 signatures): see README.md */
#ifndef PIB_STUB_PETSCVEC_H
#define PIB_STUB_PETSCVEC_H
#include <petscsys.h>
typedef struct _p_Vec *Vec;
#ifdef __cplusplus
extern "C" {
#endif
PetscErrorCode VecGetArray(Vec x, PetscScalar **a);
PetscErrorCode VecRestoreArray(Vec x, PetscScalar **a);
PetscErrorCode VecGetArrayRead(Vec x, const PetscScalar **a);
PetscErrorCode VecRestoreArrayRead(Vec x, const PetscScalar **a);
#ifdef __cplusplus
}
#endif
#endif
