/* declarations-only PETSc stub (3.16 signatures): see README.md (syntax check of the boundary headers, nothing else) */
#ifndef PIB_STUB_PETSCSYS_H
#define PIB_STUB_PETSCSYS_H
#include <mpi.h>
typedef int PetscErrorCode;
typedef int PetscInt;        /* default configuration: 32-bit indices */
typedef int PetscMPIInt;
typedef double PetscReal;
typedef double PetscScalar;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef struct _p_PetscObject *PetscObject;
typedef enum { PETSC_ERROR_INITIAL = 0, PETSC_ERROR_REPEAT = 1, PETSC_ERROR_IN_CXX = 2 } PetscErrorType;
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_LIB 76
#define PETSC_ERR_CONV_FAILED 82
#ifdef __cplusplus
extern "C" {
#endif
extern MPI_Comm PETSC_COMM_WORLD;
extern MPI_Comm PETSC_COMM_SELF;
PetscErrorCode PetscError(MPI_Comm comm, int line, const char *func, const char *file, PetscErrorCode n, PetscErrorType p,
                          const char *mess, ...);
PetscErrorCode PetscFinalized(PetscBool *isFinalized);
PetscErrorCode PetscPrintf(MPI_Comm comm, const char format[], ...);
PetscErrorCode PetscObjectTypeCompare(PetscObject obj, const char type_name[], PetscBool *same);
MPI_Comm PetscObjectComm(PetscObject obj);
#ifdef __cplusplus
}
#endif
#define SETERRQ(comm, ierr, s) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, PETSC_ERROR_INITIAL, s)
#define SETERRQ1(comm, ierr, s, a1) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, PETSC_ERROR_INITIAL, s, a1)
#define SETERRQ2(comm, ierr, s, a1, a2) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, PETSC_ERROR_INITIAL, s, a1, a2)
#define SETERRQ3(comm, ierr, s, a1, a2, a3) return PetscError(comm, __LINE__, __func__, __FILE__, ierr, PETSC_ERROR_INITIAL, s, a1, a2, a3)
#define CHKERRV(ierr)                                                                                        \
    do {                                                                                                     \
        if (ierr) {                                                                                          \
            PetscError(PETSC_COMM_SELF, __LINE__, __func__, __FILE__, ierr, PETSC_ERROR_REPEAT, " ");       \
            return;                                                                                          \
        }                                                                                                    \
    } while (0)
#define CHKERRQ(ierr)                                                                                              \
    do {                                                                                                           \
        if (ierr) return PetscError(PETSC_COMM_SELF, __LINE__, __func__, __FILE__, ierr, PETSC_ERROR_REPEAT, " "); \
    } while (0)
#define PetscFunctionBeginUser
#define PetscFunctionReturn(a) return (a)
#endif
