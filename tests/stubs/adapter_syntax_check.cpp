// SYNTAX CHECK ONLY (g++ -fsyntax-only against tests/stubs/petsc): a translation unit of this repository's own that
// drives the two PETSc-facing adapters with the argument types PetIBM passes at the boundary --
//   const Mat &, Vec &, PetscInt &, PetscReal &          include/petibm/linsolver.h:104-131
//   MPI_Comm, "dDDI", config path                        src/linsolver/linsolveramgx.cpp:69
// It is never linked or run; it pins nothing except that the headers parse and type-check (tests/test_boundary_headers.py).
#define PIB_WITH_PETSC
#include <petibm_amd/AmgXSolver.hpp>
#include <petibm_amd/linsolver.hpp>

// plug point 2: the six AmgXSolver members, called the way a LinSolverBase subclass wrapping them would
struct ViaAmgXWrapperSurface {
    AmgXSolver amgx;
    std::string config = "poisson_solver.info";
    PetscErrorCode init() { return amgx.initialize(PETSC_COMM_WORLD, "dDDI", config); }
    PetscErrorCode setMatrix(const Mat &A) { return amgx.setA(A); }
    PetscErrorCode solve(Vec &x, Vec &b) { return amgx.solve(x, b); }
    PetscErrorCode getIters(PetscInt &iters) { return amgx.getIters(iters); }
    PetscErrorCode getResidual(PetscReal &res)
    {
        PetscErrorCode ierr;
        PetscInt iter;
        ierr = amgx.getIters(iter); CHKERRQ(ierr);
        ierr = amgx.getResidual(iter, res); CHKERRQ(ierr);
        return 0;
    }
    PetscErrorCode destroy() { return amgx.finalize(); }
};

// plug point 1: the LinSolverBase subclass through the factory, PETSc types on the interface
PetscErrorCode via_linsolver_surface(const Mat &A, Vec &x, Vec &b)
{
    PetscErrorCode ierr;
    PetscMPIInt rank, size;
    char uid[PIB_UID_BYTES];
    ierr = petibm_amd::petsc::broadcastUniqueId(PETSC_COMM_WORLD, &rank, &size, uid); CHKERRQ(ierr);
    petibm_amd::type::LinSolver solver;
    ierr = petibm_amd::linsolver::createLinSolver("poisson", "GPU", "config/poisson_solver.info", "/case", solver, rank, size,
                                                  size > 1 ? uid : nullptr); CHKERRQ(ierr);
    ierr = solver->setMatrix(A); CHKERRQ(ierr);
    ierr = solver->solve(x, b); CHKERRQ(ierr);
    PetscInt its;
    PetscReal res;
    ierr = solver->getIters(its); CHKERRQ(ierr);
    ierr = solver->getResidual(res); CHKERRQ(ierr);
    std::string type;
    ierr = solver->getType(type); CHKERRQ(ierr);
    ierr = solver->printInfo(); CHKERRQ(ierr);
    ierr = solver->destroy(); CHKERRQ(ierr);
    return 0;
}
