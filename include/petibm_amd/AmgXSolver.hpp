// AmgXSolver.hpp -- a HIP/gfx950-backed class with the interface of AmgXWrapper's `AmgXSolver`, reduced to the six
// members PetIBM uses (plug point 2): with this header on the include path in place of AmgXWrapper's and
// HAVE_AMGX defined, src/linsolver/linsolveramgx.cpp and include/petibm/linsolveramgx.h compile UNCHANGED and
// `type: GPU` solvers run on libpetibm_amd.so:
//
//   amgx.initialize(PETSC_COMM_WORLD, "dDDI", config)     src/linsolver/linsolveramgx.cpp:69
//   amgx.setA(A)                                          :84
//   amgx.solve(x, b)                                      :96
//   amgx.getIters(iters)                                  :108,122
//   amgx.getResidual(iter, res)                           :123
//   amgx.finalize()                                       :37,47
//
// All return PetscErrorCode.  The solver file is the AmgX key=value text the reference's *_GPU examples ship
// (include/petibm_amd.h: pib_create); LinSolverAmgX::getType keeps answering "NVIDIA AmgX", so the applications pick
// the pinned-pressure convention as they do today (applications/navierstokes/navierstokes.cpp:414-420,553-558).
// An `AMG` preconditioner entry is served by the geometric multigrid: the mesh structure it needs is RECOVERED FROM THE
// MATRIX (pib_set_csr detects the 5/7-point DBNG of a tensor-product mesh and verifies the recovered operator against
// the CSR on the device), so nothing beyond setA is asked of the application.  Several ranks: setA takes the rows in
// whatever partition PetIBM's DMDAs produced -- PETSC_DECIDE cuts boxes from 4 ranks up (src/mesh/cartesianmesh.cpp:97,
// 503-519; no command-line switch changes that: PetIBM never calls DMSetFromOptions) -- and the backend moves the
// Poisson rows to natural z-slabs for its multigrid itself (csrc/partition.cpp, INTEGRATION.md "Several ranks").
//
// Needs PETSc + MPI headers: syntax-checked only in this repository (tests/stubs/petsc, tests/test_boundary_headers.py).
#pragma once
#include <string>

#include "petsc_adapter.hpp"

class AmgXSolver
{
public:
    AmgXSolver() = default;
    AmgXSolver(const MPI_Comm &comm, const std::string &modeStr, const std::string &cfgFile) { initialize(comm, modeStr, cfgFile); }
    ~AmgXSolver()
    {
        if (!petibm_amd::petsc::finalized()) finalize();
    }
    AmgXSolver(const AmgXSolver &) = delete;
    AmgXSolver &operator=(const AmgXSolver &) = delete;

    /** mode "dDDI" = device, fp64 matrix, fp64 vectors, 32-bit indices: the only mode PetIBM asks for and the only one
     *  this backend computes in; any other string is PETSC_ERR_SUP. */
    PetscErrorCode initialize(const MPI_Comm &comm, const std::string &modeStr, const std::string &cfgFile)
    {
        PetscErrorCode ierr;
        if (h_ != nullptr) SETERRQ(comm, PETSC_ERR_ORDER, "AmgXSolver (petibm_amd): already initialized");
        if (modeStr != "dDDI") SETERRQ1(comm, PETSC_ERR_SUP, "AmgXSolver (petibm_amd): mode %s is not provided (dDDI only)", modeStr.c_str());
        PetscMPIInt rank = 0, size = 1;
        char uid[PIB_UID_BYTES];
        int device = -1;
        ierr = petibm_amd::petsc::broadcastUniqueId(comm, &rank, &size, uid); CHKERRQ(ierr);
        ierr = petibm_amd::petsc::localDevice(comm, &device); CHKERRQ(ierr);
        // the solver's name is only the option prefix of a PETSc-style file; an AmgX-style file does not use it
        const int e = pib_create(&h_, "amgx", cfgFile.c_str(), (int)rank, (int)size, size > 1 ? uid : nullptr, device);
        if (e) SETERRQ1(comm, e, "%s", pib_last_error());
        comm_ = comm;
        return 0;
    }
    PetscErrorCode setA(const Mat &A)
    {
        if (h_ == nullptr) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ORDER, "AmgXSolver (petibm_amd): setA before initialize");
        return petibm_amd::petsc::setMatrix(h_, A);
    }
    /** p: in = initial guess (AmgXWrapper hands x to AmgX as the guess), out = solution; b read-only */
    PetscErrorCode solve(Vec &p, Vec &b)
    {
        if (h_ == nullptr) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ORDER, "AmgXSolver (petibm_amd): solve before initialize");
        return petibm_amd::petsc::solve(h_, p, b);
    }
    PetscErrorCode getIters(int &iter)
    {
        const int e = pib_get_iters(h_, &iter);
        if (e) SETERRQ1(PETSC_COMM_SELF, e, "%s", pib_last_error());
        return 0;
    }
    /** entry `iter` of the residual history of the last solve (L2 norm; CHANGELOG.md:23 of the reference) */
    PetscErrorCode getResidual(const int &iter, double &res)
    {
        const int e = pib_get_residual_at(h_, iter, &res);
        if (e) SETERRQ1(PETSC_COMM_SELF, e, "%s", pib_last_error());
        return 0;
    }
    PetscErrorCode finalize()
    {
        if (h_ != nullptr) {
            const int e = pib_destroy(h_);
            h_ = nullptr;
            if (e) SETERRQ1(PETSC_COMM_SELF, e, "%s", pib_last_error());
        }
        return 0;
    }

private:
    pib_solver *h_ = nullptr;
    MPI_Comm comm_ = MPI_COMM_NULL;
};
