// linsolver.hpp -- header-only C++ mirror of petibm::linsolver over the C ABI
// (include/petibm_amd.h).  Same names, argument meaning and error behaviour as
// the reference plugin interface:
//
//   petibm::linsolver::LinSolverBase          include/petibm/linsolver.h:59-147
//   petibm::linsolver::createLinSolver        src/linsolver/linsolver.cpp:57-91
//   petibm::linsolver::LinSolverAmgX          src/linsolver/linsolveramgx.cpp:20-126
//
// Two build modes:
//   * default (no PETSc): Mat/Vec are the plain structs below (CSR with local
//     rows + global columns; contiguous double arrays).  Used by the examples
//     and tests of this repository, which must build where PETSc is absent.
//   * -DPIB_WITH_PETSC: the class takes PETSc `Mat` / `Vec` exactly like the
//     reference (`setMatrix(const Mat&)`, `solve(Vec&, Vec&)`); the local
//     CSR comes from MatGetRowIJ on the matrix itself (MATSEQAIJ, np = 1) or
//     on MatMPIAIJGetLocalMat's result (MATMPIAIJ), the arrays from
//     VecGetArray (petsc_adapter.hpp).  This is the class PetIBM's factory
//     instantiates for `type: GPU`.  No PETSc exists in this
//     image: the branch is syntax-checked against a declarations-only stub
//     (tests/stubs/petsc, tests/test_boundary_headers.py), nothing more.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "../petibm_amd.h"

#ifdef PIB_WITH_PETSC
#include "petsc_adapter.hpp"
#endif

namespace petibm_amd
{
typedef int ErrorCode;  // PetscErrorCode-compatible: 0 success, PETSC_ERR_* otherwise
#ifdef PIB_WITH_PETSC
typedef PetscInt Int;    // getIters(PetscInt &), getResidual(PetscReal &): include/petibm/linsolver.h:122,131
typedef PetscReal Real;
#else
typedef int Int;
typedef double Real;
#endif

#ifndef PIB_WITH_PETSC
// minimal stand-ins for the two PETSc types on the boundary
struct Mat {
    int64_t n_local = 0, row0 = 0, n_global = 0;
    std::vector<int64_t> rowptr, col;  // local rows, GLOBAL columns (MatGetRowIJ layout)
    std::vector<double> val;
};
typedef std::vector<double> Vec;
#endif

namespace linsolver
{
/** \brief Mirror of petibm::linsolver::LinSolverBase (include/petibm/linsolver.h:59-147). */
class LinSolverBase
{
public:
    LinSolverBase() = default;
    LinSolverBase(const std::string &solverName, const std::string &file) : name(solverName), config(file) {}
    virtual ~LinSolverBase() = default;
    virtual ErrorCode destroy()
    {
        name = config = type = "";
        return 0;
    }
    ErrorCode printInfo() const
    {
        std::string info = std::string(80, '=') + "\nLinear Solver " + name + ":\n" + std::string(80, '=') + "\n";
        info += "\tType: " + type + "\n\n\tConfig file: " + config + "\n\n";
        // what actually runs, one "Runs:" line each: the effective solver and every departure from the file (pib_describe)
        const std::string runs = describeRuns();
        for (size_t b = 0; b < runs.size();) {
            size_t e = runs.find('\n', b);
            if (e == std::string::npos) e = runs.size();
            info += "\tRuns: " + runs.substr(b, e - b) + "\n";
            b = e + 1;
        }
        if (!runs.empty()) info += "\n";
        std::fputs(info.c_str(), stdout);
        return 0;
    }
    virtual std::string describeRuns() const { return std::string(); }
    ErrorCode getType(std::string &_type) const
    {
        _type = type;
        return 0;
    }
#ifdef PIB_WITH_PETSC
    virtual ErrorCode setMatrix(const ::Mat &A) = 0;
    virtual ErrorCode solve(::Vec &x, ::Vec &b) = 0;
#else
    virtual ErrorCode setMatrix(const Mat &A) = 0;
    virtual ErrorCode solve(Vec &x, Vec &b) = 0;
#endif
    virtual ErrorCode getIters(Int &iters) = 0;
    virtual ErrorCode getResidual(Real &res) = 0;

protected:
    std::string name, config, type;
    virtual ErrorCode init() = 0;
};

/** \brief Takes the place of LinSolverAmgX (src/linsolver/linsolveramgx.cpp): HIP/gfx950 backend. */
class LinSolverHIP : public LinSolverBase
{
public:
    /** rank/nranks/uid: one rank per GPU; uid from pib_comm_unique_id broadcast by the caller
     *  (MPI_Bcast over PETSC_COMM_WORLD in PetIBM). */
    LinSolverHIP(const std::string &solverName, const std::string &file, int rank = 0, int nranks = 1,
                 const void *uid = nullptr, int device = -1)
        : LinSolverBase(solverName, file), rank_(rank), nranks_(nranks), uid_(uid), device_(device)
    {
        err_ = init();
    }
    ~LinSolverHIP() override
    {
#ifdef PIB_WITH_PETSC
        if (petsc::finalized()) return;  // no-op after PetscFinalize, like ~LinSolverAmgX (linsolveramgx.cpp:28-38)
#endif
        if (h_) pib_destroy(h_);
    }
    ErrorCode constructionError() const { return err_; }
    ErrorCode destroy() override
    {
        if (h_) pib_destroy(h_);
        h_ = nullptr;
        return LinSolverBase::destroy();
    }
#ifdef PIB_WITH_PETSC
    // SeqAIJ (np = 1) or MPIAIJ: local rows, global columns -- what AmgXSolver::setA extracts (petsc_adapter.hpp)
    ErrorCode setMatrix(const ::Mat &A) override { return petsc::setMatrix(h_, A); }
    ErrorCode solve(::Vec &x, ::Vec &b) override { return petsc::solve(h_, x, b); }
#else
    ErrorCode setMatrix(const Mat &A) override
    {
        return pib_set_csr(h_, A.n_local, A.row0, A.n_global, A.rowptr.data(), A.col.data(), A.val.data());
    }
    ErrorCode solve(Vec &x, Vec &b) override { return pib_solve(h_, x.data(), b.data()); }
#endif
    ErrorCode getIters(Int &iters) override
    {
        int it = 0;
        const int e = pib_get_iters(h_, &it);
        iters = (Int)it;
        return e;
    }
    ErrorCode getResidual(Real &res) override
    {
        double r = 0.0;
        const int e = pib_get_residual(h_, &r);
        res = (Real)r;
        return e;
    }
    /** mesh structure of the Poisson operator (enables the stencil twin + multigrid) */
    ErrorCode setGridHint(int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                          const double *gx, const double *gy, const double *gz, int nullspace)
    {
        return pib_set_grid_hint(h_, dim, n, wx, wy, wz, gx, gy, gz, nullspace);
    }
    /** periodic directions of the mesh (mesh->periodic[0][d]); before setGridHint / the on-device assembly */
    ErrorCode setPeriodic(const int periodic[3]) { return pib_set_periodic(h_, periodic); }
    pib_solver *handle() { return h_; }
    std::string describeRuns() const override
    {
        char buf[4096];
        if (h_ == nullptr || pib_describe(h_, buf, (int)sizeof buf) != 0) return std::string();
        return std::string(buf);
    }

protected:
    ErrorCode init() override
    {
        int e = pib_create(&h_, name.c_str(), config.c_str(), rank_, nranks_, uid_, device_);
        if (e) return e;
        char buf[64];
        e = pib_get_type(h_, buf, sizeof buf);
        type = buf;  // "NVIDIA AmgX" for an AmgX-style file: unchanged applications pick the pinned-pressure path
        return e;
    }
    pib_solver *h_ = nullptr;
    int rank_, nranks_;
    const void *uid_;
    int device_;
    ErrorCode err_ = 0;
};
}  // namespace linsolver

namespace type
{
typedef std::shared_ptr<linsolver::LinSolverBase> LinSolver;
}

namespace linsolver
{
/** \brief Mirror of petibm::linsolver::createLinSolver (src/linsolver/linsolver.cpp:57-91) with the YAML node
 *  reduced to the three strings it reads: parameters.<name>Solver.type / .config and `directory`. */
inline ErrorCode createLinSolver(const std::string &solverName, const std::string &typeStr, std::string config,
                                 const std::string &directory, type::LinSolver &solver, int rank = 0, int nranks = 1,
                                 const void *uid = nullptr)
{
    if (!config.empty() && config[0] != '/' && config != "None") config = directory + "/" + config;
    if (config.empty()) config = "None";
    if (typeStr == "GPU") {
        auto p = std::make_shared<LinSolverHIP>(solverName, config, rank, nranks, uid);
        if (p->constructionError()) return p->constructionError();
        solver = p;
        return 0;
    }
    if (typeStr == "CPU") {
        std::fprintf(stderr, "PETSc KSP solver (type: CPU) is PetIBM's own LinSolverKSP; not part of this backend.\n");
        return PIB_ERR_ARG_WRONG;
    }
    std::fprintf(stderr, "Unrecognized value \"%s\" of the type of the linear solver \"%s\"\n", typeStr.c_str(),
                 solverName.c_str());
    return PIB_ERR_ARG_WRONG;
}
}  // namespace linsolver
}  // namespace petibm_amd
