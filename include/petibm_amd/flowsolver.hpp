// flowsolver.hpp -- header-only C++ mirror of the reference's flow-solver classes over the C ABI (include/petibm_amd.h):
//
//   NavierStokesSolver       applications/navierstokes/navierstokes.h:47-  (init, advance, write-side accessors)
//   DecoupledIBPMSolver      applications/decoupledibpm/decoupledibpm.h    (+ bodies, forces)
//   RigidKinematicsSolver    applications/rigidkinematics/rigidkinematics.h (+ moveBodies with user kinematics)
//
// The reference builds these from a YAML::Node; yaml-cpp is not part of this image, so the mirror takes the same
// information as a plain struct (`FlowConfig`: the `mesh`, `flow` and `parameters` nodes of config.yaml, field for
// field).  Everything numerical happens behind pib_ns_* on the GPU; errors are the C ABI's PETSc-numbered codes.
#pragma once
#include <cmath>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../petibm_amd.h"

namespace petibm_amd
{
typedef int ErrorCode;

/** \brief One `subDomains` entry of a mesh direction (src/parser/parser.cpp:331-356). */
struct SubDomain {
    double end;
    int64_t cells;
    double stretchRatio;
};
/** \brief One direction of the `mesh` node. */
struct MeshAxis {
    double start;
    std::vector<SubDomain> subDomains;
};
enum BCType { DIRICHLET = 0, NEUMANN = 1, CONVECTIVE = 2, PERIODIC = 3 };  // src/misc/type.cpp str2bt
enum BCLoc { XMINUS = 0, XPLUS, YMINUS, YPLUS, ZMINUS, ZPLUS };
struct BoundaryCondition {
    BCType type = DIRICHLET;
    double value = 0.0;
};
/** \brief The part of config.yaml the flow solvers read. */
struct FlowConfig {
    std::vector<MeshAxis> mesh;                 // x, y[, z]
    double nu = 0.0;
    std::vector<double> initialVelocity;        // one constant per component
    BoundaryCondition bc[3][6];                 // [component u,v,w][BCLoc]
    double dt = 0.0;
    std::string velocitySolver, poissonSolver, forcesSolver;  // the TEXT of the solver .info files
    std::string delta = "ROMA_ET_AL_1999";      // parameters.delta (decoupledibpm.cpp:162)
    int BN = 1;                                 // parameters.BN: order of the approximate inverse (navierstokes.cpp:348)
    std::string convection = "ADAMS_BASHFORTH_2", diffusion = "CRANK_NICOLSON";  // parameters.convection / diffusion
};

/** \brief parseSubDomains + stretchGrid (src/parser/parser.cpp:298-356, include/petibm/misc.h:148-163). */
inline std::vector<double> cellWidths(const MeshAxis &a, double *end = nullptr)
{
    std::vector<double> w;
    double bg = a.start;
    for (const SubDomain &s : a.subDomains) {
        const double r = s.stretchRatio;
        if (std::abs(r - 1.0) <= 1e-12) {
            for (int64_t i = 0; i < s.cells; ++i) w.push_back((s.end - bg) / s.cells);
        } else {
            double d = (s.end - bg) * (r - 1.0) / (std::pow(r, (double)s.cells) - 1.0);
            for (int64_t i = 0; i < s.cells; ++i) {
                w.push_back(d);
                d = d * r;
            }
        }
        bg = s.end;
    }
    if (end) *end = bg;
    return w;
}

/** \brief readLagrangianPoints (src/io/io.cpp:23-118): the number of points, then one coordinate set per line. */
inline ErrorCode readLagrangianPoints(const std::string &file, int dim, std::vector<double> &coords, int64_t &nPts)
{
    std::ifstream in(file);
    if (!in.good()) return PIB_ERR_FILE_OPEN;
    std::string line;
    if (!std::getline(in, line)) return 66;
    {
        std::stringstream s(line);
        if (!(s >> nPts)) return 66;
    }
    coords.clear();
    while (std::getline(in, line)) {
        std::stringstream s(line);
        double v;
        int c = 0;
        while (s >> v) {
            coords.push_back(v);
            ++c;
        }
        if (c != 0 && c != dim) return 66;  // PETSC_ERR_FILE_READ
    }
    return ((int64_t)coords.size() == nPts * dim) ? 0 : 66;
}

/** \brief Mirror of NavierStokesSolver (applications/navierstokes/navierstokes.cpp:83-266). */
class NavierStokesSolver
{
public:
    NavierStokesSolver() = default;
    virtual ~NavierStokesSolver() { destroy(); }
    NavierStokesSolver(const NavierStokesSolver &) = delete;
    NavierStokesSolver &operator=(const NavierStokesSolver &) = delete;

    virtual ErrorCode destroy()
    {
        if (ns) pib_ns_destroy(ns);
        ns = nullptr;
        return 0;
    }

    /** init(world, node): mesh, boundary conditions, operators, vectors, linear solvers, initial condition. */
    virtual ErrorCode init(const FlowConfig &cfg, int device = -1)
    {
        destroy();
        dim = (int)cfg.mesh.size();
        if (dim != 2 && dim != 3) return PIB_ERR_ARG_OUTOFRANGE;
        dt = cfg.dt;
        int64_t n[3] = {1, 1, 1};
        double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
        for (int d = 0; d < dim; ++d) {
            w[d] = cellWidths(cfg.mesh[d], &hi[d]);
            n[d] = (int64_t)w[d].size();
            lo[d] = cfg.mesh[d].start;
        }
        int bt[18];
        double bv[18];
        for (int f = 0; f < 3; ++f)
            for (int l = 0; l < 6; ++l) {
                bt[6 * f + l] = (int)cfg.bc[f][l].type;
                bv[6 * f + l] = cfg.bc[f][l].value;
            }
        ErrorCode ierr = pib_ns_create(&ns, dim, n, w[0].data(), w[1].data(), dim == 3 ? w[2].data() : nullptr, lo, hi, bt, bv,
                                       cfg.dt, cfg.nu, cfg.velocitySolver.c_str(), cfg.poissonSolver.c_str(), device);
        if (ierr) return ierr;
        if (cfg.convection != "ADAMS_BASHFORTH_2" || cfg.diffusion != "CRANK_NICOLSON") {
            ierr = pib_ns_set_time_integration(ns, cfg.convection.c_str(), cfg.diffusion.c_str());
            if (ierr) return ierr;
        }
        if (cfg.BN != 1) {
            ierr = pib_ns_set_bn_order(ns, cfg.BN);
            if (ierr) return ierr;
        }
        ierr = pib_ns_sizes(ns, &UN, &pN);
        if (ierr) return ierr;
        // flow.initialVelocity: a constant per component (solutionsimple.cpp:122-226 with constant expressions)
        bool any = false;
        for (double v : cfg.initialVelocity) any = any || v != 0.0;
        if (any) {
            std::vector<double> U((size_t)UN, 0.0);
            int64_t off = 0;
            for (int f = 0; f < dim; ++f) {
                int64_t nf = 1;  // component f has one point fewer along f unless f is periodic (cartesianmesh.cpp:249-266)
                for (int d = 0; d < dim; ++d) nf *= n[d] - ((d == f && cfg.bc[f][2 * d].type != PERIODIC) ? 1 : 0);
                for (int64_t q = 0; q < nf; ++q) U[(size_t)(off + q)] = cfg.initialVelocity[(size_t)f];
                off += nf;
            }
            ierr = pib_ns_set_state(ns, U.data(), nullptr);
        }
        t = 0.0;
        ite = 0;
        return ierr;
    }

    /** advance(): one time step (navierstokes.cpp:240-266). */
    virtual ErrorCode advance()
    {
        t += dt;
        ite++;
        return pib_ns_advance(ns, 1);
    }

    /** The packed velocity and the pressure (what write() hands to HDF5, navierstokes.cpp:618-634). */
    ErrorCode getSolution(std::vector<double> &U, std::vector<double> &p)
    {
        U.resize((size_t)UN);
        p.resize((size_t)pN);
        return pib_ns_get_state(ns, U.data(), p.data(), nullptr, nullptr);
    }

    /** One line of iterations-<start>.txt (navierstokes.cpp:766-794). */
    ErrorCode writeLinSolversInfo(std::ostream &os)
    {
        int vi = 0, pi = 0;
        double vr = 0, pr = 0;
        const ErrorCode ierr = pib_ns_get_solver_info(ns, &vi, &vr, &pi, &pr);
        if (!ierr) os << ite << '\t' << vi << '\t' << vr << '\t' << pi << '\t' << pr << '\n';
        return ierr;
    }

    double t = 0.0, dt = 0.0;
    int ite = 0, dim = 0;
    int64_t UN = 0, pN = 0;

protected:
    pib_ns *ns = nullptr;
    std::vector<double> w[3];
};

/** \brief Mirror of DecoupledIBPMSolver (applications/decoupledibpm/decoupledibpm.cpp:54-131,437-465). */
class DecoupledIBPMSolver : public NavierStokesSolver
{
public:
    /** bodies: one coordinate array per body (point-major, dim values per point): src/body/bodypack.cpp. */
    virtual ErrorCode init(const FlowConfig &cfg, const std::vector<std::vector<double>> &bodies, int device = -1)
    {
        ErrorCode ierr = NavierStokesSolver::init(cfg, device);
        if (ierr) return ierr;
        nPts.clear();
        coords.clear();
        for (const auto &b : bodies) {
            nPts.push_back((int64_t)b.size() / dim);
            coords.insert(coords.end(), b.begin(), b.end());
        }
        ierr = pib_ns_set_bodies(ns, (int)bodies.size(), nPts.data(), coords.data(), cfg.delta.c_str(), cfg.forcesSolver.c_str());
        if (ierr) return ierr;
        int nb = 0;
        return pib_ns_num_forces(ns, &nf, &nb);
    }

    /** bodies->calculateAvgForces (one row of forces-<start>.txt, decoupledibpm.cpp:437-465): [body][direction] */
    ErrorCode getBodyForces(std::vector<double> &fAvg)
    {
        fAvg.assign(nPts.size() * (size_t)dim, 0.0);
        return pib_ns_get_forces(ns, nullptr, fAvg.data());
    }

    ErrorCode writeForcesASCII(std::ostream &os)
    {
        std::vector<double> f;
        const ErrorCode ierr = getBodyForces(f);
        if (ierr) return ierr;
        os << t << '\t';
        for (double v : f) os << v << '\t';
        os << '\n';
        return 0;
    }

    int64_t nf = 0;

protected:
    std::vector<int64_t> nPts;
    std::vector<double> coords;
};

/** \brief Mirror of IBPMSolver (applications/ibpm/ibpm.h:30-111): the coupled immersed-boundary projection method --
 *  pressure and Lagrangian forces solved as one unknown; same construction as the decoupled solver. */
class IBPMSolver : public DecoupledIBPMSolver
{
public:
    ErrorCode init(const FlowConfig &cfg, const std::vector<std::vector<double>> &bodies, int device = -1) override
    {
        ErrorCode ierr = DecoupledIBPMSolver::init(cfg, bodies, device);
        if (ierr) return ierr;
        return pib_ns_set_coupled(ns, 1);
    }
};

/** \brief Mirror of RigidKinematicsSolver (applications/rigidkinematics/rigidkinematics.cpp:68-160): the user supplies
 *  the kinematics by overriding setCoordinatesBodies / setVelocityBodies exactly as in the reference's API example
 *  (examples/api_examples/oscillatingcylinder2dRe100_GPU/oscillatingcylinder.cpp). */
class RigidKinematicsSolver : public DecoupledIBPMSolver
{
public:
    ErrorCode advance() override
    {
        ErrorCode ierr = moveBodies(t + dt);  // note: `t + dt`, t is updated in the Navier-Stokes step (:75-79)
        if (ierr) return ierr;
        return DecoupledIBPMSolver::advance();
    }

protected:
    virtual ErrorCode setCoordinatesBodies(const double &ti, std::vector<double> &xyz) = 0;
    virtual ErrorCode setVelocityBodies(const double &ti, std::vector<double> &UB) = 0;
    ErrorCode moveBodies(const double &ti)
    {
        UB.resize((size_t)nf);
        ErrorCode ierr = setCoordinatesBodies(ti, coords);
        if (ierr) return ierr;
        ierr = setVelocityBodies(ti, UB);
        if (ierr) return ierr;
        return pib_ns_move_bodies(ns, coords.data(), UB.data());
    }
    std::vector<double> UB;
};

}  // namespace petibm_amd
