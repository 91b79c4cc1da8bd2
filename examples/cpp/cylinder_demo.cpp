// cylinder_demo.cpp -- the reference's decoupled-IBPM application loop (applications/decoupledibpm/main.cpp:45-90: init,
// then `advance(); write();` per step) from C++ over the C ABI, no PETSc / yaml-cpp / Python: impulsively started cylinder
// at Re = 40 (the parameters of examples/decoupledibpm/cylinder2dRe40_GPU), 300 steps, prints the drag coefficient.
//   g++ -std=c++14 -I include examples/cpp/cylinder_demo.cpp -L petibm_amd/lib -lpetibm_amd -Wl,-rpath,$PWD/petibm_amd/lib
#include <cstdio>
#include <iostream>

#include "petibm_amd/flowsolver.hpp"

using namespace petibm_amd;

int main(int argc, char **argv)
{
    const int nt = argc > 1 ? std::atoi(argv[1]) : 300;
    FlowConfig cfg;
    const MeshAxis axis{-15.0, {{-0.6, 69, 0.952380952}, {0.6, 48, 1.0}, {15.0, 69, 1.05}}};
    cfg.mesh = {axis, axis};
    cfg.nu = 0.025;
    cfg.dt = 0.01;
    cfg.initialVelocity = {1.0, 0.0};
    for (int l = 0; l < 4; ++l) {
        cfg.bc[0][l] = {l == XPLUS ? CONVECTIVE : DIRICHLET, 1.0};
        cfg.bc[1][l] = {l == XPLUS ? CONVECTIVE : DIRICHLET, l == XPLUS ? 1.0 : 0.0};
    }
    cfg.velocitySolver = "-velocity_ksp_type bcgs\n-velocity_ksp_atol 1.0E-06\n-velocity_ksp_rtol 0.0\n"
                         "-velocity_ksp_max_it 1000\n-velocity_pc_type jacobi\n";
    cfg.poissonSolver = "config_version=2\nsolver(solv)=PCG\nsolv:max_iters=1000\nsolv:monitor_residual=1\n"
                        "solv:convergence=ABSOLUTE\nsolv:tolerance=1.0E-06\nsolv:norm=L2\nsolv:preconditioner(prec)=AMG\n"
                        "prec:cycle=V\nprec:presweeps=1\nprec:postsweeps=1\nprec:smoother(smooth)=BLOCK_JACOBI\n"
                        "smooth:relaxation_factor=0.9\n";
    cfg.forcesSolver = "-forces_ksp_type preonly\n-forces_pc_type lu\n";
    std::vector<double> circle;
    const int npts = 126;
    for (int k = 0; k < npts; ++k) {
        const double a = 2.0 * M_PI * k / npts;
        circle.push_back(0.5 * std::cos(a));
        circle.push_back(0.5 * std::sin(a));
    }
    DecoupledIBPMSolver solver;
    ErrorCode ierr = solver.init(cfg, {circle});
    if (ierr) { std::printf("init failed: %d: %s\n", ierr, pib_last_error()); return 1; }
    std::printf("UN = %lld, pN = %lld, force unknowns = %lld\n", (long long)solver.UN, (long long)solver.pN, (long long)solver.nf);
    std::vector<double> f;
    for (int it = 0; it < nt; ++it) {
        ierr = solver.advance();
        if (ierr) { std::printf("advance failed: %d: %s\n", ierr, pib_last_error()); return 1; }
    }
    solver.getBodyForces(f);
    std::printf("t = %.2f  cd = %.6f  cl = %.3e\n", solver.t, 2.0 * f[0], 2.0 * f[1]);
    solver.writeLinSolversInfo(std::cout);
    return 0;
}
