// poisson_demo.cpp -- the reference's API example (examples/api_examples/liddrivencavity2d/main.cpp:245-256,
// 308,324: createLinSolver -> setMatrix -> solve -> getIters/getResidual) against the HIP backend, from C++,
// with no PETSc and no Python: builds the 2-D 5-point DBNG of a 64x64 cavity on the host, solves
// DBNG x = b with multigrid-PCG on the GPU and checks ||b - A x|| on the host.
//   g++ -std=c++14 -I include examples/cpp/poisson_demo.cpp -L petibm_amd/lib -lpetibm_amd -Wl,-rpath,$PWD/petibm_amd/lib
#include <cmath>
#include <cstdio>
#include <fstream>

#include "petibm_amd/linsolver.hpp"

using namespace petibm_amd;

int main()
{
    const int64_t nx = 64, ny = 48;
    const double dt = 0.01;
    std::vector<double> wx(nx, 1.0 / nx), wy(ny, 1.0 / ny), gx(nx - 1), gy(ny - 1);
    for (int64_t i = 0; i + 1 < nx; ++i) gx[i] = dt * (1.0 / (0.5 * (wx[i + 1] + wx[i])));
    for (int64_t j = 0; j + 1 < ny; ++j) gy[j] = dt * (1.0 / (0.5 * (wy[j + 1] + wy[j])));
    Mat A;
    A.n_local = A.n_global = nx * ny;
    A.rowptr.push_back(0);
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            const int64_t r = i + nx * j;
            double d = 0.0;
            const double oym = j > 0 ? wx[i] * gy[j - 1] : 0, oxm = i > 0 ? wy[j] * gx[i - 1] : 0;
            const double oxp = i < nx - 1 ? wy[j] * gx[i] : 0, oyp = j < ny - 1 ? wx[i] * gy[j] : 0;
            d = -(oxm + oxp + oym + oyp);
            if (j > 0) { A.col.push_back(r - nx); A.val.push_back(oym); }
            if (i > 0) { A.col.push_back(r - 1); A.val.push_back(oxm); }
            A.col.push_back(r); A.val.push_back(d);
            if (i < nx - 1) { A.col.push_back(r + 1); A.val.push_back(oxp); }
            if (j < ny - 1) { A.col.push_back(r + nx); A.val.push_back(oyp); }
            A.rowptr.push_back((int64_t)A.col.size());
        }
    // config in the AmgX syntax of the reference's GPU examples
    const char *cfg = "poisson_demo_solver.info";
    {
        std::ofstream f(cfg);
        f << "config_version=2\nsolver(solv)=PCG\nsolv:max_iters=100\nsolv:monitor_residual=1\n"
             "solv:convergence=RELATIVE_INI\nsolv:tolerance=1e-10\nsolv:norm=L2\nsolv:store_res_history=1\n"
             "solv:preconditioner(prec)=AMG\nprec:cycle=V\nprec:presweeps=1\nprec:postsweeps=1\n"
             "prec:smoother(smooth)=BLOCK_JACOBI\nsmooth:relaxation_factor=0.9\npib_initial_guess_nonzero=0\n";
    }
    type::LinSolver solver;
    int ierr = linsolver::createLinSolver("poisson", "GPU", cfg, ".", solver);
    if (ierr) { std::printf("createLinSolver failed: %d: %s\n", ierr, pib_last_error()); return 1; }
    std::remove(cfg);
    solver->printInfo();
    ierr = solver->setMatrix(A);
    if (ierr) { std::printf("setMatrix failed: %d: %s\n", ierr, pib_last_error()); return 1; }
    const int64_t n[3] = {nx, ny, 1};
    ierr = static_cast<linsolver::LinSolverHIP *>(solver.get())
               ->setGridHint(2, n, wx.data(), wy.data(), nullptr, gx.data(), gy.data(), nullptr, PIB_NULLSPACE_CONSTANT);
    if (ierr) { std::printf("setGridHint failed: %d: %s\n", ierr, pib_last_error()); return 1; }
    // b = A x*, x* = cos(pi x) cos(pi y) (zero mean)
    Vec xs(nx * ny), b(nx * ny, 0.0), x(nx * ny, 0.0);
    const double pi = std::acos(-1.0);
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) xs[i + nx * j] = std::cos(pi * (i + 0.5) / nx) * std::cos(pi * (j + 0.5) / ny);
    auto matmult = [&](const Vec &in, Vec &out) {
        for (int64_t r = 0; r < nx * ny; ++r) {
            double s = 0;
            for (int64_t p = A.rowptr[r]; p < A.rowptr[r + 1]; ++p) s += A.val[p] * in[A.col[p]];
            out[r] = s;
        }
    };
    matmult(xs, b);
    ierr = solver->solve(x, b);
    if (ierr) { std::printf("solve failed: %d: %s\n", ierr, pib_last_error()); return 1; }
    int its;
    double res;
    solver->getIters(its);
    solver->getResidual(res);
    Vec ax(nx * ny);
    matmult(x, ax);
    double rn = 0, bn = 0;
    for (int64_t r = 0; r < nx * ny; ++r) { rn += (b[r] - ax[r]) * (b[r] - ax[r]); bn += b[r] * b[r]; }
    std::printf("iterations %d, solver residual %.3e, recomputed ||b-Ax||/||b|| %.3e\n", its, res, std::sqrt(rn / bn));
    solver->destroy();
    return (std::sqrt(rn / bn) < 2e-10 && its < 40) ? 0 : 2;
}
