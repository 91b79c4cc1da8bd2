// gmg.hip -- K2/K5/K6/K7 structured (matrix-free) operator and geometric multigrid.
#include "pib_internal.hpp"

namespace pib {

void gmg_release(pib_solver *s)
{
    for (auto &L : s->levels) {
        for (int d = 0; d < 3; ++d) {
            if (L.w[d]) (void)hipFree(L.w[d]);
            if (L.g[d]) (void)hipFree(L.g[d]);
        }
        if (L.dinv) (void)hipFree(L.dinv);
        if (L.x) (void)hipFree(L.x);
        if (L.b) (void)hipFree(L.b);
        if (L.r) (void)hipFree(L.r);
    }
    s->levels.clear();
    s->has_grid = false;
}

int grid_register(pib_solver *s, int dim, const int64_t n[3], const double *const w[3], const double *const g[3],
                  int nullspace)
{
    (void)dim; (void)n; (void)w; (void)g;
    s->nullspace = nullspace;
    s->has_grid = false;
    return 0;
}

int gmg_setup(pib_solver *) { return 0; }

int gmg_apply(pib_solver *s, const double *, double *, hipStream_t)
{
    return fail(PIB_ERR_SUP, "solver %s: multigrid preconditioner not available", s->name.c_str());
}

int solve_bicgstab(pib_solver *s, double *, const double *)
{
    return fail(PIB_ERR_SUP, "solver %s: BiCGStab not available yet", s->name.c_str());
}

}  // namespace pib
