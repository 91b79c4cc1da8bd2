/*
 * gmg.c -- CPU restatement of the build's geometric-multigrid preconditioner.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * The reference preconditions its Poisson solve with third-party algebraic
 * multigrid (PCGAMG / hypre BoomerAMG through PETSc, or AmgX CLASSICAL/
 * AGGREGATION AMG: examples/navierstokes/liddrivencavity2dRe100/config/
 * poisson_solver.info:6-9, examples/.../liddrivencavity2dRe1000_GPU/config/
 * poisson_solver.info:20-42) -- none of which exists in this image.  The build
 * replaces it with a geometric V-cycle on the stretched Cartesian mesh
 * (BASELINE.json north_star).  A different preconditioner gives different
 * iterates, so this file is the oracle of the BUILD's V-cycle (kernel parity),
 * while parity with the reference is argued on the solver-independent
 * residual contract ||b - A x|| <= tol.
 *
 * Algorithm (identical, operation for operation, to petibm_amd/csrc/gmg.hip):
 *   - grid (nx, ny, nz) in natural ordering, 2-D grids are stored as
 *     (nx, 1, ny) so the slab axis is always the last one;
 *   - level operator: rediscretised finite-volume 7-point operator
 *       (A x)_c = sum_faces coef_f (x_nb - x_c),  coef = area_perp * g_d[s]   (evaluated through the volume-scaled rows:
 *       see face_coefs below),
 *       g_d[s] = dt / (0.5 (w_d[s] + w_d[s+1]))      (the DBNG of
 *       applications/navierstokes/navierstokes.cpp:349-356 for BN order 1);
 *   - selective coarsening per direction: walking the cells of a direction, two neighbours are merged only if
 *     their combined width is <= 1.5 * hmin * 2^(level+1) (hmin = smallest fine cell); cells that are already
 *     larger stay alone, so the stretched far field catches up with the refined region and the level grids
 *     become more and more uniform (on a uniform mesh this is plain pairing); coarse width = sum of children;
 *   - transfer: cell-centred (tri-)linear prolongation with width-based weights: a child at distance b/2 from
 *     its parent's centre (b = its sibling's width) takes t = b / (W_parent + W_neighbour) from the neighbouring
 *     coarse cell on its side and 1 - t from its parent (3/4, 1/4 on a uniform mesh, exactly); a lone child and
 *     a child at a wall take the parent's value; restriction = transpose;
 *   - smoother: damped Jacobi, omega = relaxation_factor; pre-smoothing starts
 *     from a zero guess; V(nu1, nu2);
 *   - coarsest level (<= 2 cells per direction): `coarsest_sweeps` Jacobi sweeps;
 *   - null space: the operator is singular (constants).  CONSTANT mode leaves
 *     z un-projected (the caller subtracts the mean lazily); PINNED mode maps
 *     r~ -> r' with r'[0] = -sum_{i>0} r~_i, and the caller shifts by z[0].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
#define MAXLEV 40

typedef struct {
    i64 n[3];
    i64 N;
    double *w[3]; /* widths, n[d] */
    double *g[3]; /* face factors incl. dt, n[d]-1 (+ the wrap face n[d]-1 <-> 0 at index n[d]-1 when periodic) */
    double *cm[3], *cp[3], *rw[3]; /* volume-scaled rows: g[s-1] / w[s], g[s] / w[s] (0 at a wall), 1 / w[s] */
    int per[3];   /* periodic direction: the level operator wraps */
    int tper[3];  /* ... and so do the transfers TOWARDS THE NEXT COARSER level (needs >= 4 cells: the four fine cells a
                     coarse cell gathers from must be distinct) */
    double *x, *x2, *b, *r;
    /* transfer tables towards the next coarser level, per direction (NULL on the coarsest level) */
    int32_t *par[3], *oth[3]; /* [n[d]] parent / other coarse index of fine cell s (oth == par: none) */
    double *wpar[3], *woth[3]; /* [n[d]] their weights */
    int32_t *fst[3];           /* [nc[d]+1] first child of coarse cell I */
} level_t;

typedef struct {
    int nlev;
    level_t L[MAXLEV];
    int pre, post, coarsest_sweeps;
    double omega;
    int nullspace; /* 0 none 1 constant 2 pinned */
    double hmin;      /* smallest fine cell width over the coarsenable directions */
    int target_shift; /* extra doublings of the merge target (levels on which nothing merged are skipped) */
    int smoother;  /* 0 damped Jacobi, 1 Chebyshev-Jacobi (pre/post = polynomial degree) */
    double cheb_lmax, cheb_ratio; /* eigenvalue window [lmax/ratio, lmax] of D^-1 A */
    double *d[MAXLEV];
} gmg_t;

static double vec_sum(i64 n, const double *a);
static void shift_vec(i64 n, double *a, double m);

/* Fused multiply-adds, spelled out (round 4): the library and this file are built -ffp-contract=off, so these four helpers are
 * the only places where a product is not rounded before it is added -- fma() of <math.h> here (vfmadd under
 * -march=x86-64-v3; the C library's correctly rounded software fma on a host without FMA3: the same bits), v_fma_f64 in
 * petibm_amd/csrc/gmg.hip, the same calls in the same order: facc one face of the scaled row sum, resid the
 * residual's last factor, tacc one term of an interpolation / restriction sum (and nacc / jrelax below). */
static inline double facc(double s, double c, double nb, double xc) { return fma(c, nb - xc, s); }
/* The damped-Jacobi step in its weighted-average form (gmg.hip, "weighted-average form"): with sum_faces c = -d,
 *     x + omega (bs - sum c (x_nb - x)) / d  =  (1 - omega) x + (omega / d) (bs - sum c x_nb);
 * jweight: wr = omega / d; nacc: one face, t - c x_nb, starting from t = bs in the order -x +x -y +y -z +z;
 * jrelax: fma(wr, t, (1 - omega) x); a step from a zero guess is wr * bs.  The same calls in the same order as the kernels. */
static inline double jweight(double omega, double d) { return omega / d; }
static inline double nacc(double t, double c, double nb) { return fma(-c, nb, t); }
static inline double jrelax(double x, double omc, double wr, double t) { return fma(wr, t, omc * x); }
static inline double resid(double b, double t, double w) { return fma(-t, w, b); }
static inline double tacc(double s, double w, double v) { return fma(w, v, s); }

static inline i64 idx(const level_t *l, i64 i, i64 j, i64 k) { return i + l->n[0] * (j + l->n[1] * k); }

/* The rows of the level operator DIVIDED BY THE CELL VOLUME: towards +d of cell s the coefficient (w_a w_b) g_d[s]
 * becomes g_d[s] / w_d[s], a function of one index -- the tables cm / cp made by make_g (0 at a wall, the wrap face on a
 * periodic direction).  Jacobi uses D^-1 (b - A x), which a row scaling leaves unchanged, so the smoothers work on the
 * scaled row (t, d, b / volume); the residual and the operator multiply the volume back in.  Same expressions, same
 * order as petibm_amd/csrc/gmg.hip ("level operator"). */
static inline void face_coefs(const level_t *l, i64 i, i64 j, i64 k, double c[6])
{
    c[0] = l->cm[0][i];
    c[1] = l->cp[0][i];
    c[2] = l->cm[1][j];
    c[3] = l->cp[1][j];
    c[4] = l->cm[2][k];
    c[5] = l->cp[2][k];
}
static inline double scale_b(const level_t *l, i64 i, i64 j, i64 k, double b) { return (b * (l->rw[0][i] * l->rw[1][j])) * l->rw[2][k]; }
static inline double unscale(const level_t *l, i64 i, i64 j, i64 k, double t) { return (t * (l->w[0][i] * l->w[1][j])) * l->w[2][k]; }

/* the scaled row sum t = sum_faces c (x_nb - x_c) at one cell, and the scaled diagonal */
static inline double apply_cell(const level_t *l, const double *x, i64 i, i64 j, i64 k, double *diag)
{
    double c[6];
    face_coefs(l, i, j, k, c);
    const i64 p = idx(l, i, j, k), sx = 1, sy = l->n[0], sz = l->n[0] * l->n[1];
    const double xc = x[p];
    double s = 0.0;
    if (i > 0) s = facc(s, c[0], x[p - sx], xc);
    else if (l->per[0]) s = facc(s, c[0], x[p + (l->n[0] - 1) * sx], xc);
    if (i < l->n[0] - 1) s = facc(s, c[1], x[p + sx], xc);
    else if (l->per[0]) s = facc(s, c[1], x[p - (l->n[0] - 1) * sx], xc);
    if (j > 0) s = facc(s, c[2], x[p - sy], xc);
    else if (l->per[1]) s = facc(s, c[2], x[p + (l->n[1] - 1) * sy], xc);
    if (j < l->n[1] - 1) s = facc(s, c[3], x[p + sy], xc);
    else if (l->per[1]) s = facc(s, c[3], x[p - (l->n[1] - 1) * sy], xc);
    if (k > 0) s = facc(s, c[4], x[p - sz], xc);
    else if (l->per[2]) s = facc(s, c[4], x[p + (l->n[2] - 1) * sz], xc);
    if (k < l->n[2] - 1) s = facc(s, c[5], x[p + sz], xc);
    else if (l->per[2]) s = facc(s, c[5], x[p - (l->n[2] - 1) * sz], xc);
    *diag = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
    return s;
}

/* bs - sum_faces c x_nb at one cell (what jrelax takes), and the scaled diagonal */
static inline double relax_cell(const level_t *l, const double *x, i64 i, i64 j, i64 k, double bs, double *diag)
{
    double c[6];
    face_coefs(l, i, j, k, c);
    const i64 p = idx(l, i, j, k), sx = 1, sy = l->n[0], sz = l->n[0] * l->n[1];
    double t = bs;
    if (i > 0) t = nacc(t, c[0], x[p - sx]);
    else if (l->per[0]) t = nacc(t, c[0], x[p + (l->n[0] - 1) * sx]);
    if (i < l->n[0] - 1) t = nacc(t, c[1], x[p + sx]);
    else if (l->per[0]) t = nacc(t, c[1], x[p - (l->n[0] - 1) * sx]);
    if (j > 0) t = nacc(t, c[2], x[p - sy]);
    else if (l->per[1]) t = nacc(t, c[2], x[p + (l->n[1] - 1) * sy]);
    if (j < l->n[1] - 1) t = nacc(t, c[3], x[p + sy]);
    else if (l->per[1]) t = nacc(t, c[3], x[p - (l->n[1] - 1) * sy]);
    if (k > 0) t = nacc(t, c[4], x[p - sz]);
    else if (l->per[2]) t = nacc(t, c[4], x[p + (l->n[2] - 1) * sz]);
    if (k < l->n[2] - 1) t = nacc(t, c[5], x[p + sz]);
    else if (l->per[2]) t = nacc(t, c[5], x[p - (l->n[2] - 1) * sz]);
    *diag = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
    return t;
}

static void lvl_free(level_t *l)
{
    for (int d = 0; d < 3; ++d) { free(l->w[d]); free(l->g[d]); free(l->cm[d]); free(l->cp[d]); free(l->rw[d]); free(l->par[d]); free(l->oth[d]); free(l->wpar[d]); free(l->woth[d]); free(l->fst[d]); }
    free(l->x); free(l->x2); free(l->b); free(l->r);
}

static void make_g(level_t *l, double dt)
{
    for (int d = 0; d < 3; ++d) {
        i64 n = l->n[d];
        l->g[d] = malloc(sizeof(double) * (size_t)(n > 1 ? n : 1));
        for (i64 s = 0; s + 1 < n; ++s) {
            const double dl = 0.5 * (l->w[d][s + 1] + l->w[d][s]);
            const double v = 1.0 / dl;
            l->g[d][s] = dt * v;
        }
        if (l->per[d] && n > 1) { /* wrap face: dL[d][d] of the periodic velocity mesh, 0.5*(w[0] + w[n-1]) (cartesianmesh.cpp:259-266) */
            const double dl = 0.5 * (l->w[d][0] + l->w[d][n - 1]);
            const double v = 1.0 / dl;
            l->g[d][n - 1] = dt * v;
        }
        l->cm[d] = calloc((size_t)(n > 1 ? n : 1), sizeof(double));
        l->cp[d] = calloc((size_t)(n > 1 ? n : 1), sizeof(double));
        l->rw[d] = malloc(sizeof(double) * (size_t)(n > 1 ? n : 1));
        for (i64 s = 0; s < n; ++s) {
            const double wq = l->w[d][s];
            l->rw[d][s] = 1.0 / wq;
            if (n > 1) {
                if (s > 0) l->cm[d][s] = l->g[d][s - 1] / wq;
                else if (l->per[d]) l->cm[d][s] = l->g[d][n - 1] / wq;
                if (s < n - 1 || l->per[d]) l->cp[d][s] = l->g[d][s] / wq;
            }
        }
    }
}

void *orc_gmg_create_periodic(int dim, const i64 *n_in, const double *wx, const double *wy, const double *wz, double dt,
                              int nullspace, int pre, int post, double omega, int coarsest_sweeps, int max_levels,
                              const int *periodic);
void *orc_gmg_create(int dim, const i64 *n_in, const double *wx, const double *wy, const double *wz, double dt,
                     int nullspace, int pre, int post, double omega, int coarsest_sweeps, int max_levels)
{
    const int none[3] = {0, 0, 0};
    return orc_gmg_create_periodic(dim, n_in, wx, wy, wz, dt, nullspace, pre, post, omega, coarsest_sweeps, max_levels, none);
}

/* periodic[d] (problem directions x, y[, z]): the level operators get the wrap face n-1 <-> 0, rediscretised on every
 * level like the interior faces (a direction coarsened down to ONE cell has no face left); the aggregates are unchanged
 * (pairs never straddle the seam) and the interpolation / restriction reach across the seam while the level has >= 4
 * cells in that direction. */
void *orc_gmg_create_periodic(int dim, const i64 *n_in, const double *wx, const double *wy, const double *wz, double dt,
                              int nullspace, int pre, int post, double omega, int coarsest_sweeps, int max_levels,
                              const int *periodic)
{
    gmg_t *G = calloc(1, sizeof(gmg_t));
    G->pre = pre; G->post = post; G->omega = omega; G->coarsest_sweeps = coarsest_sweeps; G->nullspace = nullspace;
    level_t *l = &G->L[0];
    const double one = 1.0;
    const double *ws[3];
    if (dim == 3) { l->n[0] = n_in[0]; l->n[1] = n_in[1]; l->n[2] = n_in[2]; ws[0] = wx; ws[1] = wy; ws[2] = wz; }
    else { l->n[0] = n_in[0]; l->n[1] = 1; l->n[2] = n_in[1]; ws[0] = wx; ws[1] = &one; ws[2] = wy; }
    int per[3];
    if (dim == 3) { per[0] = periodic[0]; per[1] = periodic[1]; per[2] = periodic[2]; }
    else { per[0] = periodic[0]; per[1] = 0; per[2] = periodic[1]; }
    for (int d = 0; d < 3; ++d) l->per[d] = per[d] && l->n[d] > 1;
    for (int d = 0; d < 3; ++d) {
        l->w[d] = malloc(sizeof(double) * (size_t)l->n[d]);
        memcpy(l->w[d], ws[d], sizeof(double) * (size_t)l->n[d]);
    }
    int nl = 1;
    if (max_levels > MAXLEV) max_levels = MAXLEV;
    G->hmin = 0.0;
    for (int d = 0; d < 3; ++d)
        if (l->n[d] > 1)
            for (i64 q = 0; q < l->n[d]; ++q)
                if (G->hmin == 0.0 || l->w[d][q] < G->hmin) G->hmin = l->w[d][q];
    for (;;) {
        l = &G->L[nl - 1];
        l->N = l->n[0] * l->n[1] * l->n[2];
        make_g(l, dt);
        l->x = calloc((size_t)l->N, 8); l->x2 = calloc((size_t)l->N, 8);
        l->b = calloc((size_t)l->N, 8); l->r = calloc((size_t)l->N, 8);
        G->d[nl - 1] = calloc((size_t)l->N, 8);
        if (nl >= max_levels) break;
        if (l->n[0] <= 2 && l->n[1] <= 2 && l->n[2] <= 2) break;
        level_t *c = &G->L[nl];
        /* selective coarsening: raise the target until at least one direction merges something */
        int merged_any = 0;
        for (int tries = 0; tries < 64 && !merged_any; ++tries, ++G->target_shift) {
            const double target = 1.5 * G->hmin * ldexp(1.0, nl + G->target_shift);
            for (int d = 0; d < 3; ++d) {
                const i64 n = l->n[d];
                l->tper[d] = l->per[d] && n >= 4;
                free(l->par[d]); free(l->oth[d]); free(l->wpar[d]); free(l->woth[d]); free(l->fst[d]); free(c->w[d]);
                l->par[d] = malloc(sizeof(int32_t) * (size_t)n);
                l->oth[d] = malloc(sizeof(int32_t) * (size_t)n);
                l->wpar[d] = malloc(sizeof(double) * (size_t)n);
                l->woth[d] = malloc(sizeof(double) * (size_t)n);
                l->fst[d] = malloc(sizeof(int32_t) * (size_t)(n + 1));
                c->w[d] = malloc(sizeof(double) * (size_t)n);
                i64 I = 0;
                for (i64 s = 0; s < n; ++I) {
                    l->fst[d][I] = (int32_t)s;
                    if (n > 2 && s + 1 < n && l->w[d][s] + l->w[d][s + 1] <= target) {
                        l->par[d][s] = l->par[d][s + 1] = (int32_t)I;
                        c->w[d][I] = l->w[d][s] + l->w[d][s + 1];
                        s += 2;
                        merged_any = 1;
                    } else {
                        l->par[d][s] = (int32_t)I;
                        c->w[d][I] = l->w[d][s];
                        s += 1;
                    }
                }
                l->fst[d][I] = (int32_t)n;
                c->n[d] = I;
                /* width-based linear interpolation weights */
                for (i64 s = 0; s < n; ++s) {
                    const i64 P = l->par[d][s];
                    const i64 f0 = l->fst[d][P], f1 = l->fst[d][P + 1];
                    i64 O = P;
                    double t = 0.0;
                    if (f1 - f0 == 2) {
                        const int left = (s == f0);
                        O = left ? P - 1 : P + 1;
                        if (l->tper[d]) O = (O + I) % I; /* across the periodic seam */
                        if (O < 0 || O >= I) O = P;
                        else {
                            const double sib = left ? l->w[d][s + 1] : l->w[d][s - 1];
                            t = sib / (c->w[d][P] + c->w[d][O]);
                        }
                    }
                    l->oth[d][s] = (int32_t)O;
                    l->wpar[d][s] = 1.0 - t;
                    l->woth[d][s] = t;
                }
            }
        }
        if (!merged_any) break;
        G->target_shift--; /* the loop's ++ after the successful try */
        for (int d = 0; d < 3; ++d) c->per[d] = per[d] && c->n[d] > 1;
        nl++;
    }
    G->nlev = nl;
    return G;
}

void orc_gmg_destroy(void *h)
{
    gmg_t *G = h;
    for (int i = 0; i < G->nlev; ++i) { lvl_free(&G->L[i]); free(G->d[i]); }
    free(G);
}

int orc_gmg_num_levels(void *h) { return ((gmg_t *)h)->nlev; }
void orc_gmg_level_size(void *h, int lev, i64 *n) { for (int d = 0; d < 3; ++d) n[d] = ((gmg_t *)h)->L[lev].n[d]; }

/* y = A_level x (the matrix-free stencil twin, K2) */
void orc_gmg_apply_operator(void *h, int lev, const double *x, double *y)
{
    const level_t *l = &((gmg_t *)h)->L[lev];
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < l->n[2]; ++k)
        for (i64 j = 0; j < l->n[1]; ++j)
            for (i64 i = 0; i < l->n[0]; ++i) {
                double d;
                y[idx(l, i, j, k)] = unscale(l, i, j, k, apply_cell(l, x, i, j, k, &d));
            }
}

/* xo = (1 - omega) xi + (omega / diag) (b - sum c x_nb) ; zero_guess: xo = (omega / diag) b */
static void smooth(const level_t *l, double omega, const double *b, const double *xi, double *xo, int zero_guess)
{
    const double omc = 1.0 - omega;
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < l->n[2]; ++k)
        for (i64 j = 0; j < l->n[1]; ++j)
            for (i64 i = 0; i < l->n[0]; ++i) {
                const i64 p = idx(l, i, j, k);
                double d;
                if (zero_guess) {
                    double c[6];
                    face_coefs(l, i, j, k, c);
                    d = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
                    xo[p] = jweight(omega, d) * scale_b(l, i, j, k, b[p]);
                } else {
                    const double t = relax_cell(l, xi, i, j, k, scale_b(l, i, j, k, b[p]), &d);
                    xo[p] = jrelax(xi[p], omc, jweight(omega, d), t);
                }
            }
}

static void residual(const level_t *l, const double *b, const double *x, double *r)
{
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < l->n[2]; ++k)
        for (i64 j = 0; j < l->n[1]; ++j)
            for (i64 i = 0; i < l->n[0]; ++i) {
                double d;
                const i64 p = idx(l, i, j, k);
                r[p] = resid(b[p], apply_cell(l, x, i, j, k, &d) * (l->w[0][i] * l->w[1][j]), l->w[2][k]);
            }
}

/* xf += P xc (table driven) */
static void prolong_add(const level_t *f, const level_t *c, const double *xc, double *xf)
{
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < f->n[2]; ++k)
        for (i64 j = 0; j < f->n[1]; ++j)
            for (i64 i = 0; i < f->n[0]; ++i) {
                const i64 I[2] = {f->par[0][i], f->oth[0][i]}, J[2] = {f->par[1][j], f->oth[1][j]},
                          K[2] = {f->par[2][k], f->oth[2][k]};
                const double wi[2] = {f->wpar[0][i], f->woth[0][i]}, wj[2] = {f->wpar[1][j], f->woth[1][j]},
                             wk[2] = {f->wpar[2][k], f->woth[2][k]};
                double s = 0.0;
                for (int c2 = 0; c2 < 2; ++c2)
                    for (int b2 = 0; b2 < 2; ++b2)
                        for (int a2 = 0; a2 < 2; ++a2) {
                            const double wgt = (wk[c2] * wj[b2]) * wi[a2];
                            if (wgt != 0.0) s = tacc(s, wgt, xc[idx(c, I[a2], J[b2], K[c2])]);
                        }
                xf[idx(f, i, j, k)] += s;
            }
}

/* weight with which fine cell s feeds coarse cell I in one direction */
static inline double rw(const level_t *f, int d, i64 s, i64 I)
{
    double w = 0.0;
    if (f->par[d][s] == I) w = f->wpar[d][s];
    else if (f->oth[d][s] == I) w = f->woth[d][s];
    return w;
}

/* bc = P^T rf : gather form over coarse cells (what the HIP kernel does): fine cells fst[I]-1 .. fst[I+1] */
static void restrict_t(const level_t *f, const level_t *c, const double *rf, double *bc)
{
#pragma omp parallel for schedule(static)
    for (i64 K = 0; K < c->n[2]; ++K)
        for (i64 J = 0; J < c->n[1]; ++J)
            for (i64 I = 0; I < c->n[0]; ++I) {
                double s = 0.0;
                const i64 k0 = f->fst[2][K] - 1, k1 = f->fst[2][K + 1];
                const i64 j0 = f->fst[1][J] - 1, j1 = f->fst[1][J + 1];
                const i64 i0 = f->fst[0][I] - 1, i1 = f->fst[0][I + 1];
                for (i64 kr = k0; kr <= k1; ++kr) {
                    i64 k = kr;
                    if (f->tper[2]) k = (kr + f->n[2]) % f->n[2];
                    if (k < 0 || k >= f->n[2]) continue;
                    const double wz = rw(f, 2, k, K);
                    if (wz == 0.0) continue;
                    /* direction by direction (gmg.hip: rsum_x / restrict_plane): t = sum_x wx r, u = sum_y wy t, s = sum_z wz u */
                    double u = 0.0;
                    for (i64 jr = j0; jr <= j1; ++jr) {
                        i64 j = jr;
                        if (f->tper[1]) j = (jr + f->n[1]) % f->n[1];
                        if (j < 0 || j >= f->n[1]) continue;
                        const double wy = rw(f, 1, j, J);
                        if (wy == 0.0) continue;
                        double t = 0.0;
                        for (i64 ir = i0; ir <= i1; ++ir) {
                            i64 i = ir;
                            if (f->tper[0]) i = (ir + f->n[0]) % f->n[0];
                            if (i < 0 || i >= f->n[0]) continue;
                            const double wx = rw(f, 0, i, I);
                            if (wx == 0.0) continue;
                            t = tacc(t, wx, rf[idx(f, i, j, k)]);
                        }
                        u = tacc(u, wy, t);
                    }
                    s = tacc(s, wz, u);
                }
                bc[idx(c, I, J, K)] = s;
            }
}

/* Chebyshev-Jacobi: `deg` steps of the three-term recurrence on D^-1 A over [lmax/ratio, lmax];
 * x in/out (zero_guess: x starts at 0), d = work vector, tmp = ping-pong buffer for x */
static double *cheby(gmg_t *G, level_t *l, int lev, int deg, const double *b, double *x, double *x2, int zero_guess)
{
    const double lmax = G->cheb_lmax, lmin = lmax / G->cheb_ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma;
    double *d = G->d[lev], *cur = x, *nxt = x2;
    for (int s = 0; s < deg; ++s) {
        const double c_d = (s == 0) ? 0.0 : 0.0; (void)c_d;
        double rho_new = (s == 0) ? rho : 1.0 / (2.0 * sigma - rho);
        const double a_d = (s == 0) ? 0.0 : rho_new * rho;
        const double a_z = (s == 0) ? 1.0 / theta : 2.0 * rho_new / delta;
        const int zg = zero_guess && s == 0;
#pragma omp parallel for schedule(static)
        for (i64 k = 0; k < l->n[2]; ++k)
            for (i64 j = 0; j < l->n[1]; ++j)
                for (i64 i = 0; i < l->n[0]; ++i) {
                    const i64 p = idx(l, i, j, k);
                    double dg, z;
                    if (zg) {
                        double c[6];
                        face_coefs(l, i, j, k, c);
                        dg = -(((((c[0] + c[1]) + c[2]) + c[3]) + c[4]) + c[5]);
                        z = scale_b(l, i, j, k, b[p]) / dg;
                        d[p] = a_z * z;
                        nxt[p] = d[p];
                    } else {
                        const double ax = apply_cell(l, cur, i, j, k, &dg);
                        z = (scale_b(l, i, j, k, b[p]) - ax) / dg;
                        const double dn = a_d * d[p] + a_z * z;
                        d[p] = dn;
                        nxt[p] = cur[p] + dn;
                    }
                }
        if (s > 0) rho = rho_new;
        double *t = cur; cur = nxt; nxt = t;
    }
    return cur;
}

static void vcycle(gmg_t *G, int lev)
{
    level_t *l = &G->L[lev];
    if (lev == G->nlev - 1) {
        /* coarsest: Jacobi sweeps from zero */
        double *a = l->x, *b2 = l->x2;
        smooth(l, G->omega, l->b, NULL, a, 1);
        for (int s = 1; s < G->coarsest_sweeps; ++s) {
            smooth(l, G->omega, l->b, a, b2, 0);
            double *t = a; a = b2; b2 = t;
        }
        if (a != l->x) memcpy(l->x, a, sizeof(double) * (size_t)l->N);
        return;
    }
    double *a = l->x, *b2 = l->x2;
    if (G->smoother == 1) {
        a = cheby(G, l, lev, G->pre, l->b, l->x2, l->x, 1);  /* first step writes into l->x */
        b2 = (a == l->x) ? l->x2 : l->x;
    } else {
    smooth(l, G->omega, l->b, NULL, a, 1);
    for (int s = 1; s < G->pre; ++s) {
        smooth(l, G->omega, l->b, a, b2, 0);
        double *t = a; a = b2; b2 = t;
    }
    }
    residual(l, l->b, a, l->r);
    restrict_t(l, &G->L[lev + 1], l->r, G->L[lev + 1].b);
    vcycle(G, lev + 1);
    prolong_add(l, &G->L[lev + 1], G->L[lev + 1].x, a);
    if (G->smoother == 1) {
        a = cheby(G, l, lev, G->post, l->b, a, b2, 0);
    } else
    for (int s = 0; s < G->post; ++s) {
        smooth(l, G->omega, l->b, a, b2, 0);
        double *t = a; a = b2; b2 = t;
    }
    if (a != l->x) memcpy(l->x, a, sizeof(double) * (size_t)l->N);
}

/* z = V-cycle(r), raw (no mean removal); PINNED: r[0] replaced by -sum_{i>0} r_i */
void orc_gmg_apply(void *h, const double *r, double *z)
{
    gmg_t *G = h;
    level_t *l = &G->L[0];
    memcpy(l->b, r, sizeof(double) * (size_t)l->N);
    if (G->nullspace == 2) l->b[0] = r[0] - vec_sum(l->N, r);
    vcycle(G, 0);
    memcpy(z, l->x, sizeof(double) * (size_t)l->N);
}

/* PCG (KSPCG recurrences, see oracle.c) on the CSR matrix, preconditioned by
 * the V-cycle; normtype 0: ||z|| (after projection), 1: ||r||.
 * nullspace 1: z <- z - mean(z); 2: z <- z - z[0], z[0] = r[0].            */
void orc_spmv(i64 n, const i64 *rowptr, const i64 *col, const double *val, const double *x, double *y);

static double vec_sum(i64 n, const double *a)
{
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (i64 i = 0; i < n; ++i) s += a[i];
    return s;
}
static void shift_vec(i64 n, double *a, double m)
{
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) a[i] -= m;
}

static double ddot(i64 n, const double *a, const double *b)
{
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (i64 i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* single_reduction != 0: the recurrences of KSPCGUseSingleReduction (oracle.c:orc_cg_single_reduction): s = A z, delta = z's,
 * w = s + b w and dpi = delta - beta^2 dpiold / betaold^2 from the second iteration on. */
static int pcg_gmg(void *h, i64 n, const i64 *rowptr, const i64 *col, const double *val, int normtype, double rtol,
                   double atol, int maxit, int guess_nonzero, const double *b, double *x, int *its_out,
                   double *rnorm_out, double *history, int single_reduction)
{
    gmg_t *G = h;
    double *R = malloc((size_t)n * 8), *Z = malloc((size_t)n * 8), *P = malloc((size_t)n * 8), *W = malloc((size_t)n * 8);
    double *Sv = single_reduction ? malloc((size_t)n * 8) : NULL;
    double beta, betaold = 1.0, dpi = 0.0, dpiold, dp, a, ttol, rnorm0, delta = 0.0;
    int reason = 0, i = 0;
    if (!guess_nonzero) { memset(x, 0, (size_t)n * 8); memcpy(R, b, (size_t)n * 8); }
    else {
        orc_spmv(n, rowptr, col, val, x, R);
#pragma omp parallel for schedule(static)
        for (i64 q = 0; q < n; ++q) R[q] = b[q] - R[q];
    }
#define PCAPPLY()                                                        \
    do {                                                                 \
        orc_gmg_apply(G, R, Z);                                          \
        if (G->nullspace == 1) {                                         \
            shift_vec(n, Z, vec_sum(n, Z) / (double)n);                  \
        } else if (G->nullspace == 2) {                                  \
            shift_vec(n, Z, Z[0]);                                       \
            Z[0] = R[0];                                                 \
        }                                                                \
    } while (0)
    PCAPPLY();
    dp = (normtype == 0) ? sqrt(ddot(n, Z, Z)) : sqrt(ddot(n, R, R));
    rnorm0 = dp;
    ttol = fmax(rtol * rnorm0, atol);
    if (history) history[0] = dp;
    *its_out = 0;
    if (dp <= ttol) { reason = 2; goto done; }
    if (single_reduction) {
        orc_spmv(n, rowptr, col, val, Z, Sv);
        delta = ddot(n, Z, Sv);
    }
    beta = ddot(n, Z, R);
    do {
        *its_out = i + 1;
        if (beta == 0.0) { reason = 3; break; }
        if (i > 0 && ((beta > 0) != (betaold > 0))) { reason = -8; break; }
        if (i == 0) memcpy(P, Z, (size_t)n * 8);
        else {
            const double bb = beta / betaold;
#pragma omp parallel for schedule(static)
            for (i64 q = 0; q < n; ++q) P[q] = Z[q] + bb * P[q];
        }
        dpiold = dpi;
        if (!single_reduction || i == 0) {
            orc_spmv(n, rowptr, col, val, P, W);
            dpi = ddot(n, P, W);
        } else {
            const double bb = beta / betaold;
#pragma omp parallel for schedule(static)
            for (i64 q = 0; q < n; ++q) W[q] = Sv[q] + bb * W[q];
            dpi = delta - beta * beta * dpiold / (betaold * betaold);
        }
        betaold = beta;
        if (dpi == 0.0 || (i > 0 && ((dpi > 0) != (dpiold > 0)))) { reason = -10; break; }
        a = beta / dpi;
#pragma omp parallel for schedule(static)
        for (i64 q = 0; q < n; ++q) { x[q] = x[q] + a * P[q]; R[q] = R[q] - a * W[q]; }
        if (normtype == 1) {
            dp = sqrt(ddot(n, R, R));
            if (history) history[i + 1] = dp;
            if (dp <= ttol) { reason = 2; break; }
            PCAPPLY();
        } else {
            PCAPPLY();
            dp = sqrt(ddot(n, Z, Z));
            if (history) history[i + 1] = dp;
            if (dp <= ttol) { reason = 2; break; }
        }
        if (single_reduction) {
            orc_spmv(n, rowptr, col, val, Z, Sv);
            delta = ddot(n, Z, Sv);
        }
        beta = ddot(n, Z, R);
        i++;
    } while (i < maxit);
    if (!reason && i >= maxit) reason = -3;
done:
    *rnorm_out = dp;
    free(R); free(Z); free(P); free(W); free(Sv);
    return reason;
}

int orc_pcg_gmg(void *h, i64 n, const i64 *rowptr, const i64 *col, const double *val, int normtype, double rtol,
                double atol, int maxit, int guess_nonzero, const double *b, double *x, int *its_out,
                double *rnorm_out, double *history)
{
    return pcg_gmg(h, n, rowptr, col, val, normtype, rtol, atol, maxit, guess_nonzero, b, x, its_out, rnorm_out, history, 0);
}
int orc_pcg_gmg_single_reduction(void *h, i64 n, const i64 *rowptr, const i64 *col, const double *val, int normtype, double rtol,
                                 double atol, int maxit, int guess_nonzero, const double *b, double *x, int *its_out,
                                 double *rnorm_out, double *history)
{
    return pcg_gmg(h, n, rowptr, col, val, normtype, rtol, atol, maxit, guess_nonzero, b, x, its_out, rnorm_out, history, 1);
}


/* ---- DBNG = D (dt I) G assembled directly from the cell widths, entry for entry the same arithmetic as
 * oracle/operators.py:create_poisson_operator (createdivergence.cpp:140-151, creategradient.cpp:70-86,
 * createbn.cpp:49, navierstokes.cpp:349-356); used for full-size CPU baselines where the numpy COO route
 * would need hundreds of GB.  2-D grids: nz = 1.  int32 output (the device CSR's index width). */
static i64 nnz_before(i64 g, int dim, i64 nx, i64 ny, i64 nz)
{
    const i64 pl = nx * ny;
    i64 c = g;
    c += g - (g + nx - 1) / nx;
    c += g - g / nx;
    const i64 kq = g / pl, rem = g % pl;
    c += g - (kq * nx + (rem < nx ? rem : nx));
    const i64 top = rem - (ny - 1) * nx;
    c += g - (kq * nx + (top > 0 ? top : 0));
    if (dim == 3) {
        c += g - (g < pl ? g : pl);
        const i64 last = g - (nz - 1) * pl;
        c += g - (last > 0 ? last : 0);
    }
    return c;
}

i64 orc_poisson_nnz(int dim, i64 nx, i64 ny, i64 nz) { return nnz_before(nx * ny * nz, dim, nx, ny, nz); }

void orc_assemble_poisson32(int dim, i64 nx, i64 ny, i64 nz, const double *wx, const double *wy, const double *wz,
                            double dt, int32_t *rowptr, int32_t *col, double *val)
{
    const i64 n = nx * ny * nz, pl = nx * ny;
    const double one = 1.0;
    if (dim == 2) wz = &one;
    double *g[3];
    const double *w[3] = {wx, wy, wz};
    const i64 nn[3] = {nx, ny, nz};
    for (int d = 0; d < 3; ++d) {
        g[d] = malloc(sizeof(double) * (size_t)(nn[d] > 1 ? nn[d] - 1 : 1));
        for (i64 s = 0; s + 1 < nn[d]; ++s) {
            const double dl = 0.5 * (w[d][s + 1] + w[d][s]);
            const double v = 1.0 / dl;
            g[d][s] = dt * v;
        }
    }
#pragma omp parallel for schedule(static)
    for (i64 r = 0; r <= n; ++r) {
        i64 p = nnz_before(r, dim, nx, ny, nz);
        rowptr[r] = (int32_t)p;
        if (r == n) continue;
        const i64 i = r % nx, j = (r / nx) % ny, k = r / pl;
        const double ax = wy[j] * wz[k], ay = wx[i] * wz[k], az = wx[i] * wy[j];
        const int hxm = i > 0, hxp = i < nx - 1, hym = j > 0, hyp = j < ny - 1;
        const int hzm = (dim == 3) && k > 0, hzp = (dim == 3) && k < nz - 1;
        const double oxm = hxm ? ax * g[0][i - 1] : 0.0, oxp = hxp ? ax * g[0][i] : 0.0;
        const double oym = hym ? ay * g[1][j - 1] : 0.0, oyp = hyp ? ay * g[1][j] : 0.0;
        const double ozm = hzm ? az * g[2][k - 1] : 0.0, ozp = hzp ? az * g[2][k] : 0.0;
        double d = 0.0;
        int first = 1;
        const double t[6] = {oxm, oxp, oym, oyp, ozm, ozp};
        const int has[6] = {hxm, hxp, hym, hyp, hzm, hzp};
        for (int q = 0; q < 6; ++q)
            if (has[q]) {
                if (first) { d = -t[q]; first = 0; }
                else d = d + (-t[q]);
            }
        if (hzm) { col[p] = (int32_t)(r - pl); val[p] = ozm; ++p; }
        if (hym) { col[p] = (int32_t)(r - nx); val[p] = oym; ++p; }
        if (hxm) { col[p] = (int32_t)(r - 1); val[p] = oxm; ++p; }
        col[p] = (int32_t)r; val[p] = d; ++p;
        if (hxp) { col[p] = (int32_t)(r + 1); val[p] = oxp; ++p; }
        if (hyp) { col[p] = (int32_t)(r + nx); val[p] = oyp; ++p; }
        if (hzp) { col[p] = (int32_t)(r + pl); val[p] = ozp; ++p; }
    }
    for (int d = 0; d < 3; ++d) free(g[d]);
}

/* the same PCG+V-cycle on the int32 CSR (what the cpu_baseline leg of bench.py times) */
void orc_spmv32(i64 n, const int32_t *rowptr, const int32_t *col, const double *val, const double *x, double *y);

int orc_pcg_gmg32(void *h, i64 n, const int32_t *rowptr, const int32_t *col, const double *val, double rtol, int maxit,
                  const double *b, double *x, int *its_out, double *rnorm_out)
{
    gmg_t *G = h;
    double *R = malloc((size_t)n * 8), *Z = malloc((size_t)n * 8), *P = malloc((size_t)n * 8), *W = malloc((size_t)n * 8);
    double beta, betaold = 1.0, dpi, dp, a, ttol;
    int reason = 0, i = 0;
#pragma omp parallel for schedule(static)
    for (i64 q = 0; q < n; ++q) { x[q] = 0.0; R[q] = b[q]; }
    orc_gmg_apply(G, R, Z);
    if (G->nullspace == 1) shift_vec(n, Z, vec_sum(n, Z) / (double)n);
    dp = sqrt(ddot(n, R, R));
    ttol = rtol * dp;
    *its_out = 0;
    beta = ddot(n, Z, R);
    if (dp <= ttol) { reason = 2; goto done; }
    do {
        *its_out = i + 1;
        if (i == 0) memcpy(P, Z, (size_t)n * 8);
        else {
            const double bb = beta / betaold;
#pragma omp parallel for schedule(static)
            for (i64 q = 0; q < n; ++q) P[q] = Z[q] + bb * P[q];
        }
        orc_spmv32(n, rowptr, col, val, P, W);
        dpi = ddot(n, P, W);
        betaold = beta;
        a = beta / dpi;
#pragma omp parallel for schedule(static)
        for (i64 q = 0; q < n; ++q) { x[q] = x[q] + a * P[q]; R[q] = R[q] - a * W[q]; }
        dp = sqrt(ddot(n, R, R));
        if (dp <= ttol) { reason = 2; break; }
        orc_gmg_apply(G, R, Z);
        if (G->nullspace == 1) shift_vec(n, Z, vec_sum(n, Z) / (double)n);
        beta = ddot(n, Z, R);
        i++;
    } while (i < maxit);
    if (!reason) reason = -3;
done:
    *rnorm_out = dp;
    free(R); free(Z); free(P); free(W);
    return reason;
}

void orc_gmg_set_chebyshev(void *h, int on, double lmax, double ratio)
{
    gmg_t *G = h;
    G->smoother = on;
    G->cheb_lmax = lmax;
    G->cheb_ratio = ratio;
}
