// ns_internal.hpp -- device-side mesh / ghost-point description and the engine state shared by the time-step
// kernels (navierstokes.hip) and the immersed-boundary operators (ibm.hip).
#pragma once
#include "pib_internal.hpp"

struct pib_ns;

namespace pib {

struct NsField {
    int64_t n[3];
    int64_t off;            // first entry of the field's block in the packed vector
    const double *dl[3];    // dL[f][d], index s+1
    const double *co[3];    // coord[f][d], index s+1
    // per-direction tables of the point-wise quotients (same expressions as the kernels evaluate, hoisted: an fp64
    // division costs ~40 instructions): Laplacian coefficients 1/((co[s]-co[s-1])*dl[s]), 1/((co[s+1]-co[s])*dl[s])
    // (createlaplacian.cpp:134-148) and, along the component's own direction, the gradient entry 1/dl[s]; index s
    const double *lneg[3], *lpos[3], *ginv;
    // ghost points per boundary location: ghost = a0*target + a1 (a0 is uniform over a face); face arrays are
    // indexed a + na*b over the two perpendicular axes in natural order (misc.cpp:154-196)
    double a0[6];
    int type[6];            // 0 DIRICHLET, 1 NEUMANN, 2 CONVECTIVE, 3 PERIODIC (no ghost points: indices wrap)
    double bcv[6];          // the BC value (Dirichlet value, Neumann gradient, convective speed)
    double gdl[6];          // distance ghost - target
    int64_t goff[6];        // offset of the face in the ghost arrays
    int64_t gcnt[6];        // points of the face
};
// time-integration coefficients (include/petibm/timeintegration.h:107-166): the explicit coefficients of the convective
// scheme and the implicit / explicit coefficients of the diffusive one
struct NsTime {
    int nconv, ndiff;       // explicit terms kept: EULER_EXPLICIT 1, EULER_IMPLICIT 0, ADAMS_BASHFORTH_2 2, CRANK_NICOLSON 1
    double cc[2], dc[2];
    double cimpl;           // implicit coefficient of the diffusive scheme (A = I/dt - cimpl nu L)
};
struct NsDev {
    int dim;
    int per;                // bit d: direction d periodic (every component: misc.cpp checkPeriodicBC)
    NsField f[3];
    int64_t pn[3];          // pressure cells
    const double *pw[3];    // pressure-cell widths
    int64_t UN, pN;
    int64_t nghost;
    double *a1, *a1n, *gv;  // [nghost] current / next ghost equations' a1, ghost values
    // several ranks on a periodic slab axis: planes [0, seam_lo) and [seam_hi, ...) of the extended slab along seam_axis are
    // images from across the seam (-1: none)
    int seam_axis;
    int64_t seam_lo, seam_hi;
};

// index of the ghost point of boundary `loc` facing (i,j,k) within its face
__device__ __forceinline__ int64_t face_index(const NsField &F, int loc, int64_t i, int64_t j, int64_t k)
{
    const int axis = loc >> 1;
    if (axis == 0) return F.goff[loc] + j + F.n[1] * k;
    if (axis == 1) return F.goff[loc] + i + F.n[0] * k;
    return F.goff[loc] + i + F.n[0] * j;
}

__device__ __forceinline__ int64_t fidx(const NsField &F, int64_t i, int64_t j, int64_t k)
{
    return F.off + i + F.n[0] * (j + F.n[1] * k);
}

struct IbState;
void ib_release(IbState *ib);
// coupled IBPM (applications/ibpm): the pressure / forces solve and the projection of one step (ibm.hip)
bool ib_is_coupled(const pib_ns *ns);
int ib_coupled_solve_and_project(pib_ns *ns);
// out = BNG phi = dt G phi on the whole velocity vector ; w -= D t (row 0 untouched when the pressure is pinned)
int ns_bng_apply(pib_ns *ns, const double *phi, double *out, hipStream_t q);
int ns_div_sub(pib_ns *ns, const double *t, double *w, hipStream_t q);
// z-slabs, BN order > 1: t (extended slab) <- sum_k (dt c nu)^(k-1) L^(k-1) t on the owned points (navierstokes.hip; collective)
int ns_bn_series_slab(pib_ns *ns, double *t);
// the extra stages of DecoupledIBPMSolver::advance (decoupledibpm.cpp:105-131)
int ib_spread_forces(pib_ns *ns);   // rhs1 += H f
// what the engine's stream has enqueued so far must be done before `sol` starts: an event the solver's stream waits for --
// not a host synchronisation (three per time step, each an idle gap of 15-20 us on the GPU)
int ns_before_solve(pib_ns *ns, pib_solver *sol);
int ib_solve_forces(pib_ns *ns);    // rhsf = -E u ; EBNH df = rhsf ; u += BNH df
int ib_update_forces(pib_ns *ns);   // f += df

}  // namespace pib

struct pib_ns {
    pib::NsDev D;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_dep = nullptr;  // what a solver's stream waits for before a solve (instead of a host synchronisation)
    pib_solver *vsol = nullptr, *psol = nullptr;
    double dt = 0, nu = 0;
    double *U = nullptr, *p = nullptr, *dP = nullptr, *rhs1 = nullptr, *rhs2 = nullptr, *conv[2] = {nullptr, nullptr};
    double *diff0 = nullptr;  // explicit diffusion term of the last step (restart files carry it: navierstokes.cpp:672-680)
    double *diff1 = nullptr;  // the one before (a two-term diffusive scheme only)
    pib::NsTime T = {2, 1, {1.5, -0.5}, {0.5, 0.0}, 0.5};  // ADAMS_BASHFORTH_2 + CRANK_NICOLSON
    std::vector<double *> owned;
    int pinned = 0;
    int v_iters = 0, p_iters = 0;
    double v_res = 0, p_res = 0;
    int64_t steps = 0;
    // immersed bodies (ibm.hip); null when the flow has none
    pib::IbState *ib = nullptr;
    std::vector<double> h_vtx[3];  // vertex coordinates (coord[4][d]) and u-widths (dL[0][d], ghosted) on the host:
    std::vector<double> h_dlu[3];  // background-cell search and kernel widths of the bodies
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int f_iters = 0;
    double f_res = 0;
    int periodic[3] = {0, 0, 0};
    bool ring = false;  // several ranks and a periodic slab axis: both ends of the extended slab are cuts, the plane exchanges wrap
    // parameters.BN > 1 (pib_ns_set_bn_order): the projection multiplies by the assembled BNG
    int bn_order = 1;
    bool neumann_normal = false;  // a NEUMANN condition on a normal velocity component: D carries a ghost fold, DBNG comes from the product chain (bn.hip)
    int32_t *bng_rowptr = nullptr, *bng_col = nullptr;
    double *bng_val = nullptr;
    int64_t bng_nnz = 0;
    // ... and the immersed-boundary operators by the assembled BN (BNH = BN H, EBNH = E BNH: decoupledibpm.cpp:194-205)
    int32_t *bn_rowptr = nullptr, *bn_col = nullptr;
    double *bn_val = nullptr;
    int64_t bn_nnz = 0;
    double *bn_tmp = nullptr;  // z-slabs: t = dt G dP on the extended slab (the projection applies BN term by term)
    // z-slab (y-slab) decomposition: the engine's mesh is this rank's EXTENDED slab (navierstokes.hip: ns_create_impl)
    // stage timers under the reference's PetscLogStage names (navierstokes.cpp:186-199, decoupledibpm.cpp:93-97): off by default;
    // when on, events on the engine's stream bracket the stages of every step (pib_ns_stage_timers)
    bool stage_timing = false;
    hipEvent_t ev_stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double stage_ms[6] = {0, 0, 0, 0, 0, 0};
    int64_t stage_steps = 0;
    int rank = 0, nranks = 1;
    int64_t slab_pk0 = 0, slab_pk1 = 0, slab_e0 = 0;          // owned pressure planes [pk0, pk1), first plane of the extended slab
    int64_t fld_plane[3] = {0, 0, 0}, fld_own_lo[3] = {0, 0, 0}, fld_own_cnt[3] = {0, 0, 0}, fld_pk_off[3] = {0, 0, 0};
    int64_t UN_owned = 0, pN_owned = 0, p_plane = 0;
    double *Upk = nullptr, *rhs1pk = nullptr;                  // packed owned [u | v | w] (the velocity solver's vectors)
    int64_t h_n[3] = {1, 1, 1};
    std::vector<double> h_w[3];
    double h_a0[18] = {0};
};
