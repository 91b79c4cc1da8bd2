// navierstokes.hip -- SURVEY.md 8f-1: device-resident right-hand-side assembly and projection around the two
// linear solves, i.e. one NavierStokesSolver::advance (applications/navierstokes/navierstokes.cpp:240-266) with
// every vector in HBM and no matrix but the two system matrices:
//
//   assembleRHSVelocity (:432-521)  rhs1 = -G p + u/dt + sum c_i conv_i + sum d_i diff_i + c nu Lbc
//   solveVelocity       (:524-537)  A u* = rhs1                      -> pib_solve (BiCGStab / CG + Jacobi)
//   assembleRHSPoisson  (:540-563)  rhs2 = D u* + Dbc ; rhs2[0] = 0 when the pressure is pinned
//   solvePoisson        (:566-580)  DBNG dP = rhs2                   -> pib_solve (multigrid-PCG)
//   applyDivergenceFreeVelocity (:583-598)  u = u* - BNG dP ; updatePressure (:601-615)  p += dP
//
// G, D, L, BNG are applied matrix-free from the 1-D mesh arrays, entry for entry and in the same summation
// order as the assembled AIJ matrices of the reference (creategradient.cpp:64-128, createdivergence.cpp:135-223,
// createlaplacian.cpp:108-263); N(u) is src/operators/createconvection.cpp:40-195 with ghost values computed on
// the fly (ghost = a0*target + a1: singleboundarydirichlet.cpp:35-44, singleboundaryneumann.cpp:27-28); the BC
// correction shells (createlaplacian.cpp:45-78, createdivergence.cpp:45-78) are the constant vectors
// sum coeff*a1.  AB2 convection {1.5,-0.5} + Crank-Nicolson diffusion (timeintegration.h:107-166), BN order 1.
// Time-independent ghost equations only (Dirichlet, Neumann); single GPU.
// The oracle (oracle/navierstokes.py) performs the same VecScale/VecAXPY sequence; the explicit parts agree
// bit for bit, the solves to solver tolerance.
#include <cstring>

#include "pib_internal.hpp"

namespace pib {

struct NsField {
    int64_t n[3];
    int64_t off;            // first entry of the field's block in the packed vector
    const double *dl[3];    // dL[f][d], index s+1
    const double *co[3];    // coord[f][d], index s+1
    double a0[6], a1[6];    // ghost = a0*target + a1 per boundary location
};
struct NsDev {
    int dim;
    NsField f[3];
    int64_t pn[3];          // pressure cells
    const double *pw[3];    // pressure-cell widths
    int64_t UN, pN;
};

__device__ __forceinline__ int64_t fidx(const NsField &F, int64_t i, int64_t j, int64_t k)
{
    return F.off + i + F.n[0] * (j + F.n[1] * k);
}

// velocity value at (i,j,k) of field f; an index one step outside is the ghost point a0*target + a1
__device__ __forceinline__ double vel(const NsDev &D, const double *__restrict__ U, int f, int64_t i, int64_t j, int64_t k)
{
    const NsField &F = D.f[f];
    int loc = -1;
    if (i < 0) { loc = 0; i = 0; } else if (i >= F.n[0]) { loc = 1; i = F.n[0] - 1; }
    if (j < 0) { loc = 2; j = 0; } else if (j >= F.n[1]) { loc = 3; j = F.n[1] - 1; }
    if (D.dim == 3) {
        if (k < 0) { loc = 4; k = 0; } else if (k >= F.n[2]) { loc = 5; k = F.n[2] - 1; }
    }
    const double t = U[fidx(F, i, j, k)];
    return loc < 0 ? t : F.a0[loc] * t + F.a1[loc];
}

// -N(u) of createconvection.cpp at one velocity point (the caller scales by -1)
__device__ __forceinline__ double convection_at(const NsDev &D, const double *__restrict__ U, int f, int64_t i, int64_t j,
                                                int64_t k)
{
    const NsField &F = D.f[f];
    const bool three = D.dim == 3;
    const double self = vel(D, U, f, i, j, k);
    const double W = (self + vel(D, U, f, i - 1, j, k)) / 2.0, E = (self + vel(D, U, f, i + 1, j, k)) / 2.0;
    const double S = (self + vel(D, U, f, i, j - 1, k)) / 2.0, N = (self + vel(D, U, f, i, j + 1, k)) / 2.0;
    double B = 0.0, Fw = 0.0;
    if (three) {
        B = (self + vel(D, U, f, i, j, k - 1)) / 2.0;
        Fw = (self + vel(D, U, f, i, j, k + 1)) / 2.0;
    }
    const double dLx = F.dl[0][i + 1], dLy = F.dl[1][j + 1];
    double r;
    if (f == 0) {
        const double vS = (vel(D, U, 1, i, j - 1, k) + vel(D, U, 1, i + 1, j - 1, k)) / 2.0;
        const double vN = (vel(D, U, 1, i, j, k) + vel(D, U, 1, i + 1, j, k)) / 2.0;
        r = (E * E - W * W) / dLx + (vN * N - vS * S) / dLy;
        if (three) {
            const double wB = (vel(D, U, 2, i, j, k - 1) + vel(D, U, 2, i + 1, j, k - 1)) / 2.0;
            const double wF = (vel(D, U, 2, i, j, k) + vel(D, U, 2, i + 1, j, k)) / 2.0;
            r = r + (wF * Fw - wB * B) / F.dl[2][k + 1];
        }
    } else if (f == 1) {
        const double uW = (vel(D, U, 0, i - 1, j, k) + vel(D, U, 0, i - 1, j + 1, k)) / 2.0;
        const double uE = (vel(D, U, 0, i, j, k) + vel(D, U, 0, i, j + 1, k)) / 2.0;
        r = (uE * E - uW * W) / dLx + (N * N - S * S) / dLy;
        if (three) {
            const double wB = (vel(D, U, 2, i, j, k - 1) + vel(D, U, 2, i, j + 1, k - 1)) / 2.0;
            const double wF = (vel(D, U, 2, i, j, k) + vel(D, U, 2, i, j + 1, k)) / 2.0;
            r = r + (wF * Fw - wB * B) / F.dl[2][k + 1];
        }
    } else {
        const double uW = (vel(D, U, 0, i - 1, j, k) + vel(D, U, 0, i - 1, j, k + 1)) / 2.0;
        const double uE = (vel(D, U, 0, i, j, k) + vel(D, U, 0, i, j, k + 1)) / 2.0;
        const double vS = (vel(D, U, 1, i, j - 1, k) + vel(D, U, 1, i, j - 1, k + 1)) / 2.0;
        const double vN = (vel(D, U, 1, i, j, k) + vel(D, U, 1, i, j, k + 1)) / 2.0;
        r = (uE * E - uW * W) / dLx + (vN * N - vS * S) / dLy + (Fw * Fw - B * B) / F.dl[2][k + 1];
    }
    return r;
}

// (L u)_row in the CSR order of createLaplacian (z-, y-, x-, diag, x+, y+, z+) and the BC correction lc = sum coeff*a1
__device__ __forceinline__ void laplacian_at(const NsDev &D, const double *__restrict__ U, int f, int64_t i, int64_t j,
                                             int64_t k, double *lu, double *lc)
{
    const NsField &F = D.f[f];
    const int64_t ijk[3] = {i, j, k};
    double v[6] = {0, 0, 0, 0, 0, 0};
    bool interior[6] = {false, false, false, false, false, false};
    double acc = 0.0;
    for (int d = 0; d < D.dim; ++d) {
        const int64_t s = ijk[d];
        const double dLSelf = F.dl[d][s + 1];
        const double dLNeg = F.co[d][s + 1] - F.co[d][s];
        const double dLPos = F.co[d][s + 2] - F.co[d][s + 1];
        v[2 * d] = 1.0 / (dLNeg * dLSelf);
        v[2 * d + 1] = 1.0 / (dLPos * dLSelf);
        interior[2 * d] = s > 0;
        interior[2 * d + 1] = s < F.n[d] - 1;
        acc = acc + v[2 * d];
        acc = acc + v[2 * d + 1];
    }
    double diag = -acc, corr = 0.0;
    for (int q = 0; q < 2 * D.dim; ++q)
        if (!interior[q]) {
            const double t = v[q] * F.a0[q];
            if (t != 0.0) diag = diag + t;
            corr = corr + v[q] * F.a1[q];
        }
    const int64_t st[3] = {1, F.n[0], F.n[0] * F.n[1]};
    const int64_t p = fidx(F, i, j, k);
    double s = 0.0;
    for (int d = D.dim - 1; d >= 0; --d)
        if (interior[2 * d]) s = s + v[2 * d] * U[p - st[d]];
    s = s + diag * U[p];
    for (int d = 0; d < D.dim; ++d)
        if (interior[2 * d + 1]) s = s + v[2 * d + 1] * U[p + st[d]];
    *lu = s;
    *lc = corr;
}

// rhs1 and the new convective term (navierstokes.cpp:432-521), one velocity point per lane
__global__ __launch_bounds__(256) void k_ns_rhs_velocity(NsDev D, double dt, double nu, double c0, double c1, double d0,
                                                         double cimpl, const double *__restrict__ U,
                                                         const double *__restrict__ p, const double *__restrict__ conv1,
                                                         double *__restrict__ conv0, double *__restrict__ rhs1)
{
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < D.UN; g += (int64_t)gridDim.x * 256) {
        int f = 0;
        if (D.dim > 1 && g >= D.f[1].off) f = 1;
        if (D.dim > 2 && g >= D.f[2].off) f = 2;
        const NsField &F = D.f[f];
        const int64_t q = g - F.off;
        const int64_t i = q % F.n[0], j = (q / F.n[0]) % F.n[1], k = q / (F.n[0] * F.n[1]);
        // G p: row {-1/dL at the cell, +1/dL at the + neighbour}, dL = dL[f][f][idx]
        const int64_t ijk[3] = {i, j, k};
        const double gv = 1.0 / F.dl[f][ijk[f] + 1];
        const int64_t pst[3] = {1, D.pn[0], D.pn[0] * D.pn[1]};
        const int64_t pc = i + D.pn[0] * (j + D.pn[1] * k);
        double r = 0.0 + (-gv) * p[pc];
        r = r + gv * p[pc + pst[f]];
        r = -1.0 * r;
        r = r + (1.0 / dt) * U[g];
        const double cn = -1.0 * convection_at(D, U, f, i, j, k);
        conv0[g] = cn;
        r = r + c0 * cn;
        r = r + c1 * conv1[g];
        double lu, lc;
        laplacian_at(D, U, f, i, j, k, &lu, &lc);
        double df = lu + lc;
        df = nu * df;
        r = r + d0 * df;
        const double b1 = nu * lc;
        r = r + cimpl * b1;
        rhs1[g] = r;
    }
}

// rhs2 = D u + Dbc (navierstokes.cpp:540-563); D row in packed-column order u(i-1), u(i), v(j-1), v(j), w(k-1), w(k)
__global__ __launch_bounds__(256) void k_ns_rhs_poisson(NsDev D, int pinned, const double *__restrict__ U,
                                                        double *__restrict__ rhs2)
{
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < D.pN; c += (int64_t)gridDim.x * 256) {
        const int64_t i = c % D.pn[0], j = (c / D.pn[0]) % D.pn[1], k = c / (D.pn[0] * D.pn[1]);
        const int64_t ijk[3] = {i, j, k};
        const double wx = D.pw[0][i], wy = D.pw[1][j], wz = (D.dim == 3) ? D.pw[2][k] : 1.0;
        const double area[3] = {wy * wz, wx * wz, wx * wy};
        double s = 0.0, corr = 0.0;
        for (int f = 0; f < D.dim; ++f) {
            const NsField &F = D.f[f];
            const int64_t st = (f == 0) ? 1 : (f == 1 ? F.n[0] : F.n[0] * F.n[1]);
            const int64_t s_ = ijk[f];
            // velocity point with the cell's own index = the + face; index s-1 = the - face
            int64_t fi[3] = {i, j, k};
            const bool has_m = s_ > 0, has_p = s_ < F.n[f];
            fi[f] = has_p ? s_ : s_ - 1;
            const int64_t base = fidx(F, fi[0], fi[1], fi[2]);
            double vm = -area[f], vp = area[f];
            if (!has_m) {  // ghost - face folds onto the + face (its target): D[row,target] += coeff*a0
                const double t = (-area[f]) * F.a0[2 * f];
                if (t != 0.0) vp = vp + t;
                corr = corr + (-area[f]) * F.a1[2 * f];
            }
            if (!has_p) {
                const double t = area[f] * F.a0[2 * f + 1];
                if (t != 0.0) vm = vm + t;
            }
            if (has_m) s = s + vm * U[has_p ? base - st : base];
            if (has_p) s = s + vp * U[base];
            if (!has_p) corr = corr + area[f] * F.a1[2 * f + 1];
        }
        double r = s + corr;
        if (pinned && c == 0) r = 0.0;
        rhs2[c] = r;
    }
}

// u = u - BNG dP (BNG = dt*G) ; p = p + dP
__global__ __launch_bounds__(256) void k_ns_project(NsDev D, double dt, const double *__restrict__ dP, double *__restrict__ U,
                                                    double *__restrict__ p)
{
    const int64_t total = D.UN > D.pN ? D.UN : D.pN;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        if (g < D.UN) {
            int f = 0;
            if (D.dim > 1 && g >= D.f[1].off) f = 1;
            if (D.dim > 2 && g >= D.f[2].off) f = 2;
            const NsField &F = D.f[f];
            const int64_t q = g - F.off;
            const int64_t i = q % F.n[0], j = (q / F.n[0]) % F.n[1], k = q / (F.n[0] * F.n[1]);
            const int64_t ijk[3] = {i, j, k};
            const double gv = 1.0 / F.dl[f][ijk[f] + 1];
            const int64_t pst[3] = {1, D.pn[0], D.pn[0] * D.pn[1]};
            const int64_t pc = i + D.pn[0] * (j + D.pn[1] * k);
            double r = 0.0 + (dt * (-gv)) * dP[pc];
            r = r + (dt * gv) * dP[pc + pst[f]];
            U[g] = U[g] + (-1.0) * r;
        }
        if (g < D.pN) p[g] = p[g] + 1.0 * dP[g];
    }
}

}  // namespace pib

struct pib_ns {
    pib::NsDev D;
    int device = 0;
    hipStream_t stream = nullptr;
    pib_solver *vsol = nullptr, *psol = nullptr;
    double dt = 0, nu = 0;
    double *U = nullptr, *p = nullptr, *dP = nullptr, *rhs1 = nullptr, *rhs2 = nullptr, *conv[2] = {nullptr, nullptr};
    std::vector<double *> owned;
    int pinned = 0;
    int v_iters = 0, p_iters = 0;
    double v_res = 0, p_res = 0;
    int64_t steps = 0;
};

extern "C" {

int pib_ns_destroy(pib_ns *ns)
{
    if (ns == nullptr) return 0;
    (void)hipSetDevice(ns->device);
    if (ns->vsol) pib_destroy(ns->vsol);
    if (ns->psol) pib_destroy(ns->psol);
    for (double *q : ns->owned) (void)hipFree(q);
    if (ns->stream) (void)hipStreamDestroy(ns->stream);
    delete ns;
    return 0;
}

int pib_ns_create(pib_ns **out, int dim, const int64_t n[3], const double *wx, const double *wy, const double *wz,
                  const double lo[3], const double hi[3], const int bc_type[18], const double bc_value[18], double dt,
                  double nu, const char *velocity_cfg, const char *poisson_cfg, int device)
{
    using namespace pib;
    if (out == nullptr || n == nullptr || lo == nullptr || hi == nullptr || bc_type == nullptr || bc_value == nullptr)
        return fail(PIB_ERR_ARG_NULL, "pib_ns_create: null argument");
    *out = nullptr;
    if (dim != 2 && dim != 3) return fail(PIB_ERR_ARG_OUTOFRANGE, "pib_ns_create: dim must be 2 or 3");
    const double *w[3] = {wx, wy, wz};
    std::vector<double> hdl[3][3], hco[3][3];
    int64_t fn[3][3];
    velocity_mesh_arrays(dim, n, w, lo, hi, hdl, hco, fn);
    pib_ns *ns = new pib_ns();
    ns->dt = dt;
    ns->nu = nu;
    int err = 0;
    auto bail = [&](int e) {
        pib_ns_destroy(ns);
        return e;
    };
    // the two linear solvers (same device)
    if ((err = pib_create_from_string(&ns->vsol, "velocity", velocity_cfg, 0, 1, nullptr, device))) return bail(err);
    if ((err = pib_create_from_string(&ns->psol, "poisson", poisson_cfg, 0, 1, nullptr, device))) return bail(err);
    ns->device = ns->vsol->device;
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamCreateWithFlags(&ns->stream, hipStreamNonBlocking));
    // ghost-equation tables: ghost = a0*target + a1
    double a0[18], a1[18];
    for (int f = 0; f < 3; ++f)
        for (int loc = 0; loc < 6; ++loc) {
            a0[6 * f + loc] = a1[6 * f + loc] = 0.0;
            if (f >= dim || loc >= 2 * dim) continue;
            const int t = bc_type[6 * f + loc];
            const double v = bc_value[6 * f + loc];
            const int axis = loc / 2;
            if (t == 0) {  // DIRICHLET (singleboundarydirichlet.cpp:35-44)
                a0[6 * f + loc] = (axis == f) ? 0.0 : -1.0;
                a1[6 * f + loc] = (axis == f) ? v : 2.0 * v;
            } else if (t == 1 && axis == f) {
                // a Neumann condition on the NORMAL component folds a0 = 1 into D (createdivergence.cpp:231-242) and
                // hence into DBNG; pib_assemble_poisson builds the a0 = 0 operator only
                return bail(fail(PIB_ERR_SUP, "pib_ns_create: NEUMANN on the normal velocity component (field %d, boundary %d) "
                                              "is not supported by the on-device Poisson assembly", f, loc));
            } else if (t == 1) {  // NEUMANN (singleboundaryneumann.cpp:27-28; dL = ghost-target distance, misc.cpp:187-190)
                const std::vector<double> &c = hco[f][axis];
                const int64_t nf = fn[f][axis];
                const double d = (loc % 2 == 1) ? c[(size_t)nf + 1] - c[(size_t)nf] : c[1] - c[0];
                a0[6 * f + loc] = 1.0;
                a1[6 * f + loc] = ((loc % 2 == 1) ? 1.0 : -1.0) * d * v;
            } else {
                return bail(fail(PIB_ERR_SUP, "pib_ns_create: boundary type %d is not supported (0 DIRICHLET, 1 NEUMANN)", t));
            }
        }
    // matrices: A = I/dt - c nu L (CN: c = 1/2), DBNG; null-space convention from the Poisson solver's flavour
    // (navierstokes.cpp:395-429)
    if ((err = pib_assemble_velocity(ns->vsol, dim, n, wx, wy, wz, lo, hi, a0, dt, 0.5 * nu))) return bail(err);
    char tbuf[64];
    pib_get_type(ns->psol, tbuf, sizeof tbuf);
    ns->pinned = (std::strcmp(tbuf, "NVIDIA AmgX") == 0) ? 1 : 0;
    if ((err = pib_assemble_poisson(ns->psol, dim, n, wx, wy, wz, dt, ns->pinned ? PIB_NULLSPACE_PINNED : PIB_NULLSPACE_CONSTANT)))
        return bail(err);
    // device mesh arrays
    NsDev &D = ns->D;
    D.dim = dim;
    int64_t off = 0;
    for (int f = 0; f < 3; ++f) {
        NsField &F = D.f[f];
        for (int d = 0; d < 3; ++d) {
            F.n[d] = fn[f][d];
            F.dl[d] = F.co[d] = nullptr;
            if (f < dim && d < dim) {
                double *p1 = nullptr, *p2 = nullptr;
                if ((err = upload_vec(hdl[f][d], &p1)) || (err = upload_vec(hco[f][d], &p2))) return bail(err);
                ns->owned.push_back(p1);
                ns->owned.push_back(p2);
                F.dl[d] = p1;
                F.co[d] = p2;
            }
        }
        F.off = off;
        if (f < dim) off += fn[f][0] * fn[f][1] * fn[f][2];
        for (int q = 0; q < 6; ++q) {
            F.a0[q] = a0[6 * f + q];
            F.a1[q] = a1[6 * f + q];
        }
    }
    D.UN = off;
    D.pN = 1;
    for (int d = 0; d < 3; ++d) {
        D.pn[d] = (d < dim) ? n[d] : 1;
        D.pN *= D.pn[d];
        D.pw[d] = nullptr;
        if (d < dim) {
            std::vector<double> hw(w[d], w[d] + n[d]);
            double *p1 = nullptr;
            if ((err = upload_vec(hw, &p1))) return bail(err);
            ns->owned.push_back(p1);
            D.pw[d] = p1;
        }
    }
    auto alloc = [&](double **q, int64_t cnt) -> int {
        PIB_HIP(hipMalloc(q, sizeof(double) * (size_t)cnt));
        PIB_HIP(hipMemset(*q, 0, sizeof(double) * (size_t)cnt));
        ns->owned.push_back(*q);
        return 0;
    };
    if ((err = alloc(&ns->U, D.UN)) || (err = alloc(&ns->rhs1, D.UN)) || (err = alloc(&ns->conv[0], D.UN)) ||
        (err = alloc(&ns->conv[1], D.UN)) || (err = alloc(&ns->p, D.pN)) || (err = alloc(&ns->dP, D.pN)) ||
        (err = alloc(&ns->rhs2, D.pN)))
        return bail(err);
    *out = ns;
    return 0;
}

int pib_ns_sizes(pib_ns *ns, int64_t *UN, int64_t *pN)
{
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    if (UN) *UN = ns->D.UN;
    if (pN) *pN = ns->D.pN;
    return 0;
}

int pib_ns_set_state(pib_ns *ns, const double *U, const double *p)
{
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    if (U) PIB_HIP(hipMemcpy(ns->U, U, sizeof(double) * (size_t)ns->D.UN, hipMemcpyHostToDevice));
    if (p) PIB_HIP(hipMemcpy(ns->p, p, sizeof(double) * (size_t)ns->D.pN, hipMemcpyHostToDevice));
    return 0;
}

int pib_ns_get_state(pib_ns *ns, double *U, double *p, double *rhs1, double *rhs2)
{
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    PIB_HIP(hipStreamSynchronize(ns->stream));
    if (U) PIB_HIP(hipMemcpy(U, ns->U, sizeof(double) * (size_t)ns->D.UN, hipMemcpyDeviceToHost));
    if (p) PIB_HIP(hipMemcpy(p, ns->p, sizeof(double) * (size_t)ns->D.pN, hipMemcpyDeviceToHost));
    if (rhs1) PIB_HIP(hipMemcpy(rhs1, ns->rhs1, sizeof(double) * (size_t)ns->D.UN, hipMemcpyDeviceToHost));
    if (rhs2) PIB_HIP(hipMemcpy(rhs2, ns->rhs2, sizeof(double) * (size_t)ns->D.pN, hipMemcpyDeviceToHost));
    return 0;
}

int pib_ns_advance(pib_ns *ns, int nsteps)
{
    using namespace pib;
    if (ns == nullptr) return fail(PIB_ERR_ARG_NULL, "null engine");
    PIB_HIP(hipSetDevice(ns->device));
    const NsDev &D = ns->D;
    const int gu = (int)std::min<int64_t>(4096, (D.UN + 255) / 256), gp = (int)std::min<int64_t>(4096, (D.pN + 255) / 256);
    const int gt = gu > gp ? gu : gp;
    for (int it = 0; it < nsteps; ++it) {
        // VecSwap chain of the convective terms (navierstokes.cpp:452-458)
        std::swap(ns->conv[0], ns->conv[1]);
        hipLaunchKernelGGL(k_ns_rhs_velocity, dim3(gu), dim3(256), 0, ns->stream, D, ns->dt, ns->nu, 1.5, -0.5, 0.5, 0.5, ns->U,
                           ns->p, ns->conv[1], ns->conv[0], ns->rhs1);
        PIB_HIP(hipGetLastError());
        PIB_HIP(hipStreamSynchronize(ns->stream));
        PIB_CHK(pib_solve(ns->vsol, ns->U, ns->rhs1));  // vSolver->solve(UGlobal, rhs1)  (:532)
        hipLaunchKernelGGL(k_ns_rhs_poisson, dim3(gp), dim3(256), 0, ns->stream, D, ns->pinned, ns->U, ns->rhs2);
        PIB_HIP(hipGetLastError());
        PIB_HIP(hipStreamSynchronize(ns->stream));
        PIB_CHK(pib_solve(ns->psol, ns->dP, ns->rhs2));  // pSolver->solve(dP, rhs2)      (:575)
        hipLaunchKernelGGL(k_ns_project, dim3(gt), dim3(256), 0, ns->stream, D, ns->dt, ns->dP, ns->U, ns->p);
        PIB_HIP(hipGetLastError());
        ns->steps++;
    }
    PIB_HIP(hipStreamSynchronize(ns->stream));
    pib_get_iters(ns->vsol, &ns->v_iters);
    pib_get_residual(ns->vsol, &ns->v_res);
    pib_get_iters(ns->psol, &ns->p_iters);
    pib_get_residual(ns->psol, &ns->p_res);
    return 0;
}

/* the columns of iterations-<start>.txt (navierstokes.cpp:766-794): vIters, vRes, pIters, pRes of the last step */
int pib_ns_get_solver_info(pib_ns *ns, int *v_iters, double *v_res, int *p_iters, double *p_res)
{
    if (ns == nullptr) return pib::fail(PIB_ERR_ARG_NULL, "null engine");
    if (v_iters) *v_iters = ns->v_iters;
    if (v_res) *v_res = ns->v_res;
    if (p_iters) *p_iters = ns->p_iters;
    if (p_res) *p_res = ns->p_res;
    return 0;
}

}  // extern "C"
