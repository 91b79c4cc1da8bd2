// config.cpp -- solver configuration files.
//
// Two syntaxes, auto-detected, both restricted to the subset every reference
// example uses:
//   * AmgX "legacy" key=value files read by AmgXSolver::initialize
//     (src/linsolver/linsolveramgx.cpp:62-72), e.g.
//     examples/navierstokes/liddrivencavity2dRe1000_GPU/config/poisson_solver.info
//         solver(solv)=PCG ; solv:max_iters=1000 ; solv:preconditioner(prec)=AMG ;
//         prec:smoother(smooth)=BLOCK_JACOBI ; smooth:relaxation_factor=0.9
//     A key looked up in a scope falls back to the "default" scope, then to
//     the AmgX built-in default.
//   * PETSc options files inserted by LinSolverKSP::init with prefix -<name>_
//     (src/linsolver/linsolverksp.cpp:56-66), e.g.
//     examples/navierstokes/liddrivencavity2dRe100/config/poisson_solver.info
//         -poisson_ksp_type cg ; -poisson_ksp_atol 1.0E-06 ; -poisson_pc_type gamg
// AMG-type preconditioners (AmgX AMG, PCGAMG, hypre) map onto this library's
// geometric multigrid (needs pib_set_grid_hint / pib_assemble_poisson).
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <new>
#include <sstream>
#include <stdexcept>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include "pib_internal.hpp"

namespace pib {

static thread_local char g_err[1024] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char *last_error() { return g_err; }

// PIB_CRASH_BACKTRACE=1 (tests/conftest.py sets it): on SIGSEGV / SIGABRT / SIGBUS the faulting thread's NATIVE frames go to stderr
// before the handler that was installed before ours runs (Python's faulthandler prints the Python frames only -- which is how
// the intermittent crash inside pib_destroy went unexplained for a round).  Diagnostics: backtrace() is not async-signal-safe
// to the letter, the process is dying anyway.
static struct sigaction g_prev_action[3];
static const int g_crash_signals[3] = {SIGSEGV, SIGABRT, SIGBUS};
static void crash_backtrace(int sig, siginfo_t *info, void *ctx)
{
    static const char head[] = "\n[petibm_amd] fatal signal: native frames of the faulting thread\n";
    if (write(2, head, sizeof(head) - 1) < 0) {}
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    for (int k = 0; k < 3; ++k) {
        if (g_crash_signals[k] != sig) continue;
        const struct sigaction &prev = g_prev_action[k];
        sigaction(sig, &prev, nullptr);
        if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction != nullptr) {
            prev.sa_sigaction(sig, info, ctx);
            return;
        }
        if (prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler != nullptr) {
            prev.sa_handler(sig);
            return;
        }
    }
    raise(sig);  // (default action restored above)
}
void install_crash_backtrace()
{
    static std::once_flag once;
    std::call_once(once, []() {
        const char *e = std::getenv("PIB_CRASH_BACKTRACE");
        if (e == nullptr || std::atoi(e) == 0) return;
        void *warm[4];
        (void)backtrace(warm, 4);  // (loads the unwinder now, not inside the handler)
        for (int k = 0; k < 3; ++k) {
            struct sigaction sa;
            std::memset(&sa, 0, sizeof sa);
            sa.sa_sigaction = crash_backtrace;
            sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
            sigemptyset(&sa.sa_mask);
            sigaction(g_crash_signals[k], &sa, &g_prev_action[k]);
        }
    });
}

// The C ABI's catch-all (every extern "C" entry point is a function-try-block ending in this): an exception must not unwind into
// a C / ctypes / cgo caller -- that is std::terminate, i.e. the host application aborted by its linear solver.
int fail_exception(const char *where) noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(PIB_ERR_MEM, "%s: out of host memory (std::bad_alloc)", where);
    } catch (const std::exception &e) {
        return fail(PIB_ERR_LIB, "%s: C++ exception: %s", where, e.what());
    } catch (...) {
        return fail(PIB_ERR_LIB, "%s: unknown C++ exception", where);
    }
}

static std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
static std::string upper(std::string s)
{
    for (auto &c : s) c = (char)std::toupper((unsigned char)c);
    return s;
}
static std::string lower(std::string s)
{
    for (auto &c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}
static bool truthy(const std::string &v)
{
    std::string u = upper(trim(v));
    return !(u == "0" || u == "FALSE" || u == "NO" || u == "OFF" || u.empty());
}

// ---------------------------------------------------------------- AmgX syntax
struct AmgxDoc {
    std::map<std::string, std::map<std::string, std::string>> scopes;  // scope -> key -> value
    std::map<std::string, std::string> child;                          // "scope:key" -> new scope name
    bool has(const std::string &scope, const std::string &key) const
    {
        auto it = scopes.find(scope);
        if (it != scopes.end() && it->second.count(key)) return true;
        auto d = scopes.find("default");
        return d != scopes.end() && d->second.count(key);
    }
    std::string get(const std::string &scope, const std::string &key, const std::string &dflt) const
    {
        auto it = scopes.find(scope);
        if (it != scopes.end()) {
            auto k = it->second.find(key);
            if (k != it->second.end()) return k->second;
        }
        auto d = scopes.find("default");
        if (d != scopes.end()) {
            auto k = d->second.find(key);
            if (k != d->second.end()) return k->second;
        }
        return dflt;
    }
    std::string child_scope(const std::string &scope, const std::string &key) const
    {
        auto it = child.find(scope + ":" + key);
        return it == child.end() ? std::string("default") : it->second;
    }
};

static int parse_amgx(const std::string &text, AmgxDoc &doc)
{
    std::string norm = text;
    std::replace(norm.begin(), norm.end(), ',', '\n');
    std::replace(norm.begin(), norm.end(), ';', '\n');
    std::istringstream in(norm);
    std::string line;
    while (std::getline(in, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        line = trim(line);
        if (line.empty()) continue;
        size_t eq = line.find('=');
        if (eq == std::string::npos) return fail(PIB_ERR_ARG_WRONG, "config: cannot parse line \"%s\"", line.c_str());
        std::string lhs = trim(line.substr(0, eq)), rhs = trim(line.substr(eq + 1));
        std::string scope = "default", key = lhs, newscope;
        size_t colon = lhs.find(':');
        if (colon != std::string::npos) {
            scope = trim(lhs.substr(0, colon));
            key = trim(lhs.substr(colon + 1));
        }
        size_t lp = key.find('(');
        if (lp != std::string::npos) {
            size_t rp = key.find(')', lp);
            if (rp == std::string::npos) return fail(PIB_ERR_ARG_WRONG, "config: unbalanced '(' in \"%s\"", line.c_str());
            newscope = trim(key.substr(lp + 1, rp - lp - 1));
            key = trim(key.substr(0, lp));
            doc.child[scope + ":" + key] = newscope;
        }
        doc.scopes[scope][key] = rhs;
    }
    return 0;
}

static int apply_amgx(const AmgxDoc &d, Config &c)
{
    c.flavor = Flavor::AMGX;
    c.norm = NormType::UNPRECONDITIONED;  // norm=L2 of the true residual
    c.initial_guess_nonzero = true;
    c.dtol = 0.0;                         // AmgX has no divergence tolerance
    // AmgX never reports non-convergence to PetIBM (linsolveramgx.cpp:90-99);
    // this library defaults to an error unless the key says otherwise.
    c.error_if_not_converged = true;

    std::string top = "default";
    std::string solver = upper(d.get("default", "solver", "PCG"));
    std::string ss = d.child_scope("default", "solver");
    if (solver == "PCG" || solver == "CG" || solver == "PCGF")
        c.method = Method::CG;
    else if (solver == "PBICGSTAB" || solver == "BICGSTAB")
        c.method = Method::BICGSTAB;
    else if (solver == "CHEBYSHEV")
        c.method = Method::CHEBYSHEV;
    else if (solver == "DENSE_LU_SOLVER") {
        c.method = Method::PREONLY;
        c.pc = Precond::LU;
        return 0;
    } else
        return fail(PIB_ERR_SUP, "config: solver=%s is not supported (PCG, PBICGSTAB, CHEBYSHEV, DENSE_LU_SOLVER)", solver.c_str());
    (void)top;

    c.max_iters = std::atoi(d.get(ss, "max_iters", "100").c_str());
    double tol = std::atof(d.get(ss, "tolerance", "1e-12").c_str());
    std::string conv = upper(d.get(ss, "convergence", "ABSOLUTE"));
    if (conv == "ABSOLUTE") {
        c.atol = tol;
        c.rtol = 0.0;
    } else if (conv == "RELATIVE_INI" || conv == "RELATIVE_INI_CORE") {
        c.atol = 0.0;
        c.rtol = tol;
    } else {
        return fail(PIB_ERR_SUP, "config: convergence=%s is not supported (ABSOLUTE, RELATIVE_INI[_CORE])", conv.c_str());
    }
    std::string nrm = upper(d.get(ss, "norm", "L2"));
    if (nrm != "L2") return fail(PIB_ERR_SUP, "config: norm=%s is not supported (L2)", nrm.c_str());
    c.monitor_residual = truthy(d.get(ss, "monitor_residual", "0"));
    c.store_res_history = truthy(d.get(ss, "store_res_history", "0"));
    if (d.has(ss, "error_if_not_converged")) c.error_if_not_converged = truthy(d.get(ss, "error_if_not_converged", "1"));

    if (c.method == Method::CHEBYSHEV && (d.has(ss, "cheby_min_lambda") || d.has(ss, "cheby_max_lambda"))) {
        c.cheb_emin = std::atof(d.get(ss, "cheby_min_lambda", "0").c_str());
        c.cheb_emax = std::atof(d.get(ss, "cheby_max_lambda", "0").c_str());
        if (!(c.cheb_emin > 0.0) || !(c.cheb_emax > c.cheb_emin))
            return fail(PIB_ERR_ARG_OUTOFRANGE, "config: solver=CHEBYSHEV needs 0 < cheby_min_lambda < cheby_max_lambda (or neither: Gershgorin bounds)");
    }
    std::string pc = upper(d.get(ss, "preconditioner", "NOSOLVER"));
    std::string ps = d.child_scope(ss, "preconditioner");
    if (pc == "NOSOLVER" || pc == "NONE") {
        c.pc = Precond::NONE;
    } else if (pc == "BLOCK_JACOBI" || pc == "JACOBI" || pc == "JACOBI_L1") {
        c.pc = Precond::JACOBI;
        c.jacobi_relaxation = std::atof(d.get(ps, "relaxation_factor", "0.9").c_str());
    } else if (pc == "AMG" || pc == "GMG") {
        c.pc = Precond::GMG;
        std::string cyc = upper(d.get(ps, "cycle", "V"));
        if (cyc != "V") return fail(PIB_ERR_SUP, "config: cycle=%s is not supported (V)", cyc.c_str());
        c.presweeps = std::atoi(d.get(ps, "presweeps", "1").c_str());
        c.postsweeps = std::atoi(d.get(ps, "postsweeps", "1").c_str());
        c.max_levels = std::atoi(d.get(ps, "max_levels", "100").c_str());
        // min_coarse_rows / dense_lu_num_rows / selector / interpolator ... steer AmgX's algebraic coarsening and have no
        // counterpart in a geometric hierarchy (it always coarsens down to <= 2 cells per direction): accepted, unused
        c.coarsest_sweeps = std::max(1, std::atoi(d.get(ps, "coarsest_sweeps", "2").c_str())) * 16;
        std::string sm = upper(d.get(ps, "smoother", "BLOCK_JACOBI"));
        std::string sms = d.child_scope(ps, "smoother");
        if (sm == "BLOCK_JACOBI" || sm == "JACOBI" || sm == "JACOBI_L1" || sm == "MULTICOLOR_DILU" ||
            sm == "MULTICOLOR_GS")
            c.smoother = Smoother::JACOBI;
        else if (sm == "CHEBYSHEV" || sm == "CHEBYSHEV_POLY")
            c.smoother = Smoother::CHEBYSHEV;
        else
            return fail(PIB_ERR_SUP, "config: smoother=%s is not supported", sm.c_str());
        c.smoother_relaxation = std::atof(d.get(sms, "relaxation_factor", "0.9").c_str());
        c.cheby_degree = std::atoi(d.get(sms, "chebyshev_polynomial_order", "2").c_str());
        c.cheby_lmax = std::atof(d.get(sms, "cheby_max_lambda", "2.0").c_str());
        {
            const double lmin = std::atof(d.get(sms, "cheby_min_lambda", "0.5").c_str());
            if (!(lmin > 0.0) || !(c.cheby_lmax > lmin)) return fail(PIB_ERR_ARG_OUTOFRANGE, "config: need 0 < cheby_min_lambda < cheby_max_lambda");
            c.cheby_ratio = c.cheby_lmax / lmin;
        }
    } else {
        return fail(PIB_ERR_SUP, "config: preconditioner=%s is not supported (NOSOLVER, BLOCK_JACOBI, AMG)", pc.c_str());
    }
    // library-specific keys (default scope)
    c.check_every = std::atoi(d.get("default", "pib_check_every", "0").c_str());
    c.graph_max_rows = std::atoll(d.get("default", "pib_graph_max_rows", "4194304").c_str());
    c.fuse_presmooth = std::atoi(d.get("default", "pib_fuse_presmooth", "1").c_str());
    c.fuse_residual_restrict = std::atoi(d.get("default", "pib_fuse_residual_restrict", "1").c_str());
    c.fuse_down_march = std::atoi(d.get("default", "pib_fuse_down_march", "1").c_str());
    c.fuse_post_pair = std::atoi(d.get("default", "pib_fuse_post_pair", "1").c_str());
    c.fuse_prolong = std::atoi(d.get("default", "pib_fuse_prolong", "1").c_str());
    c.march = std::atoi(d.get("default", "pib_march", "1").c_str());
    c.march_min_cells = std::atoi(d.get("default", "pib_march_min_cells", "12582912").c_str());
    c.matrix_free_velocity = std::atoi(d.get("default", "pib_matrix_free_velocity", "1").c_str());
    c.fuse_velocity_product = std::atoi(d.get("default", "pib_fuse_velocity_product", "1").c_str());
    c.bicgstab_form = std::atoi(d.get("default", "pib_bicgstab_form", "3").c_str());
    c.fuse_residual_update = std::atoi(d.get("default", "pib_fuse_residual_update", "1").c_str());
    c.pin_sum_local = std::atoi(d.get("default", "pib_pin_sum_local", "-1").c_str());
    c.compress_columns = std::atoi(d.get("default", "pib_compress_columns", "2").c_str());
    c.place_update_vector = std::atoi(d.get("default", "pib_place_update_vector", "1").c_str());
    c.place_min_rows = std::atoll(d.get("default", "pib_place_min_rows", "33554432").c_str());
    c.cg_single_reduction = std::atoi(d.get("default", "pib_cg_single_reduction", "0").c_str());
    c.sweep_pairs = std::atoi(d.get("default", "pib_sweep_pairs", "1").c_str());
    c.fuse_chebyshev_update = std::atoi(d.get("default", "pib_fuse_chebyshev_update", "1").c_str());
    c.matrix_free_poisson = std::atoi(d.get("default", "pib_matrix_free_poisson", "-1").c_str());
    c.agglomerate_below = std::atoi(d.get("default", "pib_agglomerate_below", "300000").c_str());
    c.detect_structure = std::atoi(d.get("default", "pib_detect_structure", "1").c_str());
    c.deep_halo = std::atoi(d.get("default", "pib_deep_halo", "2").c_str());
    c.overlap_min_bytes = std::atoi(d.get("default", "pib_overlap_min_bytes", "1048576").c_str());
    c.coarse_tail = std::atoi(d.get("default", "pib_coarse_tail", "-1").c_str());
    c.coarse_tail_lds = std::atoi(d.get("default", "pib_coarse_tail_lds", "1").c_str());
    c.fuse_small_levels = std::atoi(d.get("default", "pib_fuse_small_levels", "1").c_str());
    if (d.has("default", "pib_initial_guess_nonzero"))
        c.initial_guess_nonzero = truthy(d.get("default", "pib_initial_guess_nonzero", "1"));
    if (d.has("default", "pib_norm")) {
        std::string v = upper(d.get("default", "pib_norm", ""));
        c.norm = (v == "PRECONDITIONED") ? NormType::PRECONDITIONED : NormType::UNPRECONDITIONED;
    }
    if (c.max_iters < 0) return fail(PIB_ERR_ARG_OUTOFRANGE, "config: max_iters < 0");
    return 0;
}

// --------------------------------------------------------------- PETSc syntax
static int apply_petsc(const std::string &text, const std::string &name, Config &c)
{
    c.flavor = Flavor::KSP;
    c.method = Method::CG;  // KSPSetType(ksp, KSPCG): linsolverksp.cpp:64
    c.pc = Precond::JACOBI; // PETSc's own default (ILU) is not provided; see INTEGRATION.md
    c.norm = NormType::PRECONDITIONED;
    c.max_iters = 10000;
    c.rtol = 1e-5;
    c.atol = 1e-50;
    c.dtol = 1e4;
    c.monitor_residual = true;
    c.store_res_history = true;
    c.error_if_not_converged = true;  // linsolverksp.cpp:96-104
    c.initial_guess_nonzero = false;
    c.jacobi_relaxation = 1.0;

    std::map<std::string, std::string> opt;
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line = line.substr(0, h);
        std::istringstream ls(line);
        std::string tok, pending;
        while (ls >> tok) {
            if (tok.size() > 1 && tok[0] == '-' && !std::isdigit((unsigned char)tok[1]) && tok[1] != '.') {
                if (!pending.empty()) opt[pending] = "true";
                pending = tok.substr(1);
            } else if (!pending.empty()) {
                opt[pending] = tok;
                pending.clear();
            }
        }
        if (!pending.empty()) opt[pending] = "true";
    }
    const std::string pre = name + "_";
    auto get = [&](const std::string &k, std::string &v) {
        auto it = opt.find(pre + k);
        if (it == opt.end()) return false;
        v = it->second;
        return true;
    };
    std::string v;
    if (get("ksp_type", v)) {
        v = lower(v);
        if (v == "cg") c.method = Method::CG;
        else if (v == "bcgs" || v == "bicg" || v == "bcgsl") c.method = Method::BICGSTAB;
        else if (v == "preonly") c.method = Method::PREONLY;
        else if (v == "chebyshev") c.method = Method::CHEBYSHEV;
        else return fail(PIB_ERR_SUP, "config: -%sksp_type %s is not supported (cg, bcgs, chebyshev, preonly)", pre.c_str(), v.c_str());
    }
    if (get("ksp_chebyshev_eigenvalues", v)) {  // emin,emax of the preconditioned operator (KSPChebyshevSetEigenvalues)
        double lo = 0.0, hi = 0.0;
        if (std::sscanf(v.c_str(), "%lf , %lf", &lo, &hi) != 2 || !(lo > 0.0) || !(hi > lo))
            return fail(PIB_ERR_ARG_OUTOFRANGE, "config: -%sksp_chebyshev_eigenvalues wants emin,emax with 0 < emin < emax", pre.c_str());
        c.cheb_emin = lo;
        c.cheb_emax = hi;
    }
    if (get("ksp_atol", v)) c.atol = std::atof(v.c_str());
    if (get("ksp_rtol", v)) c.rtol = std::atof(v.c_str());
    if (get("ksp_divtol", v)) c.dtol = std::atof(v.c_str());
    if (get("ksp_max_it", v)) c.max_iters = std::atoi(v.c_str());
    if (get("ksp_initial_guess_nonzero", v)) c.initial_guess_nonzero = truthy(v);
    if (get("ksp_norm_type", v)) {
        v = lower(v);
        if (v == "preconditioned") c.norm = NormType::PRECONDITIONED;
        else if (v == "unpreconditioned") c.norm = NormType::UNPRECONDITIONED;
        else return fail(PIB_ERR_SUP, "config: -%sksp_norm_type %s is not supported", pre.c_str(), v.c_str());
    }
    if (get("pc_type", v)) {
        v = lower(v);
        if (v == "none") c.pc = Precond::NONE;
        else if (v == "jacobi") c.pc = Precond::JACOBI;
        else if (v == "gamg" || v == "hypre" || v == "mg" || v == "ml") {
            c.pc = Precond::GMG;
            c.smoother = Smoother::CHEBYSHEV;  // PCGAMG's default level smoother is Chebyshev/Jacobi
            c.presweeps = c.postsweeps = 1;
            c.cheby_degree = 2;
        } else if (v == "lu" || v == "cholesky") {
            c.pc = Precond::LU;  // -pc_factor_mat_solver_type (superlu_dist, mumps ...) selects PETSc's back end only
        } else
            return fail(PIB_ERR_SUP, "config: -%spc_type %s is not supported (none, jacobi, gamg, hypre, lu)", pre.c_str(),
                        v.c_str());
    }
    if ((c.method == Method::PREONLY) != (c.pc == Precond::LU))
        return fail(PIB_ERR_SUP, "config: -%sksp_type preonly and -%spc_type lu go together (direct solve)", pre.c_str(),
                    pre.c_str());
    if (get("pib_check_every", v)) c.check_every = std::atoi(v.c_str());
    if (get("pib_graph_max_rows", v)) c.graph_max_rows = std::atoll(v.c_str());
    if (get("pib_fuse_presmooth", v)) c.fuse_presmooth = std::atoi(v.c_str());
    if (get("pib_fuse_residual_restrict", v)) c.fuse_residual_restrict = std::atoi(v.c_str());
    if (get("pib_fuse_down_march", v)) c.fuse_down_march = std::atoi(v.c_str());
    if (get("pib_fuse_post_pair", v)) c.fuse_post_pair = std::atoi(v.c_str());
    if (get("pib_fuse_prolong", v)) c.fuse_prolong = std::atoi(v.c_str());
    if (get("pib_march", v)) c.march = std::atoi(v.c_str());
    if (get("pib_march_min_cells", v)) c.march_min_cells = std::atoi(v.c_str());
    if (get("pib_matrix_free_velocity", v)) c.matrix_free_velocity = std::atoi(v.c_str());
    if (get("pib_fuse_velocity_product", v)) c.fuse_velocity_product = std::atoi(v.c_str());
    if (get("pib_bicgstab_form", v)) c.bicgstab_form = std::atoi(v.c_str());
    if (get("pib_fuse_residual_update", v)) c.fuse_residual_update = std::atoi(v.c_str());
    if (get("pib_pin_sum_local", v)) c.pin_sum_local = std::atoi(v.c_str());
    if (get("pib_compress_columns", v)) c.compress_columns = std::atoi(v.c_str());
    if (get("pib_place_update_vector", v)) c.place_update_vector = std::atoi(v.c_str());
    if (get("pib_place_min_rows", v)) c.place_min_rows = std::atoll(v.c_str());
    if (get("ksp_cg_single_reduction", v)) c.cg_single_reduction = truthy(v) ? 1 : 0;  // PETSc's own option (KSPCGUseSingleReduction)
    if (get("pib_cg_single_reduction", v)) c.cg_single_reduction = std::atoi(v.c_str());
    if (get("pib_sweep_pairs", v)) c.sweep_pairs = std::atoi(v.c_str());
    if (get("pib_fuse_chebyshev_update", v)) c.fuse_chebyshev_update = std::atoi(v.c_str());
    if (get("pib_matrix_free_poisson", v)) c.matrix_free_poisson = std::atoi(v.c_str());
    if (get("pib_agglomerate_below", v)) c.agglomerate_below = std::atoi(v.c_str());
    if (get("pib_detect_structure", v)) c.detect_structure = std::atoi(v.c_str());
    if (get("pib_deep_halo", v)) c.deep_halo = std::atoi(v.c_str());
    if (get("pib_overlap_min_bytes", v)) c.overlap_min_bytes = std::atoi(v.c_str());
    if (get("pib_coarse_tail", v)) c.coarse_tail = std::atoi(v.c_str());
    if (get("pib_coarse_tail_lds", v)) c.coarse_tail_lds = std::atoi(v.c_str());
    if (get("pib_fuse_small_levels", v)) c.fuse_small_levels = std::atoi(v.c_str());
    if (get("pib_presweeps", v)) c.presweeps = std::atoi(v.c_str());
    if (get("pib_postsweeps", v)) c.postsweeps = std::atoi(v.c_str());
    if (get("pib_cheby_degree", v)) c.cheby_degree = std::atoi(v.c_str());
    if (get("pib_smoother", v)) c.smoother = (upper(v) == "JACOBI") ? Smoother::JACOBI : Smoother::CHEBYSHEV;
    return 0;
}

int parse_config_text(const std::string &text, const std::string &name, Config &cfg)
{
    cfg = Config();
    cfg.raw = text;
    // PETSc options files have lines that start with '-'
    bool petsc = false;
    {
        std::istringstream in(text);
        std::string line;
        while (std::getline(in, line)) {
            std::string t = trim(line);
            if (t.empty() || t[0] == '#') continue;
            petsc = (t[0] == '-');
            break;
        }
    }
    if (petsc) return apply_petsc(text, name, cfg);
    AmgxDoc doc;
    PIB_CHK(parse_amgx(text, doc));
    return apply_amgx(doc, cfg);
}

int parse_config_file(const char *path, const std::string &name, Config &cfg)
{
    if (path == nullptr || std::strcmp(path, "None") == 0 || path[0] == '\0') {
        // LinSolverAmgX::init writes an empty temporary file (linsolveramgx.cpp:62-72):
        // every key takes its AmgX default.
        return parse_config_text("", name, cfg);
    }
    std::ifstream f(path);
    if (!f) return fail(PIB_ERR_FILE_OPEN, "cannot open solver configuration file \"%s\"", path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_config_text(ss.str(), name, cfg);
}

}  // namespace pib

extern "C" const char *pib_last_error(void) { return pib::last_error(); }
