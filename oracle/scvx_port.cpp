// scvx_port.cpp — CPU twin of the device SCvx step (TEST INFRASTRUCTURE; bench.py's cpu_baseline "port").
//
// Compiles the SAME portable interior-point core the HIP kernel uses
// (successiveconvexification_amd/csrc/scvx_ipm_core.hpp) with a one-lane host executor, one OpenMP
// thread per trajectory, together with the oracle's own discretisation (scvx_oracle.c).  It is the
// "C++ fp64 restatement of the same algorithm" BASELINE.md §4 asks to be timed on the host cores, and
// the debug twin of the device kernel.  The independent check of the optimum is oracle/ipm.py.
// The product never loads this library.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>
#include "../successiveconvexification_amd/csrc/scvx_ipm_core.hpp"
#include "../successiveconvexification_amd/csrc/scvx_threedof_core.hpp"

namespace {
struct HostEx {
    double sc[2560];
    int lane() const { return 0; }
    int nlanes() const { return 1; }
    void sync() {}
    void sync_lds() {}
    double sum(double x) { return x; }
    double min(double x) { return x; }
    bool all(bool b) { return b; }
    double* scratch() { return sc; }
    static constexpr int kPrefetchRegs = 0;
    static constexpr int kLanes = 1;
    static constexpr bool kPipelineFactor = false;
    static constexpr bool kTwisted = false;
    static constexpr bool kFusedResidual = true;
    // C(14x14) = (acc ? C : 0) + alpha * A(14 x Kd) B(Kd x 14); element strides: C(i,j) = C[i*sci + j*scj],
    // A(i,k) = A[i*sai + k*sak], B(k,j) = B[k*sbk + j*sbj].  C must not alias A or B.
    void tile_gemm(double* Cm, int sci, int scj, const double* A, int sai, int sak, const double* B, int sbk, int sbj,
                   int Kd, double alpha, bool acc, int nb = 14) {
        for (int i = 0; i < 14; i++)
            for (int j = 0; j < nb; j++) {
                double s = 0;
                for (int k = 0; k < Kd; k++) s += A[i * sai + k * sak] * B[k * sbk + j * sbj];
                double* c = Cm + i * sci + j * scj;
                *c = (acc ? *c : 0.0) + alpha * s;
            }
    }
    // sums of products on one accumulator (the device keeps it in MFMA registers): see WaveExT::acc_mac
    struct Acc { double c[14][14]; };
    void acc_zero(Acc& a) { for (int i = 0; i < 14; i++) for (int j = 0; j < 14; j++) a.c[i][j] = 0.0; }
    void acc_mac(Acc& a, const double* A, int sai, int sak, const double* B, int sbk, int sbj, int Kd, double alpha, int nb = 14) {
        for (int i = 0; i < 14; i++)
            for (int j = 0; j < nb; j++) {
                double s = a.c[i][j];
                for (int k = 0; k < Kd; k++) s += (alpha * A[i * sai + k * sak]) * B[k * sbk + j * sbj];
                a.c[i][j] = s;
            }
    }
    void acc_store(const Acc& a, double* Cm, int sci, int scj, bool add, int nb = 14) {
        for (int i = 0; i < 14; i++)
            for (int j = 0; j < nb; j++) {
                double* c = Cm + i * sci + j * scj;
                *c = (add ? *c : 0.0) + a.c[i][j];
            }
    }
    void acc_store_init(const Acc& a, double* Cm, const double* H, double diag) {
        for (int i = 0; i < 14; i++)
            for (int j = 0; j < 14; j++) Cm[14 * i + j] = (H[14 * i + j] + (i == j ? diag : 0.0)) + a.c[i][j];
    }
    // L^-1 (row-major, lower) of the Cholesky factor of the SPD tile M
    bool chol_inv14(double* M, double* Li) {
        const bool ok = chol14(M);
        for (int c = 0; c < 14; c++)
            for (int i = 0; i < 14; i++) {
                if (i < c) { Li[14 * i + c] = 0.0; continue; }
                double s = (i == c) ? 1.0 : 0.0;
                for (int t = c; t < i; t++) s -= M[14 * i + t] * Li[14 * t + c];
                Li[14 * i + c] = s / M[15 * i];
            }
        return ok;
    }
    // in-place Cholesky of a 14x14 SPD tile (row-major, lower triangle on output)
    // Dynamic regularisation (as in ECOS / QDLDL-style KKT solvers): a pivot below 1e-13 * max diagonal — the Schur
    // complement is numerically semidefinite in the IPM endgame — is clamped to that value; the perturbation is
    // absorbed by the iterative refinement of the Newton solve.  Only a non-finite pivot is a failure.
    bool chol14(double* M) {
        bool ok = true;
        double dmax = 0.0;
        for (int j = 0; j < 14; j++) dmax = M[15 * j] > dmax ? M[15 * j] : dmax;
        const double floor_ = 1e-13 * dmax;
        for (int j = 0; j < 14; j++) {
            double piv = M[15 * j];
            if (!(piv == piv) || !(dmax > 0.0)) ok = false;
            if (!(piv > floor_)) piv = floor_ > 0.0 ? floor_ : 1.0;
            const double ip = 1.0 / std::sqrt(piv);
            M[15 * j] = piv;
            for (int i = j; i < 14; i++) M[14 * i + j] *= ip;
            for (int a = j + 1; a < 14; a++)
                for (int b = j + 1; b <= a; b++) M[14 * a + b] -= M[14 * a + j] * M[14 * b + j];
        }
        return ok;
    }
    template <int NR, class CP, class NP, class GP>
    void chain_n(int K, CP const (&z)[NR], NP N, GP const (&o)[NR], bool reverse) {
        for (int q = 0; q < NR; q++) chain(K, z[q], N, o[q], reverse);
    }
    // out_k = z_k + N_k out_{k-1} (forward) or out_k = z_k + N_{k+1}' out_{k+1} (reverse); 14-vectors; N_k is the negated
    // coupling tile stored transposed (element (i, j) at 14 j + i), as Solver::build_kkt writes it.  T = storage type
    // (double, or float for the f32-storage twin): the running vector is carried in double, as in the device's MFMA
    // accumulators, and rounded once when it is stored.
    template <class T, class TN>
    void chain(int K, const T* z, const TN* N, T* out, bool reverse) {
        double run[14] = {0};
        for (int step = 0; step < K; step++) {
            const int k = reverse ? K - 1 - step : step;
            double nxt[14];
            for (int i = 0; i < 14; i++) {
                double a = z[14 * k + i];
                if (step > 0) {
                    if (!reverse) {
                        const TN* col = N + (size_t)k * 196 + i;
                        for (int j = 0; j < 14; j++) a += (double)col[14 * j] * run[j];
                    } else {
                        const TN* row = N + (size_t)(k + 1) * 196 + 14 * i;
                        for (int j = 0; j < 14; j++) a += (double)row[j] * run[j];
                    }
                }
                nxt[i] = a;
            }
            for (int i = 0; i < 14; i++) { run[i] = nxt[i]; out[14 * k + i] = (T)nxt[i]; }
        }
    }
};
}  // namespace

#if defined(SCVX_COUNTERS)
extern "C" { long long scvx_counters[8] = {0}; }
#endif

extern "C" {

void scvx_port_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

// with_dp: the problem enforces the dynamic-pressure cone (one more cone group in the layout)
size_t scvx_port_work_doubles(int K, int with_dp) {
    scvx::ipm::Layout L;
    L.init(K, with_dp != 0);
    return L.work_doubles();
}
size_t scvx_port_work_doubles_nu(int K, int with_dp, int nu) {
    scvx::ipm::Layout L;
    L.init(K, with_dp != 0, nu);
    return L.work_doubles();
}

}  // extern "C"

// Solve B subproblems.  Layouts as include/scvx.h: xbar [B][K+1][14], ubar [B][K+1][NU], endpoint [B][K][14],
// deriv [B][K][14+2NU+1][14], rk [B], ic [B][6].  Outputs: sol [B][(K+1)*(14+NU)+1] = dx, du, dsigma ; nu [B][K][14];
// info [B][4] = status, iters, merit, pobj.  Stor = storage type of the linearisation and of the solver workspace.
// NU = 3 (the reference's live model) or 5 (fin extension).
template <class Stor, class DStor = Stor, int NU = 3, class FStor = SCVX_FACTOR_T>
static int port_socp(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                     const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                     int nthreads, Stor* work_all = nullptr, const int* warm = nullptr) {
    const int K = C->K;
    scvx::ipm::Layout L;
    L.init(K, C->vmax > 0.0, NU);
    const size_t nw = L.work_doubles();
    constexpr int DSZ = 14 * (14 + 2 * NU + 1);
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
    {
        std::vector<Stor> work(work_all ? 0 : nw);
        std::vector<DStor> D((size_t)K * DSZ);
        HostEx ex;
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            for (size_t i = 0; i < D.size(); i++) D[i] = (DStor)deriv[(size_t)b * K * DSZ + i];
            scvx::ipm::Solver<HostEx, Stor, DStor, NU, FStor> S(ex, *C);
            Stor* wk = work_all ? work_all + (size_t)b * nw : work.data();   // persistent per-trajectory slab, as on the device
            scvx::ipm::Result r = S.solve(xbar + (size_t)b * (K + 1) * 14, ubar + (size_t)b * (K + 1) * NU,
                                          endpoint + (size_t)b * K * 14, D.data(), rk[b], ic + (size_t)b * 6, wk,
                                          warm && warm[b]);
            double* so = sol + (size_t)b * ((K + 1) * (14 + NU) + 1);
            for (int i = 0; i < L.nx + L.nu_; i++) so[i] = S.V[i];
            so[L.nx + L.nu_] = S.V[L.iS];
            for (int i = 0; i < L.ny; i++) nu[(size_t)b * K * 14 + i] = S.V[L.nx + L.nu_ + i];
            info[4 * b + 0] = r.status; info[4 * b + 1] = r.iters; info[4 * b + 2] = r.merit; info[4 * b + 3] = r.pobj;
        }
    }
    return 0;
}

extern "C" {

int scvx_port_socp(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                   const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                   int nthreads) {
    return port_socp<double>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads);
}
// persistent workspace [B][scvx_port_work_doubles(K)] + per-trajectory warm flags: the device's warm start of the solve that
// follows a rejected step
int scvx_port_socp_ws(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                      const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                      int nthreads, double* work, const int* warm) {
    return port_socp<double>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads, work, warm);
}
// f32 linearisation (scvx_batch_set_linearization_f32): D rounded to float, workspace and arithmetic double
int scvx_port_socp_lin32(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                         const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                         int nthreads) {
    return port_socp<double, float>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads);
}
// float FACTOR only (packed L^-1 and coupling tiles; everything else double): the preconditioner experiment of VERDICT r3 item 1c
int scvx_port_socp_fac32(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                         const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                         int nthreads, double* work, const int* warm) {
    return port_socp<double, double, 3, float>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads, work, warm);
}
// ... and with the factor in double (the form of rounds 1-3), for the A/B of profiles/r04_factor_f32.md
int scvx_port_socp_fac64(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                         const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                         int nthreads, double* work, const int* warm) {
    return port_socp<double, double, 3, double>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads, work, warm);
}
// fin extension (control_dim = 5), optional persistent workspace + warm flags (work == NULL: per-call scratch)
int scvx_port_socp_fin(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                       const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                       int nthreads, double* work, const int* warm) {
    return port_socp<double, double, 5>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads, work, warm);
}
// f32 storage: the linearisation and the whole solver workspace are float, arithmetic stays double
int scvx_port_socp_f32(const scvx::ipm::Consts* C, int B, const double* xbar, const double* ubar, const double* endpoint,
                       const double* deriv, const double* rk, const double* ic, double* sol, double* nu, double* info,
                       int nthreads) {
    return port_socp<float, float, 3, float>(C, B, xbar, ubar, endpoint, deriv, rk, ic, sol, nu, info, nthreads);
}
}

// ------------------------------------------------------------------------------------------------------------------
// CPU twin of K0, the 3-DoF initialiser (scvx_threedof_core.hpp with a one-lane executor).  The independent check
// of its optimum is oracle/threedof.py on oracle/ipm.py.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct HostEx3 {
    std::vector<double> scratch;
    explicit HostEx3(int N) : scratch(scvx::td::fast_doubles(N)) {}
    int lane() const { return 0; }
    int nlanes() const { return 1; }
    void sync() {}
    void sync_lds() {}
    double sum(double x) { return x; }
    double min(double x) { return x; }
    double* fast() { return scratch.data(); }
    static constexpr int kLanes = 1;
    static constexpr bool kRegisterSweep = false;
};
}  // namespace

extern "C" {
// P: {K, alpha, tf_guess, mwet, mdry, g, Tmin, Tmax, thetaMax, gammaGs}; ic [B][6]; out [B][(K+1)*15+1] (per node
// r v ma T ga kaR ar, then nkaR); info [B][6] = status, iters, pobj, gap, pres, dres
int scvx_port_threedof(const scvx::td::Problem3* P, int B, const double* ic, double* out, double* info, double tol,
                       int max_iter, int refine, double delta, int nthreads) {
    scvx::td::HostTables H;
    if (const char* e = scvx::td::build_tables(*P, tol, max_iter, refine, delta, H)) {
        fprintf(stderr, "scvx_port_threedof: %s\n", e);
        return -1;
    }
    scvx::td::Layout L;
    L.init(P->K);
    const int no = scvx::td::out_doubles(P->K);
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
    {
        std::vector<double> work(L.total);
        HostEx3 ex(P->K);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            scvx::td::Solver<HostEx3> S(ex, H.t, work.data());
            const scvx::td::Result r = S.solve(ic + (size_t)b * 6, out + (size_t)b * no);
            double* o = info + (size_t)b * 6;
            o[0] = r.status; o[1] = r.iters; o[2] = r.pobj; o[3] = r.gap; o[4] = r.pres; o[5] = r.dres;
        }
    }
    return 0;
}
}
