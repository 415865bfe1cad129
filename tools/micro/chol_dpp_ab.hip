// WaveExT::chol_inv14 (the 14 x 14 register Cholesky + inverse of K4's factorisation chain) in isolation: correctness against a host
// Cholesky, a bitwise checksum (the variants must agree bit for bit) and cycles per call.  Build once per variant:
//   for v in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSCVX_CHOL_DPP=$v -Iinclude -Isuccessiveconvexification_amd/csrc \
//       -o /tmp/chol_dpp_$v tools/micro/chol_dpp_ab.hip && /tmp/chol_dpp_$v; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "scvx_socp.hpp"

__global__ __launch_bounds__(64) void chol_kernel(const double* Min, double* Lout, unsigned long long* cyc, int* okout, int reps) {
    __shared__ double M[196], Li[196];
    const double* src = Min + (size_t)blockIdx.x * 196;
    for (int e = threadIdx.x; e < 196; e += 64) { M[e] = src[e]; Li[e] = 0.0; }
    __syncthreads();
    scvx::WaveExT<3> ex;
    bool ok = true;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        ok = ex.chol_inv14(M, Li) && ok;
        ex.sync_lds();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int e = threadIdx.x; e < 196; e += 64) Lout[(size_t)blockIdx.x * 196 + e] = Li[e];
    if (threadIdx.x == 0) { cyc[blockIdx.x] = (t1 - t0) / reps; okout[blockIdx.x] = ok ? 1 : 0; }
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 2048, reps = 64;
    std::vector<double> M((size_t)NB * 196), L((size_t)NB * 196);
    srand(12345);
    for (int b = 0; b < NB; b++) {
        double A[14][14];
        for (int i = 0; i < 14; i++) for (int j = 0; j < 14; j++) A[i][j] = (double)rand() / RAND_MAX - 0.5;
        const double sc = pow(10.0, (b % 9) - 4);   // scales 1e-4 ... 1e4
        for (int i = 0; i < 14; i++) for (int j = 0; j < 14; j++) {
            double s = (i == j) ? 0.05 : 0.0;
            for (int k = 0; k < 14; k++) s += A[i][k] * A[j][k];
            M[(size_t)b * 196 + 14 * i + j] = s * sc;
        }
    }
    double *dM, *dL; unsigned long long* dc; int* dok;
    (void)hipMalloc((void**)&dM, M.size() * 8); (void)hipMalloc((void**)&dL, L.size() * 8); (void)hipMalloc((void**)&dc, NB * 8); (void)hipMalloc((void**)&dok, NB * 4);
    (void)hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chol_kernel, dim3(NB), dim3(64), 0, 0, dM, dL, dc, dok, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned long long> cyc(NB); std::vector<int> okv(NB);
    (void)hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(cyc.data(), dc, NB * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(okv.data(), dok, NB * 4, hipMemcpyDeviceToHost);
    // host check: Li M Li' = I
    double worst = 0; int bad = 0;
    for (int b = 0; b < NB; b++) {
        const double* m = &M[(size_t)b * 196]; const double* li = &L[(size_t)b * 196];
        double T[14][14];
        for (int i = 0; i < 14; i++) for (int j = 0; j < 14; j++) { double s = 0; for (int k = 0; k < 14; k++) s += li[14 * i + k] * m[14 * k + j]; T[i][j] = s; }
        for (int i = 0; i < 14; i++) for (int j = 0; j < 14; j++) {
            double s = 0; for (int k = 0; k < 14; k++) s += T[i][k] * li[14 * j + k];
            worst = fmax(worst, fabs(s - (i == j ? 1.0 : 0.0)));
            if (j > i && li[14 * i + j] != 0.0) bad++;
        }
        if (!okv[b]) bad++;
    }
    unsigned long long h = 1469598103934665603ull;
    for (size_t e = 0; e < L.size(); e++) { unsigned long long u; memcpy(&u, &L[e], 8); h = (h ^ u) * 1099511628211ull; }
    unsigned long long cs = 0; for (int b = 0; b < NB; b++) cs += cyc[b];
    // one wavefront alone on its SIMD: blocks 0 .. of a launch this size share SIMDs; report the mean and the minimum
    unsigned long long cmin = cyc[0]; for (int b = 0; b < NB; b++) cmin = cyc[b] < cmin ? cyc[b] : cmin;
    printf("SCVX_CHOL_DPP=%d  blocks %d  |Li M Li' - I|max %.2e  bad %d  checksum %016llx  s_memtime ticks per call: mean %.0f min %llu\n",
           SCVX_CHOL_DPP, NB, worst, bad, h, (double)cs / NB, cmin);
    return bad ? 1 : 0;
}
