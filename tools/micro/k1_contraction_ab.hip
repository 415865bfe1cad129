// north_star's condition for K1, as a measurement: "MFMA used only for the batched small dense STM x Jacobian contractions where
// rocprof shows it beating the vector path".  The contraction is  dS = F'(x) S  for the 14 x 21 sensitivity block of a segment, once
// per RK stage (40 times per segment at npts = 10).  Two kernels do exactly that product, repeatedly, on resident data:
//   sparse_vector : the form K1 ships -- one lane per sensitivity column (21 lanes per segment, 3 segments per wavefront), the
//                   column's 14 entries in registers, the ~51 structural non-zeros of F' (exo model: dr = sigma v; dv = Am c_m + Dq c_q;
//                   dq = 1/2 (Omega(w) c_q + Q(q) c_w); dw = Mw c_w) applied as straight FMAs, coefficients broadcast from an LDS record;
//   mfma_dense    : F' as a dense 14 x 14 tile in LDS (zero-padded to 16 x 16), S as two 16 x 16 column blocks kept in the MFMA
//                   accumulator layout, dS = F' S by 2 x 4 v_mfma_f64_16x16x4 per segment and stage (register r of lane (g, n) of the
//                   result IS the B operand of k-slot g + 4 r, so S never leaves the registers -- the best case for the matrix path);
//                   one segment per wavefront.
// Both update S <- S + h dS so that the product cannot be hoisted, and both are timed over the same number of segment-stages.
//   hipcc --offload-arch=gfx950 -O3 -o build/k1_contraction_ab tools/micro/k1_contraction_ab.hip && build/k1_contraction_ab
//   (under rocprofv3 --kernel-trace --stats for the per-kernel durations of profiles/r04_k1_mfma_ab.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

// record: Am[3] | Dq[12] | Om (q' = 1/2 Omega(w) q: w[3]) | Qq (q[4]) | Mw[9] | sigma  = 32 values per segment
constexpr int NREC = 32;

__global__ __launch_bounds__(256) void sparse_vector(const double* __restrict__ rec, double* __restrict__ out, int nseg, int stages, double h) {
    __shared__ double R[4][3][NREC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane / 21, col = lane - 21 * sl;
    const long seg = ((long)blockIdx.x * 4 + wave) * 3 + (sl < 3 ? sl : 0);
    const bool live = sl < 3 && seg < nseg;
    for (int e = lane; e < 3 * NREC; e += 64) {
        const long s = ((long)blockIdx.x * 4 + wave) * 3 + e / NREC;
        R[wave][e / NREC][e % NREC] = s < nseg ? rec[s * NREC + e % NREC] : 0.0;
    }
    __syncthreads();
    const double* r = R[wave][sl < 3 ? sl : 0];
    double c[14];
#pragma unroll
    for (int i = 0; i < 14; i++) c[i] = (col == i) ? 1.0 : 0.01 * (col + 1);
    for (int s = 0; s < stages; s++) {
        double d[14];
        const double sig = r[31];
        d[0] = 0.0;
        d[1] = sig * c[4]; d[2] = sig * c[5]; d[3] = sig * c[6];
#pragma unroll
        for (int i = 0; i < 3; i++)
            d[4 + i] = sig * (r[i] * c[0] + r[3 + 4 * i] * c[7] + r[4 + 4 * i] * c[8] + r[5 + 4 * i] * c[9] + r[6 + 4 * i] * c[10]);
        const double w0 = r[15], w1 = r[16], w2 = r[17], q0 = r[18], q1 = r[19], q2 = r[20], q3 = r[21];
        d[7] = 0.5 * sig * (-w0 * c[8] - w1 * c[9] - w2 * c[10] - q1 * c[11] - q2 * c[12] - q3 * c[13]);
        d[8] = 0.5 * sig * (w0 * c[7] + w2 * c[9] - w1 * c[10] + q0 * c[11] - q3 * c[12] + q2 * c[13]);
        d[9] = 0.5 * sig * (w1 * c[7] - w2 * c[8] + w0 * c[10] + q3 * c[11] + q0 * c[12] - q1 * c[13]);
        d[10] = 0.5 * sig * (w2 * c[7] + w1 * c[8] - w0 * c[9] - q2 * c[11] + q1 * c[12] + q0 * c[13]);
#pragma unroll
        for (int i = 0; i < 3; i++) d[11 + i] = sig * (r[22 + 3 * i] * c[11] + r[23 + 3 * i] * c[12] + r[24 + 3 * i] * c[13]);
#pragma unroll
        for (int i = 0; i < 14; i++) c[i] = fma(h, d[i], c[i]);
    }
    if (live) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < 14; i++) a += c[i];
        out[seg * 21 + col] = a;
    }
}

// dense F' (row-major 16 x 16, rows / columns 14, 15 zero) from the same record
__device__ void dense_from_record(const double* r, double* A) {
    for (int e = threadIdx.x & 63; e < 256; e += 64) A[e] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    if ((threadIdx.x & 63) == 0) {
        const double sig = r[31];
        for (int i = 0; i < 3; i++) {
            A[16 * (1 + i) + 4 + i] = sig;
            A[16 * (4 + i) + 0] = sig * r[i];
            for (int j = 0; j < 4; j++) A[16 * (4 + i) + 7 + j] = sig * r[3 + 4 * i + j];
            for (int j = 0; j < 3; j++) A[16 * (11 + i) + 11 + j] = sig * r[22 + 3 * i + j];
        }
        const double w0 = r[15], w1 = r[16], w2 = r[17], q0 = r[18], q1 = r[19], q2 = r[20], q3 = r[21], hs = 0.5 * sig;
        const double Om[16] = {0, -w0, -w1, -w2, w0, 0, w2, -w1, w1, -w2, 0, w0, w2, w1, -w0, 0};
        const double Qq[12] = {-q1, -q2, -q3, q0, -q3, q2, q3, q0, -q1, -q2, q1, q0};
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 4; j++) A[16 * (7 + i) + 7 + j] = hs * Om[4 * i + j];
            for (int j = 0; j < 3; j++) A[16 * (7 + i) + 11 + j] = hs * Qq[3 * i + j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

__global__ __launch_bounds__(256) void mfma_dense(const double* __restrict__ rec, double* __restrict__ out, int nseg, int stages, double h) {
    __shared__ double A[4][256];
    __shared__ double R[4][NREC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long seg = (long)blockIdx.x * 4 + wave;
    const bool live = seg < nseg;
    if (lane < NREC) R[wave][lane] = live ? rec[seg * NREC + lane] : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    dense_from_record(R[wave], A[wave]);
    const int n = lane & 15, g = lane >> 4;
    // S in accumulator layout: block b (columns 16 b + n), register r = row g + 4 r
    v4f64 S[2];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = g + 4 * r, col = 16 * b + n;
            S[b][r] = (row < 14 && col < 21) ? ((col == row) ? 1.0 : 0.01 * (col + 1)) : 0.0;
        }
    // the A operands of the four k-steps: A[row = n][k = g + 4 c] -- the same for every stage of a segment only in this benchmark; K1's
    // F' changes every stage, so they are re-read from LDS per stage as the real kernel would have to
    const double* Aw = A[wave];
    for (int s = 0; s < stages; s++) {
        double a[4];
#pragma unroll
        for (int c = 0; c < 4; c++) a[c] = Aw[16 * n + g + 4 * c];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], S[b][c], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) S[b][r] = fma(h, acc[r], S[b][r]);
        }
    }
    if (live) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int col = 16 * b + n;
            double sum = S[b][0] + S[b][1] + S[b][2] + S[b][3];
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (g == 0 && col < 21) out[seg * 21 + col] = sum;
        }
    }
}

int main(int argc, char** argv) {
    const int nseg = argc > 1 ? atoi(argv[1]) : 8192 * 50, stages = argc > 2 ? atoi(argv[2]) : 40, reps = 5;
    std::vector<double> rec((size_t)nseg * NREC);
    srand(1);
    for (auto& v : rec) v = 0.2 * ((rand() / (double)RAND_MAX) - 0.5);
    double *drec, *o1, *o2;
    hipMalloc(&drec, rec.size() * 8); hipMalloc(&o1, (size_t)nseg * 21 * 8); hipMalloc(&o2, (size_t)nseg * 21 * 8);
    hipMemcpy(drec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice);
    const double h = 1.0 / (51 * 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms1 = 0, ms2 = 0;
    const unsigned g1 = (unsigned)((nseg + 11) / 12), g2 = (unsigned)((nseg + 3) / 4);
    for (int r = 0; r < reps + 1; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(sparse_vector, dim3(g1), dim3(256), 0, 0, drec, o1, nseg, stages, h);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (r) ms1 += t / reps;
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_dense, dim3(g2), dim3(256), 0, 0, drec, o2, nseg, stages, h);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t, e0, e1); if (r) ms2 += t / reps;
    }
    std::vector<double> a((size_t)nseg * 21), b((size_t)nseg * 21);
    hipMemcpy(a.data(), o1, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 8, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < a.size(); i++) { const double d = a[i] - b[i]; md = d > md ? d : (-d > md ? -d : md); }
    const double segst = (double)nseg * stages;
    printf("| form | ms for %d segments x %d stages | ns per segment-stage | FMA-equivalent flops issued per segment-stage |\n|---|---|---|---|\n", nseg, stages);
    printf("| sparse vector (one lane per column, 51 non-zeros) | %.3f | %.3f | %d |\n", ms1, 1e6 * ms1 / segst, 2 * 51 * 21);
    printf("| dense MFMA (2 x 4 v_mfma_f64_16x16x4 per segment-stage) | %.3f | %.3f | %d |\n", ms2, 1e6 * ms2 / segst, 2 * 16 * 16 * 4 * 8);
    printf("max |difference| of the column sums between the two forms: %.2e\n", md);
    return 0;
}
