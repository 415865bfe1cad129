// What the memory system delivers to K4's access pattern: one 64-lane workgroup per "trajectory", each streaming over its OWN slab
// (84,175 doubles = the conic solver's per-trajectory workspace at K = 50) in passes of  y[i] = a x[i] + y[i]  over vectors of the
// solver's lengths, 4 elements in flight per lane as in Solver::stream, 8,192 workgroups compiled for 2 wavefronts per SIMD
// (2,048 resident: K4's launch shape).  Prints the bytes moved per second (2 reads + 1 write per element).
//   hipcc --offload-arch=gfx950 -O3 -o build/stream_ceiling tools/micro/stream_ceiling.hip && build/stream_ceiling [B] [passes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void sweep(double* work, size_t stride, int n, int nvec, int passes, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    for (int p = 0; p < passes; p++) {
        // pass p works on vectors (2 p) mod nvec and (2 p + 1) mod nvec of the slab: the whole slab is cycled through, as the solver does
        double* x = w + (size_t)((2 * p) % nvec) * n;
        double* y = w + (size_t)((2 * p + 1) % nvec) * n;
        for (int i0 = lane; i0 < n; i0 += 64 * 4) {
            double xv[4], yv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + 64 * q; xv[q] = i < n ? x[i] : 0.0; yv[q] = i < n ? y[i] : 0.0; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + 64 * q; if (i < n) y[i] = a * xv[q] + yv[q]; }
        }
        __syncthreads();
    }
}

// the same passes with 16 bytes per lane per load (each lane owns two adjacent elements): does the access width matter?
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void sweep2(double* work, size_t stride, int n, int nvec, int passes, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    const int n2 = n / 2;   // vectors are 8-byte aligned at odd offsets in general: the slab is laid out on even offsets here
    for (int p = 0; p < passes; p++) {
        double2* x = (double2*)(w + (size_t)((2 * p) % nvec) * (n + 1));
        double2* y = (double2*)(w + (size_t)((2 * p + 1) % nvec) * (n + 1));
        for (int i0 = lane; i0 < n2; i0 += 64 * 2) {
            double2 xv[2], yv[2];
#pragma unroll
            for (int q = 0; q < 2; q++) { const int i = i0 + 64 * q; xv[q] = i < n2 ? x[i] : double2{0, 0}; yv[q] = i < n2 ? y[i] : double2{0, 0}; }
#pragma unroll
            for (int q = 0; q < 2; q++) { const int i = i0 + 64 * q; if (i < n2) y[i] = double2{a * xv[q].x + yv[q].x, a * xv[q].y + yv[q].y}; }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192, passes = argc > 2 ? atoi(argv[2]) : 400;
    const int n = 1571, nvec = 52;                       // 52 x 1,571 = 81,692 doubles of an 84,176-double slab
    const size_t stride = 84176;
    double* work;
    if (hipMalloc((void**)&work, (size_t)B * stride * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(work, 0, (size_t)B * stride * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(sweep, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)B * passes * n * 24.0;
        printf("B = %d, %d passes of %d doubles: %.2f ms, %.2f TB/s (2 reads + 1 write per element)\n", B, passes, n, ms, bytes / (ms * 1e-3) / 1e12);
    }
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(sweep2, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)B * passes * (n / 2 * 2) * 24.0;
        printf("16 bytes per lane: B = %d, %d passes of %d doubles: %.2f ms, %.2f TB/s\n", B, passes, n / 2 * 2, ms, bytes / (ms * 1e-3) / 1e12);
    }
    (void)hipFree(work);
    return 0;
}
