// Round 6, gate for the K4 phase-kernel executor: what does the memory system give the SAME per-trajectory slabs when a
// streaming pass is (a) a loop inside a resident one-wavefront-per-trajectory kernel at 2 / 4 / 8 wavefronts per SIMD (the monolithic
// socp_kernel's shape is the first of these), or (b) a grid-wide launch of its own per pass (the phase-kernel shape), with 64- or
// 256-lane blocks per slab and every load of a lane issued before its first store?
//   hipcc --offload-arch=gfx950 -O3 -o build/stream_phase_shapes tools/micro/stream_phase_shapes.hip && build/stream_phase_shapes [B] [passes]
// Pass = y[i] = a x[i] + y[i] over 1,571-element vectors of an 84,176-double slab (K = 50 workspace), 2 reads + 1 write per element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OCC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void resident(double* work, size_t stride, int n, int nvec, int passes, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    for (int p = 0; p < passes; p++) {
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

// resident shape, but the lane issues ALL its loads of a pass (ceil(1571 / 64) = 25 per vector) before the first store
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void resident_deep(double* work, size_t stride, int n, int nvec, int passes, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    for (int p = 0; p < passes; p++) {
        double* x = w + (size_t)((2 * p) % nvec) * n;
        double* y = w + (size_t)((2 * p + 1) % nvec) * n;
        double xv[25], yv[25];
#pragma unroll
        for (int q = 0; q < 25; q++) { const int i = lane + 64 * q; xv[q] = i < n ? x[i] : 0.0; yv[q] = i < n ? y[i] : 0.0; }
#pragma unroll
        for (int q = 0; q < 25; q++) { const int i = lane + 64 * q; if (i < n) y[i] = a * xv[q] + yv[q]; }
        __syncthreads();
    }
}

// phase shape: ONE pass per launch; block of T lanes per slab; all loads of a lane issued before its stores
template <int T>
__global__ __launch_bounds__(T) void phase(double* work, size_t stride, int n, int nvec, int p, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    double* x = w + (size_t)((2 * p) % nvec) * n;
    double* y = w + (size_t)((2 * p + 1) % nvec) * n;
    constexpr int U = (1571 + T - 1) / T;
    double xv[U], yv[U];
#pragma unroll
    for (int q = 0; q < U; q++) { const int i = threadIdx.x + T * q; xv[q] = i < n ? x[i] : 0.0; yv[q] = i < n ? y[i] : 0.0; }
#pragma unroll
    for (int q = 0; q < U; q++) { const int i = threadIdx.x + T * q; if (i < n) y[i] = a * xv[q] + yv[q]; }
}

// phase shape with several passes' worth of vectors per launch (a fused phase: G independent vector pairs of the slab per block)
template <int T, int G>
__global__ __launch_bounds__(T) void phase_g(double* work, size_t stride, int n, int nvec, int p, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    constexpr int U = (1571 + T - 1) / T;
    for (int g = 0; g < G; g++) {
        double* x = w + (size_t)((2 * (p + g)) % nvec) * n;
        double* y = w + (size_t)((2 * (p + g) + 1) % nvec) * n;
        double xv[U], yv[U];
#pragma unroll
        for (int q = 0; q < U; q++) { const int i = threadIdx.x + T * q; xv[q] = i < n ? x[i] : 0.0; yv[q] = i < n ? y[i] : 0.0; }
#pragma unroll
        for (int q = 0; q < U; q++) { const int i = threadIdx.x + T * q; if (i < n) y[i] = a * xv[q] + yv[q]; }
    }
}

// reference: the plain grid-stride copy-like stream over the whole allocation (what "6.29 TB/s" means on this box)
__global__ __launch_bounds__(256) void flat(double* work, size_t total, double a) {
    const size_t half = total / 2;
    double2* x = (double2*)work; double2* y = (double2*)(work + half);
    const size_t n2 = half / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 xv = x[i], yv = y[i];
        y[i] = double2{a * xv.x + yv.x, a * xv.y + yv.y};
    }
}

// K4's own mix is 4.1 reads : 1 write (PMC): y = a (x1 + x2 + x3) + y, four reads and one write per element
template <int OCC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void resident41(double* work, size_t stride, int n, int nvec, int passes, double a) {
    double* w = work + (size_t)blockIdx.x * stride;
    const int lane = threadIdx.x;
    for (int p = 0; p < passes; p++) {
        const double* x1 = w + (size_t)((4 * p) % nvec) * n;
        const double* x2 = w + (size_t)((4 * p + 1) % nvec) * n;
        const double* x3 = w + (size_t)((4 * p + 2) % nvec) * n;
        double* y = w + (size_t)((4 * p + 3) % nvec) * n;
        for (int i0 = lane; i0 < n; i0 += 64 * 4) {
            double xv[4], yv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + 64 * q; xv[q] = i < n ? x1[i] + x2[i] + x3[i] : 0.0; yv[q] = i < n ? y[i] : 0.0; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + 64 * q; if (i < n) y[i] = a * xv[q] + yv[q]; }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void flat41(double* work, size_t total, double a) {
    const size_t q4 = total / 4;
    const double2* x1 = (const double2*)work; const double2* x2 = (const double2*)(work + q4); const double2* x3 = (const double2*)(work + 2 * q4);
    double2* y = (double2*)(work + 3 * q4);
    const size_t n2 = q4 / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 a1 = x1[i], a2 = x2[i], a3 = x3[i], yv = y[i];
        y[i] = double2{a * (a1.x + a2.x + a3.x) + yv.x, a * (a1.y + a2.y + a3.y) + yv.y};
    }
}
// read-only: s += x[i] (what a reduction sweep does)
__global__ __launch_bounds__(256) void flat_read(const double* work, size_t total, double* out) {
    const double2* x = (const double2*)work;
    const size_t n2 = total / 2;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = x[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192, passes = argc > 2 ? atoi(argv[2]) : 200;
    const int n = 1571, nvec = 52;
    const size_t stride = 84176;
    double* work;
    if (hipMalloc((void**)&work, (size_t)B * stride * 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(work, 0, (size_t)B * stride * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)B * passes * n * 24.0;
    auto report = [&](const char* name, float ms, double by) { printf("%-58s %9.2f ms  %.2f TB/s\n", name, ms, by / (ms * 1e-3) / 1e12); fflush(stdout); };
#define TIMED(name, by, ...) for (int rep = 0; rep < 2; rep++) { (void)hipEventRecord(e0); __VA_ARGS__; (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); if (rep) report(name, ms, by); }
    TIMED("resident, 1 wavefront per slab, 2 per SIMD, 4 in flight", bytes, hipLaunchKernelGGL(resident<2>, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5));
    TIMED("resident, 4 per SIMD", bytes, hipLaunchKernelGGL(resident<4>, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5));
    TIMED("resident, 8 per SIMD", bytes, hipLaunchKernelGGL(resident<8>, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5));
    TIMED("resident, 2 per SIMD, 25 + 25 loads in flight per lane", bytes, hipLaunchKernelGGL(resident_deep, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5));
    TIMED("phase: one launch per pass, 64 lanes per slab", bytes, for (int p = 0; p < passes; p++) hipLaunchKernelGGL(phase<64>, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, p, 0.5));
    TIMED("phase: one launch per pass, 256 lanes per slab", bytes, for (int p = 0; p < passes; p++) hipLaunchKernelGGL(phase<256>, dim3(B), dim3(256), 0, 0, work, stride, n, nvec, p, 0.5));
    TIMED("phase: one launch per pass, 512 lanes per slab", bytes, for (int p = 0; p < passes; p++) hipLaunchKernelGGL(phase<512>, dim3(B), dim3(512), 0, 0, work, stride, n, nvec, p, 0.5));
    TIMED("phase: 8 vector pairs per launch, 256 lanes per slab", bytes, for (int p = 0; p < passes; p += 8) hipLaunchKernelGGL((phase_g<256, 8>), dim3(B), dim3(256), 0, 0, work, stride, n, nvec, p, 0.5));
    TIMED("phase: 8 vector pairs per launch, 64 lanes per slab", bytes, for (int p = 0; p < passes; p += 8) hipLaunchKernelGGL((phase_g<64, 8>), dim3(B), dim3(64), 0, 0, work, stride, n, nvec, p, 0.5));
    {
        const size_t total = (size_t)B * stride;
        const double by = (double)(total / 2) * 24.0 * 4;
        TIMED("flat grid-stride y = a x + y over the whole allocation x4", by, for (int r = 0; r < 4; r++) hipLaunchKernelGGL(flat, dim3(256 * 32), dim3(256), 0, 0, work, total, 0.5));
    }
    {
        const double by = (double)B * passes * n * 40.0;
        TIMED("resident 4 reads : 1 write, 2 per SIMD", by, hipLaunchKernelGGL(resident41<2>, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5));
        TIMED("resident 4 reads : 1 write, 8 per SIMD", by, hipLaunchKernelGGL(resident41<8>, dim3(B), dim3(64), 0, 0, work, stride, n, nvec, passes, 0.5));
        const size_t total = (size_t)B * stride;
        const double byf = (double)(total / 4) * 40.0 * 4;
        TIMED("flat grid-stride, 4 reads : 1 write, x4", byf, for (int r = 0; r < 4; r++) hipLaunchKernelGGL(flat41, dim3(256 * 32), dim3(256), 0, 0, work, total, 0.5));
        const double byr = (double)total * 8.0 * 4;
        TIMED("flat grid-stride, read only, x4", byr, for (int r = 0; r < 4; r++) hipLaunchKernelGGL(flat_read, dim3(256 * 32), dim3(256), 0, 0, work, total, work));
    }
    (void)hipFree(work);
    return 0;
}
