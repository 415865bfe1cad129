// Probe: lane mapping of the gfx9 whole-wavefront DPP shifts (wave_shl:1 = 0x130, wave_shr:1 = 0x138) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe/dpp_shift.hip -o build/dpp_shift ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const int x = (int)threadIdx.x;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);
}
int main() {
    int* d; int h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("wave_shl:1 lane0..3 <- %d %d %d %d ... lane62,63 <- %d %d\n", h[0], h[1], h[2], h[3], h[62], h[63]);
    printf("wave_shr:1 lane0..3 <- %d %d %d %d ... lane62,63 <- %d %d\n", h[64], h[65], h[66], h[67], h[126], h[127]);
    return 0;
}
