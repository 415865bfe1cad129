// The conic subproblem kernels of the fin extension (control_dim = 5): the executors and kernel templates of scvx_socp.hpp
// instantiated for NU = 5, in a translation unit of their own so that they compile beside scvx_batch.hip's NU = 3 set.
#include "scvx_socp.hpp"

namespace scvx {

template <int NW>
static void launch_block5(const SocpLaunch& a) {
    if (a.deriv_f)
        hipLaunchKernelGGL((socp_block_kernel<NW, float, 5>), dim3(a.B), dim3(64 * NW), 0, a.stream, a.C, a.B, a.work_stride, a.x, a.u, a.endpoint,
                           a.deriv_f, a.rk, a.ic, a.mask, a.work, a.sol, a.nu, a.info, a.status, a.ttr, a.acc);
    else
        hipLaunchKernelGGL((socp_block_kernel<NW, double, 5>), dim3(a.B), dim3(64 * NW), 0, a.stream, a.C, a.B, a.work_stride, a.x, a.u, a.endpoint,
                           a.deriv, a.rk, a.ic, a.mask, a.work, a.sol, a.nu, a.info, a.status, a.ttr, a.acc);
}

void launch_socp_fin(const SocpLaunch& a, int waves) {
    if (waves == 4) launch_block5<4>(a);
    else if (waves == 2) launch_block5<2>(a);
    else if (a.deriv_f)
        hipLaunchKernelGGL((socp_kernel_t<float, 5>), dim3(a.B), dim3(64), 0, a.stream, a.C, a.B, a.work_stride, a.x, a.u, a.endpoint, a.deriv_f,
                           a.rk, a.ic, a.mask, a.work, a.sol, a.nu, a.info, a.status, a.ttr, a.acc);
    else
        hipLaunchKernelGGL((socp_kernel_t<double, 5>), dim3(a.B), dim3(64), 0, a.stream, a.C, a.B, a.work_stride, a.x, a.u, a.endpoint, a.deriv,
                           a.rk, a.ic, a.mask, a.work, a.sol, a.nu, a.info, a.status, a.ttr, a.acc);
}

}  // namespace scvx
