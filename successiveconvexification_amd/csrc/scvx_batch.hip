// placeholder: batched SCvx entry points (filled in by the SOCP milestone)
#include "scvx_internal.hpp"
extern "C" {
int scvx_admm_default_opts(scvx_admm_opts*) { return SCVX_ERR_STATE; }
int scvx_batch_create(scvx_ctx*, int, scvx_batch**) { return SCVX_ERR_STATE; }
void scvx_batch_destroy(scvx_batch*) {}
int scvx_batch_set_admm(scvx_batch*, const scvx_admm_opts*) { return SCVX_ERR_STATE; }
int scvx_batch_init(scvx_batch*, const double*) { return SCVX_ERR_STATE; }
int scvx_solve_step(scvx_batch*, int32_t*, double*, double*) { return SCVX_ERR_STATE; }
int scvx_solve_step_async(scvx_batch*) { return SCVX_ERR_STATE; }
int scvx_solve(scvx_batch*, int32_t*, int32_t*, double*, double*) { return SCVX_ERR_STATE; }
int scvx_batch_get_trajectory(scvx_batch*, double*) { return SCVX_ERR_STATE; }
int scvx_batch_set_trajectory(scvx_batch*, const double*) { return SCVX_ERR_STATE; }
int scvx_batch_trajectory_dev(scvx_batch*, double**, int64_t*) { return SCVX_ERR_STATE; }
int scvx_batch_get_linearization(scvx_batch*, double*, double*) { return SCVX_ERR_STATE; }
int scvx_batch_get_scalars(scvx_batch*, double*, double*, int32_t*) { return SCVX_ERR_STATE; }
int scvx_batch_set_scalars(scvx_batch*, const double*, const double*, const int32_t*) { return SCVX_ERR_STATE; }
int scvx_batch_get_solver_stats(scvx_batch*, int32_t*, double*, double*) { return SCVX_ERR_STATE; }
int scvx_socp_solve(scvx_batch*, double*, double*) { return SCVX_ERR_STATE; }
}
