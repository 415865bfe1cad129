/* abi_harness.c — replays, from plain C, the exact call sequence julia/ScvxAMD.jl makes for the reference's recipe
 * (rocketland.jl:26-32, SampleProblems.base_prob_aero_scaled):
 *
 *     cache = Cache(prob, info, lin_mod; tables=(drag, lift, trq))     scvx_ctx_create, scvx_set_nsub, scvx_set_aero_table
 *     pi    = create_initial(prob, cache)                              scvx_batch_create, scvx_batch_init(NULL),
 *                                                                       scvx_batch_get_trajectory / _linearization / _scalars
 *     pi, cnu, cdel = solve_step(pi, cache)   (x NSTEP)                scvx_solve_step + the three getters
 *     linearize_dynamics(pi.about, pi.sigma, 1/(K+1), cache)           scvx_linearize_f64_host
 *     predict_state(x4, u4, u5, sigma, dt, info, cache)                scvx_propagate_f64_host
 *
 * with the argument layouts the shim passes: the problem BY POINTER as the flat scvx_problem struct, states as
 * column-major 14 x (K+1) arrays, LinRes.derivative as column-major 14 x 21 per segment (14 x 25 and 5-row controls with the fin extension).  Julia is not in the build
 * image; this harness is how the binding's use of the ABI is executed on the GPU.  The test
 * (tests/test_abi_harness.py) writes the inputs, runs this program, and compares its outputs BIT FOR BIT with the same
 * sequence driven through the Python ctypes layer.
 *
 *     abi_harness <in.bin> <out.bin>
 * in.bin : scvx_problem | int32 n_aoa, n_mach, nsub, nstep | double aoa0, daoa, mach0, dmach | drag | lift | trq
 * out.bin: for create_initial and after each step: traj[(K+1)*(14+NU)+1] endpoint[K*14] deriv[K*14*(14+2NU+1)] rk cost (double)iter ;
 *          per step additionally (double)status nu dJ ; then linearize endpoint[K*14] deriv[K*294], predict_state x[14]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "scvx.h"

#define CHECK(ctx, call)                                                                  \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != 0) {                                                                   \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, scvx_last_error(ctx));    \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

static int dump_iteration(scvx_ctx *ctx, scvx_batch *b, int K, FILE *fo) {
    const int NU = scvx_control_dim(ctx), DSZ = 14 * (14 + 2 * NU + 1);   /* 3 / 294, or 5 / 350 with SCVX_MODEL_FINS */
    const size_t nrec = (size_t)(K + 1) * (14 + NU) + 1;
    double *rec = malloc(nrec * 8), *e = malloc((size_t)K * 14 * 8), *d = malloc((size_t)K * DSZ * 8);
    double rk, cost;
    int32_t it;
    CHECK(ctx, scvx_batch_get_trajectory(b, rec));
    CHECK(ctx, scvx_batch_get_linearization(b, e, d));
    CHECK(ctx, scvx_batch_get_scalars(b, &rk, &cost, &it));
    fwrite(rec, 8, nrec, fo); fwrite(e, 8, (size_t)K * 14, fo); fwrite(d, 8, (size_t)K * DSZ, fo);
    double sc[3] = {rk, cost, (double)it};
    fwrite(sc, 8, 3, fo);
    free(rec); free(e); free(d);
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 1; }
    scvx_problem p;
    int32_t hdr[4];
    double ax[4];
    if (fread(&p, sizeof p, 1, fi) != 1 || fread(hdr, 4, 4, fi) != 4 || fread(ax, 8, 4, fi) != 4) { fprintf(stderr, "short input\n"); return 1; }
    const int n_aoa = hdr[0], n_mach = hdr[1], nsub = hdr[2], nstep = hdr[3], K = p.K;
    const size_t nt = (size_t)n_aoa * n_mach;
    double *drag = malloc(nt * 8), *lift = malloc(nt * 8), *trq = malloc(nt * 8);
    if (fread(drag, 8, nt, fi) != nt || fread(lift, 8, nt, fi) != nt || fread(trq, 8, nt, fi) != nt) { fprintf(stderr, "short tables\n"); return 1; }
    fclose(fi);
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 1; }

    /* check_abi(): the library was compiled from the header this program was compiled with */
    int32_t asz[3];
    if (scvx_abi_version() != SCVX_ABI_VERSION || scvx_abi_struct_sizes(asz) != 0 || asz[0] != (int32_t)sizeof(scvx_problem) ||
        asz[1] != (int32_t)sizeof(scvx_solver_opts) || asz[2] != (int32_t)sizeof(scvx_threedof_opts)) {
        fprintf(stderr, "ABI mismatch: library %d, header %d\n", scvx_abi_version(), SCVX_ABI_VERSION);
        return 2;
    }
    /* Cache(prob, info, lin_mod; tables=...) */
    scvx_ctx *ctx = NULL;
    if (scvx_ctx_create(&p, 0, &ctx) != 0) { fprintf(stderr, "scvx_ctx_create failed\n"); return 2; }
    CHECK(ctx, scvx_set_nsub(ctx, nsub));
    if (p.aero_kind == 1) CHECK(ctx, scvx_set_aero_table(ctx, drag, lift, trq, n_aoa, n_mach, ax[0], ax[1], ax[2], ax[3]));

    /* create_initial(prob, cache) */
    scvx_batch *b = NULL;
    CHECK(ctx, scvx_batch_create(ctx, 1, &b));
    CHECK(ctx, scvx_batch_init(b, NULL));
    if (dump_iteration(ctx, b, K, fo)) return 2;

    /* solve_step(pi, cache) */
    for (int s = 0; s < nstep; s++) {
        int32_t st;
        double nu, dj;
        CHECK(ctx, scvx_solve_step(b, &st, &nu, &dj));
        if (st == 3 || st == 4 || st == 5) { fprintf(stderr, "Non-optimal result exiting (status %d)\n", st); return 3; }   /* rocketland.jl:273-276 */
        if (dump_iteration(ctx, b, K, fo)) return 2;
        double o[3] = {(double)st, nu, dj};
        fwrite(o, 8, 3, fo);
    }

    /* Dynamics.linearize_dynamics(pi.about, pi.sigma, 1/(K+1), cache) and predict_state on segment 4 */
    {
        const int NU = scvx_control_dim(ctx), DSZ = 14 * (14 + 2 * NU + 1);
        const size_t nrec = (size_t)(K + 1) * (14 + NU) + 1;
        double *rec = malloc(nrec * 8), *e = malloc((size_t)K * 14 * 8), *d = malloc((size_t)K * DSZ * 8);
        CHECK(ctx, scvx_batch_get_trajectory(b, rec));
        const double *x = rec, *u = rec + (size_t)(K + 1) * 14;   /* 14 x (K+1) and NU x (K+1), column-major */
        double sigma = rec[nrec - 1], dt = 1.0 / (K + 1);
        CHECK(ctx, scvx_linearize_f64_host(ctx, 1, K, x, u, &sigma, dt, e, d));
        fwrite(e, 8, (size_t)K * 14, fo); fwrite(d, 8, (size_t)K * DSZ, fo);
        double xs[28], us[10], out[14];
        memcpy(xs, x + 14 * 3, 14 * 8); memset(xs + 14, 0, 14 * 8);   /* hcat(initial_state, zeros(14)) */
        memcpy(us, u + NU * 3, (size_t)2 * NU * 8);                    /* hcat(uk, up) */
        CHECK(ctx, scvx_propagate_f64_host(ctx, 1, 1, xs, us, &sigma, dt, out));
        fwrite(out, 8, 14, fo);
        free(rec); free(e); free(d);
    }
    /* FirstRound.solve_initial(prob): the 3-DoF SOCP through scvx_threedof_solve, options struct by pointer */
    {
        scvx_threedof_opts o;
        CHECK(ctx, scvx_threedof_default_opts(&o));
        o.max_iter = 45;
        const int n = scvx_threedof_record_doubles(K);
        double *sol = malloc((size_t)n * 8), info[5];
        int32_t st3;
        CHECK(ctx, scvx_threedof_solve(ctx, 1, NULL, &o, sol, &st3, info));
        fwrite(sol, 8, (size_t)n, fo);
        double o2[6] = {(double)st3, info[0], info[1], info[2], info[3], info[4]};
        fwrite(o2, 8, 6, fo);
        free(sol);
    }
    fclose(fo);
    scvx_batch_destroy(b);
    scvx_ctx_destroy(ctx);
    free(drag); free(lift); free(trq);
    return 0;
}
