/*
 * scvx.h — C ABI of the MI355X-native successive-convexification (SCvx) hot path.
 *
 * This is the drop-in boundary for the SCvx inner loop of BenChung/SuccessiveConvexification
 * (reference citations are file:line into that repository):
 *
 *   Dynamics.linearize_dynamics   dynamics.jl:321-334  ->  scvx_linearize_f64[_host]
 *   Dynamics.predict_state        dynamics.jl:315-317  ->  scvx_propagate_f64[_host]
 *   Rocketland.create_initial     rocketland.jl:34-39  ->  scvx_batch_create + scvx_batch_init
 *   FirstRound.solve_initial      initial_solve.jl:17-110 -> scvx_threedof_solve, scvx_batch_init_threedof
 *   Rocketland.solve_step         rocketland.jl:226-321->  scvx_solve_step
 *   Rocketland.solve_problem      rocketland.jl:432-443->  scvx_solve
 *   MOI.optimize! (conic solve)   rocketland.jl:271    ->  scvx_socp_solve (batched interior-point, device)
 *
 * Conventions (reference: dynamics.jl:13-19, 136-139):
 *   state  x[14] = [m, r(3), v(3), q(4, scalar first), w(3)]
 *   control u[3] = thrust in the body frame
 *   augmented input inp[21] = [x; u_k; u_{k+1}; sigma]
 *   LinRes.derivative is Julia column-major 14x21: element (i,j) at j*14 + i   (master.jl:90-93)
 *
 * FIN EXTENSION (model_flags & SCVX_MODEL_FINS; BASELINE configs[4] "6-DoF + fin aero").  The reference carries the fin model
 * only as commented-out code, so the model is DEFINED BY THIS BUILD from exactly those comments: control_dim = 5,
 * u[4:5] = coordinates of the fin force along fd1 = normalize((C(q) e2) x v) and fd2 = fd1 x v (dynamics.jl:60-63), the force
 * added to the aerodynamic force (:66) and its torque cross(rFB, ff) to the body torque (:69), and the cone
 * |u[4:5]| <= finmxf at every node (rocketland.jl:203-209).  Three choices the comments leave open, stated so that configs[4]
 * numbers are not read as matching a reference model that does not exist:
 *   (1) the commented expression at dynamics.jl:69 is `cross(info.rFB, ff) + bdy_trq`; only Jinv * (rFB x ff) is enabled here, the
 *       aerodynamic body torque bdy_trq stays dropped exactly as in the live NU = 3 model (dynamics.jl:69,93);
 *   (2) rFB is a body-frame arm and ff = u4 fd1 + u5 fd2 is built from inertial-frame vectors: the cross product mixes the two
 *       frames as the comment does -- kept as written, not "fixed";
 *   (3) the linearised thrust lower bound (rocketland.jl:199-201) uses |u[1:3]|, the thrust part of the control; the reference's
 *       `norm(iterAbout[n].control)` would be the norm of all five components once the control had five.
 * Every array below then uses NU = 5:
 *   u [B][K+1][5], inp[25], derivative 14x25 ([A | B- (5) | B+ (5) | Sigma]), trajectory record [(K+1)*19 + 1].
 * scvx_control_dim(ctx) returns NU.
 *
 * All pointers named *_dev are device (HBM) pointers valid on the context's device; all others are
 * host pointers.  Device entry points are asynchronous on the context's stream (scvx_set_stream);
 * the *_host entry points copy in, run, copy out and synchronise.  Every function returns 0 on
 * success and a negative code on failure; scvx_last_error(ctx) then describes the failure.
 * No C++ types and no exceptions cross this boundary.
 */
#ifndef SCVX_H
#define SCVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCVX_NX 14
#define SCVX_NU 3
#define SCVX_NP 21 /* 14 + 3 + 3 + 1 */
#define SCVX_NU_FINS 5
#define SCVX_NP_FINS 25 /* 14 + 5 + 5 + 1 */

#define SCVX_OK 0
#define SCVX_ERR_ARG -1
#define SCVX_ERR_HIP -2
#define SCVX_ERR_STATE -3
#define SCVX_ERR_NOMEM -4
#define SCVX_ERR_COMM -5 /* RCCL missing or a collective failed */

/* per-trajectory status codes written by scvx_solve_step / scvx_solve */
#define SCVX_ST_CONVERGED 0   /* ||nu|| <= nuTol and dJ <= delTol             (rocketland.jl:436) */
#define SCVX_ST_RUNNING 1     /* accepted step, not yet converged / imax hit                      */
#define SCVX_ST_REJECTED 2    /* rho < rh0: iterate kept, radius shrunk, dJ=Inf (rocketland.jl:299-301) */
#define SCVX_ST_SOLVER 3      /* conic solver did not reach tolerance        (rocketland.jl:273-276) */
#define SCVX_ST_NONFINITE 4   /* NaN/Inf encountered                                              */
#define SCVX_ST_INFEASIBLE 5  /* the subproblem has no feasible point: a boundary value violates a path cone (solver status 5) */

/* model_flags: enforce the dynamic-pressure limit 1/2 rho |v_k|^2 <= dpMax at nodes 1..K as the second-order cone
 * |v_k| <= sqrt(2 dpMax / rho) (fields master.jl:27,30; the constraint is a "todo" at rocketland.jl:211-212).  The
 * initial velocity must satisfy it. */
#define SCVX_MODEL_DPMAX 1
/* model_flags: the fin extension described at the top of this file (control_dim = 5).  Needs finmxf > 0. */
#define SCVX_MODEL_FINS 2

/* Flat image of DescentProblem (master.jl:17-71) + the aero scalars of AtmosphericData (master.jl:10-16).
 * Angles in degrees exactly as the reference stores them.  jB is column-major 3x3. */
typedef struct scvx_problem {
    double g, mdry, mwet, Tmin, Tmax;
    double deltaMax, thetaMax, gammaGs, omMax, dpMax;
    double jB[9];
    double alpha, rho, sos;
    double rTB[3], rFB[3];
    double rIi[3], rIf[3], vIi[3], vIf[3];
    double qBIi[4], qBIf[4];
    double wBi[3], wBf[3];
    double wNu, wID, wDS, wCst, wTviol, nuTol, delTol, tf_guess;
    double ri, rh0, rh1, rh2, alph, bet;
    double force_scalar, length_scalar; /* AtmosphericData scalars; ignored when aero_kind == 0 */
    double finmxf; /* fin extension: bound of |u[4:5]| (rocketland.jl:205 pins it to 0.01 in the commented code); read only with SCVX_MODEL_FINS */
    int32_t K, imax;
    int32_t aero_kind; /* 0 = ExoatmosphericData, 1 = AtmosphericData */
    int32_t model_flags; /* SCVX_MODEL_* bits: constraints the reference sketches but never wired up; 0 = the reference's model */
} scvx_problem;

/* Tunables of the batched conic solver that replaces MOI.optimize! (rocketland.jl:271): a structure-
 * exploiting primal-dual interior-point method (the algorithm class of the reference's Mosek / ECOS),
 * one wavefront per trajectory.  DESIGN.md explains why it is not the first-order splitting first planned. */
typedef struct scvx_solver_opts {
    int32_t max_iter;  /* interior-point iteration cap per SOCP (default 60)                                   */
    int32_t refine;    /* iterative-refinement passes per Newton solve, at most (default 6; a pass is skipped  */
                       /* when the solve's residual is already below a tenth of the dual tolerance)            */
    double tol;        /* primal / dual residual and relative-gap tolerance (default 1e-8)                     */
    double accept_tol; /* a solve that stops on its numerical floor with tol <= merit < accept_tol is reported */
                       /* as solver status 4 "almost optimal" and still feeds the trust-region test (MOI's     */
                       /* ALMOST_OPTIMAL band).  DEFAULT = tol (1e-8): the reference errors on anything but     */
                       /* OPTIMAL (rocketland.jl:273-276), so such solves become SCVX_ST_SOLVER; a wider band   */
                       /* (e.g. 1e-6) is opt-in.                                                                */
    int32_t reuse_inactive_tr; /* 0 (default): every solve_step solves its subproblem, as the reference does.  1: after a   */
                       /* REJECTED step (same about / dynam, radius halved, rocketland.jl:299-301) the conic solve is skipped  */
                       /* when the optimum just found lies strictly inside the new radius -- the radius row is then inactive  */
                       /* and that optimum is provably the new subproblem's optimum too.  The SCvx iterates are unchanged;   */
                       /* on the sample problems ~6 of the 14 solves of a solve_problem are such repeats.                    */
    int32_t warm_start;        /* 1 (default): the solve that follows a REJECTED step (same about / dynam, radius halved,  */
                       /* rocketland.jl:299-301) starts from the optimum the previous solve returned, as long as that point */
                       /* lies inside the new radius: the residuals of the NEW subproblem are evaluated there and the solve */
                       /* returns at once when they meet `tol` (1 iteration, no factorisation), else it iterates on from    */
                       /* there; every solve still ends at `tol`.  0: every solve starts cold, like the reference's.        */
    int32_t retries;           /* default 5, at most 7.  A solve that ends on its numerical floor above `tol` is run again from the   */
                       /* cold start with another step rule (step fraction, centring exponent, starting shift, centring    */
                       /* floor), at most this many times, before the trajectory is frozen as SCVX_ST_SOLVER: such failures */
                       /* sit at the precision floor of the Newton system and move with the path taken.  0: one attempt.    */
                       /* The iteration count reported for a solve is the sum over its attempts.                            */
    int32_t reserved0;         /* must be 0: scvx_batch_set_solver returns SCVX_ERR_ARG otherwise (a struct from before `retries` ends here) */
} scvx_solver_opts;

/* ---- ABI guard ----------------------------------------------------------------------------- */
/* scvx_problem, scvx_solver_opts and scvx_threedof_opts carry no size member: a binding written against another revision of
 * this header would hand over a struct of another layout and nothing would fail.  SCVX_ABI_VERSION is raised with every change
 * of a struct layout or of a signature below; scvx_abi_version() returns the value the LIBRARY was compiled with and
 * scvx_abi_struct_sizes() the sizeof of the three structs as the library sees them, in that order.  A binding compares both
 * with its own image before its first call (Python: _lib.lib(); Julia: ScvxAMD.check_abi(); C: tests/abi_harness.c).
 * Neither function touches the device. */
#define SCVX_ABI_VERSION 4
int scvx_abi_version(void);
int scvx_abi_struct_sizes(int32_t out[3]);

typedef struct scvx_ctx scvx_ctx;     /* owns device, stream, problem constants, aero tables */
typedef struct scvx_batch scvx_batch; /* owns the batched iterate (ProblemIteration x B)     */

/* ---- context ------------------------------------------------------------------------------- */
int scvx_ctx_create(const scvx_problem *p, int device, scvx_ctx **out);
void scvx_ctx_destroy(scvx_ctx *ctx);
const char *scvx_last_error(const scvx_ctx *ctx);
/* control_dim of the context's model: 3, or 5 with SCVX_MODEL_FINS.  Sizes every u / derivative / trajectory array. */
int scvx_control_dim(const scvx_ctx *ctx);
/* Stream every kernel and copy of the context is enqueued on.  NULL selects the context's OWN stream (created
 * hipStreamNonBlocking: it does not synchronise with HIP's default stream) -- a caller that produces or consumes
 * device buffers on another stream orders against it with scvx_get_stream + events, or scvx_synchronize.
 * scvx_use_null_stream selects HIP's legacy default stream (handle 0) itself. */
int scvx_set_stream(scvx_ctx *ctx, void *hip_stream);
int scvx_use_null_stream(scvx_ctx *ctx);
int scvx_get_stream(const scvx_ctx *ctx, void **hip_stream); /* the effective stream (never "NULL = own") */
int scvx_synchronize(scvx_ctx *ctx);
/* RK4 substeps per segment: the `npts` keyword of Dynamics.rk4 (dynamics.jl:112, default 10). */
int scvx_set_nsub(scvx_ctx *ctx, int nsub);
int scvx_get_nsub(const scvx_ctx *ctx);
/* Aerodynamics.load_aerodata tables (aerodynamics.jl:11-28): three n_aoa x n_mach grids, cos(AoA)
 * fastest, on axes aoa0 + i*daoa, mach0 + j*dmach.  Host pointers; prefiltered on the host, uploaded. */
int scvx_set_aero_table(scvx_ctx *ctx, const double *drag, const double *lift, const double *trq,
                        int n_aoa, int n_mach, double aoa0, double daoa, double mach0, double dmach);

/* ---- discretisation: Dynamics.linearize_dynamics / predict_state ---------------------------- */
/* x [B][K+1][14], u [B][K+1][NU], sigma [B]; endpoint [B][K][14]; deriv [B][K][14+2NU+1][14]  (NU = scvx_control_dim). */
int scvx_linearize_f64(scvx_ctx *ctx, int B, int K, const double *x_dev, const double *u_dev,
                       const double *sigma_dev, double dt, double *endpoint_dev, double *deriv_dev);
int scvx_linearize_f64_host(scvx_ctx *ctx, int B, int K, const double *x, const double *u,
                            const double *sigma, double dt, double *endpoint, double *deriv);
/* xnext [B][K][14]: state at the end of each segment started from node k with FOH (u_k,u_{k+1}). */
int scvx_propagate_f64(scvx_ctx *ctx, int B, int K, const double *x_dev, const double *u_dev,
                       const double *sigma_dev, double dt, double *xnext_dev);
int scvx_propagate_f64_host(scvx_ctx *ctx, int B, int K, const double *x, const double *u,
                            const double *sigma, double dt, double *xnext);

/* fp32 forms of the two discretisation entry points (SURVEY.md 8b "_f64/_f32"; BASELINE configs[3-4] name fp32): the
 * same layouts in float, float arithmetic throughout (RK4 state + sensitivity columns), tables read from the same
 * double coefficients.  Stated tolerance against the fp64 path: 2e-5 relative on endpoint, 2e-4 on derivative at
 * npts = 10 (tests/test_gpu_discretize.py).  The conic solve has no fp32 form: its block-tridiagonal factor needs double
 * (cond(S) * eps_float ~ 1, DESIGN.md "fp32"), so the SCvx loop itself always runs in fp64. */
int scvx_linearize_f32(scvx_ctx *ctx, int B, int K, const float *x_dev, const float *u_dev,
                       const float *sigma_dev, float dt, float *endpoint_dev, float *deriv_dev);
int scvx_linearize_f32_host(scvx_ctx *ctx, int B, int K, const float *x, const float *u,
                            const float *sigma, float dt, float *endpoint, float *deriv);
int scvx_propagate_f32(scvx_ctx *ctx, int B, int K, const float *x_dev, const float *u_dev,
                       const float *sigma_dev, float dt, float *xnext_dev);
int scvx_propagate_f32_host(scvx_ctx *ctx, int B, int K, const float *x, const float *u,
                            const float *sigma, float dt, float *xnext);

/* ---- FirstRound.solve_initial: the 3-DoF lossless-convexification landing SOCP, batched --------
 * (initial_solve.jl:17-110, inside a block comment at HEAD; BASELINE configs[0]).  Per node k = 0..K: position r,
 * velocity v, mass ma, thrust T, thrust bound ga, virtual acceleration ar and its bound kaR; one global nkaR >= |kaR|;
 * minimise -ma_K + 100 nkaR under the trapezoidal point-mass recursions with the fixed mass profile of :24, the
 * boundary values of :59-65 and the cones of :69-70, :80-88.  The problem's K, alpha, tf_guess, mwet, mdry, g, Tmin,
 * Tmax, thetaMax, gammaGs are read from the context; rIi, vIi per trajectory.
 * A device interior-point solve, one wavefront per trajectory (csrc/scvx_threedof_core.hpp). */
typedef struct scvx_threedof_opts {
    int32_t max_iter;   /* 60 */
    int32_t refine;     /* refinement passes per Newton solve on the uncondensed residual: 1 */
    double tol;         /* primal / dual residual and relative gap: 1e-9 */
    double delta;       /* static regularisation of the quasi-definite KKT matrix: 1e-9 */
    int32_t attitude;   /* scvx_batch_init_threedof only.  0 (default) = the reference's rotation_between(e1, -T_k)
                         * (initial_solve.jl:98); 1 = rotation_between(e1, +T_k): the body axis the engine pushes along
                         * points along the 3-DoF thrust, which is what the 6-DoF model (control (|T|,0,0) in body axes)
                         * means -- the reference's sign makes the start fly backwards (tools/init_compare.py) */
    int32_t reserved;
} scvx_threedof_opts;
int scvx_threedof_default_opts(scvx_threedof_opts *o);
/* doubles per trajectory of a solution record: (K+1)*15 + 1 -- per node r(3) v(3) ma T(3) ga kaR ar(3), then nkaR */
int32_t scvx_threedof_record_doubles(int K);
/* ic [B][6] host = per-trajectory (rIi, vIi), NULL = the problem's own; opts NULL = defaults.  Outputs (host):
 * sol [B][record_doubles]; status [B] (0 optimal: residuals and relative gap below tol -- or, when the KKT system breaks
 * down at the numerical floor, below 10 tol / 100 tol, the band the oracle's solver reports as optimal too; 1 iteration
 * cap, 2 stalled, 3 non-finite, 4 almost optimal: breakdown with residuals and relative gap below 1e-6, 5 infeasible: the primal residual stopped falling while the gap closed); info [B][5] = iterations, objective, gap, primal and dual residual
 * (status and info may be NULL). */
int scvx_threedof_solve(scvx_ctx *ctx, int B, const double *ic, const scvx_threedof_opts *opts, double *sol,
                        int32_t *status, double *info);
/* The same on device arrays, enqueued on the context's stream: info_dev [B][6] = status, iterations, objective, gap,
 * primal and dual residual. */
int scvx_threedof_solve_dev(scvx_ctx *ctx, int B, const double *ic_dev, const scvx_threedof_opts *opts,
                            double *sol_dev, double *info_dev);

/* ---- batched SCvx: create_initial / solve_step / solve_problem ------------------------------ */
int scvx_solver_default_opts(scvx_solver_opts *o);
int scvx_batch_create(scvx_ctx *ctx, int B, scvx_batch **out);
void scvx_batch_destroy(scvx_batch *b);
int scvx_batch_set_solver(scvx_batch *b, const scvx_solver_opts *o);
/* ic [B][6] = per-trajectory (rIi, vIi) overriding the problem's (Monte-Carlo dispersions); NULL =
 * every trajectory uses the problem's own.  Builds the straight-line guess (initial_solve.jl:113-129),
 * linearises it and sets rk=100, cost=Inf, iter=0 (rocketland.jl:38). */
int scvx_batch_init(scvx_batch *b, const double *ic);
/* create_initial from FirstRound.solve_initial instead of the straight line: scvx_batch_init(b, ic), then every
 * trajectory whose 3-DoF solve is optimal starts from its LinPoints (initial_solve.jl:90-105: state (ma, r, v,
 * rotation_between(e1, -T), 0), control (|T|, 0, 0), sigma = tf_guess); the others keep the straight line.
 * status3 [B] (host, may be NULL) = the 3-DoF solver statuses.  scvx_batch_reset returns to this start. */
int scvx_batch_init_threedof(scvx_batch *b, const double *ic, const scvx_threedof_opts *opts, int32_t *status3);
/* create_initial again for the same initial conditions, entirely on the device and asynchronous on the stream: the
 * straight-line guess kept from scvx_batch_init is restored, rk=100, cost=Inf, iter=0, flags cleared, and the guess is
 * re-linearised.  (A Monte-Carlo driver that re-runs solve_problem, or a benchmark loop, needs no host round trip.) */
int scvx_batch_reset(scvx_batch *b);
/* One Rocketland.solve_step for every trajectory of the batch.  Outputs are host arrays of
 * length B (any may be NULL): status codes above, ||nu||_F and dJ (Inf on rejection).
 * Like the reference's solve_step this has no notion of convergence: SCVX_ST_CONVERGED reports that the
 * loop test of solve_problem (rocketland.jl:436) holds after this step, and the next call steps the
 * trajectory again.  A trajectory whose conic solve failed (SCVX_ST_SOLVER / SCVX_ST_NONFINITE -- where the
 * reference raises an error) is frozen: later calls skip it, its status keeps the failure code and its
 * nu_norm / dJ stay those of the failing step. */
int scvx_solve_step(scvx_batch *b, int32_t *status, double *nu_norm, double *dJ);
/* Asynchronous form for timing loops: enqueue one solve_step on the stream, no host read-back. */
int scvx_solve_step_async(scvx_batch *b);
/* Rocketland.solve_problem from the batch's current state: every trajectory is stepped until it converges
 * (then it is left alone, as the reference's loop exits), fails, or has taken imax-1 steps; iters[B] = total
 * solve_step calls applied to the trajectory since scvx_batch_init. */
int scvx_solve(scvx_batch *b, int32_t *status, int32_t *iters, double *nu_norm, double *dJ);

/* ---- iterate access (the batched ProblemIteration) ------------------------------------------ */
/* traj [B][(K+1)*(14+NU) + 1]: per trajectory x[K+1][14], u[K+1][NU], sigma  (NU = 3: 17 per node).  */
int scvx_batch_get_trajectory(scvx_batch *b, double *traj);
int scvx_batch_set_trajectory(scvx_batch *b, const double *traj);
/* device pointer to the same layout (zero-copy views; scvx_allgather_trajectories gathers it); valid until destroy */
int scvx_batch_trajectory_dev(scvx_batch *b, double **traj_dev, int64_t *n_doubles);
int scvx_batch_get_linearization(scvx_batch *b, double *endpoint, double *deriv);
/* Mixed precision (BASELINE configs[3-4] "fp32"; SURVEY 8b `_f64/_f32`, H7): keep the derivative tiles `dynam[k].derivative`
 * (LinRes, master.jl:91-93) in float.  The discretisation still integrates in double and rounds each entry once, at the
 * store; the conic solve widens on load and keeps its arithmetic, workspace, norms and pivots in double; the endpoint stays
 * double.  Halves the bytes of the one input every pass of the solve re-reads.  on = 1 / 0; an initialised batch is
 * re-linearised at once.  scvx_batch_get_linearization then returns the float values, widened. */
int scvx_batch_set_linearization_f32(scvx_batch *b, int on);
int scvx_batch_get_scalars(scvx_batch *b, double *rk, double *cost, int32_t *iter);
int scvx_batch_set_scalars(scvx_batch *b, const double *rk, const double *cost, const int32_t *iter);
/* per-trajectory flags, the rest of a checkpoint (trajectory + scalars + flags restore a batch after
 * scvx_batch_init with the same ic): status as above; active = 0 once failed (never stepped again);
 * live = active and not yet converged inside scvx_solve.  Any pointer may be NULL.
 * The conic solver's own warm-start state is NOT part of a checkpoint: scvx_batch_set_scalars / set_flags / set_trajectory
 * drop it, so the first solve of a restored (or edited) batch starts cold.  After an ACCEPTED step that is what an
 * uninterrupted run does too (bit-identical continuation); after a REJECTED step the uninterrupted run would have
 * warm-started, so the continuation agrees to the solver tolerance, not bit for bit. */
int scvx_batch_get_flags(scvx_batch *b, int32_t *status, int32_t *active, int32_t *live);
int scvx_batch_set_flags(scvx_batch *b, const int32_t *status, const int32_t *active, const int32_t *live);
/* last SOCP solve, per trajectory: solver status (0 optimal: merit < tol; 4 almost optimal: numerical floor with
 * tol <= merit < accept_tol; 1 iteration cap; 2 numerical floor / KKT breakdown at merit >= accept_tol; 3 non-finite;
 * 5 infeasible: a fixed boundary value (rIi, vIi, wBi) violates the glideslope / rate / dynamic-pressure cone of node 1),
 * interior-point iterations, final merit max(pres, dres, relgap) of the returned iterate, its objective */
int scvx_batch_get_solver_stats(scvx_batch *b, int32_t *status, int32_t *iters, double *merit, double *pobj);

/* Running totals over every solve_step enqueued since the last call with reset != 0 (what a timed region really executed):
 * out8 = {trajectory-steps, conic solves run, interior-point iterations summed over them, solves that were warm-started,
 * solves skipped by reuse_inactive_tr, steps REJECTED, steps that failed (SOLVER / NONFINITE / INFEASIBLE), steps that ended
 * CONVERGED}.  Synchronises the stream. */
int scvx_batch_get_step_stats(scvx_batch *b, double *out8, int reset);

/* ---- per-kernel device time of the solve_step chain (HIP events on the context's stream) ------ */
int scvx_batch_set_profiling(scvx_batch *b, int enable);
/* ms[5] = {socp (K4), propagate (K2), tr_update (K5), linearize (K1), glue (K3: candidate/unpack)}
 * summed over the `steps` solve_steps enqueued since the last call; synchronises and resets. */
int scvx_batch_get_profile(scvx_batch *b, double *ms, int64_t *steps);

/* ---- multi-GPU: the single exchange step of the path (no counterpart in the reference) ---- */
/* One process per GPU, one context per process, contiguous shards of the Monte-Carlo batch; nothing inside the
 * SCvx iteration communicates.  The final records are all-gathered over RCCL (xGMI inside a node) on the
 * context's stream.  Bootstrap as with NCCL: rank 0 calls scvx_comm_unique_id, the host language ships the
 * SCVX_COMM_ID_BYTES to every rank by whatever channel it has, every rank calls scvx_comm_create. */
#define SCVX_COMM_ID_BYTES 128
/* Non-collective: 0 when the RCCL library can be bound in this process (SCVX_RCCL_LIB overrides the soname), else
 * SCVX_ERR_COMM.  Ranks exchange this BEFORE scvx_comm_create, whose ncclCommInitRank blocks until every rank arrives. */
int scvx_comm_probe(void);
int scvx_comm_unique_id(void *id_out /* SCVX_COMM_ID_BYTES */);
int scvx_comm_create(scvx_ctx *ctx, const void *unique_id, int rank, int world);
int scvx_comm_destroy(scvx_ctx *ctx);
int scvx_comm_info(const scvx_ctx *ctx, int *rank, int *world); /* world = 0: no communicator */
/* out_dev [world][B][(K+1)*(14+NU)+1]: every rank's trajectory records (all ranks hold the same B), asynchronous on the
 * context's stream; status_out_dev / iters_out_dev [world][B] (either may be NULL). */
int scvx_allgather_trajectories(scvx_batch *b, double *out_dev);
int scvx_allgather_status(scvx_batch *b, int32_t *status_out_dev, int32_t *iters_out_dev);
/* raw collectives on the context's communicator and stream (count elements per rank) */
int scvx_allgather_f64(scvx_ctx *ctx, const double *send_dev, double *recv_dev, int64_t count);
int scvx_allgather_i32(scvx_ctx *ctx, const int32_t *send_dev, int32_t *recv_dev, int64_t count);

/* ---- the conic subproblem alone (replaces MOI.optimize!, rocketland.jl:271) ------------------ */
/* Solves the trust-region SOCP at the batch's current (about, dynam, rk).  sol [B][(K+1)*(14+NU)+1] as
 * the trajectory layout but sigma slot holds sigma + dsigma; nu [B][K][14] (nu_2..nu_{K+1}). */
int scvx_socp_solve(scvx_batch *b, double *sol, double *nu);

#ifdef __cplusplus
}
#endif
#endif /* SCVX_H */
