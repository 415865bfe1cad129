"""The conic subproblem and the SCvx step on the MI355X vs the oracles.

Three references, from tightest to loosest:
  * the CPU twin of the same algorithm (oracle/scvx_port.cpp): same iteration path, 1e-6 on the minimiser,
  * the INDEPENDENT interior-point oracle on the full Rocketland.build_model form (oracle/ipm.py + oracle/socp.py):
    2e-5 on the minimiser at the default tolerance 1e-8, 5e-6 with both solvers at 1e-10, objective 1e-8 relative
    (SURVEY.md 8c asked for <= 1e-5 on trajectories; DESIGN.md "parity tolerance"),
  * size-independent properties at the full batch (feasibility of what the solver returns, the linearised dynamics
    rows, the solver reaching its tolerance on every trajectory).
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(B, ic=None, npts=10):
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled, npts=npts)
    b = ScvxBatch(c, B).init(ic)
    return c, b


def test_initial_guess_matches_linear_points():
    from oracle import model
    po = model.base_prob_scaled()
    ic = model.disperse_ics(po, 5, 20261004)
    c, b = _setup(5, ic)
    x, u, s = b.trajectory()
    for t in range(5):
        xo, uo = model.linear_points(po, ic[t, :3], ic[t, 3:])
        assert np.abs(x[t] - xo).max() < 1e-14
        assert np.abs(u[t] - uo).max() < 1e-14
    assert np.all(s == po.tf_guess)
    rk, cost, it = b.scalars()
    assert np.all(rk == 100.0) and np.all(np.isinf(cost)) and np.all(it == 0)  # rocketland.jl:38


def test_socp_matches_cpu_twin_and_oracle_ipm():
    from oracle import model, port, scvx as oscvx
    po = model.base_prob_scaled()
    B = 4
    ic = model.disperse_ics(po, B, 20261004)
    c, b = _setup(B, ic)
    xb, ub, sg = b.trajectory()
    e, d = b.linearization()
    x, u, snew, nu = b.socp_solve()
    st, its, merit, pobj = b.solver_stats()
    assert np.all(st == 0), (st, merit)
    tw = port.socp(po, xb, ub, e, d, 100.0, ic)
    assert np.all(tw["status"] == 0)
    # same algorithm on both sides: identical iteration paths (1e-11 apart, tools/diag_twin.py) unless one side reaches
    # the numerical floor an iteration earlier — objectives then still agree to ~1e-8, the minimiser to the flatness
    # of the optimum.  Tight on the objective, flatness-level on the minimiser.
    K = po.K

    def obj(dx, du, ds, nv):
        return (-dx[:, K, 0] + po.wNu * np.sqrt((nv**2).sum((1, 2))) + 0.5 * np.sqrt((dx**2).sum((1, 2)) + (du**2).sum((1, 2))) + np.abs(ds))
    og, ot = obj(x - xb, u - ub, snew - sg, nu), obj(tw["dx"], tw["du"], tw["ds"], tw["nu"])
    assert np.abs(og - ot).max() < 1e-9 * np.abs(ot).max()
    assert np.abs(x - (xb + tw["dx"])).max() < 1e-6
    assert np.abs(u - (ub + tw["du"])).max() < 1e-6
    assert np.abs(snew - (sg + tw["ds"])).max() < 1e-6
    assert np.abs(nu - tw["nu"]).max() < 1e-6
    assert merit.max() < 1e-8 and tw["merit"].max() < 1e-8
    # independent oracle: the full build_model form solved by oracle.ipm (first trajectory only: seconds)
    it0 = oscvx.create_initial(po, 10, ic[0, :3], ic[0, 3:])
    sol, ix = oscvx.solve_socp(it0)
    assert sol.status == "optimal"
    z = sol.x
    # against the independent solver both sides run at ITS tolerance (1e-9): two valid answers at the device default (1e-8) sit up to
    # ~3e-5 apart in u on this flat optimum, whichever central path led there
    from successiveconvexification_amd.batch import ScvxBatch
    b9 = ScvxBatch(c, 1, tol=1e-9).init(ic[:1])
    x, u, snew, nu = b9.socp_solve()
    assert b9.solver_stats()[0][0] == 0 and b9.solver_stats()[2][0] < 1e-9
    xb, ub, sg = xb[:1], ub[:1], sg[:1]
    assert np.abs(x[0] - z[ix.xv].T).max() < 2e-5
    assert np.abs(u[0] - z[ix.uv].T).max() < 2e-5
    assert abs(snew[0] - sg[0] - z[ix.dsig]) < 2e-5
    assert np.abs(nu[0] - z[ix.nuv].T[1:]).max() < 2e-5
    # objective parity is much tighter than solution parity (the optimum is flat)
    obj = -x[0, -1, 0] + po.wNu * np.linalg.norm(nu[0]) + 0.5 * np.linalg.norm(np.concatenate([(x - xb)[0].ravel(), (u - ub)[0].ravel()])) + abs(snew[0] - sg[0])
    assert abs(obj - sol.pobj) < 1e-8 * abs(sol.pobj)


def test_both_socp_executors_agree(monkeypatch):
    """socp_kernel (one wavefront per trajectory, the large-batch form) and socp_block_kernel (four wavefronts per
    trajectory, chosen below 512 trajectories) run the same portable solver core: same iteration paths, objectives to
    1e-7 relative (the tolerance of the twin comparison above), minimisers to the flatness of the optimum.  SCVX_K4_WAVES forces either form."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 6
    ic = model.disperse_ics(po, B, 20261004)
    res = {}
    for waves in ("1", "2", "4"):
        monkeypatch.setenv("SCVX_K4_WAVES", waves)
        c, b = _setup(B, ic)
        xb, ub, sg = b.trajectory()
        x, u, snew, nu = b.socp_solve()
        st, its, merit, pobj = b.solver_stats()
        assert np.all(st == 0), (waves, st, merit)
        res[waves] = (x, u, snew, nu, its, pobj)
        b.close(); c.close()
    a = res["1"]
    for waves in ("2", "4"):   # two wavefronts: the two-ended form on two wavefronts (round 6); four: two-ended on two assembly / chain pairs
        bq = res[waves]
        assert np.abs(a[4] - bq[4]).max() <= 1, waves          # iteration counts
        assert np.abs(a[5] - bq[5]).max() < 1e-8 * np.abs(a[5]).max(), waves
        for i in range(4):
            assert np.abs(a[i] - bq[i]).max() < 2e-6, (waves, i)


def test_solve_problem_tail_executor_matches_single_wavefront(monkeypatch):
    """scvx_solve picks the conic solver's executor from the live count (one wavefront per trajectory above 4 per CU,
    2 once <= 1,024 are still stepped on 256 CUs, 4 once <= 512): same trajectories and statuses as the one-wavefront form
    forced for the whole run, on a flyable problem whose trajectories converge after different numbers of steps.  The step
    at which a trajectory passes the convergence test may differ by one: at the optimum dJ is rounding noise (1e-11), so
    "accepted" against "rejected" there is decided by the summation order of the executor."""
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import DescentProblem
    from successiveconvexification_amd.dynamics import IntegratorCache
    from successiveconvexification_amd.montecarlo import disperse_ics
    p = DescentProblem()
    p.K, p.tf_guess, p.mdry, p.mwet, p.alpha, p.imax = 20, 6.0, 1.0, 2.0, 0.05, 12
    p.rIi, p.vIi = np.array([4.0, 2.0, 0.0]), np.array([-0.5, -0.5, 0.3])
    B = 1100
    ic = disperse_ics(p, 0, B, 99, frac=0.3)
    res = {}
    for waves in (None, "1"):
        if waves is None:
            monkeypatch.delenv("SCVX_K4_WAVES", raising=False)
        else:
            monkeypatch.setenv("SCVX_K4_WAVES", waves)
        c = IntegratorCache(p, npts=4)
        b = ScvxBatch(c, B).init(ic)
        st, it, nu, dj = b.solve()
        x, u, s = b.trajectory()
        res[waves] = (st, it, x, u, s)
        b.close(); c.close()
    a, r = res[None], res["1"]
    assert len(np.unique(a[1])) > 1, "the instance should finish trajectories at different steps"
    same_status = a[0] == r[0]
    assert same_status.mean() > 0.99
    assert (np.abs(a[1] - r[1])[same_status] <= 1).mean() > 0.99
    assert ((a[1] == r[1]) & same_status).mean() > 0.9
    assert np.abs(a[2][same_status] - r[2][same_status]).max() < 1e-5 and np.abs(a[3][same_status] - r[3][same_status]).max() < 1e-5
    assert np.abs(a[4][same_status] - r[4][same_status]).max() < 1e-5


def test_f32_linearisation_mode_against_the_fp64_path():
    """scvx_batch_set_linearization_f32: K1 integrates in double and stores the derivative tiles as float, the conic
    solve reads them as float and stays double.  (i) the stored tiles are the fp64 tiles rounded once (bit for bit), the
    endpoint is untouched; (ii) one solve_step differs from the fp64 path by the rounding only (1e-6); (iii) so does a
    whole solve_problem on dispersed trajectories; (iv) switching an initialised batch re-linearises it."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 8
    ic = model.disperse_ics(po, B, 20261004)
    c, b64 = _setup(B, ic)
    from successiveconvexification_amd.batch import ScvxBatch
    b32 = ScvxBatch(c, B).set_linearization_f32(True).init(ic)
    e64, d64 = b64.linearization()
    e32, d32 = b32.linearization()
    assert np.array_equal(e32, e64)
    assert np.array_equal(d32, d64.astype(np.float32).astype(np.float64))
    st64, nu64, dj64 = b64.solve_step()
    st32, nu32, dj32 = b32.solve_step()
    assert np.array_equal(st32, st64)
    for a, r in zip(b32.trajectory(), b64.trajectory()):
        assert np.abs(a - r).max() < 1e-6
    (s64, it64, m64, _), (s32, it32, m32, _) = b64.solver_stats(), b32.solver_stats()
    assert np.all(s32 == 0) and np.all(m32 < 1e-8) and np.abs(it32 - it64).max() <= 1
    r64 = b64.solve()
    r32 = b32.solve()
    assert np.array_equal(r32[0], r64[0]) and np.array_equal(r32[1], r64[1])      # statuses, step counts
    for a, r in zip(b32.trajectory(), b64.trajectory()):
        assert np.abs(a - r).max() < 1e-4
    # (iv) back to double on a live batch: the tiles are the fp64 ones again
    x32 = b32.trajectory()
    b32.set_linearization_f32(False)
    b64.set_trajectory(*x32)
    assert np.array_equal(b32.linearization()[1], b64.linearization()[1])
    b32.close(); b64.close(); c.close()


def test_solve_step_matches_oracle_scvx_two_iterations():
    """Two full solve_step calls against oracle.scvx (IPM + exact discretisation) on one trajectory."""
    from oracle import model, scvx as oscvx
    po = model.base_prob_scaled()
    c, b = _setup(1)
    it = oscvx.create_initial(po, 10)
    for n in range(2):
        st, nun, dj = b.solve_step()
        it, cnu, cdel = oscvx.solve_step(it)
        x, u, s = b.trajectory()
        rk, cost, iters = b.scalars()
        assert iters[0] == it.iter
        assert rk[0] == it.rk
        assert abs(s[0] - it.sigma) < 5e-5
        assert np.abs(x[0] - it.x).max() < 5e-5
        assert np.abs(u[0] - it.u).max() < 5e-5
        assert abs(nun[0] - cnu) < 1e-6
        assert abs(cost[0] - it.cost) < 1e-4 * abs(it.cost)
        if np.isinf(cdel):
            assert np.isinf(dj[0])
        else:
            assert abs(dj[0] - cdel) < 1e-4 * abs(cdel)


def _check_socp_properties(po, ic, b, rk, tol=1e-6, rows=True):
    """Size-independent properties of one conic solve at the batch's current iterate: the solver reaches its
    tolerance on every trajectory, and what it returns satisfies every row of Rocketland.build_model
    (rocketland.jl:109-216): boundary rows exactly, the linearised dynamics with nu to 1e-9, every cone to `tol`."""
    xb, ub, sg = b.trajectory()
    x, u, snew, nu = b.socp_solve()
    st, its, merit, pobj = b.solver_stats()
    B, K = x.shape[0], po.K
    # solver status 0 = merit < tol (1e-8); 4 = stopped on the numerical floor inside the acceptance band.  At least
    # 99.5 % of the batch must be strictly optimal and nothing may be worse than 1e-7 (VERDICT r1 "done when")
    assert np.all((st == 0) | (st == 4)), np.unique(st, return_counts=True)
    assert (st == 0).mean() >= 0.995, np.unique(st, return_counts=True)
    assert merit.max() < 1e-7, merit.max()
    # boundary rows (rocketland.jl:109-115)
    assert np.abs(x[:, 0, 0] - po.mwet).max() < 1e-12
    assert np.abs(x[:, 0, 1:4] - ic[:, :3]).max() < 1e-12 and np.abs(x[:, 0, 4:7] - ic[:, 3:]).max() < 1e-12
    assert np.abs(x[:, K, 1:4] - po.rIf).max() < 1e-12 and np.abs(x[:, K, 7:11] - po.qBIf).max() < 1e-12
    assert np.abs(u[:, K, 1:3]).max() < 1e-12          # u[2:3, K+1] = 0 (rocketland.jl:115); fin controls, if any, are free
    # dynamics rows (:117-133)
    dx, du = x - xb, u - ub
    if rows:   # needs the whole linearisation on the host: 118 KB per trajectory at K = 50
        e, d = b.linearization()
        delta = np.concatenate([dx[:, :-1], du[:, :-1], du[:, 1:], np.broadcast_to((snew - sg)[:, None, None], (B, K, 1))], axis=-1)
        lhs = np.einsum("bkji,bkj->bki", d, delta) + nu - dx[:, 1:] + (e - xb[:, 1:])
        assert np.abs(lhs).max() < 1e-9
    # cones (:137-201)
    assert (x[:, 1:, 0] - po.mdry).min() > -tol
    assert (x[:, :K, 1] / np.tan(np.radians(po.gammaGs)) - np.linalg.norm(x[:, :K, 2:4], axis=-1)).min() > -tol
    assert (np.sqrt((1 - np.cos(np.radians(po.thetaMax))) / 2) - np.linalg.norm(x[:, :K, 9:11], axis=-1)).min() > -tol
    assert (po.omMax - np.linalg.norm(x[:, :K, 11:14], axis=-1)).min() > -tol
    un = np.linalg.norm(u[..., :3], axis=-1)
    assert (po.Tmax - un).min() > -tol and (u[..., 0] / np.cos(np.radians(po.deltaMax)) - un).min() > -tol
    ubn = np.linalg.norm(ub[..., :3], axis=-1)
    assert (np.sum(ub[..., :3] / ubn[..., None] * du[..., :3], axis=-1) - (po.Tmin - ubn)).min() > -tol
    assert (np.sqrt(np.sum(dx**2, axis=(1, 2)) + np.sum(du**2, axis=(1, 2))) - rk).max() < tol
    return x, u, snew, nu, its, merit


def test_returned_iterate_is_feasible_full_batch():
    """BASELINE configs[3] shape at the benchmarked settings (B = 8192, K = 50, rk4 npts = 10)."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 8192
    import bench
    ic = bench.disperse_ics(po, 0, B, 20261004)
    c, b = _setup(B, ic, npts=10)
    _check_socp_properties(po, ic, b, 100.0)
    # and after two accepted steps (the radius has grown to 100 * 3.2^2, the iterate is no longer the straight line): properties on
    # all 8,192, and three sampled trajectories of this batch against the INDEPENDENT oracle (oracle/scvx.py: IPM on the exact
    # build_model rows, rocketland.jl:226-321) after one and after two solve_steps -- as configs 3 and 5 have
    from oracle import scvx as oscvx
    sample = (0, 4097, 8191)
    its = {tr: oscvx.create_initial(po, 10, ic[tr, :3], ic[tr, 3:]) for tr in sample}
    for n in range(2):
        st, nun, dj = b.solve_step()
        xs, us, ss = b.trajectory()
        rk, _, _ = b.scalars()
        for tr in sample:
            its[tr], cnu, cdel = oscvx.solve_step(its[tr])
            assert rk[tr] == its[tr].rk, (n, tr)                         # same accept / reject decision and radius
            assert np.abs(xs[tr] - its[tr].x).max() < 5e-5 and np.abs(us[tr] - its[tr].u).max() < 5e-5, (n, tr)
            assert abs(ss[tr] - its[tr].sigma) < 5e-5 and abs(nun[tr] - cnu) < 1e-5, (n, tr)
    rk, _, _ = b.scalars()
    _check_socp_properties(po, ic, b, rk)
    b.close(); c.close()


def test_config5_shape_K100_B32768_properties():
    """BASELINE configs[4] shape: K = 100, B = 32768 (one GPU holds it: 1.2 MB of solver state per trajectory).  The fin
    model of that config is a stub in the reference: this runs control_dim = 3, as SURVEY 8d prescribes when
    the fin model is not built."""
    from dataclasses import replace
    from oracle import model
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    import bench
    K, B = 100, 32768
    po = replace(model.base_prob_scaled(), K=K)
    pp = replace(sp.base_prob_scaled, K=K)
    ic = bench.disperse_ics(po, 0, B, 20261005)
    c = IntegratorCache(pp, npts=10)
    # dynamics rows on a 1,024-trajectory batch (the host copy of the linearisation is 235 KB per trajectory here) ...
    bs = ScvxBatch(c, 1024).init(ic[:1024])
    _check_socp_properties(po, ic[:1024], bs, 100.0)
    bs.close()
    # ... everything else at the full size
    b = ScvxBatch(c, B).init(ic)
    x, u, snew, nu, its, merit = _check_socp_properties(po, ic, b, 100.0, rows=False)
    st, nun, dj = b.solve_step()
    assert np.all(st == 1) and np.isfinite(nun).all()
    b.close(); c.close()


def test_config5_shape_with_aero_K100_B32768(aero_tables):
    """BASELINE configs[4] at its shape WITH the aerodynamic tables (K = 100, B = 32768, seed 20261005): the conic solve's
    size-independent properties on every trajectory, one solve_step on all of them, and three sampled trajectories against the
    INDEPENDENT oracle (oracle/scvx.py at K = 100: IPM on the exact build_model rows + RK4 with the spline tables)."""
    from dataclasses import replace
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache
    import bench
    d, l, t = aero_tables
    K, B = 100, 32768
    pp = replace(sp.base_prob_aero_scaled(AtmosphericData(d, l, t)), K=K)
    po = replace(model.base_prob_scaled(model.AeroData(d, l, t)), K=K)
    ic = bench.disperse_ics(po, 0, B, 20261005)
    c = IntegratorCache(pp, npts=10)
    bs = ScvxBatch(c, 512).init(ic[:512])
    _check_socp_properties(po, ic[:512], bs, 100.0)      # incl. the linearised dynamics rows (host copy of the tiles)
    bs.close()
    b = ScvxBatch(c, B).init(ic)
    _check_socp_properties(po, ic, b, 100.0, rows=False)
    st, nun, dj = b.solve_step()
    assert np.all(st == 1) and np.isfinite(nun).all()
    xs, us, ss = b.trajectory()
    for tr in (0, 16384, 32767):
        it0 = oscvx.create_initial(po, 10, ic[tr, :3], ic[tr, 3:])
        it1, cnu, cdel = oscvx.solve_step(it0)
        assert np.abs(xs[tr] - it1.x).max() < 5e-5 and np.abs(us[tr] - it1.u).max() < 5e-5, tr
        assert abs(ss[tr] - it1.sigma) < 2e-5 and abs(nun[tr] - cnu) < 2e-6
    b.close(); c.close()


def test_aero_B256_dispersed_matches_oracle_on_a_sample(aero_tables):
    """BASELINE configs[2] as written: 6-DoF + aero tables, K = 50, B = 256 dispersed (SURVEY 8d law, seed 20261003).
    Four trajectories are checked against the INDEPENDENT oracle (oracle/scvx.py: IPM on the exact build_model rows +
    RK4 variational equations with the spline tables), all 256 through the size-independent properties."""
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache
    d, l, t = aero_tables
    pp = sp.base_prob_aero_scaled(AtmosphericData(d, l, t))
    po = model.base_prob_scaled(model.AeroData(d, l, t))
    B = 256
    ic = model.disperse_ics(po, B, 20261003)
    c = IntegratorCache(pp, npts=10)
    b = ScvxBatch(c, B).init(ic)
    x, u, snew, nu, its, merit = _check_socp_properties(po, ic, b, 100.0)
    st, nun, dj = b.solve_step()
    assert np.all(st == 1)
    xs, us, ss = b.trajectory()
    for tr in (0, 85, 170, 255):
        it0 = oscvx.create_initial(po, 10, ic[tr, :3], ic[tr, 3:])
        it1, cnu, cdel = oscvx.solve_step(it0)
        assert np.abs(xs[tr] - it1.x).max() < 5e-5 and np.abs(us[tr] - it1.u).max() < 5e-5, tr   # flat optimum: 2.4e-5 seen
        assert abs(ss[tr] - it1.sigma) < 2e-5 and abs(nun[tr] - cnu) < 1e-6
    # second step on the whole batch: still optimal everywhere
    st, nun, dj = b.solve_step()
    sst, its, merit, _ = b.solver_stats()
    assert np.all((sst == 0) | (sst == 4)) and (sst == 0).mean() >= 0.99 and merit.max() < 1e-7
    b.close(); c.close()


def test_f32_linearisation_mode_on_the_aero_model_and_small_batches(aero_tables):
    """The mixed-precision mode through the aero instantiation of K1 and the 2- / 4-wavefront executors of the conic solve
    (B = 600 / 40): one solve_step against the fp64 path, rounding-sized differences only."""
    from oracle import model
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache
    d, l, t = aero_tables
    pp = sp.base_prob_aero_scaled(AtmosphericData(d, l, t))
    po = model.base_prob_scaled(model.AeroData(d, l, t))
    c = IntegratorCache(pp, npts=4)
    for B in (40, 600):
        ic = model.disperse_ics(po, B, 20261003)
        b64 = ScvxBatch(c, B).init(ic)
        b32 = ScvxBatch(c, B).set_linearization_f32(True).init(ic)
        d64, d32 = b64.linearization()[1], b32.linearization()[1]
        assert np.array_equal(d32, d64.astype(np.float32).astype(np.float64))
        st64 = b64.solve_step()[0]
        st32 = b32.solve_step()[0]
        assert np.array_equal(st32, st64)
        for a, r in zip(b32.trajectory(), b64.trajectory()):
            assert np.abs(a - r).max() < 2e-6
        s32 = b32.solver_stats()
        assert np.all((s32[0] == 0) | (s32[0] == 4)) and s32[2].max() < 1e-7
        b32.close(); b64.close()
    c.close()


def test_tight_tolerance_agrees_with_independent_oracle_to_5e6():
    """Both solvers at 1e-10: the device minimiser and the independent oracle's agree to 5e-6 (the flatness of the
    optimum limits the default-tolerance comparison to 2e-5, not the solver)."""
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    po = model.base_prob_scaled()
    c = IntegratorCache(sp.base_prob_scaled, npts=10)
    b = ScvxBatch(c, 1, tol=1e-10).init(None)
    xb, ub, sg = b.trajectory()
    x, u, snew, nu = b.socp_solve()
    st, its, merit, pobj = b.solver_stats()
    assert st[0] in (0, 4) and merit[0] < 1e-9, (st, merit)
    it0 = oscvx.create_initial(po, 10)
    sol, ix = oscvx.solve_socp(it0, tol=1e-10)
    z = sol.x
    assert np.abs(x[0] - z[ix.xv].T).max() < 5e-6 and np.abs(u[0] - z[ix.uv].T).max() < 5e-6
    assert abs(snew[0] - sg[0] - z[ix.dsig]) < 5e-6 and np.abs(nu[0] - z[ix.nuv].T[1:]).max() < 5e-6
    b.close(); c.close()


def test_solve_runs_to_imax_and_reports_status():
    from oracle import model
    po = model.base_prob_scaled()
    B = 8
    ic = model.disperse_ics(po, B, 20261004)
    c, b = _setup(B, ic, npts=4)
    st, it, nu, dj = b.solve()
    assert np.all(it <= po.imax - 1)
    assert np.all((st >= 0) & (st <= 4))
    x, u, s = b.trajectory()
    assert np.isfinite(x).all() and np.isfinite(u).all()
    # the reference's sample problem keeps ||nu|| ~ 1e-2 (71 kg of propellant): it never meets nuTol
    assert np.all(st != 0)


def test_rocketland_mirror_single_trajectory():
    """The reference's recipe (rocketland.jl:26-32) through the host mirror: same names, same returns."""
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import rocketland as Rocketland, sample_problems as SampleProblems
    from successiveconvexification_amd.defns import ProbInfo
    from successiveconvexification_amd.dynamics import IntegratorCache, make_dynamics_module, linearize_dynamics, predict_state
    prob = SampleProblems.base_prob_scaled
    cache = IntegratorCache(prob, ProbInfo.from_problem(prob), make_dynamics_module(ProbInfo.from_problem(prob)))
    pi = Rocketland.create_initial(prob, cache)
    assert pi.iter == 0 and pi.rk == 100.0 and np.isinf(pi.cost) and len(pi.about) == prob.K + 1 and len(pi.dynam) == prob.K
    assert pi.dynam[0].derivative.shape == (14, 21)
    pi1, cnu, cdel = Rocketland.solve_step(pi, cache)
    it = oscvx.create_initial(model.base_prob_scaled(), 10)
    it1, onu, odel = oscvx.solve_step(it)
    assert pi1.iter == 1 and pi1.rk == it1.rk and abs(pi1.sigma - it1.sigma) < 2e-5 and abs(cnu - onu) < 1e-6 and np.isinf(cdel)
    assert np.abs(np.stack([p.state for p in pi1.about]) - it1.x).max() < 2e-5
    # Dynamics entry points with the reference's signatures
    lr = linearize_dynamics(pi1.about, pi1.sigma, 1 / (prob.K + 1), cache)
    assert np.abs(lr[3].derivative - pi1.dynam[3].derivative).max() < 1e-13
    xe = predict_state(pi1.about[3].state, pi1.about[3].control, pi1.about[4].control, pi1.sigma, 1 / (prob.K + 1), None, cache)
    assert np.abs(xe - lr[3].endpoint).max() < 1e-13
    # next_step (autodiff_dynamics.jl:104-107): the affine model reproduces the nonlinear map to second order
    from successiveconvexification_amd.dynamics import next_step
    a, an = pi1.about[3], pi1.about[4]
    pert = 1e-5
    pred = next_step(lr[3], a, an, a.state + pert, a.control, an.control, pi1.sigma, pi1.sigma, 0.0)
    true = predict_state(a.state + pert, a.control, an.control, pi1.sigma, 1 / (prob.K + 1), None, cache)
    assert np.abs(pred - true).max() < 1e-7
    trjs, tfs = Rocketland.run_iters(prob, 2, cache)
    assert len(trjs) == 2 and trjs[0].shape == (3, prob.K + 1) and abs(tfs[0] - it1.sigma) < 2e-5


def test_checkpoint_restore_roundtrip():
    """The batched iterate is the checkpoint: dump after one step, restore into a new batch, both continue identically."""
    from oracle import model
    po = model.base_prob_scaled()
    ic = model.disperse_ics(po, 3, 20261004)
    c, b = _setup(3, ic, npts=4)
    b.solve_step()
    x, u, s = b.trajectory()
    rk, cost, it = b.scalars()
    c2, b2 = _setup(3, ic, npts=4)
    b2.set_trajectory(x, u, s)
    b2.set_scalars(rk, cost, it)
    b2.set_flags(*b.flags())
    st1, nu1, dj1 = b.solve_step()
    st2, nu2, dj2 = b2.solve_step()
    assert np.array_equal(st1, st2) and np.array_equal(nu1, nu2) and np.array_equal(dj1, dj2)  # bitwise: same kernels, same data
    assert np.array_equal(b.trajectory()[0], b2.trajectory()[0])
    # ... and after a REJECTED step: the uninterrupted batch warm-starts its next conic solve from the iterate kept in
    # its work slab, the restored one has no such iterate and starts cold (set_scalars / set_flags drop the solver's warm state on
    # purpose).  Same subproblem, both solved to 1e-8: the continuation agrees to solver accuracy, not bit for bit.
    for _ in range(4):
        st, _, _ = b.solve_step()
        if np.any(st == 2):
            break
    assert np.any(st == 2), "the sample problem rejects from the third step on"
    c3, b3 = _setup(3, ic, npts=4)
    b3.set_trajectory(*b.trajectory())
    b3.set_scalars(*b.scalars())
    b3.set_flags(*b.flags())
    sta, nua, dja = b.solve_step()
    stb, nub, djb = b3.solve_step()
    assert np.array_equal(sta, stb) and np.array_equal(b.scalars()[0], b3.scalars()[0])      # same decisions, same radii
    assert np.abs(nua - nub).max() < 1e-6
    for p_, q_ in zip(b.trajectory(), b3.trajectory()):
        assert np.abs(p_ - q_).max() < 5e-5
    ita, itb = b.solver_stats()[1], b3.solver_stats()[1]
    rej = st == 2
    assert np.all(itb[rej] >= ita[rej])             # cold >= warm on the re-solved subproblems


def test_solve_step_with_aero_tables_matches_oracle(aero_tables):
    """BASELINE configs[2]: SampleProblems.base_prob_aero_scaled (lift_drag.csv tables) through one full solve_step."""
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache
    d, l, t = aero_tables
    pp = sp.base_prob_aero_scaled(AtmosphericData(d, l, t))
    po = model.base_prob_scaled(model.AeroData(d, l, t))
    assert pp.aero.force_scalar == pytest.approx(1.514738405e-8, rel=1e-9)
    c = IntegratorCache(pp, npts=10)
    b = ScvxBatch(c, 1).init(None)
    it = oscvx.create_initial(po, 10)
    e, dd = b.linearization()
    # The straight-line guess points the body axis exactly along -v: cos(AoA) sits ON the clamp(...,-1,1) boundary
    # (dynamics.jl:162-168) and the lift direction has zero norm (ifnz guard), where the reference model's
    # derivative is discontinuous; which side rounding picks is arbitrary, so the derivative is compared at 1e-6
    # here (values agree to rounding; generic attitudes agree to 1e-10 in test_gpu_discretize.py).
    assert np.abs(e[0] - it.endpoint).max() < 1e-12 and np.abs(dd[0] - it.deriv).max() < 1e-6
    # the table really acts: the aero linearisation differs from the exo one
    c0, b0 = _setup(1)
    assert np.abs(b0.linearization()[1] - dd).max() > 1e-6
    st, nun, dj = b.solve_step()
    it1, cnu, cdel = oscvx.solve_step(it)
    x, u, s = b.trajectory()
    assert st[0] == 1 and abs(nun[0] - cnu) < 1e-6 and abs(s[0] - it1.sigma) < 2e-5
    assert np.abs(x[0] - it1.x).max() < 2e-5 and np.abs(u[0] - it1.u).max() < 2e-5
    e1, d1 = b.linearization()
    assert np.abs(d1[0] - it1.deriv).max() < 5e-4  # linearisation about a point that itself agrees to 2e-5


def test_rejection_path_and_radius_schedule_match_oracle():
    """Five solve_steps on the reference's sample problem: the oracle accepts 1-2 and rejects 3-5 (rho < rh0,
    rocketland.jl:299-301): same statuses, same halving of rk, iterate frozen on rejection."""
    from oracle import model, scvx as oscvx
    po = model.base_prob_scaled()
    c, b = _setup(1)
    it = oscvx.create_initial(po, 10)
    seen_reject = False
    for n in range(5):
        xprev = b.trajectory()[0].copy()
        st, nun, dj = b.solve_step()
        it, cnu, cdel = oscvx.solve_step(it)
        rk, cost, iters = b.scalars()
        rejected = np.isinf(cdel) and n > 0
        assert (st[0] == 2) == rejected, (n, st, cdel)
        assert rk[0] == it.rk and iters[0] == it.iter
        if rejected:
            seen_reject = True
            assert np.array_equal(b.trajectory()[0], xprev) and np.isinf(dj[0])  # about/dynam kept (rocketland.jl:301)
        else:
            assert np.abs(b.trajectory()[0][0] - it.x).max() < 1e-4
            assert abs(cost[0] - it.cost) < 1e-4 * abs(it.cost)
    assert seen_reject


@pytest.mark.parametrize("K", [30, 100])
def test_other_horizons_match_twin_and_independent_oracle(K):
    """BASELINE configs[0] uses K=30, configs[4] K=100: same kernels, sizes from K at run time."""
    from dataclasses import replace
    from oracle import dynamics as od, model, port
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    po = replace(model.base_prob_scaled(), K=K)
    pp = replace(sp.base_prob_scaled, K=K)
    B = 3
    ic = model.disperse_ics(po, B, 20261005)
    b = ScvxBatch(IntegratorCache(pp, npts=4), B).init(ic)
    xb, ub, sg = b.trajectory()
    assert xb.shape == (B, K + 1, 14)
    e, d = b.linearization()
    e_ref, d_ref = od.linearize(od.Params(po), xb, ub, sg, 1.0 / (K + 1), 4)
    assert np.abs(d - d_ref).max() < 1e-11
    x, u, s, nu = b.socp_solve()
    st, its, merit, pobj = b.solver_stats()
    assert np.all(st == 0)
    tw = port.socp(po, xb, ub, e, d, 100.0, ic)
    assert np.all(tw["status"] == 0)
    def obj(dx, du, ds, nv):
        return (-dx[:, K, 0] + po.wNu * np.sqrt((nv**2).sum((1, 2))) + 0.5 * np.sqrt((dx**2).sum((1, 2)) + (du**2).sum((1, 2))) + np.abs(ds))
    og, ot = obj(x - xb, u - ub, s - sg, nu), obj(tw["dx"], tw["du"], tw["ds"], tw["nu"])
    assert np.abs(og - ot).max() < 1e-9 * np.abs(ot).max()
    assert np.abs(x - (xb + tw["dx"])).max() < 1e-6 and np.abs(u - (ub + tw["du"])).max() < 1e-6
    # the INDEPENDENT oracle (full build_model form, oracle/ipm.py) on the first trajectory at this horizon
    from oracle import scvx as oscvx
    it0 = oscvx.create_initial(po, 4, ic[0, :3], ic[0, 3:])
    sol, ix = oscvx.solve_socp(it0)
    assert sol.status == "optimal"
    z = sol.x
    assert np.abs(x[0] - z[ix.xv].T).max() < 2e-5 and np.abs(u[0] - z[ix.uv].T).max() < 2e-5
    assert abs(s[0] - sg[0] - z[ix.dsig]) < 2e-5 and np.abs(nu[0] - z[ix.nuv].T[1:]).max() < 2e-5
    st2, nun, dj = b.solve_step()
    assert np.all(st2 == 1)


@pytest.mark.parametrize("K", [4, 8, 9, 31])
def test_two_ended_factorisation_at_odd_and_small_horizons(K, monkeypatch):
    """The four- and (round 6) the two-wavefront executor factorise and solve the block chain from both ends (middle node K // 2; below 8
    nodes they keep the one-ended form): odd horizons put one more node in the bottom half.  Same solves as the one-wavefront executor and
    as the CPU twin."""
    from dataclasses import replace
    from oracle import model, port
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    po = replace(model.base_prob_scaled(), K=K)
    pp = replace(sp.base_prob_scaled, K=K)
    B = 3
    ic = model.disperse_ics(po, B, 20261005)
    res = {}
    for waves in ("1", "2", "4"):
        monkeypatch.setenv("SCVX_K4_WAVES", waves)
        c = IntegratorCache(pp, npts=4)
        b = ScvxBatch(c, B).init(ic)
        xb, ub, sg = b.trajectory()
        e, d = b.linearization()
        x, u, s, nu = b.socp_solve()
        st, its, merit, pobj = b.solver_stats()
        assert np.all((st == 0) | (st == 4)), (waves, st, merit)
        res[waves] = (x, u, s, nu, its, pobj, st)
        b.close(); c.close()
    r = res["1"]
    tw = port.socp(po, xb, ub, e, d, 100.0, ic)
    for waves in ("2", "4"):
        a = res[waves]
        assert np.array_equal(a[6], r[6]) and np.abs(a[4] - r[4]).max() <= 1, waves
        assert np.abs(a[5] - r[5]).max() < 1e-8 * np.abs(r[5]).max(), waves
        for i in range(4):
            assert np.abs(a[i] - r[i]).max() < 5e-6, (waves, i)
        assert np.abs(a[0] - (xb + tw["dx"])).max() < 5e-6 and np.abs(a[1] - (ub + tw["du"])).max() < 5e-6, waves


def test_flyable_problem_converges():
    """A variant of the sample problem with enough propellant (the reference's own sample never converges: 71 kg):
    every trajectory reaches SCVX_ST_CONVERGED (||nu|| <= nuTol and dJ <= delTol, rocketland.jl:436) with nu driven to
    zero, scvx_solve leaves converged trajectories alone, and the conic solver never fails in the harder endgame where the nu-cone
    collapses to its vertex (dynamic pivot regularisation)."""
    from dataclasses import replace
    import bench
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    p = replace(sp.base_prob_scaled, mdry=0.55, nuTol=1e-6, delTol=1e-3, imax=40, tf_guess=8.0)
    B = 8
    c = IntegratorCache(p, npts=10)
    b = ScvxBatch(c, B).init(bench.disperse_ics(p, 0, B, 7))
    st, it, nu, dj = b.solve()
    assert np.all(st == 0), (st, it)
    assert np.all(it < p.imax - 1) and nu.max() < 1e-6 and np.all(dj <= p.delTol)
    x, u, s = b.trajectory()
    K = p.K
    # the converged trajectories are dynamically feasible: nonlinear re-propagation closes the defects
    from successiveconvexification_amd.dynamics import propagate_batch
    xp = propagate_batch(c, x, u, s, 1.0 / (K + 1))
    assert np.abs(xp - x[:, 1:]).max() < 1e-5
    assert (x[:, -1, 0] > p.mdry).all() and (s > 5).all()
    un = np.linalg.norm(u, axis=-1)
    assert (un <= p.Tmax + 1e-6).all() and (un >= p.Tmin - 1e-4).all()
    # inside scvx_solve a converged trajectory is left alone (the reference's loop exits): live = 0, still active
    stf, act, live = b.flags()
    assert np.all(stf == 0) and np.all(act == 1) and np.all(live == 0)
    # solve_step itself has no notion of convergence (rocketland.jl:226-321): it steps them again
    _, _, it_before = b.scalars()
    st2, nu2, _ = b.solve_step()
    _, _, it_after = b.scalars()
    assert np.all(it_after == it_before + 1) and np.all((st2 == 0) | (st2 == 1) | (st2 == 2)) and nu2.max() < 1e-5


def test_nonfinite_trajectory_is_reported_and_frozen_the_rest_continue():
    """The reference throws when a solve is not optimal (rocketland.jl:273-276); a batch cannot: the offending trajectory
    gets status SOLVER (3) / NONFINITE (4), keeps its iterate and drops out, every other trajectory advances exactly
    as it does without the bad neighbour."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 5
    ic = model.disperse_ics(po, B, 20261004)
    c, good = _setup(B, ic)
    st_g, nu_g, dj_g = good.solve_step()
    xg, ug, sg = good.trajectory()
    good.close()
    c2, b = _setup(B, ic)
    x, u, s = b.trajectory()
    x[2, 7, 4] = np.nan                      # one poisoned node of trajectory 2
    b.set_trajectory(x, u, s)
    st, nu, dj = b.solve_step()
    assert st[2] in (3, 4), st
    keep = [0, 1, 3, 4]
    assert np.array_equal(st[keep], st_g[keep])
    x1, u1, s1 = b.trajectory()
    assert np.abs(x1[keep] - xg[keep]).max() < 1e-9 and np.abs(u1[keep] - ug[keep]).max() < 1e-9
    assert np.all(np.isfinite(x1[keep])) and np.all(np.isfinite(u1[keep]))
    # frozen: a second step leaves it alone and still runs the others
    st2, _, _ = b.solve_step()
    assert st2[2] in (3, 4)
    x2, _, _ = b.trajectory()
    assert np.array_equal(np.isnan(x2[2]), np.isnan(x1[2]))
    b.close(); c.close(); c2.close()


def test_batch_api_argument_and_state_errors():
    from successiveconvexification_amd import _lib, sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled)
    with pytest.raises(_lib.ScvxError):
        ScvxBatch(c, 0)                      # B >= 1
    b = ScvxBatch(c, 2)
    with pytest.raises(_lib.ScvxError):
        b.solve_step()                       # before scvx_batch_init
    with pytest.raises(ValueError):
        b.init(np.zeros((3, 6)))             # ic must be [B][6]
    with pytest.raises(_lib.ScvxError):
        ScvxBatch(c, 2, tol=-1.0)            # bad solver options
    with pytest.raises(_lib.ScvxError):
        ScvxBatch(c, 2, retries=8)           # the ladder has seven further rules
    ScvxBatch(c, 2, retries=0).close()       # one attempt only: the reference's literal behaviour
    o = _lib.ScvxSolverOpts()
    c._L.scvx_solver_default_opts(ctypes.byref(o))
    o.reserved0 = 3                          # what a caller built against the struct without `retries` / `reserved0` would pass
    assert c._L.scvx_batch_set_solver(b.handle, ctypes.byref(o)) < 0
    b.init(None)
    st, nu, dj = b.solve_step()
    assert st.shape == (2,) and np.all(st == 1)
    b.close(); c.close()


def test_reset_restores_create_initial_on_device():
    """scvx_batch_reset: the straight-line guess, its linearisation and the scalars of create_initial come back without a
    host round trip, and the next solve_problem reproduces the first bit for bit."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 6
    ic = model.disperse_ics(po, B, 20261004)
    c, b = _setup(B, ic)
    x0, u0, s0 = b.trajectory()
    e0, d0 = b.linearization()
    first = [b.solve_step() for _ in range(3)]
    x3 = b.trajectory()[0].copy()
    b.reset()
    x, u, s = b.trajectory()
    e, d = b.linearization()
    rk, cost, it = b.scalars()
    st, act, live = b.flags()
    assert np.array_equal(x, x0) and np.array_equal(u, u0) and np.array_equal(s, s0)
    assert np.array_equal(e, e0) and np.array_equal(d, d0)
    assert np.all(rk == 100.0) and np.all(np.isinf(cost)) and np.all(it == 0) and np.all(act == 1) and np.all(live == 1)
    again = [b.solve_step() for _ in range(3)]
    for a, bb in zip(first, again):
        assert all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, bb))
    assert np.array_equal(b.trajectory()[0], x3)
    b.close(); c.close()


def test_native_rccl_allgather_world_of_one():
    """The library's own communicator (scvx_comm_create / scvx_allgather_trajectories, RCCL bound at run time): a world
    of one rank on this box's GPU -- the N > 1 bootstrap and ordering are covered by the CPU gloo test, the transport by
    the driver's 8-GPU bench."""
    import ctypes as C
    import torch
    from oracle import model
    from successiveconvexification_amd import _lib
    po = model.base_prob_scaled()
    B = 4
    ic = model.disperse_ics(po, B, 20261004)
    c, b = _setup(B, ic)
    L = _lib.lib()
    ident = (C.c_char * 128)()
    assert L.scvx_comm_unique_id(ident) == 0
    out = torch.zeros((1, B, b.nrec), dtype=torch.float64, device="cuda")
    # no communicator yet: a clean state error, not a crash
    assert L.scvx_allgather_trajectories(b.handle, C.c_void_p(out.data_ptr())) == -3
    _lib.check(c.handle, L.scvx_comm_create(c.handle, ident, 0, 1), "scvx_comm_create")
    r, w = C.c_int(-1), C.c_int(-1)
    L.scvx_comm_info(c.handle, C.byref(r), C.byref(w))
    assert (r.value, w.value) == (0, 1)
    b.solve_step()
    _lib.check(c.handle, L.scvx_allgather_trajectories(b.handle, C.c_void_p(out.data_ptr())), "scvx_allgather_trajectories")
    stat = torch.zeros((1, B), dtype=torch.int32, device="cuda")
    iters = torch.zeros((1, B), dtype=torch.int32, device="cuda")
    _lib.check(c.handle, L.scvx_allgather_status(b.handle, C.c_void_p(stat.data_ptr()), C.c_void_p(iters.data_ptr())), "scvx_allgather_status")
    c.synchronize()
    assert np.array_equal(out.cpu().numpy()[0], b.trajectory_record())
    assert np.all(iters.cpu().numpy() == 1) and np.all(stat.cpu().numpy() == 1)
    assert L.scvx_comm_create(c.handle, ident, 0, 1) == -3     # one communicator per context
    assert L.scvx_comm_destroy(c.handle) == 0
    b.close(); c.close()


def test_dynamic_pressure_cone_on_device_matches_independent_oracle():
    """SCVX_MODEL_DPMAX: the optional cone |v_k| <= sqrt(2 dpMax / rho) on the device vs the independent oracle (explicit
    rows in oracle/socp.py), binding on a run of nodes; two solve_steps keep it satisfied."""
    from dataclasses import replace
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    p0 = model.base_prob_scaled()
    vm = 0.2255
    dpmax = 0.5 * p0.rho * vm**2
    po = replace(p0, enforce_dp=True, dpMax=dpmax)
    pp = replace(sp.base_prob_scaled, dpMax=dpmax, model_flags=1)
    c = IntegratorCache(pp, npts=10)
    b = ScvxBatch(c, 2).init(None)
    xb, ub, sg = b.trajectory()
    x, u, snew, nu = b.socp_solve()
    st, its, merit, pobj = b.solver_stats()
    assert np.all(st == 0) and merit.max() < 1e-8
    it = oscvx.create_initial(po, 10)
    sol, ix = oscvx.solve_socp(it)
    z = sol.x
    assert np.abs(x[0] - z[ix.xv].T).max() < 2e-5 and np.abs(u[0] - z[ix.uv].T).max() < 2e-5
    speed = np.linalg.norm(x[0, :po.K, 4:7], axis=1)
    assert speed.max() < vm + 1e-8 and (speed > vm - 1e-6).sum() >= 10
    for _ in range(2):
        stt, _, _ = b.solve_step()
        assert np.all((stt == 1) | (stt == 2))
    xs, _, _ = b.trajectory()
    assert np.linalg.norm(xs[:, :po.K, 4:7], axis=-1).max() < vm + 1e-7
    # flag off: the same problem data gives the unconstrained optimum (faster than vm)
    c0 = IntegratorCache(replace(sp.base_prob_scaled, dpMax=dpmax), npts=10)
    b0 = ScvxBatch(c0, 1).init(None)
    x0, _, _, _ = b0.socp_solve()
    assert np.linalg.norm(x0[0, :po.K, 4:7], axis=1).max() > vm + 1e-3
    b.close(); c.close(); b0.close(); c0.close()


def test_reuse_of_the_optimum_across_rejected_steps_changes_nothing():
    """scvx_solver_opts.reuse_inactive_tr: a complete solve_problem with and without the shortcut.  Same accept / reject
    sequence, same radius schedule, trajectories equal to solver accuracy; the skipped solves are visible (iters = 0)."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 16
    ic = model.disperse_ics(po, B, 20261004)
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled, npts=10)
    a = ScvxBatch(c, B).init(ic)
    r = ScvxBatch(c, B, reuse_inactive_tr=True).init(ic)
    skipped = 0
    for n in range(po.imax - 1):
        sa, nua, dja = a.solve_step()
        sr, nur, djr = r.solve_step()
        assert np.array_equal(sa, sr), n
        assert np.array_equal(a.scalars()[0], r.scalars()[0])          # radius schedule
        assert np.allclose(nua, nur, rtol=0, atol=1e-7)
        its = r.solver_stats()[1]
        skipped += int((its == 0).sum())
        assert np.all(a.solver_stats()[1] > 0)
    xa, ua, sga = a.trajectory()
    xr, ur, sgr = r.trajectory()
    assert np.abs(xa - xr).max() < 1e-6 and np.abs(ua - ur).max() < 1e-6 and np.abs(sga - sgr).max() < 1e-6
    assert skipped >= 4 * B, skipped     # the sample problem's rejection run: most of its re-solves are repeats
    a.close(); r.close(); c.close()


def test_infeasible_boundary_value_gets_its_own_status():
    """An initial position outside the glideslope cone (a constant row against a constant bound at node 1): solver status 5
    without iterating, SCVX_ST_INFEASIBLE for the trajectory, which is frozen; its neighbours are unaffected."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 4
    ic = model.disperse_ics(po, B, 20261004)
    c, good = _setup(B, ic)
    st_g, nu_g, _ = good.solve_step()
    xg = good.trajectory()[0]
    good.close()
    bad = ic.copy()
    bad[1, 1] = 5.0          # |(r_y, r_z)| = 5 > r_up / tan(20 deg) = 2.75
    c2, b = _setup(B, bad)
    x0 = b.trajectory()[0].copy()
    st, nu, dj = b.solve_step()
    sst, its, _, _ = b.solver_stats()
    assert st[1] == 5 and sst[1] == 5 and its[1] == 0
    keep = [0, 2, 3]
    assert np.array_equal(st[keep], st_g[keep]) and np.abs(b.trajectory()[0][keep] - xg[keep]).max() < 1e-9
    assert np.array_equal(b.trajectory()[0][1], x0[1])          # frozen
    st2, _, _ = b.solve_step()
    assert st2[1] == 5 and np.all(b.flags()[1] == np.array([1, 0, 1, 1]))
    b.close(); c.close(); c2.close()


def test_warm_start_after_rejected_steps_changes_nothing_but_the_iteration_count():
    """scvx_solver_opts.warm_start (default on): the solve that follows a rejected step starts from the previous solve's
    optimum while that point lies inside the halved radius.  A complete
    solve_problem with and without it: same accept / reject sequence and radius schedule, every solve still at merit < 1e-8,
    trajectories equal to solver accuracy, and the warm-started solves take a fraction of the iterations."""
    from oracle import model
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    po = model.base_prob_scaled()
    B = 16
    ic = model.disperse_ics(po, B, 20261004)
    c = IntegratorCache(sp.base_prob_scaled, npts=10)
    w = ScvxBatch(c, B).init(ic)                      # default: warm_start = 1
    k = ScvxBatch(c, B, warm_start=False).init(ic)
    its_w, its_k, after_reject = [], [], []
    prev = np.ones(B, np.int32)
    for n in range(po.imax - 1):
        sw, nuw, _ = w.solve_step()
        sk, nuk, _ = k.solve_step()
        assert np.array_equal(sw, sk), n
        assert np.array_equal(w.scalars()[0], k.scalars()[0])
        assert np.allclose(nuw, nuk, rtol=0, atol=1e-7)
        stw, iw, mw, _ = w.solver_stats()
        _, ik, mk, _ = k.solver_stats()
        assert np.all((stw == 0) | (stw == 4)) and mw.max() < 1e-7 and mk.max() < 1e-7
        its_w.append(iw); its_k.append(ik); after_reject.append(prev == 2)
        prev = sw
    xw, uw, sgw = w.trajectory()
    xk, uk, sgk = k.trajectory()
    print("warm vs cold after %d steps: x %.2e u %.2e sigma %.2e" % (po.imax - 1, np.abs(xw - xk).max(), np.abs(uw - uk).max(), np.abs(sgw - sgk).max()))
    # a warm start only ever re-uses an optimum whose radius row is inactive in the new problem, so both runs reach the same points
    # (the blended start of a solve whose radius binds, SCVX_BLEND_WARM, is off for this reason: 1.6e-4 here with it)
    assert np.abs(xw - xk).max() < 1e-6 and np.abs(uw - uk).max() < 1e-6 and np.abs(sgw - sgk).max() < 1e-6
    its_w, its_k, ar = np.array(its_w), np.array(its_k), np.array(after_reject)
    assert np.array_equal(its_w[~ar], its_k[~ar])               # a solve after an accepted step starts cold either way
    assert its_w.sum() < 0.85 * its_k.sum()                      # -23 % iterations over the sample problem's 14 steps
    assert (its_w[ar] < 0.5 * its_k[ar]).mean() > 0.6            # most solves after a rejection: under half the iterations
    w.close(); k.close(); c.close()


def test_two_rank_bench_on_one_gpu_matches_the_single_process_run(tmp_path):
    """the N > 1 path of bench.py end to end on the one GPU of this box -- two ranks (fresh child processes,
    torch.distributed over gloo: SCVX_DIST_BACKEND=gloo) sharing the card, strong scaling of a 2,048-trajectory batch, two
    solve_steps, the final all-gather -- against a single-process run of the same 2,048 trajectories.  With the conic solver's
    executor pinned (SCVX_K4_WAVES=1: 1,024 and 2,048 trajectories would otherwise pick different executors, whose sums are
    ordered differently) every trajectory's arithmetic is independent of its batch: the gathered records must be equal BIT FOR BIT."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SCVX_DIST_BACKEND="gloo", SCVX_K4_WAVES="1", MASTER_ADDR="127.0.0.1")
    common = ["--steps", "2", "--warmup", "0", "--exact-steps", "--no-cpu-baseline", "--no-traj-check", "--no-k1-sweep"]
    f2, f1 = str(tmp_path / "two.npy"), str(tmp_path / "one.npy")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--global-batch", "2048",
                         "--dump-gathered", f2] + common, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r2.returncode == 0, r2.stderr[-3000:]
    line2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["scaling"] == "strong" and line2["config"]["batch_per_gpu"] == 1024
    assert line2["config"]["all_gather_shape"] == [2, 1024, 51 * 17 + 1] and line2["config"]["traj_iters_timed"] == 2 * 2048
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "2048", "--dump-gathered", f1] + common,
                        env=dict(env, SCVX_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a, b = np.load(f2), np.load(f1)
    assert a.shape == b.shape == (2048, 51 * 17 + 1)
    assert np.array_equal(a, b)


def test_bench_gpus_2_from_a_plain_shell_launches_its_own_ranks():
    """`bench.py --gpus 2` with no WORLD_SIZE in the environment.  The parent starts the two rank processes itself
    (it never touches the GPU), they rendezvous on 127.0.0.1 over gloo and share this box's one card; rank 0's ONE line comes back
    through the parent: n_gpus 2, the headline = STRONG scaling of the global batch 8192 (BASELINE configs[3] as written: 4096 per
    rank here), the weak figure (8192 per rank) beside it, whole solve_problem periods timed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SCVX_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-traj-check", "--no-k1-sweep"], env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == 8192 and line["config"]["batch_per_gpu"] == 4096
    assert line["steps"] == 14 and line["steps_requested"] == 3 and line["timed_region"]["indices_within_solve_problem"] == list(range(14))
    assert line["config"]["traj_iters_timed"] == 14 * 8192 and line["config"]["all_gather_shape"] == [2, 4096, 51 * 17 + 1]
    assert "2 ranks" in line["config"]["all_gather"]
    w = line["weak_scaling"]
    assert w["scaling"] == "weak" and w["batch_per_gpu"] == 8192 and w["global_batch"] == 16384 and w["value"] > 0
    # a rank that fails makes the parent fail (here: an argument error inside the children)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--global-batch", "8191", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-traj-check", "--no-k1-sweep"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert bad.returncode != 0


def test_five_rank_rehearsal_of_the_drivers_n8_shape_on_one_gpu(tmp_path):
    """the first real 8-GPU run must not be the first many-rank run.  The GPU box's process guard allows at most
    SIX processes on its card and this test process is one of them, so the rehearsal is FIVE ranks (the eight-rank sharding / gather /
    clock logic runs on CPU over gloo: tests/test_distributed_cpu.py::test_eight_rank_strong_shard_and_gather_shape_of_the_driver_run):
    `bench.py --gpus 5 --global-batch 5120` from a plain shell -- 1,024 trajectories per rank, exactly the per-rank shard (and
    therefore the two-wavefront executor) of the driver's N = 8 run of the 8,192 batch -- over gloo, the five processes sharing the
    card.  The gathered records must equal, bit for bit, a single-process run of the same 5,120 trajectories with that executor pinned."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SCVX_DIST_BACKEND"] = "gloo"
    common = ["--steps", "2", "--warmup", "0", "--exact-steps", "--no-cpu-baseline", "--no-traj-check", "--no-k1-sweep"]
    f5, f1 = str(tmp_path / "five.npy"), str(tmp_path / "one.npy")
    r5 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "5", "--global-batch", "5120", "--dump-gathered", f5] + common,
                        env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r5.returncode == 0, r5.stderr[-3000:]
    lines = [l for l in r5.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 5 and line["scaling"] == "strong" and line["config"]["batch_per_gpu"] == 1024
    assert line["config"]["all_gather_shape"] == [5, 1024, 51 * 17 + 1] and line["config"]["traj_iters_timed"] == 2 * 5120
    assert "5 ranks" in line["config"]["all_gather"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "5120", "--dump-gathered", f1] + common,
                        env=dict(env, SCVX_DIST_BACKEND="nccl", SCVX_K4_WAVES="2"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a, b = np.load(f5), np.load(f1)
    assert a.shape == b.shape == (5120, 51 * 17 + 1)
    assert np.array_equal(a, b)
    # a rank that dies takes the job down with a non-zero exit code (an argument error inside the children)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "5", "--global-batch", "5119", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-traj-check", "--no-k1-sweep"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert bad.returncode != 0


def test_fuzz_class_that_used_to_stall_matches_independent_oracle():
    """class 16 of tools/k4_fuzz.py (K = 25, 41 % propellant, glideslope 36 deg, tf_guess 9.2 -- round 2's
    profiles/r02_k4_fuzz.md rows 13/16/22/23 are the classes that stalled) on the device against oracle/ipm.py.  Trajectory 11 of
    its dispersed batch ended "stalled at merit 1.09e-8" before the centring floor (SCVX_MU_FLOOR); it and two neighbours must be
    OPTIMAL now and agree with the independent solver.  5e-5 on the minimiser: the optimum of this class is flatter than the
    sample problem's (2.2e-5 seen on the twin)."""
    import os
    import sys
    from dataclasses import replace
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import k4_fuzz
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    rng = np.random.default_rng(1)
    base = model.base_prob_scaled()
    for _ in range(17):
        po = k4_fuzz.draw_class(rng, base)
    assert po.K == 25 and not po.enforce_dp
    pp = replace(sp.base_prob_scaled, K=po.K, mdry=po.mdry, Tmin=po.Tmin, deltaMax=po.deltaMax, thetaMax=po.thetaMax,
                 gammaGs=po.gammaGs, omMax=po.omMax, tf_guess=po.tf_guess)
    ic = model.disperse_ics(po, 16, 516, 0.3)
    c = IntegratorCache(pp, npts=4)
    b = ScvxBatch(c, 16).init(ic)
    xs, us, ss, nu = b.socp_solve()
    st, its, merit, _ = b.solver_stats()
    assert np.all(st == 0) and merit.max() < 1e-8, (st, merit)
    for tr in (11, 0, 2):
        it0 = oscvx.create_initial(po, 4, ic[tr, :3], ic[tr, 3:])
        sol, ix = oscvx.solve_socp(it0)
        assert sol.status == "optimal"
        assert np.abs(xs[tr] - sol.x[ix.xv].T).max() < 5e-5 and np.abs(us[tr] - sol.x[ix.uv].T).max() < 5e-5, tr
    b.close(); c.close()


@pytest.mark.parametrize("cls", [3, 8, 21, 30, 62, 77, 90])
def test_random_problem_classes_match_independent_oracle(cls):
    """Off the sample problem: classes of tools/k4_fuzz.py (horizon 12 .. 64, mass ratio, throttle range, gimbal / tilt / glideslope /
    rate limits, tf_guess, dynamic-pressure cone drawn at random) -- one solve_step of the first feasible dispersed trajectory on the
    device against one solve_step of the independent oracle (oracle/scvx.py: explicit build_model rows + its own IPM), both at 1e-9."""
    import os
    import sys
    from dataclasses import replace
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import k4_fuzz
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    rng = np.random.default_rng(1)
    base = model.base_prob_scaled()
    for _ in range(cls + 1):
        po = k4_fuzz.draw_class(rng, base)
    pp = replace(sp.base_prob_scaled, K=po.K, mdry=po.mdry, Tmin=po.Tmin, deltaMax=po.deltaMax, thetaMax=po.thetaMax,
                 gammaGs=po.gammaGs, omMax=po.omMax, tf_guess=po.tf_guess,
                 model_flags=sp.base_prob_scaled.model_flags | (1 if po.enforce_dp else 0))
    ic = model.disperse_ics(po, 4, 500 + cls, 0.3)
    c = IntegratorCache(pp, npts=4)
    b = ScvxBatch(c, 4, tol=1e-9).init(ic)
    st, nun, dj = b.solve_step()
    sst, its, merit, _ = b.solver_stats()
    assert np.all((sst == 0) | (sst == 5)), sst                      # optimal, or an initial condition outside a path cone
    tr = int(np.nonzero(sst == 0)[0][0])
    x, u, s = b.trajectory()
    rk, cost, it = b.scalars()
    o0 = oscvx.create_initial(po, 4, ic[tr, :3], ic[tr, 3:])
    o1, cnu, cdel = oscvx.solve_step(o0)
    assert rk[tr] == o1.rk and abs(nun[tr] - cnu) < 1e-5 * max(1.0, cnu)
    assert np.abs(x[tr] - o1.x).max() < 5e-5 and np.abs(u[tr] - o1.u).max() < 5e-5 and abs(s[tr] - o1.sigma) < 5e-5, \
        (cls, np.abs(x[tr] - o1.x).max(), np.abs(u[tr] - o1.u).max())
    b.close(); c.close()


def test_masked_trajectories_are_left_alone_and_cost_nothing():
    """Masking (SURVEY 8e "masked/compacted"): a trajectory whose `active` flag is 0 is not stepped -- its iterate, scalars and
    linearisation stay bit for bit what they were (K1 skips it too: its reference point did not move), and the trajectories that ARE
    stepped get exactly what they get in a batch where everything is active."""
    from oracle import model
    po = model.base_prob_scaled()
    B = 12
    ic = model.disperse_ics(po, B, 20261004)
    c, full = _setup(B, ic)
    c2, part = _setup(B, ic)
    full.solve_step(); part.solve_step()                      # one common step so that the masked ones hold a non-trivial state
    st, act, live = part.flags()
    mask = (np.arange(B) % 3 != 0).astype(np.int32)           # every third trajectory is left out
    x0, u0, s0 = part.trajectory(); e0, d0 = part.linearization(); rk0, cost0, it0 = part.scalars()
    part.set_flags(st, mask, mask)
    for n in range(2):
        sf, nuf, djf = full.solve_step()
        sp_, nup, djp = part.solve_step()
    x1, u1, s1 = part.trajectory(); e1, d1 = part.linearization(); rk1, cost1, it1 = part.scalars()
    xf, uf, sff = full.trajectory(); ef, df = full.linearization(); rkf, costf, itf = full.scalars()
    off, on = mask == 0, mask == 1
    assert np.array_equal(x1[off], x0[off]) and np.array_equal(u1[off], u0[off]) and np.array_equal(s1[off], s0[off])
    assert np.array_equal(e1[off], e0[off]) and np.array_equal(d1[off], d0[off])
    assert np.array_equal(rk1[off], rk0[off]) and np.array_equal(it1[off], it0[off])
    assert np.array_equal(x1[on], xf[on]) and np.array_equal(u1[on], uf[on]) and np.array_equal(s1[on], sff[on])
    assert np.array_equal(e1[on], ef[on]) and np.array_equal(d1[on], df[on])
    assert np.array_equal(rk1[on], rkf[on]) and np.array_equal(it1[on], itf[on])
    assert np.array_equal(sp_[on], sf[on])
    full.close(); part.close(); c.close(); c2.close()


def test_retry_ladder_rescues_floor_failures_and_matches_oracle():
    """scvx_solver_opts.retries (default 5): a conic solve that ends on its numerical floor above `tol` is run again from the cold
    start under another step rule before its trajectory is frozen.  Class 39 of tools/k4_fuzz.py (K = 50, 55 % dry mass, glideslope
    28 deg, dynamic-pressure cone on): on the second solve_step three of the 16 dispersed trajectories stall at merit 1.4e-8 ..
    1.9e-8 with a single attempt (CPU twin; the set differs with the rounding, so the device's own single-attempt run says which).
    With the ladder every solve is OPTIMAL, and the rescued trajectories agree with two steps of the independent oracle."""
    import os
    import sys
    from dataclasses import replace
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import k4_fuzz
    from oracle import model, scvx as oscvx
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    rng = np.random.default_rng(1)
    base = model.base_prob_scaled()
    for _ in range(40):
        po = k4_fuzz.draw_class(rng, base)
    assert po.K == 50 and po.enforce_dp
    pp = replace(sp.base_prob_scaled, K=po.K, mdry=po.mdry, Tmin=po.Tmin, deltaMax=po.deltaMax, thetaMax=po.thetaMax,
                 gammaGs=po.gammaGs, omMax=po.omMax, tf_guess=po.tf_guess, model_flags=sp.base_prob_scaled.model_flags | 1)
    ic = model.disperse_ics(po, 16, 539, 0.3)
    c = IntegratorCache(pp, npts=4)
    runs = {}
    for retries in (0, None):
        b = ScvxBatch(c, 16, retries=retries).init(ic)
        sts, its = [], []
        for n in range(2):
            st, nun, dj = b.solve_step()
            sst, sit, merit, _ = b.solver_stats()
            sts.append(st.copy()); its.append(sit.copy())
        runs[retries] = (np.array(sts), np.array(its), b.trajectory())
        b.close()
    st0, it0, _ = runs[0]
    st5, it5, (x5, u5, s5) = runs[None]
    single_failed = np.nonzero((st0 == 3).any(axis=0))[0]
    print("single attempt: frozen trajectories", single_failed.tolist(), "; with the ladder:", np.nonzero((st5 == 3).any(axis=0))[0].tolist(),
          "; iterations of step 2 (ladder):", it5[1].tolist())
    assert not (st5 >= 3).any(), st5                                   # nothing frozen, nothing non-finite
    assert ((st0 >= 3).sum()) >= ((st5 >= 3).sum())
    check = single_failed.tolist()[:2] if single_failed.size else [7, 13]
    for tr in check:
        it = oscvx.create_initial(po, 4, ic[tr, :3], ic[tr, 3:])
        for n in range(2):
            it, cnu, cdel = oscvx.solve_step(it)
        # 5e-4: WHICH trajectory needs the ladder moves with the executor and the rounding, and this class has trajectories whose
        # optimum is that flat -- trajectory 8 at tol 1e-8 sits 2.8e-4 from its own tol-1e-10 answer under every executor, rescued or
        # not, while rescued and unrescued runs of it agree to 4e-7 (tools/diag_w2w4.py, round 4); most are within 5e-5
        assert np.abs(x5[tr] - it.x).max() < 5e-4 and np.abs(u5[tr] - it.u).max() < 5e-4, (tr, np.abs(x5[tr] - it.x).max())
    c.close()


def test_library_communicator_next_to_torch_nccl_group(tmp_path):
    """scvx_comm_create in a process that ALSO holds a live torch.distributed NCCL (= RCCL) group -- what
    bench.py does under torchrun with the default backend -- world of one on this box: the torch group all-reduces on the GPU first
    (its RCCL is initialised), then montecarlo.bootstrap_comm creates the library's own communicator over it and the library's
    all-gather returns this rank's records.  In a fresh child process (the group and the communicator die with it)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "child.py"
    script.write_text('''
import os, sys, ctypes as C
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
c = IntegratorCache(sp.base_prob_scaled, npts=4)
b = ScvxBatch(c, 5).init(mc.disperse_ics(sp.base_prob_scaled, 0, 5, 7))
b.solve_step()
why = mc.bootstrap_comm(c, dist, 0, 1)
assert why is None, why
out = torch.empty((1, 5, b.nrec), dtype=torch.float64, device="cuda")
assert c._L.scvx_allgather_trajectories(b.handle, C.c_void_p(out.data_ptr())) == 0
c.synchronize()
assert np.array_equal(out[0].cpu().numpy(), b.trajectory_record())
t2 = torch.full((4,), 2.0, device="cuda"); dist.all_reduce(t2); torch.cuda.synchronize()      # torch's group still works
assert float(t2[0]) == 2.0
assert c._L.scvx_comm_destroy(c.handle) == 0
dist.destroy_process_group()
print("OK")
''' % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
