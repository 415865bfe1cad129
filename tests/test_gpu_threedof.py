"""K0 on the MI355X: the batched 3-DoF lossless-convexification initialiser (FirstRound.solve_initial,
initial_solve.jl:17-110; BASELINE configs[0]) against
  * the INDEPENDENT oracle oracle/threedof.py (explicit rows of initial_solve.jl on the generic IPM of oracle/ipm.py),
  * the CPU twin of the same solver core (oracle/port.py::threedof),
  * size-independent properties at a Monte-Carlo batch: every row of initial_solve.jl:59-88 re-evaluated from what the
    kernel returns, the lossless-convexification tightness |T| = ga,
and the 6-DoF start it produces (initial_solve.jl:90-107) against the oracle's discretisation of the same LinPoints.
"""
from dataclasses import replace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("T", "r", "v", "ma", "ga", "kaR", "ar", "nkaR")


def _product(po):
    """the product-side DescentProblem with the oracle problem's numbers"""
    from successiveconvexification_amd.defns import DescentProblem
    p = DescentProblem()
    for f in ("g", "mdry", "mwet", "Tmin", "Tmax", "deltaMax", "thetaMax", "gammaGs", "omMax", "alpha", "K", "tf_guess", "imax"):
        setattr(p, f, getattr(po, f))
    for f in ("rIi", "vIi", "rIf", "vIf", "rTB", "rFB", "jB", "qBIf", "wBi", "wBf"):
        setattr(p, f, np.array(getattr(po, f), float))
    return p


def _cache(po, npts=10):
    from successiveconvexification_amd.dynamics import IntegratorCache
    return IntegratorCache(_product(po), npts=npts)


def _config0():
    from oracle import model
    return replace(model.DescentProblem(), K=30)


def _flyable(K=30):
    from oracle import model
    return replace(model.DescentProblem(), K=K, tf_guess=6.0, rIi=np.array([4.0, 2.0, 0.0]), vIi=np.array([-0.5, -0.5, 0.3]),
                   mdry=1.0, mwet=2.0, alpha=0.05)


def _T_of(tnorm, q):
    """the thrust vector behind a reference-convention start: q = rotation_between(e1, -T), |T| = tnorm"""
    w, x, y, z = q
    e1_rot = np.array([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)])   # R(q) e1 = -T / |T|
    return -tnorm * e1_rot


def _linf(sol, b, o):
    return max(np.abs(np.asarray(sol[k][b]) - np.asarray(o[k])).max() for k in KEYS)


@pytest.mark.parametrize("make", [_config0, _flyable])
def test_device_matches_independent_oracle_and_twin(make):
    from oracle import port, threedof
    from successiveconvexification_amd import first_round
    po = make()
    ref, o, _ = threedof.solve_initial(po)
    assert ref.status == "optimal"
    c = _cache(po)
    sol, st, info = first_round.solve_initial_batch(c)
    assert st[0] == 0, (st, info)
    assert int(info[0, 0]) == ref.iters
    assert abs(info[0, 1] - ref.pobj) <= 1e-9 * max(1.0, abs(ref.pobj))
    assert _linf(sol, 0, o) < 1e-7
    tw, tst, tinfo = port.threedof(po)
    assert tst[0] == 0 and int(tinfo[0, 0]) == int(info[0, 0])
    assert _linf(sol, 0, {k: tw[k][0] for k in KEYS}) < 1e-8   # same arithmetic up to the order of the lane reductions


def test_device_dispersed_batch_other_horizon():
    from oracle import model, port, threedof
    from successiveconvexification_amd import first_round
    po = _flyable(K=50)
    B = 64
    ic = model.disperse_ics(po, B, 20261004)
    c = _cache(po)
    sol, st, info = first_round.solve_initial_batch(c, ic)
    assert np.all(st == 0), (st, info[:, 3:])
    tw, tst, tinfo = port.threedof(po, ic)
    assert np.all(tst == 0)
    assert np.abs(info[:, 1] - tinfo[:, 1]).max() < 1e-9
    for b in (0, 31, 63):
        ref, o, _ = threedof.solve_initial(replace(po, rIi=ic[b, :3], vIi=ic[b, 3:]))
        assert ref.status == "optimal"
        assert abs(info[b, 1] - ref.pobj) <= 1e-8 * max(1.0, abs(ref.pobj))
        assert _linf(sol, b, o) < 5e-5   # the fuel-optimal thrust profile is flat along some directions


@pytest.mark.parametrize("K", [1, 3])
def test_device_tiny_horizons(K):
    from oracle import port
    from successiveconvexification_amd import first_round
    po = _flyable(K=K)
    c = _cache(po)
    sol, st, info = first_round.solve_initial_batch(c)
    tw, tst, tinfo = port.threedof(po)
    assert st[0] == 0 and tst[0] == 0 and abs(info[0, 1] - tinfo[0, 1]) < 1e-8
    # K = 3 ends on the numerical floor (gap 2e-8) on both sides, an iteration apart: the flat thrust directions differ by 5e-5
    assert _linf(sol, 0, {k: tw[k][0] for k in KEYS}) < 2e-4


def test_device_long_horizon_K100():
    """K = 100: the long cone has 102 rows (more than a wavefront: lanes own several), the band 2,230 positions."""
    from oracle import model, port, threedof
    from successiveconvexification_amd import first_round
    po = _flyable(K=100)
    ic = model.disperse_ics(po, 6, 3)
    c = _cache(po)
    sol, st, info = first_round.solve_initial_batch(c, ic)
    assert np.all(st == 0), (st, info)
    tw, tst, tinfo = port.threedof(po, ic)
    assert np.all(tst == 0) and np.abs(info[:, 1] - tinfo[:, 1]).max() < 1e-9
    ref, o, _ = threedof.solve_initial(replace(po, rIi=ic[2, :3], vIi=ic[2, 3:]))
    assert ref.status == "optimal" and abs(info[2, 1] - ref.pobj) <= 1e-8 * max(1.0, abs(ref.pobj))
    assert _linf(sol, 2, o) < 5e-5


def test_device_monte_carlo_batch_properties():
    """B = 4096 dispersed initial conditions (more trajectories than workgroups in flight: the persistent blocks stride):
    every one optimal, and every constraint of initial_solve.jl:59-88 holds for what the kernel returns."""
    from oracle import model
    from successiveconvexification_amd import first_round
    po = _flyable(K=30)
    B = 4096
    ic = model.disperse_ics(po, B, 7)
    c = _cache(po)
    sol, st, info = first_round.solve_initial_batch(c, ic)
    assert np.all(st == 0), np.unique(st, return_counts=True)
    assert info[:, 0].max() <= 40 and info[:, 3].max() < 1e-9 and info[:, 4].max() < 1e-9
    N, dt = po.K, po.tf_guess / po.K
    mu = np.array([((N - k) / N) * po.mwet + (k / N) * po.mdry for k in range(N + 1)])
    T, r, v, ma, ga, kaR, ar, nkaR = (sol[k] for k in KEYS)
    # boundary rows (:59-65)
    assert np.abs(r[:, :, 0] - ic[:, :3]).max() < 1e-8 and np.abs(v[:, :, 0] - ic[:, 3:]).max() < 1e-8
    assert np.abs(ma[:, 0] - po.mwet).max() < 1e-8
    assert np.abs(r[:, :, -1]).max() < 1e-8 and np.abs(v[:, :, -1]).max() < 1e-8 and np.abs(T[:, 1:, -1]).max() < 1e-8
    # recursions (:72-78)
    a = T / mu + ar + np.array([-po.g, 0.0, 0.0])[None, :, None]
    assert np.abs(ma[:, 1:] - (ma[:, :-1] - po.alpha * (ga[:, :-1] + ga[:, 1:]) * dt / 2)).max() < 1e-8
    assert np.abs(r[:, :, 1:] - (r[:, :, :-1] + v[:, :, :-1] * dt + (a[:, :, :-1] + 0.5 * a[:, :, 1:]) * dt**2 / 3)).max() < 1e-8
    assert np.abs(v[:, :, 1:] - (v[:, :, :-1] + 0.5 * (a[:, :, :-1] + a[:, :, 1:]) * dt)).max() < 1e-8
    # cones (:69-70, :80-88)
    assert (ma >= po.mdry - 1e-8).all() and (ga >= po.Tmin - 1e-8).all() and (ga <= po.Tmax + 1e-8).all()
    assert (T[:, 0] - ga * np.cos(np.radians(po.thetaMax)) > -1e-7).all()
    assert (r[:, 0] / np.tan(np.radians(po.gammaGs)) - np.linalg.norm(r[:, 1:], axis=1) > -1e-7).all()
    assert (ga - np.linalg.norm(T, axis=1) > -1e-7).all() and (kaR - np.linalg.norm(ar, axis=1) > -1e-7).all()
    assert (nkaR - np.linalg.norm(kaR, axis=1) > -1e-7).all()
    # flyable: the relaxation is tight and no virtual acceleration is bought
    assert np.abs(np.linalg.norm(T, axis=1) - ga).max() < 1e-5 and nkaR.max() < 1e-6


def test_device_reports_infeasible():
    from oracle import model
    from successiveconvexification_amd import first_round
    po = model.base_prob_scaled()   # tf_guess = 1: Tmin over the whole horizon costs more fuel than mwet - mdry
    c = _cache(po)
    sol, st, info = first_round.solve_initial_batch(c)
    assert st[0] == 5 and info[0, 3] > 1e-6
    with pytest.raises(RuntimeError, match="infeasible"):
        first_round.solve_initial(c.problem, c)


def test_batch_init_from_threedof_and_first_steps():
    """Rocketland.create_initial from FirstRound.solve_initial: the trajectory records are the LinPoints of
    initial_solve.jl:90-105 built from the independent oracle's optimum, their linearisation is the C oracle's, an
    infeasible trajectory keeps the straight line, solve_step runs from there and reset returns to the same start."""
    from oracle import dynamics as od
    from oracle import model, threedof
    from successiveconvexification_amd import first_round
    from successiveconvexification_amd.batch import ScvxBatch
    po = _flyable(K=30)
    B = 3
    ic = model.disperse_ics(po, B, 11)
    ic[2, :3] *= 40.0    # far outside what tf_guess allows with Tmax: the virtual acceleration is bought, still "optimal"
    c = _cache(po)
    b = ScvxBatch(c, B)
    st3 = b.init_threedof(ic)
    assert np.all(st3 == 0)
    x, u, s = b.trajectory()
    assert np.all(s == po.tf_guess)
    for t in range(B):
        ref, o, _ = threedof.solve_initial(replace(po, rIi=ic[t, :3], vIi=ic[t, 3:]))
        assert ref.status == "optimal"
        assert np.abs(x[t, :, 0] - o["ma"]).max() < 1e-6 and np.abs(x[t, :, 1:4] - o["r"].T).max() < 1e-5
        assert np.abs(x[t, :, 4:7] - o["v"].T).max() < 1e-5 and np.abs(x[t, :, 11:14]).max() == 0.0
        assert np.abs(u[t, :, 0] - np.linalg.norm(o["T"], axis=0)).max() < 1e-5 and np.abs(u[t, :, 1:]).max() == 0.0
        for k in range(po.K + 1):
            q = first_round.rotation_between([1, 0, 0], -o["T"][:, k])
            assert min(np.abs(x[t, k, 7:11] - q).max(), np.abs(x[t, k, 7:11] + q).max()) < 1e-4
        assert np.abs(np.linalg.norm(x[t, :, 7:11], axis=1) - 1.0).max() < 1e-14
    e, d = b.linearization()
    e_ref, d_ref = od.linearize(od.Params(po), x, u, s, 1.0 / (po.K + 1), 10)
    assert np.abs(e - e_ref).max() < 1e-11 and np.abs(d - d_ref).max() < 1e-10
    st, nun, dj = b.solve_step()
    assert np.all((st == 1) | (st == 2) | (st == 0)), st
    x1, _, _ = b.trajectory()
    b.reset()
    x0, u0, s0 = b.trajectory()
    assert np.array_equal(x0, x) and np.array_equal(u0, u)
    # attitude option: the body axis along +T instead of the reference's -T; everything else identical
    b2 = ScvxBatch(c, B)
    assert np.all(b2.init_threedof(ic, align_thrust=True) == 0)
    x2, u2, _ = b2.trajectory()
    assert np.array_equal(x2[:, :, :7], x[:, :, :7]) and np.array_equal(u2, u)
    for k in (0, po.K // 2, po.K):
        q = first_round.rotation_between([1, 0, 0], _T_of(u[0, k, 0], x[0, k, 7:11]))
        assert min(np.abs(x2[0, k, 7:11] - q).max(), np.abs(x2[0, k, 7:11] + q).max()) < 1e-12
    b2.close()
    # an infeasible 3-DoF problem keeps the straight-line guess
    pi = model.base_prob_scaled()
    ci = _cache(pi)
    bi = ScvxBatch(ci, 2)
    sti = bi.init_threedof()
    assert np.all(sti == 5)
    xi, ui, _ = bi.trajectory()
    xo, uo = model.linear_points(pi, pi.rIi, pi.vIi)
    assert np.abs(xi[0] - xo).max() < 1e-14 and np.abs(ui[1] - uo).max() < 1e-14


def test_device_random_instances_match_twin():
    """Random problem classes (tools/k0_fuzz.py draws 60 of them): statuses, iteration counts and objectives of device and twin."""
    from oracle import model, port
    from successiveconvexification_amd import first_round
    rng = np.random.default_rng(3)
    for n in range(8):
        po = replace(model.DescentProblem(), K=int(rng.choice([8, 20, 30, 45])), tf_guess=float(rng.uniform(2.0, 10.0)), mdry=1.0,
                     mwet=float(rng.uniform(1.2, 3.0)), alpha=float(rng.uniform(0.01, 0.2)), Tmax=float(rng.uniform(2.0, 8.0)),
                     Tmin=float(rng.uniform(0.1, 0.8)), thetaMax=float(rng.choice([30.0, 60.0, 90.0])),
                     gammaGs=float(rng.choice([10.0, 20.0, 35.0])),
                     rIi=np.array([rng.uniform(2, 6), rng.uniform(-3, 3), rng.uniform(-1, 1)]),
                     vIi=np.array([rng.uniform(-1.5, 0.2), rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)]))
        ic = model.disperse_ics(po, 16, 200 + n)
        c = _cache(po)
        sol, st, info = first_round.solve_initial_batch(c, ic)
        tw, tst, tinfo = port.threedof(po, ic)
        assert np.array_equal(st, tst), (n, st, tst)
        ok = st == 0
        assert np.all(np.isin(st, (0, 5)))
        if ok.any():
            assert np.abs(info[ok, 0] - tinfo[ok, 0]).max() <= 1
            assert (np.abs(info[ok, 1] - tinfo[ok, 1]) / np.maximum(1.0, np.abs(tinfo[ok, 1]))).max() < 1e-8
        c.close()


def test_device_pointer_entry_point():
    """scvx_threedof_solve_dev on device arrays (another framework's tensors), enqueued on the context's stream."""
    import ctypes as C
    import torch
    from oracle import model
    from successiveconvexification_amd import first_round
    po = _flyable(K=30)
    B = 5
    ic = model.disperse_ics(po, B, 5)
    c = _cache(po)
    ref, st, info = first_round.solve_initial_batch(c, ic)
    L = c._L
    n = L.scvx_threedof_record_doubles(po.K)
    d_ic = torch.tensor(ic, dtype=torch.float64, device="cuda")
    d_sol = torch.zeros((B, n), dtype=torch.float64, device="cuda")
    d_info = torch.zeros((B, 6), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert L.scvx_threedof_solve_dev(c.handle, B, C.c_void_p(d_ic.data_ptr()), None, C.c_void_p(d_sol.data_ptr()),
                                     C.c_void_p(d_info.data_ptr())) == 0
    c.synchronize()
    raw = d_sol.cpu().numpy()
    inf = d_info.cpu().numpy()
    assert np.array_equal(inf[:, 0], st.astype(float)) and np.array_equal(inf[:, 1:], info)
    assert np.array_equal(raw[:, -1], ref["nkaR"]) and np.array_equal(raw[:, :-1].reshape(B, po.K + 1, 15)[:, :, 6], ref["ma"])


def test_first_round_solve_initial_mirror():
    from successiveconvexification_amd import first_round
    po = _flyable(K=30)
    c = _cache(po)
    pts, lin = first_round.solve_initial(c.problem, c)
    assert len(pts) == po.K + 1 and len(lin) == po.K
    assert abs(pts[0].state[0] - po.mwet) < 1e-8 and np.abs(pts[0].state[1:4] - po.rIi).max() < 1e-8
    assert np.abs(pts[-1].state[1:7]).max() < 1e-8 and pts[-1].control[1] == 0.0
