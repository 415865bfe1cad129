"""The fin extension (control_dim = 5; BUILD-DEFINED from the reference's commented-out code) in the oracle:
dynamics.jl:60-69 (fd1, fd2, ff, torque at rFB) and rocketland.jl:203-209 (|u[4:5]| <= finmxf).  CPU only."""
from dataclasses import replace

import numpy as np
import pytest

from oracle import dynamics as od, model, port, scvx as oscvx, socp


def _rand_state(rng):
    x = np.concatenate([[rng.uniform(.999, 1)], rng.uniform(0, 1, 3), rng.uniform(-.2, .2, 3), rng.normal(size=4), rng.uniform(-.1, .1, 3)])
    x[7:11] /= np.linalg.norm(x[7:11])
    u = np.concatenate([[0.03, 0.004, -0.003], rng.uniform(-.01, .01, 2)])
    return x, u


@pytest.fixture(scope="module")
def aero():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lift_drag_tables.npz"))
    return model.AeroData(z["drag"], z["lift"], z["torque"])


def test_fin_force_follows_the_commented_reference_lines():
    """ff = u4 fd1 + u5 fd2 with fd1 = normalize((C e2) x v), fd2 = fd1 x v enters the acceleration as ff / m and the rate
    equation as Jinv (rFB x ff); with u4 = u5 = 0 the model is the reference's live one."""
    p3 = model.base_prob_scaled()
    p5 = model.base_prob_fin_scaled()
    par3, par5 = od.Params(p3), od.Params(p5)
    rng = np.random.default_rng(3)
    x, u = _rand_state(rng)
    g0 = od.rhs(par5, x, np.concatenate([u[:3], [0, 0]]))
    assert np.array_equal(g0, od.rhs(par3, x, u[:3]))
    q, v = x[7:11], x[4:7]
    f1, f2, _, _ = od.fin_dirs(q, v)
    C = np.array([[1 - 2 * (q[2]**2 + q[3]**2), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2])],
                  [2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1]**2 + q[3]**2), 2 * (q[2] * q[3] - q[0] * q[1])],
                  [2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1]**2 + q[2]**2)]])
    n = np.cross(C @ [0, 1, 0], v)
    assert np.allclose(f1, n / np.linalg.norm(n), atol=1e-15) and np.allclose(f2, np.cross(f1, v), atol=1e-15)
    ff = u[3] * f1 + u[4] * f2
    g = od.rhs(par5, x, u)
    assert np.allclose(g[4:7] - g0[4:7], ff / x[0], atol=1e-15)
    assert np.allclose(g[11:14] - g0[11:14], np.linalg.inv(p5.jB) @ np.cross(p5.rFB, ff), rtol=1e-12)
    assert np.array_equal(np.delete(g, [4, 5, 6, 11, 12, 13]), np.delete(g0, [4, 5, 6, 11, 12, 13]))


@pytest.mark.parametrize("with_aero", [False, True])
def test_fin_jacobians_match_central_differences(with_aero, aero):
    p = model.base_prob_fin_scaled(aero if with_aero else None)
    par = od.Params(p)
    rng = np.random.default_rng(11)
    for _ in range(10):
        x, u = _rand_state(rng)
        A, Bu = od.jac(par, x, u)
        assert Bu.shape == (14, 5)
        h = 1e-6
        Af = np.stack([(od.rhs(par, x + h * e, u) - od.rhs(par, x - h * e, u)) / (2 * h) for e in np.eye(14)], axis=1)
        Bf = np.stack([(od.rhs(par, x, u + h * e) - od.rhs(par, x, u - h * e)) / (2 * h) for e in np.eye(5)], axis=1)
        assert np.abs(A - Af).max() < 5e-9 * max(1.0, np.abs(Af).max())
        assert np.abs(Bu - Bf).max() < 5e-9 * max(1.0, np.abs(Bf).max())


def test_fin_segment_derivative_is_the_derivative_of_the_discrete_map(aero):
    p = model.base_prob_fin_scaled(aero)
    par = od.Params(p)
    rng = np.random.default_rng(5)
    x, u = _rand_state(rng)
    inp = np.concatenate([x, u, 1.1 * u, [1.3]])
    e, d = od.segment(par, inp, 1 / 51, 10)
    assert d.shape == (14, 25)
    df = np.stack([(od.segment(par, inp + 1e-6 * c, 1 / 51, 10, False) - od.segment(par, inp - 1e-6 * c, 1 / 51, 10, False)) / 2e-6
                   for c in np.eye(25)], axis=1)
    assert np.abs(d - df).max() < 2e-8 * max(1.0, np.abs(df).max())
    # and the RK4 endpoint converges at 4th order towards a fine integration
    fine = od.segment(par, inp, 1 / 51, 160, False)
    e1, e2 = np.abs(od.segment(par, inp, 1 / 51, 2, False) - fine).max(), np.abs(od.segment(par, inp, 1 / 51, 4, False) - fine).max()
    assert e2 < e1 / 10


def test_fin_socp_sizes_and_rows():
    """build_model with the commented fin rows enabled: 2 more u and du rows per node, K+1 finmxf variables pinned by K+1
    equalities, K+1 cones of dimension 3."""
    p = replace(model.base_prob_fin_scaled(), K=30)
    it = oscvx.create_initial(p, 4)
    c, A, b, G, h, l, q, ix = socp.build(p, it.x, it.u, it.endpoint, it.deriv, it.rk)
    K = p.K
    p3 = replace(model.base_prob_scaled(), K=30)
    it3 = oscvx.create_initial(p3, 4)
    c3, A3, b3, G3, h3, l3, q3, ix3 = socp.build(p3, it3.x, it3.u, it3.endpoint, it3.deriv, it3.rk)
    assert len(c) == len(c3) + 4 * (K + 1) + (K + 1)            # u, du rows + finmxf
    assert A.shape[0] == A3.shape[0] + 2 * (K + 1) + (K + 1)     # control_base rows + the finmxf pins
    assert l == l3 and len(q) == len(q3) + (K + 1) and sorted(q)[:1] == [2]
    assert q.count(3) == q3.count(3) + (K + 1)
    assert sum(q) == sum(q3) + 3 * (K + 1) + 2 * (K + 1)         # the fin cones + two more rows of the trust-region cone per node


@pytest.mark.parametrize("with_aero", [False, True])
def test_fin_twin_matches_independent_oracle(with_aero, aero):
    """The solver core instantiated for control_dim = 5 (CPU twin, same source as the kernel) against the independent IPM on the
    explicit build_model rows."""
    p = model.base_prob_fin_scaled(aero if with_aero else None)
    it0 = oscvx.create_initial(p, 10)
    r = port.socp(p, it0.x[None], it0.u[None], it0.endpoint[None], it0.deriv[None], 100.0)
    sol, ix = oscvx.solve_socp(it0)
    assert r["status"][0] == 0 and sol.status == "optimal"
    assert np.abs(it0.x + r["dx"][0] - sol.x[ix.xv].T).max() < 2e-5
    assert np.abs(it0.u + r["du"][0] - sol.x[ix.uv].T).max() < 2e-5
    assert abs(r["ds"][0] - sol.x[ix.dsig]) < 2e-5
    fin = (it0.u + r["du"][0])[:, 3:]
    assert np.linalg.norm(fin, axis=1).max() < p.finmxf + 1e-7 and np.linalg.norm(fin, axis=1).max() > 0.5 * p.finmxf   # the cone is used


def test_twin_config5_full_run_matches_the_oracle_runs(aero_tables):
    """The CPU suite's copy of tests/test_gpu_fins.py::test_config5_full_run_matches_oracle_on_4_dispersed_trajectories: the solver core compiled
    for the host on the four fixture trajectories of BASELINE configs[4] as named (aero + fins, K = 100), complete solve_problem runs."""
    import os
    from dataclasses import replace
    from conftest import GOLDEN
    from oracle import model, port
    g = np.load(os.path.join(GOLDEN, "oracle_scvx_config5_batch4_tol1e-08.npz"))
    d, l, t = aero_tables
    p = replace(model.base_prob_fin_scaled(model.AeroData(d, l, t)), K=100)
    assert np.array_equal(model.disperse_ics(p, int(g["B"]), int(g["seed"]))[g["index"]], g["ic"]) and float(g["tol"]) == 1e-8
    log = g["log"]
    n = log.shape[1]
    o = port.scvx_steps(p, g["ic"], n, nsub=10, tol=1e-8, warm_start=True, nthreads=0)
    assert all((np.asarray(s) == 0).all() for s in o["status"])
    with np.errstate(invalid="ignore"):
        rej_oracle = (log[:, :, 6] < p.rh0).T
    assert np.array_equal(np.asarray(o["rejected"]).astype(bool), rej_oracle) and np.array_equal(o["rk"], log[:, -1, 3])
    dx = np.abs(o["x"] - g["xs"][:, -1])
    assert dx[..., :7].max() < 1e-5 and dx[..., 7:].max() < 5e-4 and np.abs(o["u"] - g["us"][:, -1]).max() < 5e-4
    assert np.abs(o["sigma"] - log[:, -1, 5]).max() < 1e-4
