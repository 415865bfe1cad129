"""The discretisation oracle pinned by mathematics (the reference has no golden vectors: SURVEY.md §4).
Runs on CPU."""
import numpy as np
import pytest
from scipy.integrate import solve_ivp

from conftest import random_segments
from oracle import dynamics as od
from oracle import model


@pytest.fixture(scope="module")
def exo():
    p = model.base_prob_scaled()
    return p, od.Params(p)


def _one_segment(p, seed):
    x, u, s = random_segments(p, 1, 1, seed)
    return np.concatenate([x[0, 0], u[0, 0], u[0, 1], [s[0]]])


def test_analytic_jacobians_match_central_differences(exo):
    p, par = exo
    for seed in range(5):
        inp = _one_segment(p, 100 + seed)
        x, u = inp[:14], inp[14:17]
        A, Bu = od.jac(par, x, u)
        h = 1e-6
        Af = np.stack([(od.rhs(par, x + h * e, u) - od.rhs(par, x - h * e, u)) / (2 * h) for e in np.eye(14)], axis=1)
        Bf = np.stack([(od.rhs(par, x, u + h * e) - od.rhs(par, x, u - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
        assert np.abs(A - Af).max() < 5e-9 and np.abs(Bu - Bf).max() < 5e-9
        # structural sparsity the kernel exploits (~48 of 196)
        assert (A != 0).sum() <= 48


def test_rhs_equations(exo):
    """dx_static (dynamics.jl:54-77) on a hand-checkable state: q = identity, w = 0."""
    p, par = exo
    x = np.zeros(14)
    x[0] = 0.9995
    x[4:7] = [0.1, -0.2, 0.05]
    x[7] = 1.0
    u = np.array([0.03, 0.002, -0.001])
    g = od.rhs(par, x, u)
    assert g[0] == pytest.approx(-p.alpha * np.linalg.norm(u))
    assert np.allclose(g[1:4], x[4:7])
    assert np.allclose(g[4:7], u / x[0] - np.array([p.g, 0, 0]))
    assert np.allclose(g[7:11], 0)
    assert np.allclose(g[11:14], np.linalg.inv(p.jB) @ np.cross(p.rTB, u))


def test_rk4_is_fourth_order_and_matches_dop853(exo):
    p, par = exo
    inp = _one_segment(p, 7)
    inp[20] = 8.0  # a realistic time dilation
    dt = 1.0 / 51

    def flow(v):
        def f(t, x):
            lkp = t / dt
            return v[20] * od.rhs(par, x, v[14:17] * (1 - lkp) + v[17:20] * lkp)
        return solve_ivp(f, (0, dt), v[:14], method="DOP853", rtol=1e-13, atol=1e-15).y[:, -1]

    ref = flow(inp)
    errs = [np.abs(od.segment(par, inp, dt, n, with_deriv=False) - ref).max() for n in (1, 2, 4, 8)]
    orders = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert np.all(orders > 3.5), (errs, orders)
    e10, d10 = od.segment(par, inp, dt, 10)
    assert np.abs(e10 - ref).max() < 1e-9
    # derivative of the exactly integrated flow (autodiff_dynamics.jl:74-92 semantics) by central differences
    Dfd = np.stack([(flow(inp + 1e-6 * e) - flow(inp - 1e-6 * e)) / 2e-6 for e in np.eye(21)], axis=1)
    assert np.abs(d10 - Dfd).max() < 5e-8


def test_variational_rk4_is_the_derivative_of_the_discrete_map(exo):
    """RK4 on the variational equations == d(RK4 map)/d(inp) — what sensitivity_zygote (dynamics.jl:311-313) computes."""
    p, par = exo
    inp = _one_segment(p, 11)
    dt = 1.0 / 51
    _, d = od.segment(par, inp, dt, 3)
    Dfd = np.stack([(od.segment(par, inp + 1e-6 * e, dt, 3, False) - od.segment(par, inp - 1e-6 * e, dt, 3, False)) / 2e-6
                    for e in np.eye(21)], axis=1)
    assert np.abs(d - Dfd).max() < 1e-8


def test_identity_columns_and_packing(exo):
    p, par = exo
    B, K = 3, 5
    x, u, s = random_segments(p, B, K, 3)
    e, d = od.linearize(par, x, u, s, 1.0 / (K + 1), 4)
    assert e.shape == (B, K, 14) and d.shape == (B, K, 21, 14)
    # position columns: d x(t)/d r_k = [0; I; 0...] exactly (nothing depends on position)
    for j in range(3):
        col = d[:, :, 1 + j, :]
        expect = np.zeros(14)
        expect[1 + j] = 1.0
        assert np.abs(col - expect).max() == 0.0
    # mass row depends on the controls and sigma only
    assert np.abs(d[:, :, 1:14, 0]).max() == 0.0 and np.all(d[:, :, 0, 0] == 1.0)
    # K2 == endpoint of K1
    assert np.abs(od.propagate(par, x, u, s, 1.0 / (K + 1), 4) - e).max() == 0.0


def test_first_order_hold_and_zero_duration(exo):
    p, par = exo
    inp = _one_segment(p, 5)
    inp[20] = 0.0  # sigma = 0: nothing moves, derivative = [I | 0 | 0 | g*dt-ish]
    e, d = od.segment(par, inp, 1.0 / 51, 4)
    assert np.abs(e - inp[:14]).max() == 0.0
    assert np.abs(d[:, :14] - np.eye(14)).max() == 0.0 and np.abs(d[:, 14:20]).max() == 0.0
    g0 = od.rhs(par, inp[:14], inp[14:17])
    g1 = od.rhs(par, inp[:14], inp[17:20])
    # d x(dt)/d sigma at sigma = 0 is the integral of g along the FOH control: Simpson on a linear-in-u RHS
    gm = od.rhs(par, inp[:14], 0.5 * (inp[14:17] + inp[17:20]))
    simpson = (g0 + 4 * gm + g1) / 6 / 51
    assert np.abs(d[1:, 20] - simpson[1:]).max() < 1e-12   # rows linear in u: one Simpson panel is exact
    assert abs(d[0, 20] - simpson[0]) < 1e-6               # mass row: -alpha*||u(t)|| is not linear in u


def test_gap_to_the_reference_live_first_order_sensitivity(exo):
    """the reference's live path declares jac = 0 (dynamics.jl:270) while paramjac carries the
    true df/d(delta); if the sensitivity solver honours it, derivative = [I 0] + int F'(x(t)) dt — first order in dt.
    The build integrates the exact variational equations; this test makes the deviation explicit: the gap is O(dt^2)
    (ratio ~4 when dt halves) and ~1e-3 relative at the operating point, far above solver tolerances."""
    p, par = exo
    inp = _one_segment(p, 21)
    inp[20] = 7.0

    def first_order(dt, n=400):
        # [I 0] + int_0^dt F'(x(t)) dt with F' = d(sigma g)/d[x,uk,up,sigma] along the exactly integrated state
        acc = np.zeros((14, 21))
        for s in range(n):
            tm = (s + 0.5) / n
            xm = od.segment(par, np.concatenate([inp[:14], inp[14:17], inp[14:17] * (1 - tm) + inp[17:20] * tm, [inp[20]]]),
                            dt * tm, 8, with_deriv=False) if tm > 0 else inp[:14]
            u = inp[14:17] * (1 - tm) + inp[17:20] * tm
            A, Bu = od.jac(par, xm, u)
            F = np.zeros((14, 21))
            F[:, :14] = inp[20] * A
            F[:, 14:17] = inp[20] * Bu * (1 - tm)
            F[:, 17:20] = inp[20] * Bu * tm
            F[:, 20] = od.rhs(par, xm, u)
            acc += F * (dt / n)
        acc[:, :14] += np.eye(14)
        return acc

    gaps = []
    for dt in (1.0 / 51, 0.5 / 51):
        _, d = od.segment(par, inp, dt, 10)
        gaps.append(np.abs(d - first_order(dt)).max())
    assert 3.0 < gaps[0] / gaps[1] < 5.0, gaps      # second order in dt
    assert gaps[0] > 1e-4                            # not a rounding-level difference at the operating point
