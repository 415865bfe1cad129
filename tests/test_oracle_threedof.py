"""BASELINE configs[0]: 3-DoF point-mass landing, K=30, single trajectory, CPU only (solver plumbing)."""
from dataclasses import replace

import numpy as np

from oracle import model, threedof


def test_config0_defaults_solves_with_virtual_acceleration():
    # DescentProblem() defaults (master.jl:65-68) at K=30: one time unit to land from r=(4,4,0) is not flyable,
    # the virtual acceleration ar (penalty 100) makes the SOCP feasible — exactly what the reference's comment says
    p = replace(model.DescentProblem(), K=30)
    sol, o, idx = threedof.solve_initial(p)
    assert sol.status == "optimal" and sol.iters < 60
    assert np.abs(o["r"][:, 0] - p.rIi).max() < 1e-8 and np.abs(o["r"][:, -1]).max() < 1e-8 and np.abs(o["v"][:, -1]).max() < 1e-8
    assert o["nkaR"] > 1.0 and (o["ma"] >= p.mdry - 1e-8).all()
    assert (np.linalg.norm(o["T"], axis=0) <= o["ga"] + 1e-7).all() and (o["ga"] <= p.Tmax + 1e-8).all() and (o["ga"] >= p.Tmin - 1e-8).all()


def test_config0_flyable_case_is_tight_and_feasible():
    # a flyable instance: the lossless-convexification property — the relaxation ||T_k|| <= Gamma_k is tight at the optimum —
    # and no virtual acceleration is used
    p = replace(model.DescentProblem(), K=30, tf_guess=6.0, rIi=np.array([4.0, 2.0, 0.0]), vIi=np.array([-0.5, -0.5, 0.3]),
                mdry=1.0, mwet=2.0, alpha=0.05)
    sol, o, idx = threedof.solve_initial(p)
    assert sol.status == "optimal"
    assert o["nkaR"] < 1e-6
    assert np.abs(np.linalg.norm(o["T"], axis=0) - o["ga"]).max() < 1e-5
    # discrete dynamics of initial_solve.jl:73-77 re-evaluated from the solution
    dt, mu, N = idx["dt"], idx["mu"], p.K
    a = o["T"] / mu + o["ar"] + np.array([[-p.g], [0.0], [0.0]])
    r_next = o["r"][:, :-1] + o["v"][:, :-1] * dt + (a[:, :-1] + 0.5 * a[:, 1:]) * dt**2 / 3
    v_next = o["v"][:, :-1] + 0.5 * (a[:, :-1] + a[:, 1:]) * dt
    assert np.abs(r_next - o["r"][:, 1:]).max() < 1e-8 and np.abs(v_next - o["v"][:, 1:]).max() < 1e-8
    assert np.abs(o["ma"][1:] - (o["ma"][:-1] - p.alpha * (o["ga"][:-1] + o["ga"][1:]) * dt / 2)).max() < 1e-9
    # glideslope and pointing cones
    assert (o["r"][0] / np.tan(np.radians(p.gammaGs)) - np.linalg.norm(o["r"][1:], axis=0)).min() > -1e-7
    assert (o["T"][0] - o["ga"] * np.cos(np.radians(p.thetaMax))).min() > -1e-7
    assert o["ma"][-1] > p.mdry
