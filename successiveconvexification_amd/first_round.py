"""FirstRound — initial_solve.jl:113-135 of the reference (the straight-line initial guess)."""
import numpy as np

from .defns import DescentProblem, LinPoint
from . import dynamics


def rotation_between(a, b):
    """Rotations.rotation_between (third-party) as a scalar-first unit quaternion."""
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    w = np.sqrt(a.dot(a) * b.dot(b)) + a.dot(b)
    if abs(w) < 100 * np.finfo(float).eps:
        e = np.zeros(3)
        e[int(np.argmin(np.abs(a)))] = 1.0
        v = np.cross(a, e)
    else:
        v = np.cross(a, b)
    q = np.array([w, v[0], v[1], v[2]])
    return q / np.linalg.norm(q)


def linear_points(problem: DescentProblem):
    """initial_solve.jl:113-129 -> K+1 LinPoints."""
    K = problem.K
    pts = []
    for k in range(K + 1):
        mk = (K - k) / K * problem.mwet + (k / K) * problem.mdry
        rIk = (K - k) / K * problem.rIi + (k / K) * problem.rIf
        vIk = (K - k) / K * problem.vIi + (k / K) * problem.vIf
        q = rotation_between([1, 0, 0], -vIk)
        pts.append(LinPoint(np.concatenate([[mk], rIk, vIk, q, [0.0, 0, 0]]), np.array([mk * problem.g, 0, 0])))
    return pts


def linear_initial(problem: DescentProblem, cache):
    """initial_solve.jl:131-135"""
    pts = linear_points(problem)
    return pts, dynamics.linearize_dynamics(pts, problem.tf_guess, 1 / (problem.K + 1), cache)
