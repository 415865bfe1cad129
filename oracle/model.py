"""Problem data of the reference, restated (oracle; see oracle/__init__.py for the rules).

Follows (file:line into /root/reference):
    DescentProblem keyword defaults      master.jl:65-70
    ProbInfo                              master.jl:73-83
    SampleProblems.normalize_problem      sample_problems.jl:5-23   (quirks reproduced, see below)
    SampleProblems.base_prob / _aero      sample_problems.jl:25-32
    FirstRound.linear_points              initial_solve.jl:113-129
    Rotations.rotation_between            third-party (Rotations.jl, unpinned): q = normalize([|a||b| + a.b, a x b])

Quirks of normalize_problem kept on purpose: vIf is built from vIi; rFB is scaled
by 1/Ut; omMax is divided by Ut and later used un-converted from degrees; nuTol is not forwarded
(so it falls back to the default 1e-10).
"""
from dataclasses import dataclass, field, replace
import numpy as np


def _v(*a):
    return np.array(a, dtype=np.float64)


@dataclass
class AeroData:
    """AtmosphericData (master.jl:10-16): raw tables on the load_aerodata axes + the two scalars."""
    drag: np.ndarray  # [n_mach][n_aoa]  (cos(AoA) fastest, as the CSV rows are ordered)
    lift: np.ndarray
    trq: np.ndarray
    aoa0: float = -1.0
    daoa: float = 1.0 / 90.0
    mach0: float = 0.0
    dmach: float = 0.025
    force_scalar: float = 1.0
    length_scalar: float = 1.0


@dataclass
class DescentProblem:
    g: float = 1.0
    mdry: float = 1.0
    mwet: float = 2.0
    Tmin: float = 0.3
    Tmax: float = 5.0
    deltaMax: float = 20.0
    thetaMax: float = 90.0
    gammaGs: float = 20.0
    dpMax: float = 50000.0
    omMax: float = 60.0
    jB: np.ndarray = field(default_factory=lambda: np.diag([1e-2, 1e-2, 1e-2]))
    alpha: float = 0.01
    rho: float = 1.225
    rTB: np.ndarray = field(default_factory=lambda: _v(-1e-2, 0, 0))
    rFB: np.ndarray = field(default_factory=lambda: _v(1e-2, 0, 0))
    rIi: np.ndarray = field(default_factory=lambda: _v(4.0, 4.0, 0.0))
    rIf: np.ndarray = field(default_factory=lambda: _v(0.0, 0.0, 0.0))
    vIi: np.ndarray = field(default_factory=lambda: _v(0, -2, 2))
    vIf: np.ndarray = field(default_factory=lambda: _v(-0.1, 0.0, 0.0))
    qBIi: np.ndarray = field(default_factory=lambda: _v(1.0, 0, 0, 0))
    qBIf: np.ndarray = field(default_factory=lambda: _v(1.0, 0, 0, 0))
    wBi: np.ndarray = field(default_factory=lambda: _v(0.0, 0.0, 0.0))
    wBf: np.ndarray = field(default_factory=lambda: _v(0.0, 0, 0))
    aero: object = None  # None = ExoatmosphericData
    K: int = 50
    imax: int = 15
    wNu: float = 1e5
    wID: float = 1e-3
    wDS: float = 1e-1
    wCst: float = 10.0
    wTviol: float = 100.0
    nuTol: float = 1e-10
    delTol: float = 1e-3
    tf_guess: float = 1.0
    ri: float = 1.0
    rh0: float = 0.0
    rh1: float = 0.25
    rh2: float = 0.90
    alph: float = 2.0
    bet: float = 3.2
    sos: float = 5.0
    enforce_dp: bool = False   # build extension: enforce 1/2 rho |v|^2 <= dpMax (the reference leaves it as a todo)
    # fin extension (BUILD-DEFINED): control_dim = 5, u[4:5] = fin force coordinates along fd1 / fd2
    # (dynamics.jl:60-63,66,69 as commented there), |u[4:5]| <= finmxf at every node (rocketland.jl:203-209: finmxf is pinned
    # to 0.01 by a Zeros row in the commented code, in the units of the normalised problem)
    fins: bool = False
    finmxf: float = 0.01

    @property
    def nu(self):
        return 5 if self.fins else 3


def normalize_problem(dp: DescentProblem) -> DescentProblem:
    """sample_problems.jl:5-23, field by field."""
    Ul = float(np.max(dp.rIi))
    Ut = dp.tf_guess
    Um = dp.mwet
    aero = dp.aero
    if aero is not None:  # rescale_aerodata aerodynamics.jl:30-32
        aero = replace(aero, force_scalar=1.0 / (Ul * Um / Ut**2), length_scalar=1.0 / Ul)
    return DescentProblem(
        g=dp.g / (Ul / Ut**2), mdry=dp.mdry / Um, mwet=dp.mwet / Um,
        Tmin=dp.Tmin / (Um * Ul / Ut**2), Tmax=dp.Tmax / (Um * Ul / Ut**2),
        omMax=dp.omMax / Ut, jB=dp.jB * (1.0 / (Um * Ul**2)),
        rTB=dp.rTB / Ul, rIi=dp.rIi / Ul, rIf=dp.rIf / Ul, vIi=dp.vIi / (Ul / Ut),
        vIf=dp.vIi / (Ul / Ut),  # sic: built from vIi (sample_problems.jl:15)
        qBIf=dp.qBIf.copy(), qBIi=dp.qBIi.copy(), wBi=dp.wBi.copy(), wBf=dp.wBf.copy(),
        rFB=dp.rFB / Ut,  # sic (sample_problems.jl:16)
        deltaMax=dp.deltaMax, thetaMax=dp.thetaMax, gammaGs=dp.gammaGs,
        alpha=dp.alpha / (Ut**2 / Ul), K=dp.K, imax=dp.imax, wNu=dp.wNu, wID=dp.wID,
        wDS=dp.wDS, wCst=dp.wCst, wTviol=dp.wTviol, delTol=dp.delTol,
        tf_guess=dp.tf_guess / Ut, ri=dp.ri, rh0=dp.rh0, rh1=dp.rh1, rh2=dp.rh2,
        alph=dp.alph, bet=dp.bet, dpMax=dp.dpMax / (Um / (Ul * Ut**2)), rho=dp.rho / (Um / Ul**3),
        sos=dp.sos / (Ul / Ut), aero=aero, enforce_dp=dp.enforce_dp,
        fins=dp.fins, finmxf=dp.finmxf)   # finmxf: a constant of build_model (rocketland.jl:205), not rescaled


def base_prob(aero=None) -> DescentProblem:
    """sample_problems.jl:26-27 (exo) / :30-31 (aero)."""
    return DescentProblem(
        g=9.82, mwet=66018.0, mdry=65947.0, Tmin=0.1 * 4.686588e6, Tmax=4.686588e6,
        jB=np.diag([72487.03125, 2.0734175e6, 2.0734175e6]), alpha=0.000345,
        rTB=_v(-4.26114, 0, 0), rFB=_v(2.0, 0, 0), rIi=_v(1000.0, 1000.0, 100.0), rIf=_v(0.0, 0.0, 0.0),
        vIi=_v(-100.0, -200.0, 0), sos=352.0, wNu=1e4, aero=aero)


def base_prob_scaled(aero=None) -> DescentProblem:
    return normalize_problem(base_prob(aero))


def base_prob_fin_scaled(aero=None) -> DescentProblem:
    """The sample problem with the fin extension (control_dim = 5; build-defined).  rFB is a length and is scaled
    by 1/Ul here (normalize_problem's 1/Ut, sample_problems.jl:16, would put the fins 2 km from the centre of mass)."""
    b = base_prob(aero)
    return replace(normalize_problem(replace(b, fins=True)), rFB=b.rFB / float(np.max(b.rIi)))


def rotation_between(a, b):
    """Rotations.rotation_between as a scalar-first unit quaternion."""
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    normprod = np.sqrt(a.dot(a) * b.dot(b))
    w = normprod + a.dot(b)
    if abs(w) < 100 * np.finfo(float).eps:
        # perpendicular_vector: any vector orthogonal to a
        k = int(np.argmin(np.abs(a)))
        e = np.zeros(3)
        e[k] = 1.0
        v = np.cross(a, e)
    else:
        v = np.cross(a, b)
    q = np.array([w, v[0], v[1], v[2]])
    return q / np.linalg.norm(q)


def linear_points(p: DescentProblem, rIi=None, vIi=None):
    """initial_solve.jl:113-129 -> x[K+1][14], u[K+1][nu].  vIf follows the problem (already the
    scaled vIi in the samples); a dispersed (rIi, vIi) replaces the problem's initial condition.  (Fin controls start at 0.)"""
    K = p.K
    rIi = p.rIi if rIi is None else np.asarray(rIi, float)
    vIi = p.vIi if vIi is None else np.asarray(vIi, float)
    x = np.zeros((K + 1, 14))
    u = np.zeros((K + 1, p.nu))
    for k in range(K + 1):
        a, b = (K - k) / K, k / K
        mk = a * p.mwet + b * p.mdry
        rk = a * rIi + b * p.rIf
        vk = a * vIi + b * p.vIf
        q = rotation_between([1.0, 0.0, 0.0], -vk)
        x[k, 0] = mk
        x[k, 1:4] = rk
        x[k, 4:7] = vk
        x[k, 7:11] = q
        u[k, 0] = mk * p.g
    return x, u


def disperse_ics(p: DescentProblem, B: int, seed: int, frac: float = 0.1):
    """SURVEY.md §8d dispersion law: rIi*(1+frac*U(-1,1)), vIi*(1+frac*U(-1,1)) per component;
    trajectory b draws from stream b of a Philox generator keyed by `seed`."""
    ic = np.zeros((B, 6))
    for b in range(B):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, b]))
        r = rng.uniform(-1.0, 1.0, size=6)
        ic[b, 0:3] = p.rIi * (1.0 + frac * r[0:3])
        ic[b, 3:6] = p.vIi * (1.0 + frac * r[3:6])
    return ic
