"""RocketlandDefns — the reference's problem-definition types, same names and field meaning.

Mirrors master.jl:1-136 of BenChung/SuccessiveConvexification:
    AerodynamicInfo / ExoatmosphericData / AtmosphericData   master.jl:6-16
    DescentProblem (keyword constructor, same defaults)       master.jl:17-71
    ProbInfo                                                  master.jl:73-83
    LinPoint / LinRes                                         master.jl:85-93
    ProblemIteration                                          master.jl:122-134
The Julia shim in julia/ScvxAMD.jl declares the same structs over the same C ABI.
"""
from dataclasses import dataclass, field
import numpy as np

from ._lib import ScvxProblem


def _v(*a):
    return np.array(a, dtype=np.float64)


class AerodynamicInfo:
    pass


class ExoatmosphericData(AerodynamicInfo):
    def __repr__(self):
        return "ExoatmosphericData()"


@dataclass
class AtmosphericData(AerodynamicInfo):
    """drag/lift/trq tables on the load_aerodata axes (aerodynamics.jl:17-21), cos(AoA) fastest."""
    drag_itrp: np.ndarray  # [n_mach][n_aoa]
    lift_itrp: np.ndarray
    trq_itrp: np.ndarray
    force_scalar: float = 1.0
    length_scalar: float = 1.0
    aoa0: float = -1.0
    daoa: float = 1.0 / 90.0
    mach0: float = 0.0
    dmach: float = 0.025


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
    aero: AerodynamicInfo = field(default_factory=ExoatmosphericData)
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
    model_flags: int = 0   # not a reference field: 1 (SCVX_MODEL_DPMAX) enforces the dpMax / rho constraint the reference leaves as a todo;
                           # 2 (SCVX_MODEL_FINS) enables the fin extension: control_dim = 5 (dynamics.jl:60-69 / rocketland.jl:203-209 as commented there)
    finmxf: float = 0.01   # fin extension: bound of |u[4:5]| (rocketland.jl:205)

    @property
    def fins(self) -> bool:
        return bool(int(self.model_flags) & 2)

    @property
    def nu(self) -> int:
        """control_dim: 3, or 5 with the fin extension."""
        return 5 if self.fins else 3

    def to_c(self) -> ScvxProblem:
        s = ScvxProblem()
        for name in ("g", "mdry", "mwet", "Tmin", "Tmax", "deltaMax", "thetaMax", "gammaGs", "omMax", "dpMax",
                     "alpha", "rho", "sos", "wNu", "wID", "wDS", "wCst", "wTviol", "nuTol", "delTol", "tf_guess",
                     "ri", "rh0", "rh1", "rh2", "alph", "bet"):
            setattr(s, name, float(getattr(self, name)))
        s.jB[:] = list(np.asarray(self.jB, float).flatten(order="F"))
        for name, n in (("rTB", 3), ("rFB", 3), ("rIi", 3), ("rIf", 3), ("vIi", 3), ("vIf", 3), ("qBIi", 4),
                        ("qBIf", 4), ("wBi", 3), ("wBf", 3)):
            a = np.asarray(getattr(self, name), float)
            assert a.shape == (n,), name
            getattr(s, name)[:] = list(a)
        s.K, s.imax = int(self.K), int(self.imax)
        s.model_flags = int(getattr(self, "model_flags", 0))
        s.finmxf = float(self.finmxf)
        if isinstance(self.aero, AtmosphericData):
            s.aero_kind = 1
            s.force_scalar, s.length_scalar = float(self.aero.force_scalar), float(self.aero.length_scalar)
        else:
            s.aero_kind = 0
            s.force_scalar = s.length_scalar = 1.0
        return s


@dataclass
class ProbInfo:
    a: float
    g0: float
    sos: float
    jB: np.ndarray
    jBi: np.ndarray
    rTB: np.ndarray
    rFB: np.ndarray
    aero: AerodynamicInfo

    @classmethod
    def from_problem(cls, p: DescentProblem) -> "ProbInfo":
        jB = np.asarray(p.jB, float)
        return cls(p.alpha, p.g, p.sos, jB, np.linalg.inv(jB), np.asarray(p.rTB, float), np.asarray(p.rFB, float),
                   p.aero)


@dataclass
class LinPoint:
    state: np.ndarray    # [14]
    control: np.ndarray  # [3]  ([5] with the fin extension)


@dataclass
class LinRes:
    endpoint: np.ndarray    # [14]
    derivative: np.ndarray  # [14][21]  ([14][25] with the fin extension)


@dataclass
class ProblemIteration:
    problem: DescentProblem
    cache: object
    sigma: float
    about: list  # of LinPoint, K+1
    dynam: list  # of LinRes, K
    model: object
    iter: int
    rk: float
    cost: float
