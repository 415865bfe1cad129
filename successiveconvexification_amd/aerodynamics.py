"""Aerodynamics — table loading side of aerodynamics.jl (the force model itself runs in the kernel).

    load_aerodata(liftdrag_csv)      aerodynamics.jl:11-28  columns aoa,mach,drag,lift,torque;
                                     reshape(col, 181, 61): cos(AoA) fastest, Mach slowest
    rescale_aerodata(data,Ul,Ut,Um)  aerodynamics.jl:30-36
The cubic B-spline prefilter (Interpolations.jl Cubic(Line(OnGrid()))) is applied inside
scvx_set_aero_table when the table is uploaded.

    aero_force(data, bv, vel, spds)  aerodynamics.jl:38-58, the NUMERIC method (T <: Number): what a caller of the reference gets when it
                                     evaluates the aerodynamic force outside the optimiser (aero/TestFlight.jl-style analysis, plots) --
                                     drag only when |bv . vel / |vel|| >= 0.95, otherwise drag + lift and the aerodynamic torque.
                                     Host-side, for analysis; the SCvx loop uses the symbolic method's form (aerodynamics.jl:60-77)
                                     inside the kernels, where tau_aero is dropped exactly as dynamics.jl:69 drops it.
"""
from dataclasses import replace
import numpy as np

from .defns import AtmosphericData, ExoatmosphericData


def load_aerodata(liftdrag: str, finforce=None) -> AtmosphericData:
    if liftdrag.endswith(".npz"):
        z = np.load(liftdrag)
        return AtmosphericData(z["drag"], z["lift"], z["torque"], 1.0, 1.0)
    data = np.genfromtxt(liftdrag, delimiter=",", names=True)
    n_aoa, n_mach = 181, 61  # cosd(180):1/90:cosd(0) x 0:0.025:1.5
    if data.shape[0] != n_aoa * n_mach:
        raise ValueError(f"expected {n_aoa * n_mach} rows, got {data.shape[0]}")

    def tab(name):
        return np.ascontiguousarray(data[name].reshape(n_mach, n_aoa))

    if finforce is not None:
        load_fin_table(finforce)   # aerodynamics.jl:23-26: the reference reads fin.csv here and uses it nowhere; neither does this
    return AtmosphericData(tab("drag"), tab("lift"), tab("torque"), 1.0, 1.0)


def load_fin_table(finforce: str):
    """aero/fin.csv (columns lift,drag,mach,aoa: 60 Mach numbers x 901 fin deflections, aero/AeroTable.jl:94-112) as
    (mach [60], aoa [901], lift [60][901], drag [60][901]).  The reference loads the file and drops it
    (aerodynamics.jl:23-26; the fin-force model is commented out, dynamics.jl:60-69) -- it is parsed here so that a
    malformed file fails where the reference's CSV.read would."""
    data = np.genfromtxt(finforce, delimiter=",", names=True)
    for col in ("lift", "drag", "mach", "aoa"):
        if col not in data.dtype.names:
            raise ValueError(f"{finforce}: no column '{col}'")
    mach = np.unique(data["mach"])
    aoa = np.unique(data["aoa"])
    if data.shape[0] != mach.size * aoa.size:
        raise ValueError(f"{finforce}: {data.shape[0]} rows are not a {mach.size} x {aoa.size} (mach x aoa) grid")
    order = np.lexsort((data["aoa"], data["mach"]))
    shape = (mach.size, aoa.size)
    return mach, aoa, data["lift"][order].reshape(shape), data["drag"][order].reshape(shape)


def fin_force_bound(fin_table, mach: float, force_scalar: float = 1.0) -> float:
    """Largest fin force the table offers at `mach`, max over the deflection axis of |lift| (linear in Mach between the table's
    rows, flat outside), times the AtmosphericData force scalar: a data-derived value for DescentProblem.finmxf, the bound of
    the fin cone |u[4:5]| <= finmxf.  (The reference's commented code pins finmxf to the constant 0.01, rocketland.jl:205,
    and never uses fin.csv; this is the one place this build gives the file a meaning, and it is opt-in:
    sample_problems.base_prob_fin_scaled(aero, fin_table=...).)"""
    mach_ax, _, lift, _ = fin_table
    peak = np.abs(lift).max(axis=1)
    return float(np.interp(mach, mach_ax, peak)) * float(force_scalar)


def rescale_aerodata(data, Ul: float, Ut: float, Um: float):
    if isinstance(data, ExoatmosphericData):
        return data
    return replace(data, force_scalar=1 / (Ul * Um / Ut**2), length_scalar=1 / Ul)


def _table_interpolant(tab, data):
    """Interpolations.jl `extrapolate(scale(interpolate(A, BSpline(Cubic(Line(OnGrid())))), aoa, mach), Flat())` of one table: the cubic spline
    interpolant with vanishing second derivative at the first and last grid point in both directions (tensor product, separable), arguments
    clamped to the grid.  tab: [n_mach][n_aoa]."""
    from scipy.interpolate import make_interp_spline
    tab = np.asarray(tab, float)
    n_mach, n_aoa = tab.shape
    aoa = data.aoa0 + data.daoa * np.arange(n_aoa)
    mach = data.mach0 + data.dmach * np.arange(n_mach)
    along_aoa = make_interp_spline(aoa, tab, k=3, bc_type="natural", axis=1)   # a spline in cos(AoA) for every Mach row

    def ev(cos_aoa, m):
        col = along_aoa(min(max(cos_aoa, aoa[0]), aoa[-1]))                    # values on the Mach grid at this cos(AoA)
        return float(make_interp_spline(mach, col, k=3, bc_type="natural")(min(max(m, mach[0]), mach[-1])))
    return ev


def aero_force(data, bv, vel, spds: float):
    """Aerodynamics.aero_force(data::AtmosphericData, bv, vel, spds) for numbers (aerodynamics.jl:38-58): (force, torque).
    bv: the body axis in the inertial frame, vel: the velocity, spds: the speed of sound.  The reference's test compares `dp` (not the
    clamped cosine) with 0.95; kept.  ExoatmosphericData: zero force (aerodynamics.jl:79-81; that method returns the force only)."""
    bv = np.asarray(bv, float)
    vel = np.asarray(vel, float)
    if isinstance(data, ExoatmosphericData):
        return np.zeros(3), np.zeros(3)
    nv = np.linalg.norm(vel)
    dp = float(bv @ vel) / nv
    cos_aoa = min(max(dp / np.linalg.norm(bv), -1.0), 1.0)
    mach = nv / spds
    drag = _table_interpolant(data.drag_itrp, data)(cos_aoa, mach) * data.force_scalar
    dragf = drag * vel / nv
    if abs(dp) >= 0.95:
        return dragf, np.zeros(3)
    lift = _table_interpolant(data.lift_itrp, data)(cos_aoa, mach) * data.force_scalar
    trq = _table_interpolant(data.trq_itrp, data)(cos_aoa, mach) * data.length_scalar * data.force_scalar
    trqd = np.cross(vel, bv)
    liftd = np.cross(-trqd, vel)
    liftd = liftd / np.linalg.norm(liftd)
    trqd = trqd / np.linalg.norm(trqd)
    return dragf + lift * liftd, trqd * trq

