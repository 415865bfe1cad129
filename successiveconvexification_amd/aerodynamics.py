"""Aerodynamics — table loading side of aerodynamics.jl (the force model itself runs in the kernel).

    load_aerodata(liftdrag_csv)      aerodynamics.jl:11-28  columns aoa,mach,drag,lift,torque;
                                     reshape(col, 181, 61): cos(AoA) fastest, Mach slowest
    rescale_aerodata(data,Ul,Ut,Um)  aerodynamics.jl:30-36
The cubic B-spline prefilter (Interpolations.jl Cubic(Line(OnGrid()))) is applied inside
scvx_set_aero_table when the table is uploaded.
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
