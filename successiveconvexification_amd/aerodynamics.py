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

    return AtmosphericData(tab("drag"), tab("lift"), tab("torque"), 1.0, 1.0)


def rescale_aerodata(data, Ul: float, Ut: float, Um: float):
    if isinstance(data, ExoatmosphericData):
        return data
    return replace(data, force_scalar=1 / (Ul * Um / Ut**2), length_scalar=1 / Ul)
