"""SampleProblems — sample_problems.jl of the reference, same names.

normalize_problem reproduces sample_problems.jl:5-23 field by field, including its quirks
(SURVEY.md §8a-9): vIf is built from vIi (:15), rFB is scaled by 1/Ut (:16), omMax is divided by Ut
and stays in degrees, nuTol is not forwarded.
"""
from dataclasses import replace
import numpy as np

from .aerodynamics import rescale_aerodata
from .defns import DescentProblem, ExoatmosphericData


def normalize_problem(dp: DescentProblem) -> DescentProblem:
    Ul = float(np.max(dp.rIi))
    Ut = dp.tf_guess
    Um = dp.mwet
    return DescentProblem(
        g=dp.g / (Ul / Ut**2), mdry=dp.mdry / Um, mwet=dp.mwet / Um,
        Tmin=dp.Tmin / (Um * Ul / Ut**2), Tmax=dp.Tmax / (Um * Ul / Ut**2),
        omMax=dp.omMax / Ut, jB=np.asarray(dp.jB, float) * (1 / (Um * Ul**2)),
        rTB=dp.rTB * (1 / Ul), rIi=dp.rIi * (1 / Ul),
        rIf=dp.rIf * (1 / Ul), vIi=dp.vIi * (1 / (Ul / Ut)),
        vIf=dp.vIi * (1 / (Ul / Ut)), qBIf=dp.qBIf.copy(), qBIi=dp.qBIi.copy(),
        wBi=dp.wBi.copy(), wBf=dp.wBf.copy(), rFB=dp.rFB * (1 / Ut),
        deltaMax=dp.deltaMax, thetaMax=dp.thetaMax, gammaGs=dp.gammaGs,
        alpha=dp.alpha / (Ut**2 / Ul), K=dp.K, imax=dp.imax, wNu=dp.wNu, wID=dp.wID,
        wDS=dp.wDS, wCst=dp.wCst, wTviol=dp.wTviol, delTol=dp.delTol,
        tf_guess=dp.tf_guess / Ut, ri=dp.ri, rh0=dp.rh0, rh1=dp.rh1,
        rh2=dp.rh2, alph=dp.alph, bet=dp.bet, dpMax=dp.dpMax / (Um / (Ul * Ut**2)), rho=dp.rho / (Um / Ul**3),
        sos=dp.sos / (Ul / Ut), aero=rescale_aerodata(dp.aero, Ul, Ut, Um),
        model_flags=dp.model_flags, finmxf=dp.finmxf)   # build extensions pass through (finmxf is a constant of build_model, rocketland.jl:205)


def _base(aero) -> DescentProblem:
    return DescentProblem(
        g=9.82, mwet=66018.0, mdry=65947.0, Tmin=0.1 * 4.686588e6, Tmax=4.686588e6,
        jB=np.diag([72487.03125, 2.0734175e6, 2.0734175e6]), alpha=0.000345,
        rTB=np.array([-4.26114, 0, 0]), rFB=np.array([2.0, 0, 0]), rIi=np.array([1000.0, 1000.0, 100.0]),
        rIf=np.array([0.0, 0.0, 0.0]), vIi=np.array([-100.0, -200.0, 0]), sos=352.0, aero=aero, wNu=1e4)


base_prob = _base(ExoatmosphericData())
base_prob_scaled = normalize_problem(base_prob)


def base_prob_aero(aero_info) -> DescentProblem:
    """sample_problems.jl:30-31; the table is passed in (load_aerodata needs a file path)."""
    return _base(aero_info)


def base_prob_aero_scaled(aero_info) -> DescentProblem:
    return normalize_problem(base_prob_aero(aero_info))


def base_prob_fin_scaled(aero_info=None, fin_table=None) -> DescentProblem:
    """BASELINE configs[4] "6-DoF + fin aero": the sample problem with the fin extension (control_dim = 5).  The model is
    DEFINED BY THIS BUILD from the reference's commented-out fin code (include/scvx.h).  One deliberate
    departure from normalize_problem: it scales rFB by 1/Ut (sample_problems.jl:16, harmless there because rFB is unused),
    which would put the fins 2 normalised length units = 2 km from the centre of mass; the fin torque arm is a length, so it
    is scaled by 1/Ul here like rTB.  fin_table (aerodynamics.load_fin_table of aero/fin.csv), if given, replaces the constant
    finmxf = 0.01 of rocketland.jl:205 by the table's largest fin force at the initial Mach number (aerodynamics.fin_force_bound)."""
    from .aerodynamics import fin_force_bound
    from .defns import AtmosphericData, ExoatmosphericData
    b = _base(aero_info if aero_info is not None else ExoatmosphericData())
    p = normalize_problem(replace(b, model_flags=b.model_flags | 2))
    p = replace(p, rFB=b.rFB * (1.0 / float(np.max(b.rIi))))
    if fin_table is not None:
        fs = p.aero.force_scalar if isinstance(p.aero, AtmosphericData) else 1.0 / (float(np.max(b.rIi)) * b.mwet / b.tf_guess**2)
        p = replace(p, finmxf=fin_force_bound(fin_table, float(np.linalg.norm(p.vIi)) / p.sos, fs))
    return p
