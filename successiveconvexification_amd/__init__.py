"""successiveconvexification_amd — MI355X-native SCvx inner loop (discretisation + trust-region SOCP)
behind the problem-definition API of BenChung/SuccessiveConvexification.

Layout
    csrc/            hand-written HIP kernels (gfx950) and the C ABI (include/scvx.h) -> libscvx_hip.so
    _lib.py          ctypes binding (fails loudly if the HIP extension is absent; no CPU fallback)
    defns.py         RocketlandDefns   (master.jl)
    aerodynamics.py  Aerodynamics      (aerodynamics.jl: load_aerodata, rescale_aerodata)
    dynamics.py      Dynamics          (dynamics.jl: IntegratorCache, linearize_dynamics, predict_state)
    first_round.py   FirstRound        (initial_solve.jl: linear_points, linear_initial)
    rocketland.py    Rocketland        (rocketland.jl: create_initial, solve_step, solve_problem)
    sample_problems.py SampleProblems  (sample_problems.jl)
    batch.py         batched / multi-GPU driver (new: the reference solves one trajectory serially)
"""
from .defns import (AerodynamicInfo, AtmosphericData, DescentProblem, ExoatmosphericData, LinPoint, LinRes,
                    ProbInfo, ProblemIteration)

__all__ = ["AerodynamicInfo", "AtmosphericData", "DescentProblem", "ExoatmosphericData", "LinPoint", "LinRes",
           "ProbInfo", "ProblemIteration"]
