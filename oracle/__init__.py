"""CPU oracle for the SCvx hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package, and only as the checker.  The product package ``successiveconvexification_amd`` never
imports it.

PARITY UNPINNED.  The reference (BenChung/SuccessiveConvexification, Julia) cannot be executed in the
build container (no ``julia`` binary, no network), ships no tests / golden vectors / recorded outputs
for this path, and delegates its numerics to un-vendored, un-versioned Julia packages
(DifferentialEquations ``BS3``, DiffEqSensitivity, Zygote, Interpolations, MathOptInterface + Mosek
with ECOS imported, Rotations).  The oracle therefore restates the reference's own equations
(citations in each module) and the published algorithms of those dependencies, and is validated by
mathematical self-consistency (finite differences, high-order ODE integration, KKT certificates),
not against reference output.

Modules
    model     problem data: DescentProblem defaults, normalize_problem, sample problems, linear_points
    dynamics  ctypes front-end of scvx_oracle.c (RHS, Jacobians, RK4 discretisation, aero tables)
    socp      the trust-region SOCP exactly as Rocketland.build_model assembles it
    ipm       primal-dual interior-point conic solver (the role Mosek/ECOS play in the reference)
    scvx      solve_step / solve_problem on top of the three
    port      ctypes front-end of scvx_port.cpp — C++ twin of the device algorithm (RK4 discretisation +
              the structure-exploiting interior-point solver, same source as the kernel), the
              ``cpu_baseline`` of bench.py
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _stale(so, srcs):
    return not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)


def build(force: bool = False) -> str:
    """Compile liboracle.so (gcc) and liboracle_port.so (g++); recipe: oracle/Makefile."""
    so = os.path.join(_HERE, "liboracle.so")
    port = os.path.join(_HERE, "liboracle_port.so")
    core = os.path.join(os.path.dirname(_HERE), "successiveconvexification_amd", "csrc", "scvx_ipm_core.hpp")
    if force or _stale(so, [os.path.join(_HERE, "scvx_oracle.c")]):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    if force or _stale(port, [os.path.join(_HERE, "scvx_port.cpp"), core]):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_port.so"], stdout=subprocess.DEVNULL)
    return so


# ---- the same two sources built for SPEED on the host they run on: bench.py's cpu_baseline -------------------------------
# The parity build above is -O2 -ffp-contract=off (bit-stable against the device's fused/unfused choices are irrelevant
# there; it is the checker).  A CPU *baseline* built like that is a soft target, so the timed leg uses
# -O3 -march=native with contraction allowed, compiled ON THE BOX THAT RUNS IT (the file name carries a hash of that CPU's
# model and flags: a binary built elsewhere is never picked up).
NATIVE_FLAGS = "-O3 -march=native -fopenmp -fPIC"
_NATIVE = False


def _cpu_tag() -> str:
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        keep = [l for l in txt.splitlines() if l.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = []
    return hashlib.sha1("|".join(keep).encode()).hexdigest()[:10]


def build_native():
    tag = _cpu_tag()
    so = os.path.join(_HERE, f"liboracle_native_{tag}.so")
    port = os.path.join(_HERE, f"liboracle_port_native_{tag}.so")
    csrc = os.path.join(os.path.dirname(_HERE), "successiveconvexification_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in ("scvx_ipm_core.hpp", "scvx_threedof_core.hpp")]
    if _stale(so, [os.path.join(_HERE, "scvx_oracle.c")]):
        subprocess.check_call(["gcc", *NATIVE_FLAGS.split(), "-std=c11", "-shared", "-o", so, os.path.join(_HERE, "scvx_oracle.c"), "-lm"])
    if _stale(port, [os.path.join(_HERE, "scvx_port.cpp")] + deps):
        subprocess.check_call(["g++", *NATIVE_FLAGS.split(), "-std=c++17", "-shared", "-o", port, os.path.join(_HERE, "scvx_port.cpp"), "-lm"])
    return so, port


def use_native(on: bool = True):
    """Route lib() / port_lib() to the -O3 -march=native builds (cpu_baseline only; parity tests use the default build)."""
    global _NATIVE, _LIB, _PORT
    if on != _NATIVE:
        _NATIVE, _LIB, _PORT = on, None, None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_native()[0] if _NATIVE else build())
    return _LIB


_PORT = None


def port_lib() -> ctypes.CDLL:
    global _PORT
    if _PORT is None:
        if _NATIVE:
            _PORT = ctypes.CDLL(build_native()[1])
        else:
            build()
            _PORT = ctypes.CDLL(os.path.join(_HERE, "liboracle_port.so"))
    return _PORT
