"""ctypes binding of libscvx_hip.so — the only door from the Python host layer to the HIP path.

There is no CPU fallback: if the shared library is missing or does not export a symbol declared in
include/scvx.h, importing this module's `lib()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscvx_hip.so")
_DEFAULT_LIB_PATH = LIB_PATH

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)


class ScvxProblem(C.Structure):
    """struct scvx_problem (include/scvx.h) — flat image of DescentProblem, master.jl:17-71."""
    _fields_ = [
        ("g", C.c_double), ("mdry", C.c_double), ("mwet", C.c_double), ("Tmin", C.c_double), ("Tmax", C.c_double),
        ("deltaMax", C.c_double), ("thetaMax", C.c_double), ("gammaGs", C.c_double), ("omMax", C.c_double),
        ("dpMax", C.c_double),
        ("jB", C.c_double * 9),
        ("alpha", C.c_double), ("rho", C.c_double), ("sos", C.c_double),
        ("rTB", C.c_double * 3), ("rFB", C.c_double * 3),
        ("rIi", C.c_double * 3), ("rIf", C.c_double * 3), ("vIi", C.c_double * 3), ("vIf", C.c_double * 3),
        ("qBIi", C.c_double * 4), ("qBIf", C.c_double * 4),
        ("wBi", C.c_double * 3), ("wBf", C.c_double * 3),
        ("wNu", C.c_double), ("wID", C.c_double), ("wDS", C.c_double), ("wCst", C.c_double),
        ("wTviol", C.c_double), ("nuTol", C.c_double), ("delTol", C.c_double), ("tf_guess", C.c_double),
        ("ri", C.c_double), ("rh0", C.c_double), ("rh1", C.c_double), ("rh2", C.c_double),
        ("alph", C.c_double), ("bet", C.c_double),
        ("force_scalar", C.c_double), ("length_scalar", C.c_double), ("finmxf", C.c_double),
        ("K", C.c_int32), ("imax", C.c_int32), ("aero_kind", C.c_int32), ("model_flags", C.c_int32),
    ]


class ScvxSolverOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("refine", C.c_int32), ("tol", C.c_double), ("accept_tol", C.c_double),
                ("reuse_inactive_tr", C.c_int32), ("warm_start", C.c_int32), ("retries", C.c_int32), ("reserved0", C.c_int32)]


class ScvxThreedofOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("refine", C.c_int32), ("tol", C.c_double), ("delta", C.c_double),
                ("attitude", C.c_int32), ("reserved", C.c_int32)]


ABI_VERSION = 4   # SCVX_ABI_VERSION of the include/scvx.h these structs and signatures were written against

_vp = C.c_void_p
# name -> (restype, argtypes); must list every symbol include/scvx.h declares (tests check this)
SIGNATURES = {
    "scvx_abi_version": (C.c_int, []),
    "scvx_abi_struct_sizes": (C.c_int, [_ip]),
    "scvx_ctx_create": (C.c_int, [C.POINTER(ScvxProblem), C.c_int, C.POINTER(_vp)]),
    "scvx_ctx_destroy": (None, [_vp]),
    "scvx_last_error": (C.c_char_p, [_vp]),
    "scvx_control_dim": (C.c_int, [_vp]),
    "scvx_set_stream": (C.c_int, [_vp, _vp]),
    "scvx_use_null_stream": (C.c_int, [_vp]),
    "scvx_get_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "scvx_synchronize": (C.c_int, [_vp]),
    "scvx_set_nsub": (C.c_int, [_vp, C.c_int]),
    "scvx_get_nsub": (C.c_int, [_vp]),
    "scvx_set_aero_table": (C.c_int, [_vp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "scvx_linearize_f64": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp, _vp]),
    "scvx_linearize_f64_host": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp]),
    "scvx_propagate_f64": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "scvx_propagate_f64_host": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp]),
    "scvx_linearize_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_float, _vp, _vp]),
    "scvx_linearize_f32_host": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_float, _fp, _fp]),
    "scvx_propagate_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_float, _vp]),
    "scvx_propagate_f32_host": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_float, _fp]),
    "scvx_solver_default_opts": (C.c_int, [C.POINTER(ScvxSolverOpts)]),
    "scvx_batch_create": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "scvx_batch_destroy": (None, [_vp]),
    "scvx_batch_set_solver": (C.c_int, [_vp, C.POINTER(ScvxSolverOpts)]),
    "scvx_batch_init": (C.c_int, [_vp, _dp]),
    "scvx_batch_reset": (C.c_int, [_vp]),
    "scvx_batch_init_threedof": (C.c_int, [_vp, _dp, C.POINTER(ScvxThreedofOpts), _ip]),
    "scvx_threedof_default_opts": (C.c_int, [C.POINTER(ScvxThreedofOpts)]),
    "scvx_threedof_record_doubles": (C.c_int32, [C.c_int]),
    "scvx_threedof_solve": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(ScvxThreedofOpts), _dp, _ip, _dp]),
    "scvx_threedof_solve_dev": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(ScvxThreedofOpts), _vp, _vp]),
    "scvx_solve_step": (C.c_int, [_vp, _ip, _dp, _dp]),
    "scvx_solve_step_async": (C.c_int, [_vp]),
    "scvx_solve": (C.c_int, [_vp, _ip, _ip, _dp, _dp]),
    "scvx_batch_get_trajectory": (C.c_int, [_vp, _dp]),
    "scvx_batch_set_trajectory": (C.c_int, [_vp, _dp]),
    "scvx_batch_trajectory_dev": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "scvx_batch_get_linearization": (C.c_int, [_vp, _dp, _dp]),
    "scvx_batch_set_linearization_f32": (C.c_int, [_vp, C.c_int]),
    "scvx_batch_get_scalars": (C.c_int, [_vp, _dp, _dp, _ip]),
    "scvx_batch_set_scalars": (C.c_int, [_vp, _dp, _dp, _ip]),
    "scvx_batch_get_flags": (C.c_int, [_vp, _ip, _ip, _ip]),
    "scvx_batch_set_flags": (C.c_int, [_vp, _ip, _ip, _ip]),
    "scvx_batch_get_solver_stats": (C.c_int, [_vp, _ip, _ip, _dp, _dp]),
    "scvx_comm_probe": (C.c_int, []),
    "scvx_comm_unique_id": (C.c_int, [_vp]),
    "scvx_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "scvx_comm_destroy": (C.c_int, [_vp]),
    "scvx_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "scvx_allgather_trajectories": (C.c_int, [_vp, _vp]),
    "scvx_allgather_status": (C.c_int, [_vp, _vp, _vp]),
    "scvx_allgather_f64": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "scvx_allgather_i32": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "scvx_socp_solve": (C.c_int, [_vp, _dp, _dp]),
    "scvx_batch_get_step_stats": (C.c_int, [_vp, _dp, C.c_int]),
    "scvx_batch_set_profiling": (C.c_int, [_vp, C.c_int]),
    "scvx_batch_get_profile": (C.c_int, [_vp, _dp, C.POINTER(C.c_int64)]),
}

_LIB = None


def _preload_torch_hip_runtime():
    """One process, one HIP / HSA runtime.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the
    same sonames as /opt/rocm's; whichever copy is loaded first serves both torch and libscvx_hip.so.  If this library came
    first (the system copy), a later `import torch` finds "No HIP GPUs": its extensions were built against the bundled
    copy.  So when torch is installed its copies are loaded here, before libscvx_hip.so, without importing torch."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


class ScvxError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libscvx_hip.so and bind every ABI symbol; raises if the HIP extension is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH) and LIB_PATH == _DEFAULT_LIB_PATH:
        # a source checkout without the built artefact: build the HIP extension (hipcc, gfx950) — never a CPU path
        try:
            from . import build as _build
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise ScvxError(f"{LIB_PATH} is missing and building it with hipcc failed: {e}") from e
    if not os.path.exists(LIB_PATH):
        raise ScvxError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m successiveconvexification_amd.build` or __graft_entry__.build().")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library lacks an ABI symbol
        fn.restype = res
        fn.argtypes = args
    # the ABI guard of include/scvx.h: a library built from another revision of the header is refused before the first call
    sizes = (C.c_int32 * 3)()
    L.scvx_abi_struct_sizes(sizes)
    mine = (C.sizeof(ScvxProblem), C.sizeof(ScvxSolverOpts), C.sizeof(ScvxThreedofOpts))
    if L.scvx_abi_version() != ABI_VERSION or tuple(sizes) != mine:
        raise ScvxError(f"{LIB_PATH}: ABI version {L.scvx_abi_version()} / struct sizes {tuple(sizes)}, this binding expects "
                        f"{ABI_VERSION} / {mine}: rebuild the library from this tree's include/scvx.h")
    _LIB = L
    return L


def check(ctx_handle, rc: int, what: str):
    if rc != 0:
        msg = lib().scvx_last_error(ctx_handle)
        raise ScvxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
