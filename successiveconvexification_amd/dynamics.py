"""Dynamics — host-side mirror of the reference's discretisation interface over the HIP path.

Same names and argument meaning as dynamics.jl of BenChung/SuccessiveConvexification:
    make_dynamics_module(info)                 dynamics.jl:141   (code generation; a no-op here — the
                                                                 Jacobians are analytic in the kernel)
    IntegratorCache(prob, info, lin_mod)       dynamics.jl:258   (owns the device context)
    linearize_dynamics(states, tf, dt, cache)  dynamics.jl:321
    predict_state(x, uk, up, sigma, dt, pinfo, cache)   dynamics.jl:315
plus the batched forms the GPU path exists for.  Everything computes in libscvx_hip.so.
"""
import ctypes as C
import numpy as np

from . import _lib
from .defns import AtmosphericData, DescentProblem, LinPoint, LinRes, ProbInfo

_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def make_dynamics_module(info: ProbInfo):
    """The reference generates and evals a `Linearizer` module here; nothing to generate on this path."""
    return None


class IntegratorCache:
    """Holds the scvx_ctx (device, stream, problem constants, aero tables) for one DescentProblem."""

    def __init__(self, prob: DescentProblem, info: ProbInfo = None, lin_mod=None, device: int = 0, npts: int = 10):
        self.problem = prob
        self.info = info if info is not None else ProbInfo.from_problem(prob)
        self._L = _lib.lib()
        self._c_prob = prob.to_c()
        h = C.c_void_p()
        rc = self._L.scvx_ctx_create(C.byref(self._c_prob), int(device), C.byref(h))
        if rc != 0:
            raise _lib.ScvxError(f"scvx_ctx_create failed ({rc}): is a HIP device visible? (rc -1: bad problem, e.g. fins without finmxf > 0)")
        self.handle = h
        self.device = device
        self.nu = int(self._L.scvx_control_dim(h))        # 3, or 5 with the fin extension (SCVX_MODEL_FINS)
        self.np = 14 + 2 * self.nu + 1
        self.set_npts(npts)
        if isinstance(prob.aero, AtmosphericData):
            a = prob.aero
            d = np.ascontiguousarray(a.drag_itrp, float)
            l = np.ascontiguousarray(a.lift_itrp, float)
            t = np.ascontiguousarray(a.trq_itrp, float)
            nm, na = d.shape
            _lib.check(h, self._L.scvx_set_aero_table(h, _p(d), _p(l), _p(t), na, nm, a.aoa0, a.daoa, a.mach0, a.dmach),
                       "scvx_set_aero_table")

    def cproblem(self):
        """The flat struct scvx_problem this context was created from (what a ccall caller passes by pointer)."""
        return self._c_prob

    def set_npts(self, npts: int):
        _lib.check(self.handle, self._L.scvx_set_nsub(self.handle, int(npts)), "scvx_set_nsub")

    @property
    def npts(self) -> int:
        return self._L.scvx_get_nsub(self.handle)

    def set_stream(self, stream_handle):
        _lib.check(self.handle, self._L.scvx_set_stream(self.handle, C.c_void_p(stream_handle or 0)), "scvx_set_stream")

    def synchronize(self):
        _lib.check(self.handle, self._L.scvx_synchronize(self.handle), "scvx_synchronize")

    def close(self):
        if getattr(self, "handle", None):
            self._L.scvx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def linearize_batch(cache: IntegratorCache, x, u, sigma, dt):
    """x [B][K+1][14], u [B][K+1][nu], sigma [B] (host) -> endpoint [B][K][14], deriv [B][K][14+2nu+1][14]  (nu = cache.nu)."""
    x = np.ascontiguousarray(x, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    sigma = np.ascontiguousarray(sigma, np.float64)
    B, K1, nx = x.shape
    K = K1 - 1
    if nx != 14 or u.shape != (B, K1, cache.nu) or sigma.shape != (B,):
        raise ValueError("shape mismatch: x [B][K+1][14], u [B][K+1][%d], sigma [B]" % cache.nu)
    e = np.empty((B, K, 14))
    d = np.empty((B, K, cache.np, 14))
    _lib.check(cache.handle, cache._L.scvx_linearize_f64_host(cache.handle, B, K, _p(x), _p(u), _p(sigma), float(dt),
                                                              _p(e), _p(d)), "scvx_linearize_f64_host")
    return e, d


def propagate_batch(cache: IntegratorCache, x, u, sigma, dt):
    x = np.ascontiguousarray(x, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    sigma = np.ascontiguousarray(sigma, np.float64)
    B, K1, nx = x.shape
    K = K1 - 1
    if nx != 14 or u.shape != (B, K1, cache.nu) or sigma.shape != (B,):
        raise ValueError("shape mismatch: x [B][K+1][14], u [B][K+1][%d], sigma [B]" % cache.nu)
    e = np.empty((B, K, 14))
    _lib.check(cache.handle, cache._L.scvx_propagate_f64_host(cache.handle, B, K, _p(x), _p(u), _p(sigma), float(dt),
                                                              _p(e)), "scvx_propagate_f64_host")
    return e


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def linearize_batch_f32(cache: IntegratorCache, x, u, sigma, dt):
    """fp32 form (scvx_linearize_f32_host): float arrays in and out, float arithmetic on the device."""
    x = np.ascontiguousarray(x, np.float32)
    u = np.ascontiguousarray(u, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    B, K1, nx = x.shape
    K = K1 - 1
    if nx != 14 or u.shape != (B, K1, cache.nu) or sigma.shape != (B,):
        raise ValueError("shape mismatch: x [B][K+1][14], u [B][K+1][%d], sigma [B]" % cache.nu)
    e = np.empty((B, K, 14), np.float32)
    d = np.empty((B, K, cache.np, 14), np.float32)
    _lib.check(cache.handle, cache._L.scvx_linearize_f32_host(cache.handle, B, K, _pf(x), _pf(u), _pf(sigma), float(dt),
                                                              _pf(e), _pf(d)), "scvx_linearize_f32_host")
    return e, d


def propagate_batch_f32(cache: IntegratorCache, x, u, sigma, dt):
    x = np.ascontiguousarray(x, np.float32)
    u = np.ascontiguousarray(u, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    B, K1, nx = x.shape
    K = K1 - 1
    if nx != 14 or u.shape != (B, K1, cache.nu) or sigma.shape != (B,):
        raise ValueError("shape mismatch: x [B][K+1][14], u [B][K+1][%d], sigma [B]" % cache.nu)
    e = np.empty((B, K, 14), np.float32)
    _lib.check(cache.handle, cache._L.scvx_propagate_f32_host(cache.handle, B, K, _pf(x), _pf(u), _pf(sigma), float(dt),
                                                              _pf(e)), "scvx_propagate_f32_host")
    return e


def make_state(a: LinPoint, b: LinPoint, sig: float):
    """dynamics.jl:318-320"""
    return np.concatenate([a.state, a.control, b.control, [sig]])


def linearize_dynamics(states, tf_guess: float, base_dt: float, cache: IntegratorCache):
    """dynamics.jl:321-334: K+1 LinPoints -> K LinRes."""
    x = np.stack([s.state for s in states])[None]
    u = np.stack([s.control for s in states])[None]
    e, d = linearize_batch(cache, x, u, np.array([tf_guess]), base_dt)
    return [LinRes(e[0, k].copy(), d[0, k].T.copy()) for k in range(len(states) - 1)]


def predict_state(initial_state, uk, up, sigma, dt, pinfo, cache: IntegratorCache):
    """dynamics.jl:315-317: state at the end of one segment."""
    x = np.zeros((1, 2, 14))
    u = np.zeros((1, 2, cache.nu))
    x[0, 0] = initial_state
    u[0, 0] = uk
    u[0, 1] = up
    return propagate_batch(cache, x, u, np.array([float(sigma)]), dt)[0, 0]


def next_step(dynam, ab, abn, state, control_k, control_kp, sigma, sigHat, relax):
    """autodiff_dynamics.jl:104-107 / old_dynamics.jl:150-153: the affine prediction of the next node from a LinRes,
    derivative * [dx; du_k; du_{k+1}; dsigma] + endpoint + relax (host-side helper, a 14x21 product)."""
    ctrl = np.concatenate([np.asarray(state) - ab.state, np.asarray(control_k) - ab.control,
                           np.asarray(control_kp) - abn.control, [sigma - sigHat]])
    return dynam.derivative @ ctrl + dynam.endpoint + relax
