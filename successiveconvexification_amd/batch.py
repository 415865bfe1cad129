"""Batched SCvx driver over libscvx_hip.so (new relative to the reference, which solves one trajectory
serially): B independent DescentProblem instances that differ in their initial condition, advanced by
Rocketland.solve_step in lock-step on one GPU; `solve_sharded` spreads a Monte-Carlo batch over the GPUs
of a node, one process per GPU, and all-gathers the final trajectories over RCCL."""
import ctypes as C
import numpy as np

from . import _lib
from .dynamics import IntegratorCache

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _p(a):
    return a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


STATUS = {0: "converged", 1: "running", 2: "rejected", 3: "solver", 4: "nonfinite", 5: "infeasible"}


class ScvxBatch:
    """The batched ProblemIteration (master.jl:122-134) living in HBM."""

    def __init__(self, cache: IntegratorCache, B: int, tol: float = None, max_iter: int = None, refine: int = None,
                 accept_tol: float = None, reuse_inactive_tr: bool = None, warm_start: bool = None, retries: int = None):
        self.cache = cache
        self.B = int(B)
        self.K = cache.problem.K
        self.nu = cache.nu                               # control_dim: 3, or 5 with the fin extension
        self.nrec = (self.K + 1) * (14 + self.nu) + 1
        self._L = cache._L
        h = C.c_void_p()
        _lib.check(cache.handle, self._L.scvx_batch_create(cache.handle, self.B, C.byref(h)), "scvx_batch_create")
        self.handle = h
        if any(v is not None for v in (tol, max_iter, refine, accept_tol, reuse_inactive_tr, warm_start, retries)):
            o = _lib.ScvxSolverOpts()
            self._L.scvx_solver_default_opts(C.byref(o))
            if tol is not None:
                o.tol = tol
                o.accept_tol = tol   # the default band is empty (accept_tol = tol), whatever tol is
            if max_iter is not None:
                o.max_iter = max_iter
            if refine is not None:
                o.refine = refine
            if accept_tol is not None:
                o.accept_tol = accept_tol
            if reuse_inactive_tr is not None:
                o.reuse_inactive_tr = 1 if reuse_inactive_tr else 0
            if warm_start is not None:
                o.warm_start = 1 if warm_start else 0
            if retries is not None:
                o.retries = int(retries)   # scvx_solver_opts.retries: the ladder of step rules behind a failed solve (0 = one attempt)
            _lib.check(cache.handle, self._L.scvx_batch_set_solver(h, C.byref(o)), "scvx_batch_set_solver")

    def _chk(self, rc, what):
        _lib.check(self.cache.handle, rc, what)

    # create_initial (rocketland.jl:34-39) for every trajectory
    def init(self, ic=None):
        if ic is not None:
            ic = np.ascontiguousarray(ic, np.float64)
            if ic.shape != (self.B, 6):
                raise ValueError("ic must be [B][6] = (rIi, vIi)")
        self._chk(self._L.scvx_batch_init(self.handle, _p(ic) if ic is not None else None), "scvx_batch_init")
        return self

    def init_threedof(self, ic=None, **opts):
        """create_initial from FirstRound.solve_initial (initial_solve.jl:17-110) instead of the straight line: the 3-DoF
        landing SOCP is solved on the device for every trajectory; those whose solve is optimal start from its LinPoints,
        the others keep the straight line.  Returns the 3-DoF solver statuses [B] (0 = optimal)."""
        from .first_round import threedof_opts
        if ic is not None:
            ic = np.ascontiguousarray(ic, np.float64)
            if ic.shape != (self.B, 6):
                raise ValueError("ic must be [B][6] = (rIi, vIi)")
        o = threedof_opts(self._L, **opts)
        st3 = np.zeros(self.B, np.int32)
        self._chk(self._L.scvx_batch_init_threedof(self.handle, _p(ic) if ic is not None else None, C.byref(o), _pi(st3)),
                  "scvx_batch_init_threedof")
        return st3

    def reset(self):
        """create_initial again on the device for the same initial conditions (asynchronous)."""
        self._chk(self._L.scvx_batch_reset(self.handle), "scvx_batch_reset")
        return self

    # solve_step (rocketland.jl:226-321)
    def solve_step(self):
        st = np.zeros(self.B, np.int32)
        nu = np.zeros(self.B)
        dj = np.zeros(self.B)
        self._chk(self._L.scvx_solve_step(self.handle, _pi(st), _p(nu), _p(dj)), "scvx_solve_step")
        return st, nu, dj

    def solve_step_async(self):
        self._chk(self._L.scvx_solve_step_async(self.handle), "scvx_solve_step_async")

    # solve_problem (rocketland.jl:432-443)
    def solve(self):
        st = np.zeros(self.B, np.int32)
        it = np.zeros(self.B, np.int32)
        nu = np.zeros(self.B)
        dj = np.zeros(self.B)
        self._chk(self._L.scvx_solve(self.handle, _pi(st), _pi(it), _p(nu), _p(dj)), "scvx_solve")
        return st, it, nu, dj

    def socp_solve(self):
        """The conic subproblem alone at the current iterate: returns (x, u, sigma_new, nu)."""
        sol = np.zeros((self.B, self.nrec))
        nu = np.zeros((self.B, self.K, 14))
        self._chk(self._L.scvx_socp_solve(self.handle, _p(sol), _p(nu)), "scvx_socp_solve")
        x, u, s = self._split(sol)
        return x, u, s, nu

    def _split(self, rec):
        K, B = self.K, self.B
        nx = (K + 1) * 14
        return (rec[:, :nx].reshape(B, K + 1, 14).copy(), rec[:, nx:nx + (K + 1) * self.nu].reshape(B, K + 1, self.nu).copy(),
                rec[:, -1].copy())

    def trajectory(self):
        rec = np.zeros((self.B, self.nrec))
        self._chk(self._L.scvx_batch_get_trajectory(self.handle, _p(rec)), "scvx_batch_get_trajectory")
        return self._split(rec)

    def trajectory_record(self):
        rec = np.zeros((self.B, self.nrec))
        self._chk(self._L.scvx_batch_get_trajectory(self.handle, _p(rec)), "scvx_batch_get_trajectory")
        return rec

    def set_trajectory(self, x, u, sigma):
        rec = np.concatenate([np.asarray(x, float).reshape(self.B, -1), np.asarray(u, float).reshape(self.B, -1),
                              np.asarray(sigma, float).reshape(self.B, 1)], axis=1)
        rec = np.ascontiguousarray(rec)
        assert rec.shape == (self.B, self.nrec)
        self._chk(self._L.scvx_batch_set_trajectory(self.handle, _p(rec)), "scvx_batch_set_trajectory")

    def trajectory_dev(self):
        ptr = C.c_void_p()
        n = C.c_int64()
        self._chk(self._L.scvx_batch_trajectory_dev(self.handle, C.byref(ptr), C.byref(n)), "scvx_batch_trajectory_dev")
        return ptr.value, n.value

    def linearization(self):
        e = np.zeros((self.B, self.K, 14))
        d = np.zeros((self.B, self.K, 14 + 2 * self.nu + 1, 14))
        self._chk(self._L.scvx_batch_get_linearization(self.handle, _p(e), _p(d)), "scvx_batch_get_linearization")
        return e, d

    def set_linearization_f32(self, on=True):
        """Keep the derivative tiles in float (double arithmetic in K1, rounded at the store; the conic solve widens on load
        and stays double): the mixed-precision form of BASELINE configs[3-4].  Re-linearises an initialised batch."""
        self._chk(self._L.scvx_batch_set_linearization_f32(self.handle, 1 if on else 0), "scvx_batch_set_linearization_f32")
        return self

    def scalars(self):
        rk = np.zeros(self.B)
        cost = np.zeros(self.B)
        it = np.zeros(self.B, np.int32)
        self._chk(self._L.scvx_batch_get_scalars(self.handle, _p(rk), _p(cost), _pi(it)), "scvx_batch_get_scalars")
        return rk, cost, it

    def set_scalars(self, rk=None, cost=None, it=None):
        rk = None if rk is None else np.ascontiguousarray(np.broadcast_to(np.asarray(rk, float), (self.B,)))
        cost = None if cost is None else np.ascontiguousarray(np.broadcast_to(np.asarray(cost, float), (self.B,)))
        it = None if it is None else np.ascontiguousarray(np.broadcast_to(np.asarray(it, np.int32), (self.B,)))
        self._chk(self._L.scvx_batch_set_scalars(self.handle, _p(rk) if rk is not None else None,
                                                 _p(cost) if cost is not None else None,
                                                 _pi(it) if it is not None else None), "scvx_batch_set_scalars")

    def flags(self):
        """(status, active, live) per trajectory — with trajectory_record() and scalars() the full checkpoint."""
        st = np.zeros(self.B, np.int32)
        ac = np.zeros(self.B, np.int32)
        lv = np.zeros(self.B, np.int32)
        self._chk(self._L.scvx_batch_get_flags(self.handle, _pi(st), _pi(ac), _pi(lv)), "scvx_batch_get_flags")
        return st, ac, lv

    def set_flags(self, status=None, active=None, live=None):
        a = [None if v is None else np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.int32), (self.B,)))
             for v in (status, active, live)]
        self._chk(self._L.scvx_batch_set_flags(self.handle, *[_pi(v) if v is not None else None for v in a]),
                  "scvx_batch_set_flags")

    def solver_stats(self):
        st = np.zeros(self.B, np.int32)
        it = np.zeros(self.B, np.int32)
        merit = np.zeros(self.B)
        pobj = np.zeros(self.B)
        self._chk(self._L.scvx_batch_get_solver_stats(self.handle, _pi(st), _pi(it), _p(merit), _p(pobj)),
                  "scvx_batch_get_solver_stats")
        return st, it, merit, pobj

    def step_stats(self, reset: bool = True):
        """Totals over the solve_steps enqueued since the last reset (scvx_batch_get_step_stats); synchronises."""
        o = np.zeros(8)
        self._chk(self._L.scvx_batch_get_step_stats(self.handle, _p(o), 1 if reset else 0), "scvx_batch_get_step_stats")
        return dict(zip(("traj_steps", "solves", "ipm_iters", "warm_started", "skipped", "rejected", "failed", "converged"), o.tolist()))

    def set_profiling(self, on: bool):
        self._chk(self._L.scvx_batch_set_profiling(self.handle, 1 if on else 0), "scvx_batch_set_profiling")

    def profile(self):
        """(dict of summed ms per kernel, steps) since the last call; synchronises."""
        ms = np.zeros(5)
        n = C.c_int64()
        self._chk(self._L.scvx_batch_get_profile(self.handle, _p(ms), C.byref(n)), "scvx_batch_get_profile")
        return dict(zip(("socp", "propagate", "tr_update", "linearize", "glue"), ms.tolist())), n.value

    def close(self):
        if getattr(self, "handle", None):
            self._L.scvx_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- multi-GPU: see montecarlo.py (kept importable from here) -------------------------------------------
from .montecarlo import gather_records, shard_range  # noqa: E402,F401


def gather_trajectories(rec, group=None):
    """All-gather of the per-rank trajectory records [B][n] over the (default) torch.distributed process group into
    [world][B][n]; without an initialised group: [1][B][n].  Thin wrapper over montecarlo.gather_records."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rec.unsqueeze(0).clone()
    if group is None:
        return gather_records(rec, dist=dist)

    class _G:  # the module's collectives bound to `group`
        is_initialized = staticmethod(dist.is_initialized)
        get_world_size = staticmethod(lambda: dist.get_world_size(group))
        all_gather_into_tensor = staticmethod(lambda o, i: dist.all_gather_into_tensor(o, i, group=group))
        all_gather = staticmethod(lambda o, i: dist.all_gather(o, i, group=group))
    return gather_records(rec, dist=_G)
