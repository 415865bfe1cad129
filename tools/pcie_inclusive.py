"""The benchmark's 14-step solve_problem mix through the HOST-buffer side of the boundary: initial conditions uploaded from host memory
(scvx_batch_init with a host pointer), per-step status / |nu| / dJ read back by every solve_step (the blocking form), and the final
trajectories, scalars and flags downloaded -- next to the same 14 steps with everything resident (the bench's `value`).
    python tools/pcie_inclusive.py [B]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = IntegratorCache(sp.base_prob_scaled)
ic = mc.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
b = ScvxBatch(c, B)
steps = sp.base_prob_scaled.imax - 1
for rep in range(2):                      # first repetition warms the library
    c.synchronize(); t0 = time.perf_counter()
    b.init(ic)                            # H2D of [B][6] + create_initial + first linearisation
    for n in range(steps):
        st, nun, dj = b.solve_step()      # blocking: D2H of status, |nu|, dJ per step
    x, u, s = b.trajectory()              # D2H of [B][(K+1)*17+1] doubles
    rk, cost, it = b.scalars()
    flags = b.flags()
    c.synchronize(); t_host = time.perf_counter() - t0
    b.init(ic); c.synchronize(); t0 = time.perf_counter()
    for n in range(steps):
        b.solve_step_async()
    c.synchronize(); t_res = time.perf_counter() - t0
print("B = %d, %d solve_steps: host-buffer boundary %.1f ms (%.0f traj-iter/s), resident %.1f ms (%.0f traj-iter/s); downloaded %.1f MB"
      % (B, steps, 1e3 * t_host, B * steps / t_host, 1e3 * t_res, B * steps / t_res, (x.nbytes + u.nbytes + s.nbytes) / 1e6))
