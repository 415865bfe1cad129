"""B-sweep of the conic-solve executor: trajectories per GPU x wavefronts per trajectory.
    python tools/bsweep.py [--lib variants/x.so] [--B 512,1024] [--waves 1,2] > gpurun_out/bsweep.md      (on the GPU box)
One solve_problem-like run per cell: 2 warm-up + 6 timed solve_steps from create_initial, per-kernel device time from
the library's HIP events."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import _lib, montecarlo as mc, sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache

def _arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
if "--lib" in sys.argv:
    _lib._LIB = None
    _lib.LIB_PATH = os.path.join(ROOT, _arg("--lib", ""))
    print("library:", _lib.LIB_PATH)
BS = tuple(int(v) for v in _arg("--B", "1,64,256,512,1024,1536,2048,3072,4096,8192").split(","))
WS = tuple(int(v) for v in _arg("--waves", "1,2,4").split(","))
p = sp.base_prob_scaled
c = IntegratorCache(p, npts=10)
print("| B | waves / trajectory | socp ms / step | step ms | traj-iter/s | vs best of row |")
print("|---|---|---|---|---|---|")
for B in BS:
    ic = mc.disperse_ics(p, 0, B, 20261004)
    rows = []
    for w in WS:
        if (w == 4 and B > 4096) or (w == 1 and B < 64 and False):
            continue
        os.environ["SCVX_K4_WAVES"] = str(w)
        b = ScvxBatch(c, B).init(ic)
        for _ in range(2):
            b.solve_step_async()
        c.synchronize()
        b.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(6):
            b.solve_step_async()
        c.synchronize()
        t = time.perf_counter() - t0
        prof, n = b.profile()
        rows.append((w, prof["socp"] / n, 1e3 * t / 6, B * 6 / t))
        b.close()
    best = max(r[3] for r in rows)
    for w, ms, st, v in rows:
        print("| %d | %d | %.2f | %.2f | %.0f | %.2f |" % (B, w, ms, st, v, v / best), flush=True)
os.environ.pop("SCVX_K4_WAVES", None)
