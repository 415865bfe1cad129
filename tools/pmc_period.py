"""One solve_problem period (14 solve_steps from create_initial) of the bench workload for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_period.py [lib.so] [cold]
2 untimed steps, reset, 14 steps.  Prints the device-side counters of the 14 steps (conic solves, interior-point iterations), which
tools/pmc_model.py needs to turn the byte counts into bytes per iteration."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import _lib, sample_problems as sp
args = sys.argv[1:]
cold = "cold" in args
libs = [a for a in args if a.endswith(".so")]
if libs:
    _lib.LIB_PATH = os.path.join(ROOT, libs[0])
import bench
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
B = int(os.environ.get("B", "8192"))
ic = bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
c = IntegratorCache(sp.base_prob_scaled)
b = ScvxBatch(c, B, **({"warm_start": False} if cold else {})).init(ic)
b.step_stats(reset=True)
b.solve_step_async(); b.solve_step_async()
c.synchronize()
tw = b.step_stats(reset=True)     # the two warm-up launches are in the counter pass too
b.reset(); c.synchronize()
for _ in range(14):
    b.solve_step_async()
c.synchronize()
ts = b.step_stats(reset=True)
print("PMC_PERIOD " + json.dumps({"lib": libs[0] if libs else "default", "cold": cold, "B": B, "launches_counted": 14, "launches_total": 16,
                                  "solves": ts["solves"], "ipm_iters": ts["ipm_iters"], "warm_started": ts["warm_started"],
                                  "warmup_solves": tw["solves"], "warmup_ipm_iters": tw["ipm_iters"]}), flush=True)
b.close(); c.close()
