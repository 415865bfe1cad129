"""B-sweep on the bench's own timed region: for each batch size one solve_problem period (14 solve_steps from
create_initial) with the default solver options and with every solve started cold; the executor is the library's own choice (4 wavefronts
per trajectory up to 512, 2 up to 1,024, 1 above).  Prints the table of profiles/r05_bsweep.md and, from it, the strong-scaling prediction
for the 8,192 batch on 2 / 4 / 8 GPUs (no communication inside the loop: efficiency = per-GPU rate at B / N over the rate at 8,192).
    python tools/bsweep_mix.py [B list]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
BS = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,64,256,512,1024,2048,4096,8192").split(",")]
p = sp.base_prob_scaled
c = IntegratorCache(p)
rows = {}
for B in BS:
    ic = bench.disperse_ics(p, 0, B, 20261004)
    res = {}
    for label, kw in (("mix", {}), ("cold", {"warm_start": False})):
        b = ScvxBatch(c, B, **kw).init(ic)
        b.solve_step_async(); b.solve_step_async()
        best = None
        for rep in range(3):
            b.reset(); c.synchronize()
            t0 = time.perf_counter()
            for _ in range(14):
                b.solve_step_async()
            c.synchronize()
            t = time.perf_counter() - t0
            best = t if best is None or t < best else best
        res[label] = 1e3 * best / 14
        b.close()
    rows[B] = res
    print("B %5d  mix %.3f ms/step  cold %.3f ms/step  %.0f traj-iter/s" % (B, res["mix"], res["cold"], B / (1e-3 * res["mix"])), flush=True)
c.close()
full = max(BS)
rf = full / rows[full]["mix"]
print("\n| B per GPU | ms per step (mix) | ms per cold step | traj-iter/s | of the B = %d rate |" % full)
print("|---|---|---|---|---|")
for B in BS:
    r = rows[B]
    print("| %d | %.2f | %.2f | %.0f | %.0f %% |" % (B, r["mix"], r["cold"], B / (1e-3 * r["mix"]), 100 * (B / r["mix"]) / rf))
print("\n| GPUs | B per GPU | predicted traj-iter/s (whole job) | strong-scaling efficiency |")
print("|---|---|---|---|")
for n in (1, 2, 4, 8):
    if full // n in rows:
        r = rows[full // n]
        print("| %d | %d | %.0f | %.0f %% |" % (n, full // n, n * (full // n) / (1e-3 * r["mix"]), 100 * ((full // n) / r["mix"]) / rf))
