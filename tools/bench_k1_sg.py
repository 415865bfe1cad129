"""K1 at rk4 npts 1/2/4/10 with the substep- and the stage-granular producer/consumer pipeline (SCVX_K1_SG=0/1)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sg in ("0", "1"):
    env = dict(os.environ, SCVX_K1_SG=sg)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-traj-check"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    import json
    d = json.loads(out)["roofline_k1_by_npts"]
    print("SCVX_K1_SG=" + sg, {k: (round(v["ms"], 3), round(v["frac"], 4)) for k, v in d.items()}, flush=True)
