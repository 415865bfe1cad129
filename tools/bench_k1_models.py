"""K1 (linearize) timing per model -- exo / aero / exo+fins / aero+fins -- and npts, across library variants.
HIP events through torch on the library's stream; B x K segments of random-but-physical nodes (SURVEY 8d law).
    python tools/bench_k1_models.py [--B 8192] [--K 50] lib1.so [lib2.so ...]      (paths relative to the repo root)"""
import argparse, os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from successiveconvexification_amd import _lib, sample_problems as sp
from successiveconvexification_amd.defns import AtmosphericData
from conftest import random_segments
from oracle import model
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8192)
ap.add_argument("--K", type=int, default=50)
ap.add_argument("--npts", default="1,2,10")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
B, K = a.B, a.K
z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))
aero = AtmosphericData(z["drag"], z["lift"], z["torque"])
x, u3, s = random_segments(model.base_prob_scaled(), B, K, 20261006)
fin = 0.01 * np.random.default_rng(7).uniform(-0.7, 0.7, (B, K + 1, 2))
models = {"exo": (sp.base_prob_scaled, u3), "aero": (sp.base_prob_aero_scaled(aero), u3),
          "exo+fins": (sp.base_prob_fin_scaled(), np.concatenate([u3, fin], -1)), "aero+fins": (sp.base_prob_fin_scaled(aero), np.concatenate([u3, fin], -1))}
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
print("| library | model | npts | ms | GB/s (algorithmic) | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for path in a.libs:
    _lib._LIB = None; _lib.LIB_PATH = os.path.join(ROOT, path)
    from successiveconvexification_amd.dynamics import IntegratorCache
    for name, (p, u) in models.items():
        nu = u.shape[-1]
        alg = ((K + 1) * (14 + nu) + 1 + K * (14 + 14 * (14 + 2 * nu + 1))) * 8
        xd, ud, sd = (torch.tensor(np.ascontiguousarray(v), device="cuda") for v in (x, u, s))
        e = torch.empty((B, K, 14), dtype=torch.float64, device="cuda"); d = torch.empty((B, K, 14 + 2 * nu + 1, 14), dtype=torch.float64, device="cuda")
        for npts in [int(v) for v in a.npts.split(",")]:
            try:
                c = IntegratorCache(p, npts=npts)
            except Exception as ex:   # a variant built before the fin extension
                print("|", path, "|", name, "|", npts, "| n/a (%s) | | |" % type(ex).__name__); continue
            c.set_stream(torch.cuda.current_stream().cuda_stream)
            L = c._L
            call = lambda: L.scvx_linearize_f64(c.handle, B, K, C.c_void_p(xd.data_ptr()), C.c_void_p(ud.data_ptr()), C.c_void_p(sd.data_ptr()), C.c_double(1 / (K + 1)), C.c_void_p(e.data_ptr()), C.c_void_p(d.data_ptr()))
            for _ in range(3): assert call() == 0
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10): call()
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 10
            print("| %s | %s | %d | %.3f | %.0f | %.3f |" % (path, name, npts, ms, alg * B / ms / 1e6, alg * B / ms / 1e6 / 8000), flush=True)
            c.close()
