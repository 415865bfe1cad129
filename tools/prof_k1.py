"""Build first:  mkdir -p variants && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCVX_K1_PROF -I include \
    -I successiveconvexification_amd/csrc -o variants/libscvx_hip_k1prof.so successiveconvexification_amd/csrc/*.hip
Barrier-wait shares of the persistent K1 kernel, per wavefront of block 0 (diagnostic build variants/libscvx_hip_k1prof.so:
-DSCVX_K1_PROF).  Wave 0 is the producer, 1..7 the consumers.    python tools/prof_k1.py [npts]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from successiveconvexification_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "variants", "libscvx_hip_k1prof.so")
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.dynamics import IntegratorCache
from conftest import random_segments
from oracle import model

args = [a for a in sys.argv[1:] if not a.startswith("--")]
aero = "--aero" in sys.argv
fins = "--fins" in sys.argv
npts = int(args[0]) if args else 10
B, K = 8192, 50
x, u, s = random_segments(model.base_prob_scaled(), B, K, 20261006)
xd, ud, sd = (torch.tensor(a, device="cuda") for a in (x, u, s))
e = torch.empty((B, K, 14), dtype=torch.float64, device="cuda")
d = torch.empty((B, K, 21, 14), dtype=torch.float64, device="cuda")
tabs = None
if aero:
    from successiveconvexification_amd.defns import AtmosphericData
    z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))
    tabs = AtmosphericData(z["drag"], z["lift"], z["torque"])
if fins:
    c = IntegratorCache(sp.base_prob_fin_scaled(tabs) if tabs is not None else sp.base_prob_fin_scaled(), npts=npts)
    fin = 0.01 * np.random.default_rng(7).uniform(-0.7, 0.7, (B, K + 1, 2))
    ud = torch.tensor(np.concatenate([u, fin], -1), device="cuda")
    d = torch.empty((B, K, 25, 14), dtype=torch.float64, device="cuda")
elif aero:
    c = IntegratorCache(sp.base_prob_aero_scaled(tabs), npts=npts)
else:
    c = IntegratorCache(sp.base_prob_scaled, npts=npts)
L = c._L
for _ in range(3):
    assert L.scvx_linearize_f64(c.handle, B, K, C.c_void_p(xd.data_ptr()), C.c_void_p(ud.data_ptr()), C.c_void_p(sd.data_ptr()), C.c_double(1 / 51),
                                C.c_void_p(e.data_ptr()), C.c_void_p(d.data_ptr())) == 0
c.synchronize()
out = np.zeros(32)
L.scvx_debug_k1_prof.argtypes = [C.POINTER(C.c_double)]
assert L.scvx_debug_k1_prof(out.ctypes.data_as(C.POINTER(C.c_double))) == 0
print("npts = %d, %s%s, block 0 of the persistent kernel" % (npts, "aero" if aero else "exo", " + fins" if fins else ""))
print("| wavefront | role | in barriers | total (s_memtime ticks) |")
print("|---|---|---|---|")
for w in range(8):
    print("| %d | %s | %.1f %% | %.0f |" % (w, "producer" if w == 0 else "consumer", 100 * out[2 * w] / max(out[2 * w + 1], 1), out[2 * w + 1]))
