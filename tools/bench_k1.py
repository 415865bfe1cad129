"""K1 timing A/B across library variants and npts, HIP events through torch on the library's stream."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from successiveconvexification_amd import _lib, sample_problems as sp
from conftest import random_segments
from oracle import model
B, K = 8192, 50
x, u, s = random_segments(model.base_prob_scaled(), B, K, 20261006)
xd, ud, sd = (torch.tensor(a, device="cuda") for a in (x, u, s))
e = torch.empty((B, K, 14), dtype=torch.float64, device="cuda"); d = torch.empty((B, K, 21, 14), dtype=torch.float64, device="cuda")
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
for path in sys.argv[1:]:
    _lib._LIB = None; _lib.LIB_PATH = os.path.join(ROOT, path)
    from successiveconvexification_amd.dynamics import IntegratorCache
    for npts in (1, 2, 10):
        c = IntegratorCache(sp.base_prob_scaled, npts=npts)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        L = c._L
        call = lambda: L.scvx_linearize_f64(c.handle, B, K, C.c_void_p(xd.data_ptr()), C.c_void_p(ud.data_ptr()), C.c_void_p(sd.data_ptr()), C.c_double(1/51), C.c_void_p(e.data_ptr()), C.c_void_p(d.data_ptr()))
        for _ in range(3): assert call() == 0
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10): call()
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        print(path, "npts", npts, "ms %.3f" % ms, "GB/s %.0f" % (130144 * B / ms / 1e6), "frac %.3f" % (130144 * B / ms / 1e6 / 8000), flush=True)
        c.close()
