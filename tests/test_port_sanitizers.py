"""The interior-point core shared by the HIP kernel and the CPU twin, under AddressSanitizer + UBSan on the host
(GPU ASan is not available on the pool; the same source runs on the device, so an out-of-bounds index found here
is a device fault avoided)."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT


def test_ipm_core_is_clean_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    so = tmp_path / "liboracle_port_asan.so"
    subprocess.check_call(["g++", "-O1", "-g", "-fPIC", "-fopenmp", "-ffp-contract=off", "-std=c++17",
                           "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared", "-o", str(so),
                           os.path.join(ROOT, "oracle", "scvx_port.cpp"), "-lm"])
    script = tmp_path / "run.py"
    script.write_text(textwrap.dedent(f"""
        import ctypes as C, sys
        import numpy as np
        sys.path.insert(0, {ROOT!r})
        import oracle
        oracle._PORT = C.CDLL({str(so)!r})
        from dataclasses import replace
        from oracle import model, port, dynamics as od
        for K in (50, 7):
            p = replace(model.base_prob_scaled(), K=K)
            B = 2
            ic = model.disperse_ics(p, B, 20261004)
            x = np.zeros((B, K + 1, 14)); u = np.zeros((B, K + 1, 3))
            for b in range(B):
                x[b], u[b] = model.linear_points(p, ic[b, :3], ic[b, 3:])
            e, d = od.linearize(od.Params(p), x, u, np.full(B, p.tf_guess), 1 / (K + 1), 2)
            r = port.socp(p, x, u, e, d, 100.0, ic, nthreads=2)
            assert (r["status"] == 0).all(), r["status"]
        print("SANITIZED_OK")
        """))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SANITIZED_OK" in out.stdout, out.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-2000:]
