"""WaveExT::chol_inv14 -- the 14 x 14 register Cholesky + inverse on the factorisation chain of the conic kernel (K4) -- in isolation
(tools/micro/chol_dpp_ab.hip): the DPP forms (row_newbcast by v_mov_b64_dpp, and folded into v_fmac_f64_dpp) against the v_readlane form of
rounds 1-5 on 2,048 random SPD tiles scaled 1e-4 ... 1e4: L^-1 M L^-T = I to 1e-12 and BIT-IDENTICAL factors across the three variants.
Replaces the 14 x 14 block factorisations inside MOI.optimize! (rocketland.jl:271) on the device."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dpp_cholesky_variants_are_bit_identical_and_correct(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    out = {}
    for v in (0, 1, 2):
        exe = tmp_path / f"chol_{v}"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-DSCVX_CHOL_DPP={v}", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "successiveconvexification_amd", "csrc"), "-o", str(exe),
                        os.path.join(ROOT, "tools", "micro", "chol_dpp_ab.hip")], check=True, capture_output=True, timeout=300)
        r = subprocess.run([str(exe), "2048"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"max (\S+)\s+bad (\d+)\s+checksum (\w+)", r.stdout)
        assert m, r.stdout
        out[v] = (float(m.group(1)), int(m.group(2)), m.group(3))
        print(r.stdout.strip())
    for v, (err, bad, _) in out.items():
        assert err < 1e-12 and bad == 0, (v, err, bad)
    assert out[0][2] == out[1][2] == out[2][2], out
