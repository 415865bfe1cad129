"""C-ABI surface and host-side logic that need no GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "scvx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scvx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from successiveconvexification_amd import _lib, build
    so = build.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = set(re.findall(r"\bT (scvx_[a-z0-9_]+)", out))
    declared = _header_symbols()
    assert len(declared) >= 30
    assert set(declared) <= exported, set(declared) - exported
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    # the shared object loads and binds without touching a GPU
    _lib.lib()


def test_struct_layouts_match_the_header(tmp_path):
    from successiveconvexification_amd import _lib
    c = tmp_path / "sz.c"
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "scvx.h"\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(scvx_problem), '
                 'offsetof(scvx_problem, K), offsetof(scvx_problem, wNu), sizeof(scvx_solver_opts));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    a, b, cc, d = map(int, subprocess.check_output([str(exe)], text=True).split())
    P = _lib.ScvxProblem
    assert (ctypes.sizeof(P), P.K.offset, P.wNu.offset, ctypes.sizeof(_lib.ScvxSolverOpts)) == (a, b, cc, d)


def test_abi_guard_reports_the_header_and_refuses_another_revision(monkeypatch):
    """scvx_abi_version / scvx_abi_struct_sizes (no device call): the library's values are the header's and the binding's; a binding
    written for another revision (version or a struct size) is refused at load, before any struct crosses the boundary."""
    from successiveconvexification_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "scvx.h")).read()
    assert int(re.search(r"#define SCVX_ABI_VERSION (\d+)", hdr).group(1)) == L.scvx_abi_version() == _lib.ABI_VERSION
    sz = (ctypes.c_int32 * 3)()
    assert L.scvx_abi_struct_sizes(sz) == 0
    assert tuple(sz) == (ctypes.sizeof(_lib.ScvxProblem), ctypes.sizeof(_lib.ScvxSolverOpts), ctypes.sizeof(_lib.ScvxThreedofOpts))
    assert L.scvx_abi_struct_sizes(None) == -1   # SCVX_ERR_ARG
    # the Julia shim carries the same number
    jl = open(os.path.join(ROOT, "julia", "ScvxAMD.jl")).read()
    assert int(re.search(r"const ABI_VERSION = (\d+)", jl).group(1)) == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.ScvxError, match="ABI version"):
        _lib.lib()


def test_conic_kernels_keep_the_solver_object_out_of_private_memory(tmp_path):
    """Kernel metadata of the built library (no GPU): the conic kernels spill nothing and their private segment holds only the register
    saves around non-inlined calls -- the solver object lives in an LDS frame (DESIGN 'what comes next' item 0: as an automatic object it
    cost 1,472 B of private memory per lane and 7 % of the kernel's HBM traffic, unnoticed for three rounds).  K0 likewise, with its
    known spills."""
    import shutil
    from successiveconvexification_amd import build
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    lib = tmp_path / "lib.so"
    shutil.copy(build.build(), lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    kernels = {}
    for co in sorted(tmp_path.glob("lib.so.*gfx950")):
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            k, v = m.groups()
            if k == "args" or (k == "agpr_count" and cur.get("name")):   # a new kernel record starts
                if cur.get("name"):
                    kernels[cur["name"]] = cur
                cur = {}
            cur[k] = v
        if cur.get("name"):
            kernels[cur["name"]] = cur
    conic = {n: k for n, k in kernels.items() if re.search(r"socp_(kernel|lin32_kernel|kernel_t|block_kernel)", n)}
    assert len(conic) >= 12, sorted(kernels)     # 2 + 4 (control_dim 3) and 2 + 4 (fins)
    for n, k in conic.items():
        # Round 6: <= 11.  The DPP Cholesky lets the factorisation routines take all 256 VGPRs, so the kernel BODY no longer has a
        # callee-saved register for the VGPR that carries its spilled SGPRs and saves / reloads that one register around its call sites
        # (once per solve attempt, never inside an interior-point iteration); the solver's routines themselves spill nothing
        # (checked on the ISA: no `Folded Spill` between prologue and epilogue of any Solver:: routine of the single-wavefront kernel).
        assert int(k["vgpr_spill_count"]) <= 24, (n, k)   # (21 in the two-wavefront kernels: two more call sites in the kernel body)
        # 832 (one wavefront) ... 1428 (two wavefronts, two-ended: build_kkt -> twc_top / twc_bot is one more non-inlined level, whose prologue
        # saves the callee-saved registers it uses -- once per factorisation): register saves around calls, never the solver object
        assert int(k["private_segment_fixed_size"]) <= 1536, (n, k)
        # four-wavefront blocks carry two tile sets of the two-ended factorisation (round 6: with the border's slices, rings and running t):
        # 70-74 KB of the CU's 160 KB -- two blocks per CU, which is all B <= 512 asks for; everything else stays below 40 KB (four per CU)
        assert int(k["group_segment_fixed_size"]) <= (76 if "block_kernelILi4" in n else 40) * 1024, (n, k)
    one = [k for n, k in conic.items() if "socp_kernelE" in n][0]
    # round 5: 496 -> 832 B (the solve and its refinement check are one routine now: more values live across the non-inlined passes;
    # ~200 scratch accesses per interior-point iteration, 3 % of its traffic); LDS: eight single-wavefront blocks per CU (two per SIMD)
    # round 6: the tiles of the switched-off build_kkt(res) path are no longer allocated: nine blocks' worth of LDS per CU again (the kernel
    # itself is VGPR-limited at eight)
    # (928 B since the carried residuals of round 6: three more values live across the non-inlined passes of attempt_solve)
    assert int(one["private_segment_fixed_size"]) <= 960 and int(one["group_segment_fixed_size"]) <= 160 * 1024 // 9, one
    k0 = [k for n, k in kernels.items() if "threedof_kernelE" in n][0]
    assert int(k0["private_segment_fixed_size"]) <= 1200, k0


def test_missing_extension_fails_loudly(monkeypatch):
    from successiveconvexification_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libscvx_hip.so")
    with pytest.raises(_lib.ScvxError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "successiveconvexification_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f


def test_sample_problem_numbers():
    """SampleProblems.base_prob through normalize_problem (Ul=1000, Ut=1, Um=66018)."""
    from successiveconvexification_amd import sample_problems as sp
    p = sp.base_prob_scaled
    assert p.g == pytest.approx(0.00982) and p.mwet == 1.0 and p.mdry == pytest.approx(0.998924535733, rel=1e-10)
    assert p.Tmax == pytest.approx(0.0709895483, rel=1e-8) and p.Tmin == pytest.approx(0.00709895483, rel=1e-8)
    assert p.alpha == pytest.approx(0.345) and p.sos == pytest.approx(0.352)
    assert np.allclose(np.diag(p.jB), [1.09798890e-6, 3.14068512e-5, 3.14068512e-5], rtol=1e-7)
    assert np.allclose(p.rTB, [-0.00426114, 0, 0]) and np.allclose(p.rIi, [1, 1, 0.1])
    assert np.allclose(p.vIi, [-0.1, -0.2, 0]) and np.allclose(p.vIf, p.vIi)  # quirk: vIf built from vIi
    assert np.allclose(p.rFB, [2.0, 0, 0])                                    # quirk: rFB scaled by 1/Ut
    assert p.omMax == 60.0 and p.nuTol == 1e-10 and p.wNu == 1e4 and p.K == 50 and p.imax == 15


def test_product_and_oracle_problem_definitions_agree():
    from dataclasses import fields
    from oracle import model
    from successiveconvexification_amd import sample_problems as sp
    po, pp = model.base_prob_scaled(), sp.base_prob_scaled
    for f in fields(po):
        if f.name in ("aero", "enforce_dp"):   # enforce_dp (oracle) = model_flags bit 0 (product): build extension
            continue
        a, b = getattr(po, f.name), getattr(pp, f.name)
        assert np.all(np.asarray(a) == np.asarray(b)), f.name
    d = model.DescentProblem()
    from successiveconvexification_amd.defns import DescentProblem
    q = DescentProblem()
    assert (d.wNu, d.Tmax, d.K, d.imax, d.bet, d.sos) == (q.wNu, q.Tmax, q.K, q.imax, q.bet, q.sos) == (1e5, 5.0, 50, 15, 3.2, 5.0)


def test_linear_points_and_rotation_between():
    from oracle import model
    p = model.base_prob_scaled()
    x, u = model.linear_points(p)
    K = p.K
    assert np.allclose(x[0, 1:4], p.rIi) and np.allclose(x[K, 1:4], p.rIf) and x[0, 0] == p.mwet and x[K, 0] == p.mdry
    assert np.allclose(u[:, 0], x[:, 0] * p.g) and np.all(u[:, 1:] == 0)
    # the quaternion rotates e1 onto -v/|v| (initial_solve.jl:121)
    from oracle import dynamics as od
    for k in (0, 17, K):
        q = x[k, 7:11]
        assert np.linalg.norm(q) == pytest.approx(1.0)
        C = np.array([[1 - 2 * (q[2]**2 + q[3]**2), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2])],
                      [2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1]**2 + q[3]**2), 2 * (q[2] * q[3] - q[0] * q[1])],
                      [2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1]**2 + q[2]**2)]])
        v = -x[k, 4:7]
        assert np.allclose(C @ [1, 0, 0], v / np.linalg.norm(v), atol=1e-12)
    assert np.allclose(model.rotation_between([1, 0, 0], [1, 0, 0]), [1, 0, 0, 0])


def test_dispersion_is_deterministic_per_trajectory():
    import bench
    from oracle import model
    p = model.base_prob_scaled()
    a = model.disperse_ics(p, 10, 20261004)
    assert np.array_equal(a[3:7], bench.disperse_ics(p, 3, 7, 20261004))  # bench shards by global index
    assert np.abs(a[:, :3] / p.rIi - 1).max() <= 0.1 + 1e-15
    assert not np.array_equal(a[0], a[1])


def test_shard_range_covers_everything():
    from successiveconvexification_amd.batch import shard_range
    for total, world in [(8192, 8), (10, 3), (5, 8), (0, 2)]:
        cuts = [shard_range(total, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def test_bench_roofline_byte_model_matches_survey():
    """SURVEY.md §8d / BASELINE.md §3: algorithmic HBM bytes of K1 per trajectory."""
    import bench
    assert bench.k1_alg_bytes(50, 3, 8) == 130144
    assert bench.k1_alg_bytes(50, 3, 4) == 65072
    assert bench.k1_alg_bytes(100, 3, 4) == 130072
    assert bench.k1_alg_bytes(100, 5, 4) == 153280
    assert bench.HBM_PEAK == 8.0e12
    t = bench.k1_measured_traffic(8192)
    assert t is None or 0.9 < t["bytes"] / (bench.k1_alg_bytes(50) * 8192) < 1.2  # PMC: no wasted re-reads


def test_plot_solution_data_matches_the_reference_formulas(tmp_path):
    """rocketland.jl:454-478 restated literally (per node, scalar code) against the vectorised dump."""
    from successiveconvexification_amd import rocketland as R, sample_problems as sp
    from successiveconvexification_amd.defns import LinPoint, ProblemIteration
    rng = np.random.default_rng(3)
    about = []
    for k in range(7):
        x = rng.normal(size=14)
        x[7:11] /= np.linalg.norm(x[7:11])
        about.append(LinPoint(x, rng.normal(size=3)))
    ip = ProblemIteration(sp.base_prob_scaled, None, 1.25, about, [], None, 3, 1.0, 0.0)
    d = R.plot_solution_data(ip)

    def DCM(q):  # dynamics.jl:29-44
        q0, q1, q2, q3 = q
        return np.array([[1 - 2 * (q2**2 + q3**2), 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
                         [2 * (q1 * q2 + q0 * q3), 1 - 2 * (q1**2 + q3**2), 2 * (q2 * q3 - q0 * q1)],
                         [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 1 - 2 * (q1**2 + q2**2)]])
    for k, pt in enumerate(about):
        dv = DCM(pt.state[7:11]) @ np.array([1.0, 0, 0])
        assert d["dp"][k] == pytest.approx(dv @ pt.state[4:7] / np.linalg.norm(pt.state[4:7]), rel=1e-13)
        assert d["thr"][k] == pytest.approx(np.linalg.norm(pt.control) / sp.base_prob_scaled.Tmax, rel=1e-13)
        assert np.allclose(d["xls"][k], [[pt.state[2], pt.state[3]], [pt.state[2] + dv[1] / 3, pt.state[3] + dv[2] / 3]], rtol=0, atol=1e-15)
        assert np.allclose(d["yls"][k], [[pt.state[1], pt.state[1]], [pt.state[1] + dv[0] / 3, pt.state[1] + dv[0] / 3]], rtol=0, atol=1e-15)
        assert np.array_equal(d["xs"][k], [pt.state[2], pt.state[3]]) and np.array_equal(d["ys"][k], [pt.state[1], pt.state[1]])
    up = np.array([pt.state[1] for pt in about]); ry = np.array([pt.state[2] for pt in about])
    assert d["xlims"] == (min(up.min(), ry.min()), max(up.max(), ry.max()))
    f = R.dump_solution(ip, str(tmp_path / "sol.csv"))
    tab = np.loadtxt(f, delimiter=",", skiprows=1)
    assert tab.shape == (7, 9) and np.array_equal(tab[:, 4], d["thr"])
    z = np.load(R.dump_solution(ip, str(tmp_path / "sol.npz")))
    assert np.array_equal(z["dp"], d["dp"]) and float(z["sigma"]) == 1.25


def test_aero_csv_loader_and_fin_table(tmp_path, aero_tables):
    """Aerodynamics.load_aerodata (aerodynamics.jl:11-28) on the CSV layout of aero/lift_drag.csv -- written here from the
    golden tables -- and the fin table the reference reads and drops (aerodynamics.jl:23-26)."""
    from successiveconvexification_amd import aerodynamics as ae
    d, l, t = aero_tables                     # [61 mach][181 aoa]
    n_mach, n_aoa = d.shape
    aoa = np.tile(np.arange(n_aoa), n_mach)   # cos(AoA) index fastest, Mach slowest: reshape(col, 181, 61) in the reference
    mach = np.repeat(np.arange(n_mach) * 0.025, n_aoa)
    csv = tmp_path / "lift_drag.csv"
    np.savetxt(csv, np.column_stack([aoa, mach, d.ravel(), l.ravel(), t.ravel()]), delimiter=",",
               header="aoa,mach,drag,lift,torque", comments="")
    fin = tmp_path / "fin.csv"
    fm, fa = np.array([0.01, 0.035, 0.06]), np.array([0.0, 0.1, 0.2, 0.3])
    rows = [(m * a, -m * a * a, m, a) for m in fm for a in fa]
    np.savetxt(fin, np.array(rows), delimiter=",", header="lift,drag,mach,aoa", comments="")
    a = ae.load_aerodata(str(csv), str(fin))
    assert np.array_equal(a.drag_itrp, d) and np.array_equal(a.lift_itrp, l) and np.array_equal(a.trq_itrp, t)
    m2, a2, lift, drag = ae.load_fin_table(str(fin))
    assert np.allclose(m2, fm) and np.allclose(a2, fa) and lift.shape == (3, 4)
    assert np.allclose(lift, fm[:, None] * fa[None, :]) and np.allclose(drag, -fm[:, None] * fa[None, :] ** 2)
    # the fin table's one use: a data-derived bound for the fin cone (opt-in; the reference's commented code has the constant 0.01)
    assert np.isclose(ae.fin_force_bound((m2, a2, lift, drag), 0.035, 2.0), 2.0 * 0.035 * 0.3)
    assert np.isclose(ae.fin_force_bound((m2, a2, lift, drag), 0.0475, 1.0), 0.0475 * 0.3)      # linear in Mach between rows
    assert np.isclose(ae.fin_force_bound((m2, a2, lift, drag), 9.0, 1.0), 0.06 * 0.3)           # flat outside
    from successiveconvexification_amd import sample_problems as sp
    pf = sp.base_prob_fin_scaled(a, fin_table=(m2, a2, lift, drag))
    assert pf.nu == 5 and pf.model_flags & 2 and 0 < pf.finmxf < 1e-6 and sp.base_prob_fin_scaled().finmxf == 0.01
    assert np.allclose(pf.rFB, [0.002, 0, 0])
    bad = tmp_path / "bad.csv"
    np.savetxt(bad, np.array(rows[:-1]), delimiter=",", header="lift,drag,mach,aoa", comments="")
    with pytest.raises(ValueError):
        ae.load_aerodata(str(csv), str(bad))


def test_bench_gpus_n_launches_its_own_ranks_and_fails_with_them():
    """`python bench.py --gpus 2` from a plain environment: the parent starts two rank processes itself, never
    imports torch, and exits non-zero when its ranks fail -- here they must fail, loudly: there is no GPU in this container and the
    HIP path has no CPU fallback.  (The same command on the GPU box is tests/test_gpu_scvx.py::test_bench_gpus_2_from_a_plain_shell_…)"""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by the -m gpu test")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2 and "a rank process failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no line, rather than a wrong one


def test_bench_rounds_the_timed_steps_to_whole_solve_problem_periods():
    """ADVICE r3 / VERDICT r3 item 3: the timed region is whole periods of imax - 1 = 14 solve_steps (the source states the rule once)."""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert "steps = ((args.steps + period - 1) // period) * period if whole else args.steps" in src
    period = 14
    for asked, timed in ((1, 14), (14, 14), (15, 28), (20, 28), (28, 28)):
        assert ((asked + period - 1) // period) * period == timed
