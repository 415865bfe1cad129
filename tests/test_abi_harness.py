"""The Julia binding's use of the C ABI, executed without Julia: tests/abi_harness.c replays the ccall sequence of
julia/ScvxAMD.jl on the reference's recipe (rocketland.jl:26-32, the aero problem) and must reproduce the Python-driven
run bit for bit (same library, same kernels, same data)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "successiveconvexification_amd")
SRC = os.path.join(ROOT, "tests", "abi_harness.c")


def _compile(tmp_path, link=True):
    exe = str(tmp_path / "abi_harness")
    cmd = ["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"), SRC]
    cmd += ["-o", exe, "-L", PKG, "-l:libscvx_hip.so", "-Wl,-rpath," + PKG] if link else ["-c", "-o", exe + ".o"]
    subprocess.check_call(cmd)
    return exe


def test_harness_compiles_against_the_header(tmp_path):
    """-m "not gpu": the harness is valid C against include/scvx.h (prototypes, struct, constants), warnings as errors."""
    _compile(tmp_path, link=False)


def test_julia_shim_struct_matches_header():
    """julia/ScvxAMD.jl's CProblem lists the fields of struct scvx_problem in the same order (names and multiplicity)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "scvx.h")).read()
    body = hdr[hdr.index("typedef struct scvx_problem {"):hdr.index("} scvx_problem;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hfields = []
    for decl in re.findall(r"(?:double|int32_t)\s+([^;]+);", body):
        for name in decl.split(","):
            m = re.match(r"\s*(\w+)(?:\[(\d+)\])?", name)
            hfields.append((m.group(1), int(m.group(2) or 1)))
    jl = open(os.path.join(ROOT, "julia", "ScvxAMD.jl")).read()
    sb = jl[jl.index("struct CProblem"):jl.index("\nend", jl.index("struct CProblem"))]
    jfields = []
    for name, typ in re.findall(r"(\w+)::(\w+(?:\{\d+,\w+\})?)", sb):
        n = re.match(r"NTuple\{(\d+),", typ)
        jfields.append((name, int(n.group(1)) if n else 1))
    assert jfields == hfields


def test_julia_shim_solver_opts_match_header():
    """julia/ScvxAMD.jl's SolverOpts and ThreedofOpts list the fields of scvx_solver_opts / scvx_threedof_opts in the header's order,
    with the header's types (int32_t <-> Int32, double <-> Cdouble)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "scvx.h")).read()
    jl = open(os.path.join(ROOT, "julia", "ScvxAMD.jl")).read()
    for cname, jname in (("scvx_solver_opts", "SolverOpts"), ("scvx_threedof_opts", "ThreedofOpts")):
        body = hdr[hdr.index("typedef struct %s {" % cname):hdr.index("} %s;" % cname)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        hfields = [(n, {"int32_t": "Int32", "double": "Cdouble"}[t]) for t, n in re.findall(r"(double|int32_t)\s+(\w+)\s*;", body)]
        sb = jl[jl.index("struct %s\n" % jname):jl.index("\nend", jl.index("struct %s\n" % jname))]
        jfields = re.findall(r"(\w+)::(\w+)", sb)
        assert len(hfields) >= 6 and [t for _, t in jfields] == [t for _, t in hfields], (cname, jfields, hfields)
        assert [n for n, _ in jfields][:4] == [n for n, _ in hfields][:4]


@pytest.mark.gpu
@pytest.mark.parametrize("fins", [False, True])
def test_harness_replays_the_julia_call_sequence_bit_for_bit(tmp_path, aero_tables, fins):
    from successiveconvexification_amd import _lib, sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache
    exe = _compile(tmp_path)
    d, l, t = aero_tables
    prob = sp.base_prob_fin_scaled(AtmosphericData(d, l, t)) if fins else sp.base_prob_aero_scaled(AtmosphericData(d, l, t))
    K, nsub, nstep = prob.K, 10, 2
    NU = prob.nu
    DSZ = 14 * (14 + 2 * NU + 1)
    cache = IntegratorCache(prob, npts=nsub)
    cp = cache.cproblem()                       # the flat struct the Python layer hands to scvx_ctx_create
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(bytes(cp))
        f.write(np.array([d.shape[1], d.shape[0], nsub, nstep], np.int32).tobytes())   # tables are [n_mach][n_aoa], AoA fastest
        f.write(np.array([-1.0, 1.0 / 90.0, 0.0, 0.025]).tobytes())
        for tab in (d, l, t):
            f.write(np.ascontiguousarray(tab, np.float64).tobytes())
    env = dict(os.environ)
    r = subprocess.run([exe, fin, fout], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = np.fromfile(fout)
    nrec = (K + 1) * (14 + NU) + 1
    per_iter = nrec + K * 14 + K * DSZ + 3
    n3 = (K + 1) * 15 + 1
    assert out.size == per_iter * (nstep + 1) + 3 * nstep + K * 14 + K * DSZ + 14 + n3 + 6

    b = ScvxBatch(cache, 1).init(None)

    def expect():
        e, dd = b.linearization()
        rk, cost, it = b.scalars()
        return np.concatenate([b.trajectory_record()[0], e.ravel(), dd.ravel(), [rk[0], cost[0], float(it[0])]])
    pos = 0
    ref = expect()
    assert np.array_equal(out[pos:pos + per_iter], ref, equal_nan=True)
    pos += per_iter
    for s in range(nstep):
        st, nu, dj = b.solve_step()
        ref = expect()
        assert np.array_equal(out[pos:pos + per_iter], ref, equal_nan=True), s
        pos += per_iter
        assert np.array_equal(out[pos:pos + 3], [float(st[0]), nu[0], dj[0]], equal_nan=True)
        pos += 3
    # Dynamics.linearize_dynamics / predict_state through the host entry points
    from successiveconvexification_amd.dynamics import linearize_batch, propagate_batch
    x, u, s = b.trajectory()
    e2, d2 = linearize_batch(cache, x, u, s, 1.0 / (K + 1))
    assert np.array_equal(out[pos:pos + K * 14], e2.ravel()); pos += K * 14
    assert np.array_equal(out[pos:pos + K * DSZ], d2.ravel()); pos += K * DSZ
    xs = np.zeros((1, 2, 14)); xs[0, 0] = x[0, 3]
    xp = propagate_batch(cache, xs, u[:, 3:5], s, 1.0 / (K + 1))
    assert np.array_equal(out[pos:pos + 14], xp[0, 0]); pos += 14
    # FirstRound.solve_initial through scvx_threedof_solve (whatever its status on this problem: the same bits)
    from successiveconvexification_amd import first_round
    L = cache._L
    o = first_round.threedof_opts(L, max_iter=45)
    rec = np.zeros(n3); st3 = np.zeros(1, np.int32); info = np.zeros(5)
    dp = C.POINTER(C.c_double)
    assert L.scvx_threedof_solve(cache.handle, 1, None, C.byref(o), rec.ctypes.data_as(dp), st3.ctypes.data_as(C.POINTER(C.c_int32)),
                                 info.ctypes.data_as(dp)) == 0
    assert np.array_equal(out[pos:pos + n3], rec, equal_nan=True); pos += n3
    assert np.array_equal(out[pos:pos + 6], np.concatenate([[float(st3[0])], info]), equal_nan=True)
    b.close(); cache.close()


# the reference's own function lines for the hot path (dynamics.jl:141, 258, 315, 321; rocketland.jl:34, 226, 432): name -> argument list
_REF_SIGNATURES = {
    "make_dynamics_module": "info::ProbInfo",
    "(::Type{IntegratorCache})": "prob::DescentProblem, info::ProbInfo, lin_mod",
    "predict_state": "initial_state, uk, up, sigma, dt, pinfo, cache",
    "linearize_dynamics": "states::Array{LinPoint,1}, tf_guess::Float64, base_dt::Float64, cache::IntegratorCache",
    "create_initial": "problem::DescentProblem, linear_cache::IntegratorCache",
    "solve_step": "iteration::ProblemIteration, linear_cache::IntegratorCache",
    "solve_problem": "iprob::DescentProblem, cache::IntegratorCache",   # the reference types cache::LinearCache, undefined at HEAD
}


def test_julia_install_defines_the_references_own_signatures():
    """ScvxAMD.install!() adds methods to Dynamics / Rocketland with the reference's exact argument lists,
    so the recipe of rocketland.jl:26-32 runs unedited.  Parsed, not executed (no Julia in the image); the expected argument
    lists are checked against /root/reference when it is present (this container), and held above otherwise."""
    import re
    jl = open(os.path.join(ROOT, "julia", "ScvxAMD.jl")).read()
    body = jl[jl.index("function install!()"):]
    found = dict(re.findall(r"^\s+function (\(::Type\{IntegratorCache\}\)|\w+)\(([^)]*)\)\s*$", body, flags=re.M))
    assert found == _REF_SIGNATURES, found
    assert "@eval HOST.Dynamics" in body and "@eval HOST.Rocketland" in body
    ref = "/root/reference"
    if os.path.isdir(ref):
        src = {f: open(os.path.join(ref, f)).read() for f in ("dynamics.jl", "rocketland.jl")}
        for name, args in _REF_SIGNATURES.items():
            fn = "dynamics.jl" if name in ("make_dynamics_module", "(::Type{IntegratorCache})", "predict_state", "linearize_dynamics") else "rocketland.jl"
            want = args if name != "solve_problem" else args.replace("cache::IntegratorCache", "cache::LinearCache")
            assert ("function " + name + "(" + want + ")") in src[fn], (name, fn)
