"""oracle/bcr_proto.py: block cyclic reduction of a block-tridiagonal SPD system (the reordering DESIGN's "what comes next" plans for the
conic kernel's factorisation chain) solves what the sequential block Cholesky solves, to the same accuracy.  CPU, numpy."""
import numpy as np

from oracle import bcr_proto


def _system(K, rng, cond):
    # SPD and block tridiagonal by construction: diagonal blocks Q diag(1 .. 1/cond) Q' + 2 c I, couplings c x (an orthogonal matrix):
    # S >= the ill-conditioned diagonal part (the couplings are dominated by the 2 c I)
    nb, c = 14, 0.2
    Sd = np.zeros((K, nb, nb)); So = np.zeros((max(K - 1, 0), nb, nb))
    for k in range(K):
        Q, _ = np.linalg.qr(rng.normal(size=(nb, nb)))
        Sd[k] = Q @ np.diag(np.logspace(0, -np.log10(cond), nb)) @ Q.T + 2 * c * np.eye(nb) * (1.0 if 0 < k < K - 1 else 0.5)
        if k + 1 < K:
            So[k] = c * np.linalg.qr(rng.normal(size=(nb, nb)))[0]
    return Sd, So


def test_cyclic_reduction_matches_sequential_cholesky():
    rng = np.random.default_rng(3)
    for K in (1, 2, 3, 7, 8, 30, 50, 100):
        for cond in (1e2, 1e8):
            Sd, So = _system(K, rng, cond)
            S = bcr_proto.dense(Sd, So)
            xt = rng.normal(size=(K, 14))
            r = (S @ xt.ravel()).reshape(K, 14)
            xs = bcr_proto.seq_cholesky_solve(Sd, So, r)
            levels = []
            xb = bcr_proto.bcr_solve(Sd, So, r, levels)
            res_s = np.linalg.norm(S @ xs.ravel() - r.ravel()) / np.linalg.norm(r)
            res_b = np.linalg.norm(S @ xb.ravel() - r.ravel()) / np.linalg.norm(r)
            assert res_b < 1e-13 and res_b < 20 * res_s + 1e-15, (K, cond, res_s, res_b)
            assert np.linalg.norm(xb - xt) <= 20 * np.linalg.norm(xs - xt) + 1e-12 * np.linalg.norm(xt), (K, cond)
            assert len(levels) == int(np.ceil(np.log2(K))) if K > 1 else levels == []


def test_cyclic_reduction_on_the_schur_complements_of_a_real_solve():
    """The matrices that matter: the block-tridiagonal Schur complements of every interior-point iteration of the sample problem's first
    subproblem (captured from the numpy design twin, cond up to ~1e8).  The reordered elimination is as accurate as the sequential
    block Cholesky the device runs, iteration by iteration."""
    from oracle import ipm_struct, model, scvx
    p = model.base_prob_scaled()
    it0 = scvx.create_initial(p, 10)
    cap = []
    ipm_struct.CAPTURE = cap
    try:
        ipm_struct.solve(p, it0.x, it0.u, it0.endpoint, it0.deriv, it0.rk, tol=1e-9)
    finally:
        ipm_struct.CAPTURE = None
    assert len(cap) >= 15
    rng = np.random.default_rng(0)
    conds = []
    for Sd, So in cap[:28]:      # (the twin's last iterations sit on its numerical floor: both methods alike, see profiles/r04_bcr_numerics.txt)
        S = bcr_proto.dense(Sd, So)
        conds.append(np.linalg.cond(S))
        xt = rng.normal(size=(Sd.shape[0], 14))
        r = (S @ xt.ravel()).reshape(-1, 14)
        xs, xb = bcr_proto.seq_cholesky_solve(Sd, So, r), bcr_proto.bcr_solve(Sd, So, r)
        es, eb = np.linalg.norm(xs - xt) / np.linalg.norm(xt), np.linalg.norm(xb - xt) / np.linalg.norm(xt)
        assert eb < 10 * es + 1e-14, (conds[-1], es, eb)
        assert np.linalg.norm(S @ xb.ravel() - r.ravel()) / np.linalg.norm(r) < 1e-12
    assert max(conds) > 1e7
