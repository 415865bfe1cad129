"""Block cyclic reduction for the block-tridiagonal Schur complement of the conic solve -- NUMERICS PROTOTYPE (test infrastructure, numpy).

DESIGN "what comes next" (end of round 4): the conic kernel's state can only move on chip if the 50-step factorisation / substitution
chain over the segments gets shorter.  Block cyclic reduction eliminates every other node at once (a nested-dissection order of the same
symmetric elimination): ceil(log2 K) = 6 dependent levels instead of 50 (26 two-ended) steps.  Before that is written as a kernel, this
file answers the one question that can be answered on the CPU: does the reordered elimination keep the accuracy of the sequential block
Cholesky on the ILL-CONDITIONED matrices the interior-point iteration really produces (cond ~ 1e7 .. 1e12 in the endgame)?

    python oracle/bcr_proto.py          # table: per interior-point iteration, cond(S), relative error / residual of both methods

S = E Hb^-1 E' with 14 x 14 blocks: Sd[k] (diagonal), So[k] = block (k+1, k).  The matrices are captured from oracle/ipm_struct.py (the
numpy design twin of the device solver) on the sample problem's first subproblems.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def seq_cholesky_solve(Sd, So, r):
    """The shipped order: block Cholesky k = 0 .. K-1, forward / backward substitution (Solver::build_kkt, S_solve)."""
    K = Sd.shape[0]
    L = np.zeros_like(Sd); Wb = np.zeros_like(So)
    for k in range(K):
        M = Sd[k].copy()
        if k > 0:
            M -= Wb[k - 1] @ Wb[k - 1].T
        L[k] = np.linalg.cholesky(M)
        if k + 1 < K:
            Wb[k] = np.linalg.solve(L[k], So[k].T).T
    t = np.zeros_like(r)
    for k in range(K):
        rr = r[k].copy()
        if k > 0:
            rr -= Wb[k - 1] @ t[k - 1]
        t[k] = np.linalg.solve(L[k], rr)
    x = np.zeros_like(r)
    for k in range(K - 1, -1, -1):
        rr = t[k].copy()
        if k + 1 < K:
            rr -= Wb[k].T @ x[k + 1]
        x[k] = np.linalg.solve(L[k].T, rr)
    return x


def bcr_solve(Sd, So, r, stats=None):
    """Block cyclic reduction: at every level the odd-indexed nodes of the current system are eliminated (all of them independent of each
    other), the even ones form the next system; each pivot block is SPD (a Schur complement of S) and is applied through its Cholesky
    factor.  Lo[i] = block (i, i-1) of the current system (Lo[0] unused)."""
    D = [Sd[k].copy() for k in range(Sd.shape[0])]
    Lo = [None] + [So[k].copy() for k in range(So.shape[0])]
    rr = [r[k].copy() for k in range(r.shape[0])]
    levels = []
    idx = list(range(len(D)))          # original node index of each row of the current system
    while len(D) > 1:
        n = len(D)
        odd = list(range(1, n, 2))
        chol = {i: np.linalg.cholesky(D[i]) for i in odd}

        def inv_apply(i, X):
            return np.linalg.solve(chol[i].T, np.linalg.solve(chol[i], X))
        Dn, Ln, rn = [], [None], []
        for j in range(0, n, 2):
            Dj, rj = D[j].copy(), rr[j].copy()
            if j - 1 >= 0:       # neighbour j-1 (odd): block (j, j-1) = Lo[j]
                G = inv_apply(j - 1, Lo[j].T)                # D_{j-1}^-1 Lo[j]'
                Dj -= Lo[j] @ G
                rj -= Lo[j] @ inv_apply(j - 1, rr[j - 1])
            if j + 1 < n:        # neighbour j+1 (odd): block (j+1, j) = Lo[j+1]
                G = inv_apply(j + 1, Lo[j + 1])              # D_{j+1}^-1 Lo[j+1]
                Dj -= Lo[j + 1].T @ G
                rj -= Lo[j + 1].T @ inv_apply(j + 1, rr[j + 1])
            Dn.append(Dj); rn.append(rj)
            if j >= 2:           # new coupling (j, j-2) through the eliminated j-1:  - Lo[j] D_{j-1}^-1 Lo[j-1]
                Ln.append(-Lo[j] @ inv_apply(j - 1, Lo[j - 1]))
        levels.append((D, Lo, rr, chol, odd))
        if stats is not None:
            stats.append(len(odd))
        D, Lo, rr = Dn, Ln, rn
    x_top = [np.linalg.solve(D[0], rr[0])]
    # back substitution, level by level
    for (D, Lo, rr, chol, odd) in reversed(levels):
        n = len(D)
        x = [None] * n
        for q, j in enumerate(range(0, n, 2)):
            x[j] = x_top[q]
        for i in odd:
            rhs = rr[i] - Lo[i] @ x[i - 1]
            if i + 1 < n:
                rhs = rhs - Lo[i + 1].T @ x[i + 1]
            x[i] = np.linalg.solve(chol[i].T, np.linalg.solve(chol[i], rhs))
        x_top = x
    return np.stack(x_top)


def dense(Sd, So):
    K = Sd.shape[0]
    S = np.zeros((14 * K, 14 * K))
    for k in range(K):
        S[14 * k:14 * k + 14, 14 * k:14 * k + 14] = Sd[k]
        if k + 1 < K:
            S[14 * (k + 1):14 * (k + 2), 14 * k:14 * k + 14] = So[k]
            S[14 * k:14 * k + 14, 14 * (k + 1):14 * (k + 2)] = So[k].T
    return S


def main():
    from oracle import ipm_struct, model, scvx
    rng = np.random.default_rng(0)
    p = model.base_prob_scaled()
    it0 = scvx.create_initial(p, 10)
    it1, _, _ = scvx.solve_step(it0)          # the second subproblem (after an accepted step) as well
    print("| subproblem | IPM iteration | cond(S) | seq. Cholesky: rel. error / rel. residual | cyclic reduction: rel. error / rel. residual |")
    print("|---|---|---|---|---|")
    worst = [0.0, 0.0]
    for tag, itx in (("first", it0), ("second", it1)):
        cap = []
        ipm_struct.CAPTURE = cap
        ipm_struct.solve(p, itx.x, itx.u, itx.endpoint, itx.deriv, itx.rk, tol=1e-9)
        ipm_struct.CAPTURE = None
        for n, (Sd, So) in enumerate(cap):
            S = dense(Sd, So)
            xt = rng.normal(size=(Sd.shape[0], 14))
            r = (S @ xt.ravel()).reshape(-1, 14)
            out = []
            for q, fn in enumerate((seq_cholesky_solve, bcr_solve)):
                x = fn(Sd, So, r)
                err = np.linalg.norm(x - xt) / np.linalg.norm(xt)
                res = np.linalg.norm(S @ x.ravel() - r.ravel()) / np.linalg.norm(r)
                out.append((err, res))
                if n > 0:
                    worst[q] = max(worst[q], res)
            if n in (0, 1) or n % 3 == 0 or n == len(cap) - 1:
                print("| %s | %d | %.1e | %.1e / %.1e | %.1e / %.1e |" % (tag, n, np.linalg.cond(S), out[0][0], out[0][1], out[1][0], out[1][1]))
    lv = []
    bcr_solve(Sd, So, r, lv)
    print("\nlevels of the reduction at K = %d (nodes eliminated per level, all independent): %s + the last node" % (Sd.shape[0], lv))
    print("worst relative residual over all captured factorisations: sequential %.1e, cyclic reduction %.1e" % tuple(worst))


if __name__ == "__main__":
    main()
