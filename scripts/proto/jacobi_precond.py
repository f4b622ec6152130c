"""Sweep counts of one-sided cyclic Jacobi on X = G (what the device does) against X = L^T / L of a Cholesky factorisation
G = L L^T (with and without diagonal pivoting): does a Cholesky pre-step shorten the tournament?  CPU prototype, numpy."""
import numpy as np, sys

def rr_pairs(n):
    idx = list(range(n))
    for _ in range(n - 1):
        yield [(idx[i], idx[n - 1 - i]) for i in range(n // 2)]
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]

def jacobi_sweeps(X, tol=1e-9, max_sweeps=40):
    """columns of X are rotated until mutually orthogonal; returns (sweeps, history of max |cos|)"""
    X = X.copy()
    n = X.shape[1]
    hist = []
    for sweep in range(max_sweeps):
        worst = 0.0
        for pairs in rr_pairs(n):
            p = np.array([a for a, b in pairs]); q = np.array([b for a, b in pairs])
            xp, xq = X[:, p], X[:, q]
            app = (xp * xp).sum(0); aqq = (xq * xq).sum(0); apq = (xp * xq).sum(0)
            c = np.abs(apq) / np.sqrt(app * aqq + 1e-300)
            worst = max(worst, c.max())
            rot = c > 1e-15
            zeta = (aqq - app) / (2.0 * np.where(rot, apq, 1.0))
            t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(1.0 + zeta * zeta))
            t = np.where(zeta == 0, 1.0, t)
            cs = 1.0 / np.sqrt(1.0 + t * t); sn = cs * t
            cs = np.where(rot, cs, 1.0); sn = np.where(rot, sn, 0.0)
            X[:, p], X[:, q] = cs * xp - sn * xq, sn * xp + cs * xq
        hist.append(worst)
        if worst < tol:
            return sweep + 1, hist
    return max_sweeps, hist

def chol_pivoted(G):
    n = G.shape[0]
    A = G.copy(); perm = np.arange(n); L = np.zeros_like(G)
    for k in range(n):
        j = k + int(np.argmax(np.diag(A)[k:]))
        if j != k:
            A[[k, j], :] = A[[j, k], :]; A[:, [k, j]] = A[:, [j, k]]
            L[[k, j], :] = L[[j, k], :]; perm[[k, j]] = perm[[j, k]]
        L[k, k] = np.sqrt(A[k, k])
        L[k + 1:, k] = A[k + 1:, k] / L[k, k]
        A[k + 1:, k + 1:] -= np.outer(L[k + 1:, k], L[k + 1:, k])
    return L, perm

rng = np.random.default_rng(0)
for (N, M, label) in ((128, 1152, "wishart 128 x 1152"), (256, 2304, "wishart 256 x 2304"), (288, 384, "deit-like 288 x 384"), (256, 2304, "decaying")):
    A = rng.standard_normal((N, M))
    if label == "decaying":
        u, s, vt = np.linalg.svd(A, full_matrices=False); A = (u * (s * np.exp(-np.arange(N) / 30.0))) @ vt
    G = A @ A.T
    G /= np.linalg.norm(G, 2)
    s_g, h_g = jacobi_sweeps(G)
    L = np.linalg.cholesky(G)
    s_l, h_l = jacobi_sweeps(L)          # columns of L: G = L L^T -> L = U S V^T, rotations on the right give U S
    s_lt, h_lt = jacobi_sweeps(L.T.copy())
    Lp, perm = chol_pivoted(G)
    s_p, h_p = jacobi_sweeps(Lp)
    s_pt, h_pt = jacobi_sweeps(Lp.T.copy())
    print(label, "| sweeps: G", s_g, "| L", s_l, "| L^T", s_lt, "| pivoted L", s_p, "| pivoted L^T", s_pt, flush=True)
    print("   G  ", ["%.1e" % x for x in h_g]); print("   L^T", ["%.1e" % x for x in h_lt]); print("   pL^T", ["%.1e" % x for x in h_pt])
