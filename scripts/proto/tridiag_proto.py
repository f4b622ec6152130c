"""numpy model of the single-workgroup eigen-solver planned for csrc/tridiag.hip: Householder tridiagonalisation,
bisection for the leading r eigenvalues, inverse iteration, back-transformation.  Written with the loops the
kernel will have (per-eigenvalue lanes), to settle the numerics before the HIP version."""
import numpy as np


def tridiagonalize(A):
    """LAPACK dsytd2('L') on a full symmetric copy.  Returns d, e, V (Householder vectors, v[k+1] = 1 implicit -> stored
    explicitly), tau."""
    A = A.copy()
    n = A.shape[0]
    d = np.zeros(n); e = np.zeros(max(n - 1, 0)); tau = np.zeros(max(n - 1, 0))
    V = np.zeros((n, n))          # V[k] = Householder vector of step k (length n, zero for i <= k)
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        alpha = x[0]
        xnorm2 = float(np.dot(x[1:], x[1:]))
        if xnorm2 == 0.0:
            tau[k] = 0.0
            e[k] = alpha
            V[k, k + 1] = 1.0
        else:
            beta = -np.copysign(np.sqrt(alpha * alpha + xnorm2), alpha)
            tau[k] = (beta - alpha) / beta
            v = x / (alpha - beta)
            v[0] = 1.0
            e[k] = beta
            V[k, k + 1:] = v
            A22 = A[k + 1:, k + 1:]
            p = tau[k] * (A22 @ v)
            w = p - (0.5 * tau[k] * np.dot(p, v)) * v
            A22 -= np.outer(v, w) + np.outer(w, v)
        d[k] = A[k, k]
    if n >= 2:
        d[n - 2] = A[n - 2, n - 2]
        e[n - 2] = A[n - 1, n - 2]
    d[n - 1] = A[n - 1, n - 1]
    return d, e, V, tau


def sturm_count(d, e2, x, pivmin):
    """number of eigenvalues < x"""
    cnt = 0
    q = d[0] - x
    if abs(q) < pivmin: q = -pivmin
    cnt += q < 0
    for i in range(1, len(d)):
        q = d[i] - x - e2[i - 1] / q
        if abs(q) < pivmin: q = -pivmin
        cnt += q < 0
    return cnt


def bisect_top(d, e, r, bits=40, probes=8):
    """the r largest eigenvalues (descending) by multi-section: `probes` interior points per round"""
    n = len(d)
    e2 = e * e
    gl = min(d[i] - (abs(e[i - 1]) if i else 0) - (abs(e[i]) if i < n - 1 else 0) for i in range(n))
    gu = max(d[i] + (abs(e[i - 1]) if i else 0) + (abs(e[i]) if i < n - 1 else 0) for i in range(n))
    tn = max(abs(gl), abs(gu))
    pivmin = np.finfo(float).tiny * max(1.0, float(e2.max()) if n > 1 else 1.0)
    gl -= 2 * tn * np.finfo(float).eps * n + 2 * pivmin
    gu += 2 * tn * np.finfo(float).eps * n + 2 * pivmin
    lam = np.zeros(r)
    rounds = int(np.ceil(bits / np.log2(probes + 1)))
    for j in range(r):            # eigenvalue index from the top: k-th smallest with k = n - 1 - j
        k = n - 1 - j             # want lambda with exactly k eigenvalues below it
        lo, hi = gl, gu           # invariant: count(lo) <= k < count(hi)
        for _ in range(rounds):
            xs = lo + (hi - lo) * (np.arange(1, probes + 1) / (probes + 1))
            cs = np.array([sturm_count(d, e2, x, pivmin) for x in xs])
            below = np.nonzero(cs <= k)[0]          # probes with count <= k are still <= lambda
            nlo = xs[below[-1]] if len(below) else lo
            above = np.nonzero(cs > k)[0]
            nhi = xs[above[0]] if len(above) else hi
            lo, hi = nlo, nhi
        lam[j] = 0.5 * (lo + hi)
    return lam, tn


def solve_shifted(d, e, lam, b, tn):
    """(T - lam I) x = b by Gaussian elimination with partial pivoting on the tridiagonal (dlagtf/dlagts style)."""
    n = len(d)
    eps = np.finfo(float).eps
    tol = max(eps * tn, np.finfo(float).tiny)
    a = d - lam                 # diagonal
    bsup = e.copy()             # super-diagonal
    c = e.copy()                # sub-diagonal
    dsup2 = np.zeros(max(n - 2, 0))     # second super-diagonal (fill-in)
    x = b.copy()
    for k in range(n - 1):
        if abs(a[k]) >= abs(c[k]):              # no interchange
            piv = a[k] if abs(a[k]) > tol else np.copysign(tol, a[k] if a[k] != 0 else 1.0)
            a[k] = piv
            m = c[k] / piv
            a[k + 1] -= m * bsup[k]
            x[k + 1] -= m * x[k]
            # row k: (a_k, bsup_k, 0)
        else:                                   # swap rows k and k+1
            m = a[k] / c[k]
            ak1, bk1 = a[k + 1], (bsup[k + 1] if k + 1 < n - 1 else 0.0)
            # new row k = old row k+1: (c_k, a_{k+1}, b_{k+1}); new row k+1 = old row k - m * old row k+1
            newa = bsup[k] - m * ak1
            newb = -m * bk1
            a[k] = c[k]
            bs = ak1
            if k < n - 2:
                dsup2[k] = bk1
                bsup[k + 1] = newb
            bsup[k] = bs
            a[k + 1] = newa
            x[k], x[k + 1] = x[k + 1], x[k] - m * x[k + 1]
    if abs(a[n - 1]) <= tol:
        a[n - 1] = np.copysign(tol, a[n - 1] if a[n - 1] != 0 else 1.0)
    # back substitution
    x[n - 1] /= a[n - 1]
    if n >= 2:
        x[n - 2] = (x[n - 2] - bsup[n - 2] * x[n - 1]) / a[n - 2]
    for k in range(n - 3, -1, -1):
        x[k] = (x[k] - bsup[k] * x[k + 1] - dsup2[k] * x[k + 2]) / a[k]
    return x


def inverse_iteration(d, e, lam, tn, iters=3, cluster_tol=1e-3):
    n, r = len(d), len(lam)
    Z = np.zeros((r, n))
    rng = np.random.default_rng(1)
    clusters = 0
    for j in range(r):
        # LAPACK perturbs equal/close shifts: keep shifts at least 10 eps tn apart inside a cluster
        shift = lam[j]
        start = j
        while start > 0 and abs(lam[start - 1] - lam[start]) <= cluster_tol * tn:
            start -= 1
        if start < j:
            clusters += 1
            sep = 10 * np.finfo(float).eps * tn
            if lam[j - 1] - shift < sep:
                shift = lam[j - 1] - sep if j - 1 >= start else shift
        x = rng.uniform(-1, 1, n)
        for it in range(iters):
            x = solve_shifted(d, e, shift, x, tn)
            for i in range(start, j):               # MGS inside the cluster
                x -= np.dot(Z[i], x) * Z[i]
            x /= np.linalg.norm(x)
        Z[j] = x
    return Z, clusters


def back_transform(Z, V, tau):
    """rows of Z are eigenvectors of T; apply Q = H_0 H_1 ... H_{n-3}: u = Q z"""
    U = Z.copy()
    n = V.shape[0]
    for k in range(n - 3, -1, -1):
        v = V[k]
        U -= tau[k] * np.outer(U @ v, v)
    return U


def eig_top(G, r, bits=40):
    d, e, V, tau = tridiagonalize(G)
    lam, tn = bisect_top(d, e, r, bits=bits)
    Z, clusters = inverse_iteration(d, e, lam, tn)
    theta = np.array([z @ (d * z) + 2 * np.dot(e * z[:-1], z[1:]) for z in Z])      # Rayleigh quotients
    U = back_transform(Z, V, tau)
    return theta, U, clusters


def check(name, G, r):
    theta, U, clusters = eig_top(G, r)
    w, v = np.linalg.eigh(G)
    w, v = w[::-1], v[:, ::-1]
    orth = np.abs(U @ U.T - np.eye(r)).max()
    resid = np.abs(G @ U.T - U.T * theta).max() / max(abs(w).max(), 1e-300)
    ev = np.abs(theta - w[:r]).max() / max(abs(w).max(), 1e-300)
    P, Pr = U.T @ U, v[:, :r] @ v[:, :r].T
    gap = (w[r - 1] - w[r]) / abs(w).max() if r < len(w) else 1.0
    print("%-28s n=%3d r=%3d clusters=%3d  orth %.1e  resid %.1e  eval %.1e  proj %.1e  (rel gap at cut %.1e)" %
          (name, G.shape[0], r, clusters, orth, resid, ev, np.abs(P - Pr).max(), gap))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    A = rng.standard_normal((64, 576)).astype(np.float32).astype(np.float64)
    check("gram 64x576", A @ A.T, 25)
    check("gram 64x576 all", A @ A.T, 64)
    A = rng.standard_normal((64, 27)); check("rank 27 of 64, top 25", A @ A.T, 25)
    check("rank 27 of 64, top 30", A @ A.T, 30)
    check("constant", np.ones((32, 32)) * 0.25, 4)
    check("identity", np.eye(32), 8)
    D = np.diag(np.concatenate([np.linspace(2, 1, 16), np.linspace(2, 1, 16) + 1e-9]))
    Q, _ = np.linalg.qr(rng.standard_normal((32, 32)))
    check("close pairs 1e-9", Q @ D @ Q.T, 12)
    A = rng.standard_normal((192, 4608)); G = A @ A.T
    Qb, _ = np.linalg.qr(rng.standard_normal((192, 192)))
    check("wishart 192 r=105", G, 105)
    A = rng.standard_normal((480, 4608)); G = A @ A.T
    w, v = np.linalg.eigh(G); Qs = v[:, -192:] @ np.linalg.qr(rng.standard_normal((192, 192)))[0]
    check("ritz H 192 (top of 480)", Qs.T @ G @ Qs, 105)


def sturm_count_prod(ds, e2s, x):
    """product form with periodic rescaling (the kernel's version): sign changes of p_0 = 1, p_1, ..., p_n"""
    p0, p1 = 1.0, ds[0] - x
    cnt = int(p1 < 0 or p1 == 0)
    for i in range(1, len(ds)):
        p2 = (ds[i] - x) * p1 - e2s[i - 1] * p0
        neg1 = p1 < 0 or (p1 == 0 and False)
        s1 = -1 if p1 < 0 else (1 if p1 > 0 else 0)
        s2 = -1 if p2 < 0 else (1 if p2 > 0 else 0)
        if s2 == 0:
            s2 = -s1 if s1 != 0 else -1
            p2v = p2
        cnt += (s1 != s2) if s1 != 0 else (s2 < 0)
        p0, p1 = p1, (p2 if p2 != 0 else (np.copysign(1e-300, -p0 if p0 != 0 else -1.0)))
        if i % 8 == 0:
            m = max(abs(p0), abs(p1))
            if m > 0:
                ex = np.frexp(m)[1]
                p0, p1 = np.ldexp(p0, -ex), np.ldexp(p1, -ex)
    return cnt


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    for n, mk in ((64, lambda: rng.standard_normal((64, 576))), (64, lambda: rng.standard_normal((64, 27))), (192, lambda: rng.standard_normal((192, 300)))):
        A = mk(); G = A @ A.T
        d, e, V, tau = tridiagonalize(G)
        tn = max(abs(d).max(), 1e-300) + 2 * abs(e).max()
        ds, e2s = d / tn, (e / tn) ** 2
        pivmin = np.finfo(float).tiny
        bad = 0
        for x in np.concatenate([rng.uniform(-0.1, 1.0, 300), np.linalg.eigvalsh(G) / tn * (1 + 1e-15)]):
            bad += sturm_count(ds, e2s, x, pivmin) != sturm_count_prod(ds, e2s, x)
        print("sturm product-form vs ratio-form mismatches:", bad, "of", 300 + n)
    # inverse iterations needed with eigenvalues accurate to `bits`
    A = rng.standard_normal((64, 576)); G = A @ A.T
    d, e, V, tau = tridiagonalize(G)
    for bits in (30, 40, 46):
        lam, tn = bisect_top(d, e, 25, bits=bits)
        for iters in (1, 2, 3):
            Z, _ = inverse_iteration(d, e, lam, tn, iters=iters)
            T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
            th = np.array([z @ T @ z for z in Z])
            res = np.abs(T @ Z.T - Z.T * th).max() / tn
            print("bits %d iters %d: resid %.1e orth %.1e" % (bits, iters, res, np.abs(Z @ Z.T - np.eye(25)).max()))
