"""numpy prototype of the two-stage tridiagonalisation (dense -> band -> tridiagonal) that csrc/sbr.hip implements.
Checks the index arithmetic of the kernels: panel QR in WY form, two-sided trailing update, bulge chasing on the
lower band, back-transformation.  Not used by the product or the tests."""
import numpy as np

def house(x):
    """LAPACK slarfg: H = I - tau v v^T, v[0] = 1, H x = beta e1."""
    alpha = x[0]
    xn2 = float(np.dot(x[1:], x[1:]))
    if xn2 == 0.0:
        return np.concatenate([[1.0], np.zeros(len(x) - 1)]).astype(x.dtype), x.dtype.type(0), alpha
    beta = -np.copysign(np.sqrt(alpha * alpha + xn2), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1
    return v.astype(x.dtype), x.dtype.type(tau), x.dtype.type(beta)

def stage1(A, b):
    """A symmetric n x n (n % b == 0) -> band (half-bandwidth b) in place; returns V (n x n: panel k in rows
    (k+1)b.., cols kb..kb+b-1), T (n/b-1, b, b)."""
    n = A.shape[0]
    nb = n // b
    V = np.zeros_like(A)
    T = np.zeros((nb - 1, b, b), A.dtype)
    for k in range(nb - 1):
        r0 = (k + 1) * b
        m = n - r0
        P = A[r0:, k * b:(k + 1) * b].copy()           # m x b
        Vk = np.zeros((m, b), A.dtype)
        Tk = np.zeros((b, b), A.dtype)
        for j in range(min(b, m - 1) if m > 1 else 0):
            v, tau, beta = house(P[j:, j].copy())
            Vk[j:, j] = v
            P[j, j] = beta
            P[j + 1:, j] = 0
            # apply H_j to the remaining columns
            if j + 1 < b:
                s = tau * (v @ P[j:, j + 1:])
                P[j:, j + 1:] -= np.outer(v, s)
            # T: T[:j, j] = -tau * T[:j,:j] @ (V[:, :j]^T v) ; T[j, j] = tau
            Tk[j, j] = tau
            if j > 0:
                Tk[:j, j] = -tau * (Tk[:j, :j] @ (Vk[:, :j].T @ Vk[:, j]))
        # (m == b: the last panel is b x b; its R is upper triangular, columns b-1.. keep v = e? handled: j < m-1)
        if m >= 1 and min(b, m - 1) < b:
            for j in range(min(b, m - 1), b):
                if j < m:
                    Vk[j, j] = 1.0               # H = I (tau = 0)
        A[r0:, k * b:(k + 1) * b] = P
        A[k * b:(k + 1) * b, r0:] = P.T
        V[r0:, k * b:(k + 1) * b] = Vk
        T[k] = Tk
        # two-sided update of the trailing block
        A22 = A[r0:, r0:]
        Y = A22 @ Vk @ Tk
        W = Y - 0.5 * Vk @ (Tk.T @ (Vk.T @ Y))
        A22 -= Vk @ W.T + W @ Vk.T
    return V, T

def to_band(A, b):
    n = A.shape[0]
    AB = np.zeros((b + 1, n), A.dtype)
    for d in range(b + 1):
        AB[d, :n - d] = np.diagonal(A, -d)
    return AB

def chase(AB, b):
    """Lower band (AB[d, j] = A[j+d, j]) -> tridiagonal by Householder bulge chasing.  Returns d, e and the
    reflector list [(row0, v, tau)] in application order (A <- H A H each)."""
    n = AB.shape[1]
    dt = AB.dtype
    # work on a dense copy restricted to band + bulge, for clarity of the prototype
    A = np.zeros((n, n), dt)
    for d in range(b + 1):
        for j in range(n - d):
            A[j + d, j] = AB[d, j]
            A[j, j + d] = AB[d, j]
    refl = []
    for i in range(n - 2):
        # task 0: annihilate column i below the subdiagonal
        st = i + 1
        ln = min(b, n - st)
        if ln <= 1:
            continue
        v, tau, beta = house(A[st:st + ln, i].copy())
        A[st, i] = beta; A[i, st] = beta
        A[st + 1:st + ln, i] = 0; A[i, st + 1:st + ln] = 0
        while True:
            refl.append((st, v, tau))
            # two-sided on the diagonal block
            D = A[st:st + ln, st:st + ln]
            p = tau * (D @ v)
            w = p - 0.5 * tau * (p @ v) * v
            D -= np.outer(v, w) + np.outer(w, v)
            # off-diagonal block below: rows st+ln .. st+ln+l2, cols st..st+ln
            r2 = st + ln
            l2 = min(b, n - r2)
            if l2 <= 0:
                break
            Bk = A[r2:r2 + l2, st:st + ln]
            s = tau * (Bk @ v)
            Bk -= np.outer(s, v)                       # right-apply H
            if l2 <= 1:
                A[st:st + ln, r2:r2 + l2] = Bk.T
                break
            v2, tau2, beta2 = house(Bk[:, 0].copy())
            Bk[0, 0] = beta2; Bk[1:, 0] = 0
            s2 = tau2 * (v2 @ Bk[:, 1:])
            Bk[:, 1:] -= np.outer(v2, s2)              # left-apply H' to the remaining columns
            A[st:st + ln, r2:r2 + l2] = Bk.T
            st, ln, v, tau = r2, l2, v2, tau2
    d = np.diagonal(A).copy()
    e = np.diagonal(A, -1).copy()
    return d, e, refl, A

def back(V, T, refl, Z, b):
    """X = Q1 Q2 Z for columns of Z (eigenvectors of the tridiagonal)."""
    X = Z.copy()
    for (r0, v, tau) in reversed(refl):
        seg = X[r0:r0 + len(v)]
        seg -= tau * np.outer(v, v @ seg)
    n = V.shape[0]
    nb = n // b
    for k in reversed(range(nb - 1)):
        r0 = (k + 1) * b
        Vk = V[r0:, k * b:(k + 1) * b]
        X[r0:] -= Vk @ (T[k] @ (Vk.T @ X[r0:]))
    return X

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n, b in [(64, 32), (96, 32), (384, 32), (128, 16)]:
        Z0 = rng.standard_normal((4 * n, n))
        G = (Z0.T @ Z0 / (4 * n))
        for dt in (np.float64, np.float32):
            A = G.astype(dt).copy()
            V, T = stage1(A, b)
            off = np.abs(np.tril(A, -b - 1)).max()
            AB = to_band(A, b)
            d, e, refl, Afin = chase(AB, b)
            fill = np.abs(np.tril(Afin, -2)).max()
            Tm = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
            lam_ref = np.linalg.eigvalsh(G)
            lam, Zt = np.linalg.eigh(Tm.astype(np.float64))
            X = back(V.astype(np.float64), T.astype(np.float64), [(r, v.astype(np.float64), float(t)) for r, v, t in refl], Zt, b)
            res = np.abs(G @ X - X * lam).max()
            print(n, b, dt.__name__, "below band", off, "fill", fill, "eig err", np.abs(lam - lam_ref).max() / lam_ref.max(),
                  "resid", res, "orth", np.abs(X.T @ X - np.eye(n)).max(), "nrefl", len(refl))
