"""Cook-Toom F(m, 4) along the four time taps: fp32 error of the transformed form against fp64, at the sizes of CDAE layers 2 / 3
(52 channels x kf frequency taps per output, ReLU inputs), for candidate point sets.  CPU only (numpy); the choice of points for
an F(3, 4) arm of csrc/cdae_wino.h was priced with it."""
import itertools, sys
import numpy as np
from fractions import Fraction as Fr


def cook_toom(m, r, pts):
    """AT (m x n), G (n x r), BT (n x n) for finite points pts (n - 1 of them) + infinity; y = AT [(G g) * (BT d)]."""
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fr(p) for p in pts]
    # evaluation matrices of polynomials of degree < k at the points (+ infinity -> leading coefficient)
    def ev(k):
        M = [[p ** j for j in range(k)] for p in pts]
        M.append([Fr(0)] * (k - 1) + [Fr(1)])
        return M
    # Toom-Cook for the linear convolution s = V^-1 [(ev(r) g) * (ev(m) d)], transposed into the FIR filter (the transposition
    # principle): AT = ev(m)^T, G = ev(r), BT = (V^-1)^T with V = ev(n)
    V = ev(n)
    Vinv = inv_frac(V)
    AT = [[ev(m)[i][j] for i in range(n)] for j in range(m)]
    G = ev(r)
    BT = [[Vinv[j][i] for j in range(n)] for i in range(n)]      # (V^-1)^T
    return AT, G, BT


def inv_frac(M):
    n = len(M)
    A = [list(row) + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(M)]
    for c in range(n):
        p = next(i for i in range(c, n) if A[i][c] != 0)
        A[c], A[p] = A[p], A[c]
        d = A[c][c]
        A[c] = [x / d for x in A[c]]
        for i in range(n):
            if i != c and A[i][c] != 0:
                f = A[i][c]
                A[i] = [x - f * y for x, y in zip(A[i], A[c])]
    return [row[n:] for row in A]


def balance(AT, G, BT):
    """Move the denominators of BT's rows into G (BT integer), as csrc/cdae_wino.h does for F(2, 4)."""
    from math import lcm
    n = len(BT)
    for i in range(n):
        L = 1
        for x in BT[i]:
            L = lcm(L, x.denominator)
        BT[i] = [x * L for x in BT[i]]
        G[i] = [x / L for x in G[i]]
    return AT, G, BT


def check(m, pts, trials=3, C=52, kf=3, N=4096, seed=0):
    r = 4
    AT, G, BT = balance(*cook_toom(m, r, pts))
    n = m + r - 1
    ATd, Gd, BTd = (np.array([[float(x) for x in row] for row in M]) for M in (AT, G, BT))
    # exactness in rational arithmetic: y_k = sum_dt g[dt] d[k + dt]
    rng = np.random.default_rng(seed)
    errs, refs = [], []
    for _ in range(trials):
        w = (rng.standard_normal((kf * C, r)) * (1.0 / np.sqrt(kf * C * r))).astype(np.float32)
        d = np.maximum(rng.standard_normal((N, kf * C, n)), 0).astype(np.float32)
        y64 = np.zeros((N, m))
        for k in range(m):
            y64[:, k] = np.einsum("ct,nct->n", w.astype(np.float64), d[:, :, k:k + r].astype(np.float64))
        U = (Gd @ w.astype(np.float64).T).T.astype(np.float32)              # (kfC, n): host fp64 -> fp32
        V = np.zeros((N, kf * C, n), np.float32)
        for j in range(n):                                                   # fp32 input transform
            acc = np.zeros((N, kf * C), np.float32)
            for i in range(n):
                if BTd[j, i] != 0:
                    acc = (acc + np.float32(BTd[j, i]) * d[:, :, i]).astype(np.float32)
            V[:, :, j] = acc
        M = np.zeros((N, n), np.float32)
        for c in range(kf * C):                                              # fp32 accumulation over the channels
            M = (M + V[:, c, :] * U[c, :]).astype(np.float32)
        y = np.zeros((N, m), np.float32)
        for k in range(m):
            acc = np.zeros(N, np.float32)
            for j in range(n):
                if ATd[k, j] != 0:
                    acc = (acc + np.float32(ATd[k, j]) * M[:, j]).astype(np.float32)
            y[:, k] = acc
        ydir = np.zeros((N, m), np.float32)
        for k in range(m):
            acc = np.zeros(N, np.float32)
            for c in range(kf * C):
                for t in range(r):
                    acc = (acc + d[:, c, k + t] * w[c, t]).astype(np.float32)
            ydir[:, k] = acc
        errs.append((np.sqrt(np.mean((y - y64) ** 2)), np.abs(y - y64).max(), np.sqrt(np.mean((ydir - y64) ** 2))))
        refs.append(np.sqrt(np.mean(y64 ** 2)))
    e = np.mean([x[0] for x in errs]); mx = np.max([x[1] for x in errs]); dd = np.mean([x[2] for x in errs])
    return e, mx, dd, np.mean(refs), BT, AT, G


if __name__ == "__main__":
    sets = {"F(2,4) {0,1,-1,2}": (2, [0, 1, -1, 2]),
            "F(3,4) {0,1,-1,2,-2}": (3, [0, 1, -1, 2, -2]),
            "F(3,4) {0,1,-1,2,1/2}": (3, [0, 1, -1, 2, Fr(1, 2)]),
            "F(3,4) {0,1,-1,1/2,-1/2}": (3, [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
            "F(3,4) {0,1,-1,2,-1/2}": (3, [0, 1, -1, 2, Fr(-1, 2)]),
            "F(4,4) {0,1,-1,2,-2,1/2}": (4, [0, 1, -1, 2, -2, Fr(1, 2)]),
            "F(4,4) {0,1,-1,2,-1/2,1/2}": (4, [0, 1, -1, 2, Fr(-1, 2), Fr(1, 2)])}
    for name, (m, pts) in sets.items():
        e, mx, dd, ref, BT, AT, G = check(m, pts)
        print(f"{name:32s} rms {e:.2e} max {mx:.2e}  direct fp32 rms {dd:.2e}  (|y| rms {ref:.2f})")
        if "-v" in sys.argv:
            for nm, M in (("BT", BT), ("AT", AT), ("G", G)):
                print(" ", nm, [[str(x) for x in row] for row in M])
