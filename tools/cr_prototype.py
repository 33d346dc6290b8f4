"""Index-level prototype of bandchol3.hpp (block odd-even reduction of the band + arrow system), in numpy.

Same data flow as the kernels: super-blocks of `bw` frames (m = 6 bw columns; a block-banded matrix with frame band width bw is
block TRIDIAGONAL in these super-blocks), level l eliminates the active blocks with odd index i (block id i * 2**l), every active
block first PULLS the updates of the previous level's eliminations of its two neighbours from their stored panels, an eliminated
block derives its couplings to its new neighbours from the same panels, the arrow-block contributions are kept per eliminated
block and summed in block order at the end.  Used by tests/test_cr_prototype.py to pin the scheme against a dense solve.
"""
import numpy as np


def band_to_blocks(Sband, Sarrow, F, bw, NA):
    """Sband[f][dd][6][6] = block (row frame f, column frame f - dd), Sarrow[a][0 : 6F + NA + 1] (row NA = rhs).
    Returns D[k] (m x m, lower valid), E0[k] = S[k, k-1] (m x m), A[k] ((NA+1) x m), AA ((NA+1) x (NA+1), lower), nb, m."""
    m = 6 * bw
    nb = (F + bw - 1) // bw
    Fp = nb * bw
    D = np.zeros((nb, m, m)); E0 = np.zeros((nb, m, m)); A = np.zeros((nb, NA + 1, m))
    for f in range(Fp):
        k, lf = divmod(f, bw)
        if f >= F:   # padding frames: identity rows
            D[k, 6 * lf:6 * lf + 6, 6 * lf:6 * lf + 6] = np.eye(6)
            continue
        for dd in range(min(bw, f) + 1):
            g = f - dd
            kg, lg = divmod(g, bw)
            blk = Sband[f, dd]
            if kg == k:
                D[k, 6 * lf:6 * lf + 6, 6 * lg:6 * lg + 6] = np.tril(blk) if dd == 0 else blk
            else:
                E0[k, 6 * lf:6 * lf + 6, 6 * lg:6 * lg + 6] = blk
        A[k, :, 6 * lf:6 * lf + 6] = Sarrow[:, 6 * f:6 * f + 6]
    AA = np.tril(Sarrow[:, 6 * F:6 * F + NA + 1])
    return D, E0, A, AA, nb, m


def sym(L):
    return np.tril(L) + np.tril(L, -1).T


def cr_solve(Sband, Sarrow, F, bw, NA):
    D, E0, A, AA, nb, m = band_to_blocks(Sband, Sarrow, F, bw, NA)
    D = np.array([sym(d) for d in D])
    NAx = NA + 1
    panels = {}        # eliminated block e -> dict(Pl, Pr, Pa, LinvT, left, right, level)
    order = []         # levels: list of lists of eliminated blocks
    s = 1
    level = 0
    while True:
        active = list(range(0, nb, s))
        if len(active) == 1:
            break
        half = s // 2
        newD = {}; newA = {}
        for b in active:
            Db = D[b].copy(); Ab = A[b].copy()
            if level > 0:   # pull the previous level's updates
                e1, e2 = b - half, b + half
                if e1 >= 0 and e1 in panels:
                    P = panels[e1]
                    Db -= P["Pr"] @ P["Pr"].T; Ab -= P["Pa"] @ P["Pr"].T
                if e2 < nb and e2 in panels:
                    P = panels[e2]
                    Db -= P["Pl"] @ P["Pl"].T; Ab -= P["Pa"] @ P["Pl"].T
            newD[b] = Db; newA[b] = Ab
        for b in active:
            D[b] = newD[b]; A[b] = newA[b]
        elim = [b for i, b in enumerate(active) if i % 2 == 1]
        for e in elim:
            l, r = e - s, e + s
            has_r = r < nb
            # couplings S[e, l] and S[r, e]: the original band at level 0, the previous level's panels otherwise
            if level == 0:
                Cl = E0[e]
                Cr = E0[r] if has_r else None
            else:
                P1 = panels[e - half]                      # eliminated between l and e
                Cl = -(P1["Pr"] @ P1["Pl"].T)              # rows of e, columns of l
                if has_r:
                    P2 = panels[e + half]                  # eliminated between e and r
                    Cr = -(P2["Pr"] @ P2["Pl"].T)          # rows of r, columns of e
                else:
                    Cr = None
            Lc = np.linalg.cholesky(D[e])
            LinvT = np.linalg.inv(Lc).T
            Pl = Cl.T @ LinvT                               # rows: variables of l
            Pr = (Cr @ LinvT) if has_r else np.zeros((m, m))
            Pa = A[e] @ LinvT                               # (NA+1) x m, last row = y_e
            panels[e] = dict(Pl=Pl, Pr=Pr, Pa=Pa, LinvT=LinvT, left=l, right=r if has_r else -1, level=level, AApart=np.tril(Pa @ Pa.T))
        order.append(elim)
        s *= 2
        level += 1
    # last level's updates of block 0, then the final dense system [D0 | arrow]
    half = s // 2
    if half >= 1 and half in panels:
        P = panels[half]
        D[0] = D[0] - P["Pl"] @ P["Pl"].T; A[0] = A[0] - P["Pa"] @ P["Pl"].T
    AAf = AA.copy()
    for e in sorted(panels):
        AAf -= panels[e]["AApart"]
    n = m + NA
    M = np.zeros((n, n)); rhs = np.zeros(n)
    M[:m, :m] = sym(D[0]); M[m:, :m] = A[0][:NA]; M[:m, m:] = A[0][:NA].T; M[m:, m:] = sym(AAf[:NA, :NA])
    rhs[:m] = A[0][NA]; rhs[m:] = AAf[NA, :NA]
    Lf = np.linalg.cholesky(M)
    xf = np.linalg.solve(Lf.T, np.linalg.solve(Lf, rhs))
    x = np.zeros((nb, m)); x[0] = xf[:m]; xa = xf[m:]
    for elim in reversed(order):
        for e in elim:
            P = panels[e]
            t = P["Pa"][NA].copy()                         # y_e
            t -= P["Pl"].T @ x[P["left"]]
            if P["right"] >= 0:
                t -= P["Pr"].T @ x[P["right"]]
            t -= P["Pa"][:NA].T @ xa
            x[e] = P["LinvT"] @ t
    return np.concatenate([x.reshape(-1)[:6 * F], xa])


def random_system(F, bw, NA, seed=0):
    rng = np.random.default_rng(seed)
    n = 6 * F + NA
    J = np.zeros((4 * n, n))
    # random sparse-ish rows touching frames within a window of bw+1 and the arrow columns
    for i in range(4 * n):
        f0 = rng.integers(0, max(1, F - bw))
        fs = rng.choice(np.arange(f0, min(F, f0 + bw + 1)), size=min(3, bw + 1), replace=False)
        for f in fs:
            J[i, 6 * f:6 * f + 6] = rng.normal(size=6)
        J[i, 6 * F:] = rng.normal(size=NA) * 0.3
    S = J.T @ J + 1e-3 * np.eye(n)
    rhs = rng.normal(size=n)
    Sband = np.zeros((F, bw + 1, 6, 6)); Sarrow = np.zeros((NA + 1, 6 * F + NA + 1))
    for f in range(F):
        for dd in range(min(bw, f) + 1):
            Sband[f, dd] = S[6 * f:6 * f + 6, 6 * (f - dd):6 * (f - dd) + 6]
    Sarrow[:NA, :6 * F] = S[6 * F:, :6 * F]
    Sarrow[:NA, 6 * F:6 * F + NA] = S[6 * F:, 6 * F:]
    Sarrow[NA, :6 * F] = rhs[:6 * F]
    Sarrow[NA, 6 * F:6 * F + NA] = rhs[6 * F:]
    return S, rhs, Sband, Sarrow


if __name__ == "__main__":
    for (F, bw, NA) in [(40, 3, 5), (37, 4, 17), (334, 9, 17), (9, 2, 3), (8, 2, 3), (5, 1, 1)]:
        S, rhs, Sband, Sarrow = random_system(F, bw, NA, seed=F)
        x = cr_solve(Sband, Sarrow, F, bw, NA)
        xr = np.linalg.solve(S, rhs)
        print(F, bw, NA, "max rel err", np.max(np.abs(x - xr)) / np.max(np.abs(xr)))
