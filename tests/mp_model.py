"""Arbitrary-precision (mpmath, 40 digits) restatement of LiFCal's residual, written from the equations of
SURVEY.md Appendix A / reference src/CameraModel.h:86-264 and src/BundleAdjustment/BundleAdjustment.h:120-195,
independently of oracle/ (different language, rotation composed from plain matrices, no dual numbers).
Derivatives are taken by mpmath's high-order numerical differentiation, so an error in either restatement
shows up as a disagreement at the 1e-12 level instead of hiding behind finite-difference noise."""
import mpmath as mp

mp.mp.dps = 40


def _dist(x, y, k, p):
    r2 = x * x + y * y
    g = mp.mpf(0)
    ri = r2
    for i, ki in enumerate(k):
        if i:
            ri = ri * r2
        g += ki * ri
    dx, dy = x * g, y * g
    if p is not None:
        dx += p[0] * (r2 + 2 * x * x) + 2 * p[1] * x * y
        dy += p[1] * (r2 + 2 * y * y) + 2 * p[0] * x * y
    return dx, dy


def residual(x26, config, u, v, mcx, mcy, spx, scale):
    """x26 = camera[17] | view[6] | point[3] (mp numbers); returns (rx, ry)."""
    cam, view, P = x26[:17], x26[17:23], x26[23:26]
    n_rad = config & 3
    tan = bool(config & 4)
    adj = bool(config & 0x800)
    fL, bL0, B = abs(cam[0]), abs(cam[1]), abs(cam[2])
    sp = mp.mpf(spx) / mp.mpf(scale)
    craw = [abs((cam[3] + mp.mpf("0.5")) * scale - mp.mpf("0.5")), abs((cam[4] + mp.mpf("0.5")) * scale - mp.mpf("0.5"))]
    k = [cam[5 + i] for i in range(n_rad)]
    p = [cam[5 + n_rad], cam[6 + n_rad]] if tan else None
    a0, a1, a2 = view[0], view[1], view[2]
    Rx = mp.matrix([[1, 0, 0], [0, mp.cos(a0), -mp.sin(a0)], [0, mp.sin(a0), mp.cos(a0)]])
    Ry = mp.matrix([[mp.cos(a1), 0, mp.sin(a1)], [0, 1, 0], [-mp.sin(a1), 0, mp.cos(a1)]])
    Rz = mp.matrix([[mp.cos(a2), -mp.sin(a2), 0], [mp.sin(a2), mp.cos(a2), 0], [0, 0, 1]])
    pc = Rx * Ry * Rz * mp.matrix(P) + mp.matrix(view[3:6])
    cd = [(mp.mpf(mcx) - craw[0]) * sp, (mp.mpf(mcy) - craw[1]) * sp]
    cu = list(cd)
    if n_rad or tan:
        for _ in range(10):
            dx, dy = _dist(cu[0], cu[1], k, p)
            cu = [cd[0] - dx, cd[1] - dy]
    if adj:
        cu = [c * bL0 / (bL0 + B) for c in cu]
    D = fL - bL0
    zc0 = fL * bL0 / D
    zq = pc[2] + zc0
    q = [(pc[0] + cu[0] * fL / D) / zq, (pc[1] + cu[1] * fL / D) / zq]
    ml = [(q[i] - cu[i] / fL) * fL * B / D for i in range(2)]
    if adj:
        pr = [ml[0] + cu[0], ml[1] + cu[1]]
        if n_rad or tan:
            dx, dy = _dist(pr[0], pr[1], k, p)
            pr = [pr[0] + dx, pr[1] + dy]
    else:
        pr = [ml[0] + cd[0], ml[1] + cd[1]]
    return pr[0] / sp + craw[0] - mp.mpf(u), pr[1] / sp + craw[1] - mp.mpf(v)


def residual_and_jacobian(x26, config, u, v, mcx, mcy, spx, scale, cols):
    xs = [mp.mpf(float(t)) for t in x26]
    r = residual(xs, config, u, v, mcx, mcy, spx, scale)
    J = {}
    for c in cols:
        for a in range(2):
            def f(t, c=c, a=a):
                y = list(xs)
                y[c] = t
                return residual(y, config, u, v, mcx, mcy, spx, scale)[a]
            J[(a, c)] = mp.diff(f, xs[c], h=mp.mpf(10) ** -12)
    return r, J
