#!/usr/bin/env python3
"""(CPU, mpmath) The minimax polynomials of exp_nonpos (csrc/sk_internal.h): 1 + r + r^2/2 + sum_{k=3..DEG} c_k r^k on |r| <= ln2/2, relative
error, by Remez exchange in 80-digit arithmetic; prints the coefficients as doubles and the error of the DOUBLE-ROUNDED polynomial.
    degree 11: 2.4e-17   degree 12: 7.5e-20   degree 13: 3.8e-21   (Taylor degree 13: 4e-18, degree 10: 2e-13)   degree 9: 1.1e-13   degree 10: 4.1e-16"""
import mpmath as mp
mp.mp.dps = 80
a = mp.log(2) / 2 * mp.mpf('1.0000001')
f = lambda r: mp.e ** r


def fit(deg, fixed=3):
    n = deg - fixed + 1
    base = lambda r: sum(r ** k / mp.factorial(k) for k in range(fixed))
    pts = [-a * mp.cos(mp.pi * i / n) for i in range(n + 1)]
    c, mx = None, None
    for _ in range(40):
        A, b = mp.matrix(n + 1, n + 1), mp.matrix(n + 1, 1)
        for i, r in enumerate(pts):
            for j in range(n): A[i, j] = r ** (fixed + j)
            A[i, n] = (-1) ** i * f(r)
            b[i] = f(r) - base(r)
        sol = mp.lu_solve(A, b)
        c = [sol[j] for j in range(n)]
        err = lambda r: (base(r) + sum(c[j] * r ** (fixed + j) for j in range(n)) - f(r)) / f(r)
        N = 6000
        xs = [-a + 2 * a * i / N for i in range(N + 1)]
        es = [err(x) for x in xs]
        cand = [0] + [i for i in range(1, N) if (es[i] - es[i - 1]) * (es[i + 1] - es[i]) <= 0] + [N]
        merged = []
        for i in cand:
            x, v = xs[i], es[i]
            if merged and (merged[-1][1] > 0) == (v > 0):
                if abs(v) > abs(merged[-1][1]): merged[-1] = (x, v)
            else: merged.append((x, v))
        while len(merged) > n + 1:
            merged.pop(0) if abs(merged[0][1]) < abs(merged[-1][1]) else merged.pop()
        mx = max(abs(e) for e in es)
        if len(merged) < n + 1: break
        pts = [x for x, v in merged]
        if mx / min(abs(v) for x, v in merged) < mp.mpf('1.001'): break
    return c, mx


for deg in (9, 11):
    c, mx = fit(deg)
    cd = [mp.mpf(float(x)) for x in reversed(c)]      # r^deg .. r^3, rounded to double
    def p(r):
        v = cd[0]
        for q in cd[1:]: v = v * r + q
        return ((v * r + mp.mpf('0.5')) * r + 1) * r + 1
    N = 4000
    e = max(abs((p(r) - f(r)) / f(r)) for r in [-a + 2 * a * i / N for i in range(N + 1)])
    print("degree %d: minimax %s, with double coefficients %s" % (deg, mp.nstr(mx, 4), mp.nstr(e, 4)))
    print("   {" + ", ".join(repr(float(x)) for x in reversed(c)) + "}")
