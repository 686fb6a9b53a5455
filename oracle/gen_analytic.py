"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Known-answer vectors for the residual polynomials (the `return u_t + ...` line of every net_f body):
exact solutions of the PDEs AS CODED in the reference, with their derivatives evaluated symbolically
(sympy) at 40 digits (mpmath) and stored as float64 -> tests/golden/analytic_solutions.npz.  The residual
of an exact solution is 0, whatever network the scripts train: this pins the formulas and coefficient sets
independently of any net.  (The reference holds exact solutions only as data files and as the plotting
helpers soliton.py:8-25 — the tau-function form u = d^2/dx^2 ln f used below — and NLS/testGNSS.py:34-45.)

  burgers     u_t + u u_x - nu u_xx                  Burgers/Burgers_equation(PINN).py:131   travelling front
  kdv5        u_t + 180u^2u_x + 30u_xu_xx + 30uu_xxx + u_5x          5kdv/PI-51.py:140
              = Sawada-Kotera for 2u: u = d^2/dx^2 ln(1 + e^a + e^b + C e^(a+b)), a = k1 x - k1^5 t  (2-soliton)
  kdv7_1sol   Lax 7th-order KdV                      7kdv/PI-7kdv1soliton.py:132
              u = 2 d^2/dx^2 ln(1 + e^a + e^b + ((k1-k2)/(k1+k2))^2 e^(a+b)), a = k1 x - k1^7 t      (2-soliton)
  kdv7_3sol   the same equation for u/14             7kdv/PI-7kdv3soliton.py:132
  gnss        Hirota: i psi_t + psi_xx/2 + |psi|^2 psi - 0.1 i (psi_xxx + 6|psi|^2 psi_x) = 0, psi = u + iv
              NLS/sw-GNSS-adamW.py:178-179           psi = a sech(a(x + 0.1 a^2 t)) e^(i a^2 t/2)
  nls         the same with 0.1 -> 0                 NLS/sw-PINN-rogue.py:179-180
              psi = a sech(a(x - c t)) e^(i(c x + (a^2 - c^2) t/2))                               (boosted soliton)
Layout of `derivs_<eq>`: [n][nout][J] = d^k u/dx^k (k = 0..K) then d/dt of x-orders 0..KT-1, like
pinn_residual_fwd's derivs_out.  Run:  python oracle/gen_analytic.py
"""
import os

import mpmath as mp
import numpy as np
import sympy as sp

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "analytic_solutions.npz")
x, t = sp.symbols("x t", real=True)
mp.mp.dps = 40
NU = float(np.float32(0.01 / np.pi))


def tau2(k1, k2, C, order):
    a = k1 * x - k1 ** order * t
    b = k2 * x - k2 ** order * t
    return 1 + sp.exp(a) + sp.exp(b) + C * sp.exp(a + b)


def solutions():
    k1, k2 = sp.Rational(1), sp.Rational(7, 5)
    sk = sp.diff(sp.log(tau2(k1, k2, (k1 - k2) ** 2 * (k1 ** 2 - k1 * k2 + k2 ** 2) / ((k1 + k2) ** 2 * (k1 ** 2 + k1 * k2 + k2 ** 2)), 5)), x, 2)
    q1, q2 = sp.Rational(4, 5), sp.Rational(6, 5)
    lax = 2 * sp.diff(sp.log(tau2(q1, q2, ((q1 - q2) / (q1 + q2)) ** 2, 7)), x, 2)
    a_, c_ = sp.Rational(3, 10), sp.Rational(1, 5)
    nu = sp.Float(NU, 40)
    am, cb = sp.Rational(6, 5), sp.Rational(1, 2)
    hir = am * sp.sech(am * (x + sp.Rational(1, 10) * am ** 2 * t))
    ph = am ** 2 * t / 2
    nl = am * sp.sech(am * (x - cb * t))
    ph2 = cb * x + (am ** 2 - cb ** 2) * t / 2
    return {
        # name: (outputs, K, KT, x range, t range)
        "burgers": ([c_ - a_ * sp.tanh(a_ * (x - c_ * t) / (2 * nu))], 2, 1, (-0.05, 0.05), (0.0, 0.2)),
        "kdv5": ([sk], 5, 1, (-8.0, 8.0), (-1.0, 1.0)),
        "kdv7_1sol": ([lax], 7, 1, (-8.0, 8.0), (-1.0, 1.0)),
        "kdv7_3sol": ([14 * lax], 7, 1, (-8.0, 8.0), (-1.0, 1.0)),
        "gnss": ([hir * sp.cos(ph), hir * sp.sin(ph)], 3, 1, (-4.0, 4.0), (-2.0, 0.0)),
        "nls": ([nl * sp.cos(ph2), nl * sp.sin(ph2)], 3, 1, (-4.0, 4.0), (-1.0, 1.0)),
    }


def residual(eq, D, T):
    """the net_f bodies of SURVEY 8(a), in mpmath, on D[o][k] = d^k u_o/dx^k, T[o][k] = d/dt of it"""
    u = D[0]
    if eq == "burgers":
        return [T[0][0] + u[0] * u[1] - mp.mpf(NU) * u[2]]
    if eq == "kdv5":
        return [T[0][0] + 180 * u[0] ** 2 * u[1] + 30 * u[1] * u[2] + 30 * u[0] * u[3] + u[5]]
    if eq in ("kdv7_1sol", "kdv7_3sol"):
        c = [140, 70, 280, 70, 70, 42, 14, 1] if eq == "kdv7_1sol" else \
            [mp.mpf(5) / 98, mp.mpf(5) / 14, mp.mpf(10) / 7, mp.mpf(5) / 14, 5, 3, 1, 1]
        return [T[0][0] + c[0] * u[0] ** 3 * u[1] + c[1] * u[1] ** 3 + c[2] * u[0] * u[1] * u[2] + c[3] * u[0] ** 2 * u[3]
                + c[4] * u[2] * u[3] + c[5] * u[1] * u[4] + c[6] * u[0] * u[5] + c[7] * u[7]]
    v = D[1]
    cc = mp.mpf(1) / 10 if eq == "gnss" else 0
    rho = u[0] ** 2 + v[0] ** 2
    return [T[0][0] + v[2] / 2 + rho * v[0] - cc * (u[3] + 6 * rho * u[1]),
            T[1][0] - u[2] / 2 - rho * u[0] - cc * (v[3] + 6 * rho * v[1])]


def main():
    rng = np.random.default_rng(7)
    out = {}
    n = 48
    for eq, (outs, K, KT, xr, tr) in solutions().items():
        J = K + 1 + KT
        fns = []
        for o in outs:
            dx = [o]
            for _ in range(K):
                dx.append(sp.diff(dx[-1], x))
            fns.append([sp.lambdify((x, t), e, "mpmath") for e in dx + [sp.diff(dx[k], t) for k in range(KT)]])
        X = np.stack([xr[0] + (xr[1] - xr[0]) * rng.random(n), tr[0] + (tr[1] - tr[0]) * rng.random(n)], axis=1)
        der = np.zeros((n, len(outs), J))
        worst = mp.mpf(0)
        scale = mp.mpf(0)
        for p in range(n):
            xv, tv = mp.mpf(float(X[p, 0])), mp.mpf(float(X[p, 1]))
            vals = [[f(xv, tv) for f in fo] for fo in fns]
            D = [v[:K + 1] for v in vals]
            T = [v[K + 1:] for v in vals]
            for r in residual(eq, D, T):
                worst = max(worst, abs(r))
            scale = max(scale, max(abs(vv) for v in vals for vv in v))
            der[p] = np.array([[float(vv) for vv in v] for v in vals])
        print("%-10s max |residual| of the exact solution at 40 digits: %s   (largest derivative %s)" % (
            eq, mp.nstr(worst, 3), mp.nstr(scale, 3)))
        assert worst < mp.mpf(10) ** -25 * max(scale, 1), eq
        out["X_" + eq], out["derivs_" + eq] = X, der
    np.savez_compressed(OUT, source=np.array("oracle/gen_analytic.py: exact solutions of the PDEs as coded in the reference"), **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
