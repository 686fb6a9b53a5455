"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes binding of oracle/libpinn_oracle.so (the plain-C CPU restatement of the
reference's net_u / net_f / loss / weight-gradient path, see pinn_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  pde_amd/ never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpinn_oracle.so")

EQ_NONE, EQ_BURGERS, EQ_KDV5, EQ_KDV7, EQ_CH, EQ_GNSS, EQ_LPD, EQ_NLSREG = range(8)

# residual coefficient sets, cited from the reference
COEF = {
    "burgers": (EQ_BURGERS, [float(np.float32(0.01 / np.pi))]),  # Burgers/Burgers_equation(PINN).py:96,206 (nu is a float32 tensor)
    "kdv5": (EQ_KDV5, [180.0, 30.0, 30.0, 1.0]),             # 5kdv/PI-51.py:140
    "kdv7_3sol": (EQ_KDV7, [5 / 98, 5 / 14, 10 / 7, 5 / 14, 5.0, 3.0, 1.0, 1.0]),  # 7kdv/PI-7kdv3soliton.py:132
    "kdv7_1sol": (EQ_KDV7, [140.0, 70.0, 280.0, 70.0, 70.0, 42.0, 14.0, 1.0]),      # 7kdv/PI-7kdv1soliton.py:132
    "ch": (EQ_CH, [0.0]),                                    # CH/PI-CH-1soliton.py:138
    "ch_bc": (EQ_CH, [2.0]),                                 # CH/bc-CH-1soliton.py:278
    "gnss": (EQ_GNSS, [0.5, 1.0, 0.1]),                      # NLS/sw-GNSS-adamW.py:178-179
    "nls": (EQ_GNSS, [0.5, 1.0, 0.0]),                       # NLS/sw-PINN-rogue.py:179-180
    "lpd": (EQ_LPD, [0.5, 1.0, 0.125, 0.0625, 6.0, 10.0]),   # NLS/Ibc-test.py:163-164, NLS/testGNSS.py:139-140
    "nls_reg": (EQ_NLSREG, [0.5, 1.0, 1e-9]),                # NLS/sw-test.py:159-189 (three residual rows)
    "none": (EQ_NONE, []),
}
ORDERS = {EQ_NONE: (0, 0), EQ_BURGERS: (2, 1), EQ_KDV5: (5, 1), EQ_KDV7: (7, 1), EQ_CH: (3, 3), EQ_GNSS: (3, 1), EQ_LPD: (4, 1), EQ_NLSREG: (2, 1)}


class _Desc(ctypes.Structure):
    _fields_ = [
        ("eq", ctypes.c_int),
        ("n_lin", ctypes.c_int),
        ("widths", ctypes.c_int * 17),
        ("lb", ctypes.c_double * 2),
        ("ub", ctypes.c_double * 2),
        ("coef", ctypes.c_double * 8),
    ]


def build():
    """Compile the oracle with the committed Makefile (gcc only)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_f32_precise_top(on):
    """float32 entry points only: model the kernels' "precise top" arithmetic (last hidden Linear layer, output
    layer and residual polynomial accumulated / evaluated in double; pinn_oracle.c).  Process-wide switch."""
    lib().orc_set_f32_precise_top(ctypes.c_int(1 if on else 0))


class Oracle:
    """One network + equation.  layers: the script's `layers` list; lb/ub are
    rounded to float32 first because the reference's Scale stores them as
    float32 (7kdv/PI-7kdv3soliton.py:37-38)."""

    def __init__(self, eq, layers, lb, ub, coef=None):
        if isinstance(eq, str):
            eq_id, default_coef = COEF[eq]
            coef = default_coef if coef is None else coef
        else:
            eq_id = eq
            coef = [] if coef is None else coef
        self.eq = eq_id
        self.layers = list(layers)
        self.K, self.KT = ORDERS[eq_id]
        self.J = self.K + 1 + self.KT
        self.nout = self.layers[-1]
        self.nf = {EQ_NONE: self.nout, EQ_GNSS: 2, EQ_LPD: 2, EQ_NLSREG: 3}.get(eq_id, 1)
        d = _Desc()
        d.eq = eq_id
        d.n_lin = len(layers) - 1
        for i, w in enumerate(layers):
            d.widths[i] = int(w)
        lb32 = np.asarray(lb, dtype=np.float32).astype(np.float64)
        ub32 = np.asarray(ub, dtype=np.float32).astype(np.float64)
        for i in range(2):
            d.lb[i] = lb32[i]
            d.ub[i] = ub32[i]
        for i, c in enumerate(coef):
            d.coef[i] = float(c)
        self.desc = d
        self.P = sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))

    def _prep(self, dtype, theta, x, t):
        theta = np.ascontiguousarray(np.asarray(theta, dtype=dtype).reshape(-1))
        assert theta.size == self.P, (theta.size, self.P)
        x = np.ascontiguousarray(np.asarray(x, dtype=dtype).reshape(-1))
        t = np.ascontiguousarray(np.asarray(t, dtype=dtype).reshape(-1))
        assert x.size == t.size
        return theta, x, t

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def forward(self, theta, x, t, dtype=np.float64):
        """-> (derivs [n, nout, J] = D_0..D_K then T_0..T_{KT-1}; f [n, nf])"""
        theta, x, t = self._prep(dtype, theta, x, t)
        n = x.size
        der = np.zeros((n, self.nout, self.J), dtype=dtype)
        f = np.zeros((n, self.nf), dtype=dtype)
        fn = lib().orc_forward_f64 if dtype == np.float64 else lib().orc_forward_f32
        rc = fn(ctypes.byref(self.desc), self._p(theta), self._p(x), self._p(t),
                ctypes.c_long(n), self._p(der), self._p(f))
        assert rc == 0, rc
        return der, f

    def residual_from_derivs(self, derivs, dtype=np.float64):
        """the residual polynomial alone on true derivatives [n, nout, J] -> f [n, nf]"""
        derivs = np.ascontiguousarray(np.asarray(derivs, dtype=dtype).reshape(-1, self.nout, self.J))
        n = derivs.shape[0]
        f = np.zeros((n, self.nf), dtype=dtype)
        fn = lib().orc_residual_derivs_f64 if dtype == np.float64 else lib().orc_residual_derivs_f32
        rc = fn(ctypes.byref(self.desc), self._p(derivs), ctypes.c_long(n), self._p(f))
        assert rc == 0, rc
        return f

    def loss_grad(self, theta, x, t, scale=1.0, gbar=None, dtype=np.float64):
        """-> (loss_sums [nf] = sum_p f_c^2 (zeros in vjp mode), grad [P]).
        loss mode: grad = d/dtheta scale * sum_c sum_p f_c^2.
        vjp  mode: grad = sum_p sum_c gbar[p,c] df_c/dtheta."""
        theta, x, t = self._prep(dtype, theta, x, t)
        n = x.size
        grad = np.zeros(self.P, dtype=dtype)
        ls = np.zeros(3, dtype=np.float64)
        if gbar is not None:
            gbar = np.ascontiguousarray(np.asarray(gbar, dtype=dtype).reshape(n, self.nf))
            gp = self._p(gbar)
        else:
            gp = ctypes.c_void_p(0)
        fn = lib().orc_loss_grad_f64 if dtype == np.float64 else lib().orc_loss_grad_f32
        rc = fn(ctypes.byref(self.desc), self._p(theta), self._p(x), self._p(t),
                ctypes.c_long(n), gp, ctypes.c_double(scale), self._p(ls), self._p(grad))
        assert rc == 0, rc
        return ls[: self.nf].copy(), grad
