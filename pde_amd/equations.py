"""Equation registry: which `net_f` body of the reference a name stands for.

Every entry cites the reference lines that define the residual; `coef` is the
coefficient vector handed to the kernels (include/pinnjet.h, PINN_EQ_*).
K = highest x-derivative carried, KT = number of x-orders that also carry a
t-derivative (1: u_t only; 3: u_t, u_xt, u_xxt as Camassa-Holm needs).
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

EQ_NONE, EQ_BURGERS, EQ_KDV5, EQ_KDV7, EQ_CH, EQ_GNSS, EQ_LPD, EQ_NLSREG = range(8)


@dataclass(frozen=True)
class Equation:
    name: str
    eq_id: int
    K: int
    KT: int
    nout: int
    nf: int
    coef: List[float] = field(default_factory=list)
    ref: str = ""

    @property
    def J(self):
        return self.K + 1 + self.KT


_REG = {}


def _add(e):
    _REG[e.name] = e


_add(Equation("none", EQ_NONE, 0, 0, 0, 0, [], "self.dnn(X): 7kdv/PI-7kdv3soliton.py:99"))
# nu is stored as a float32 tensor by the reference (Burgers/Burgers_equation(PINN).py:96,206)
_add(Equation("burgers", EQ_BURGERS, 2, 1, 1, 1, [float(np.float32(0.01 / np.pi))],
              "Burgers/Burgers_equation(PINN).py:126-131"))
_add(Equation("kdv5", EQ_KDV5, 5, 1, 1, 1, [180.0, 30.0, 30.0, 1.0], "5kdv/PI-51.py:132-140"))
_add(Equation("kdv7_3sol", EQ_KDV7, 7, 1, 1, 1, [5 / 98, 5 / 14, 10 / 7, 5 / 14, 5.0, 3.0, 1.0, 1.0],
              "7kdv/PI-7kdv3soliton.py:122-132 (also 2-soliton scripts)"))
_add(Equation("kdv7_1sol", EQ_KDV7, 7, 1, 1, 1, [140.0, 70.0, 280.0, 70.0, 70.0, 42.0, 14.0, 1.0],
              "7kdv/PI-7kdv1soliton.py:122-132"))
_add(Equation("ch", EQ_CH, 3, 3, 1, 1, [0.0], "CH/PI-CH-1soliton.py:131-138"))
_add(Equation("ch_bc", EQ_CH, 3, 3, 1, 1, [2.0], "CH/bc-CH-1soliton.py:140-148,270-278"))
_add(Equation("gnss", EQ_GNSS, 3, 1, 2, 2, [0.5, 1.0, 0.1], "NLS/sw-GNSS-adamW.py:159-180"))
_add(Equation("nls", EQ_GNSS, 3, 1, 2, 2, [0.5, 1.0, 0.0], "NLS/sw-PINN-rogue.py:179-180"))
# plain NLS + the 1e-9 derivative regulariser, which rides as a third residual row (include/pinnjet.h)
_add(Equation("nls_reg", EQ_NLSREG, 2, 1, 2, 3, [0.5, 1.0, 1e-9], "NLS/sw-test.py:159-189"))
# 4th-order NLS of the scratch scripts (an optional id of SURVEY 8(a) a7), coefficients as coded there
_add(Equation("lpd", EQ_LPD, 4, 1, 2, 2, [0.5, 1.0, 0.125, 0.0625, 6.0, 10.0], "NLS/Ibc-test.py:163-164, NLS/testGNSS.py:139-140"))


def get(name, coef=None):
    """Look an equation up by name; `coef` overrides the default coefficients."""
    e = _REG[name]
    if coef is not None:
        e = Equation(e.name, e.eq_id, e.K, e.KT, e.nout, e.nf, [float(c) for c in coef], e.ref)
    return e


def names():
    return sorted(_REG)
