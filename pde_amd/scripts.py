"""One drop-in class per reference script family.

Usage inside a reference script: delete its Scale/DNN/PhysicsInformedNN class
definitions and write e.g.
    from pde_amd.scripts import kdv7_3soliton_PI as _m
    PhysicsInformedNN = _m.PhysicsInformedNN
(see INTEGRATION.md).  Literals cited from the reference: IC time `t0`
(SURVEY 8(a) quirks), optimizer and L-BFGS settings.
"""
from types import SimpleNamespace

from .pinn import PhysicsInformedNN, bcPhysicsInformedNN, make_class

_LB_DEFAULT = dict(lr=1.0, max_eval=5000, history_size=50, tolerance_grad=1e-5)


def _family(equation, t0, bc="periodic", lbfgs=None, adamw_lbfgs=None):
    lb = dict(_LB_DEFAULT, **(lbfgs or {}))
    ns = SimpleNamespace()
    ns.PI = SimpleNamespace(
        PhysicsInformedNN=make_class("PhysicsInformedNN", equation=equation, t0=t0, bc=bc, lbfgs=lb))
    ns.sw = SimpleNamespace(
        PhysicsInformedNN=make_class("PhysicsInformedNN", equation=equation, t0=t0, bc=bc, style="sw", lbfgs=lb))
    ns.sw_adamW = SimpleNamespace(
        PhysicsInformedNN=make_class("PhysicsInformedNN", equation=equation, t0=t0, bc=bc, style="sw",
                                     optimizer="adamw", weight_decay=1e-2, lbfgs=dict(lb, **(adamw_lbfgs or {}))))
    ns.bc = SimpleNamespace(
        PhysicsInformedNN=make_class("PhysicsInformedNN", equation=equation, t0=t0, bc=bc, lbfgs=lb),
        bcPhysicsInformedNN=make_class("bcPhysicsInformedNN", base=bcPhysicsInformedNN, equation=equation,
                                       t0=t0, bc=bc, style="bc", lbfgs=lb))
    return ns


# 7kdv/PI-7kdv3soliton.py:95,138-148 ; sw-73.py ; sw73-adamW.py:170 ; bc-7kdv3soliton.py
kdv7_3soliton = _family("kdv7_3sol", -0.5)
kdv7_2soliton = _family("kdv7_3sol", -0.5)          # same residual coefficients as the 3-soliton scripts
kdv7_1soliton = _family("kdv7_1sol", -2.0)          # 7kdv/PI-7kdv1soliton.py:95,132
# 5kdv/PI-51.py:101 (t0=-2), PI-52/53 use -0.5
kdv5_1soliton = _family("kdv5", -2.0)
kdv5_2soliton = _family("kdv5", -0.5)
kdv5_3soliton = _family("kdv5", -0.5)
# Burgers/Burgers_equation(PINN).py:100,109: t0 = 0, Dirichlet boundary loss
burgers = _family("burgers", 0.0, bc="dirichlet")
# CH/PI-CH-1soliton.py:101,147-150: lr 0.5, history 150 ; bc script carries the +2u_x term
camassa_holm = _family("ch", -1.0, lbfgs=dict(lr=0.5, history_size=150))
camassa_holm_bc = _family("ch_bc", -1.0, lbfgs=dict(lr=0.5, history_size=150))
# NLS/sw-GNSS-adamW.py:102,186-190: max_eval 50000, tolerance_grad 1e-7
gnss = _family("gnss", -2.0, lbfgs=dict(max_eval=50000, tolerance_grad=1e-7))
nls = _family("nls", -1.0, lbfgs=dict(max_eval=50000, tolerance_grad=1e-9))   # NLS/sw-PINN-rogue.py:103,194

# flat aliases used in INTEGRATION.md
kdv7_3soliton_PI = kdv7_3soliton.PI
kdv7_3soliton_sw = kdv7_3soliton.sw
kdv7_3soliton_sw_adamW = kdv7_3soliton.sw_adamW
kdv7_3soliton_bc = kdv7_3soliton.bc
