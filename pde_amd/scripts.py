"""One drop-in class per class definition of every hot-path script of the reference.

Usage inside a reference script: delete its Scale/DNN/PhysicsInformedNN class
definitions and write e.g.
    from pde_amd import scripts
    PhysicsInformedNN = scripts.for_script("7kdv/sw-73.py").PhysicsInformedNN
(see INTEGRATION.md).  What differs from script to script is a literal table: the
residual body, the IC time `t0`, the boundary-loss form, Adam vs
AdamW, the L-BFGS settings.  Every row below cites the reference lines it was
read from, and tests/test_host_logic.py::test_script_literals_match_reference compares the
whole table with tests/golden/script_literals.json, which oracle/gen_script_literals.py
extracts from the reference's source text.
"""
from types import SimpleNamespace

from .pinn import PhysicsInformedNN, bcPhysicsInformedNN, make_class


def _lb(max_eval, tolerance_grad, lr=1.0, history_size=50):
    return dict(lr=lr, max_eval=max_eval, history_size=history_size, tolerance_grad=tolerance_grad)


def _pi(eq, t0, lbfgs, bc="periodic"):
    """a PI-script class, or the segment-0 class of a bc script: set_training_data(x0, u0, tb, Xf)"""
    return dict(equation=eq, t0=t0, bc=bc, style="pi", lbfgs=lbfgs)


def _sw(eq, t0, lbfgs, adamw=False):
    """sliding-window class: set_training_data(..., X_prev=None, u_prev=None), loss_prev, 'Loss0' log format"""
    return dict(equation=eq, t0=t0, style="sw", lbfgs=lbfgs, **(dict(optimizer="adamw", weight_decay=1e-2) if adamw else {}))


def _bc(eq, lbfgs):
    """bcPhysicsInformedNN: t0 is an argument of set_training_data, extra loss1 over the earlier segments"""
    return dict(equation=eq, t0=0.0, style="bc", lbfgs=lbfgs)


# reference script -> {class name: variant}.  Line numbers: t0 literal / optimizer_Adam / LBFGS(...) block.
_TABLE = {
    # ---- 7th-order KdV (7kdv/) ----
    "7kdv/PI-7kdv1soliton.py": {"PhysicsInformedNN": _pi("kdv7_1sol", -2.0, _lb(5000, 1e-5))},      # :95,138,139-148
    "7kdv/PI-7kdv2soliton.py": {"PhysicsInformedNN": _pi("kdv7_3sol", -0.5, _lb(5000, 1e-5))},      # :95,137,138-147
    "7kdv/PI-7kdv3soliton.py": {"PhysicsInformedNN": _pi("kdv7_3sol", -0.5, _lb(5000, 1e-5))},      # :95,138,139-148
    "7kdv/sw-71.py": {"PhysicsInformedNN": _sw("kdv7_1sol", -2.0, _lb(50000, 1e-7))},               # :102,170,171-180
    "7kdv/sw-72.py": {"PhysicsInformedNN": _sw("kdv7_3sol", -0.5, _lb(50000, 1e-7))},               # :102,170,171-180
    "7kdv/sw-73.py": {"PhysicsInformedNN": _sw("kdv7_3sol", -0.5, _lb(50000, 1e-7))},               # :102,170,171-180
    "7kdv/sw71-adamW.py": {"PhysicsInformedNN": _sw("kdv7_1sol", -2.0, _lb(50000, 1e-7), adamw=True)},   # :102,170
    "7kdv/sw72-adamW.py": {"PhysicsInformedNN": _sw("kdv7_3sol", -0.5, _lb(50000, 1e-7), adamw=True)},   # :102,170
    "7kdv/sw73-adamW.py": {"PhysicsInformedNN": _sw("kdv7_3sol", -0.5, _lb(50000, 1e-7), adamw=True)},   # :102,170
    "7kdv/bc-7kdv1soliton.py": {"PhysicsInformedNN": _pi("kdv7_1sol", -2.0, _lb(50000, 1e-5)),      # :94,148,149-158
                                "bcPhysicsInformedNN": _bc("kdv7_1sol", _lb(50000, 1e-5))},         # :221,282,283-292
    "7kdv/bc-7kdv2soliton.py": {"PhysicsInformedNN": _pi("kdv7_3sol", -0.5, _lb(50000, 1e-5)),
                                "bcPhysicsInformedNN": _bc("kdv7_3sol", _lb(50000, 1e-5))},
    "7kdv/bc-7kdv3soliton.py": {"PhysicsInformedNN": _pi("kdv7_3sol", -0.5, _lb(50000, 1e-5)),
                                "bcPhysicsInformedNN": _bc("kdv7_3sol", _lb(50000, 1e-5))},
    # ---- 5th-order KdV (5kdv/) ----
    "5kdv/PI-51.py": {"PhysicsInformedNN": _pi("kdv5", -2.0, _lb(5000, 1e-5))},                     # :101,146,147-156
    "5kdv/PI-52.py": {"PhysicsInformedNN": _pi("kdv5", -0.5, _lb(500, 1e-5))},                      # :101,145,150 (max_eval=500)
    "5kdv/PI-53.py": {"PhysicsInformedNN": _pi("kdv5", -0.5, _lb(500, 1e-5))},                      # :101,145,150
    "5kdv/sw-51.py": {"PhysicsInformedNN": _sw("kdv5", -2.0, _lb(50000, 1e-7))},                    # :102,167,168-177
    "5kdv/sw-52.py": {"PhysicsInformedNN": _sw("kdv5", -0.5, _lb(50000, 1e-7))},
    "5kdv/sw-53.py": {"PhysicsInformedNN": _sw("kdv5", -0.5, _lb(50000, 1e-7))},
    "5kdv/sw51-adamw.py": {"PhysicsInformedNN": _sw("kdv5", -2.0, _lb(50000, 1e-7), adamw=True)},   # :102,167
    "5kdv/sw52-adamw.py": {"PhysicsInformedNN": _sw("kdv5", -0.5, _lb(50000, 1e-7), adamw=True)},
    "5kdv/sw53-adamW.py": {"PhysicsInformedNN": _sw("kdv5", -0.5, _lb(50000, 1e-7), adamw=True)},
    "5kdv/bc-51.py": {"PhysicsInformedNN": _pi("kdv5", -2.0, _lb(50000, 1e-5)),                     # :103,155,156-165
                      "bcPhysicsInformedNN": _bc("kdv5", _lb(50000, 1e-5))},                        # :228,293,294-303
    "5kdv/bc-52.py": {"PhysicsInformedNN": _pi("kdv5", -0.5, _lb(50000, 1e-5)),
                      "bcPhysicsInformedNN": _bc("kdv5", _lb(50000, 1e-5))},
    "5kdv/bc-53.py": {"PhysicsInformedNN": _pi("kdv5", -0.5, _lb(50000, 1e-5)),
                      "bcPhysicsInformedNN": _bc("kdv5", _lb(50000, 1e-5))},
    # ---- Burgers: t0 = 0 (:100), Dirichlet boundary loss (:109), max_eval 50000 (:142) ----
    "Burgers/Burgers_equation(PINN).py": {"PhysicsInformedNN": _pi("burgers", 0.0, _lb(50000, 1e-5), bc="dirichlet")},
    # ---- Camassa-Holm: PI script lr 0.5 / max_eval 75000 / history 150 / 1e-7 (:147-151); the bc script
    #      carries the +2u_x term in BOTH classes (:148,278) and plain L-BFGS settings (:157-161,287-291) ----
    "CH/PI-CH-1soliton.py": {"PhysicsInformedNN": _pi("ch", -1.0, _lb(75000, 1e-7, lr=0.5, history_size=150))},
    "CH/bc-CH-1soliton.py": {"PhysicsInformedNN": _pi("ch_bc", -1.0, _lb(50000, 1e-5)),             # :103
                             "bcPhysicsInformedNN": _bc("ch_bc", _lb(50000, 1e-5))},
    # ---- NLS / GNSS (two outputs) ----
    "NLS/PI-GNSS-torch.py": {"PhysicsInformedNN": _pi("gnss", -2.0, _lb(50000, 1e-5))},             # :94,160,161-170
    "NLS/sw-GNSS.py": {"PhysicsInformedNN": _sw("gnss", -2.0, _lb(50000, 1e-7))},                   # :102,185,186-195
    "NLS/sw-GNSS-adamW.py": {"PhysicsInformedNN": _sw("gnss", -2.0, _lb(50000, 1e-7), adamw=True)},  # :102,185
    "NLS/bc-GNSS-torch.py": {"PhysicsInformedNN": _pi("gnss", -1.0, _lb(50000, 1e-5)),              # :102 (t0 = -1 here)
                             "bcPhysicsInformedNN": _bc("gnss", _lb(50000, 1e-5))},                 # :245,319,320-329
    "NLS/sw-PINN-rogue.py": {"PhysicsInformedNN": _sw("nls", -1.0, _lb(50000, 1e-9))},              # :103,187,194
    # scratch copy, kept for its derivative regulariser (lambda_reg = 1e-9, :172-180, added to the loss :147): t0 = 0 (:102)
    "NLS/sw-test.py": {"PhysicsInformedNN": _sw("nls_reg", 0.0, _lb(50000, 1e-7))},                 # :102,194,195-204
    # scratch copy, kept for its 4th-order (LPD-type) residual id (:163-164; NLS/testGNSS.py:139-140 has the same body)
    "NLS/Ibc-test.py": {"PhysicsInformedNN": _pi("lpd", -5.0, _lb(50000, 1e-5)),                    # :102,171,172-181
                        "bcPhysicsInformedNN": _bc("lpd", _lb(50000, 1e-5))},                       # :245,319,320-329
}


def _build(script):
    ns = SimpleNamespace(script=script)
    for cname, var in _TABLE[script].items():
        base = bcPhysicsInformedNN if cname == "bcPhysicsInformedNN" else PhysicsInformedNN
        setattr(ns, cname, make_class(cname, base=base, **var))
    return ns


BY_SCRIPT = {script: _build(script) for script in _TABLE}


def for_script(path):
    """The classes of one reference script, by its path relative to the reference root
    (e.g. "7kdv/sw-73.py"): a namespace with `PhysicsInformedNN` (and `bcPhysicsInformedNN`)."""
    return BY_SCRIPT[path]


def variant_of(path, cname="PhysicsInformedNN"):
    return getattr(BY_SCRIPT[path], cname).variant


def _family(PI=None, sw=None, sw_adamW=None, bc=None):
    ns = SimpleNamespace()
    for k, v in (("PI", PI), ("sw", sw), ("sw_adamW", sw_adamW), ("bc", bc)):
        if v is not None:
            setattr(ns, k, BY_SCRIPT[v])
    return ns


# families: the PI / sliding-window / sliding-window-AdamW / bc script of one problem
kdv7_1soliton = _family("7kdv/PI-7kdv1soliton.py", "7kdv/sw-71.py", "7kdv/sw71-adamW.py", "7kdv/bc-7kdv1soliton.py")
kdv7_2soliton = _family("7kdv/PI-7kdv2soliton.py", "7kdv/sw-72.py", "7kdv/sw72-adamW.py", "7kdv/bc-7kdv2soliton.py")
kdv7_3soliton = _family("7kdv/PI-7kdv3soliton.py", "7kdv/sw-73.py", "7kdv/sw73-adamW.py", "7kdv/bc-7kdv3soliton.py")
kdv5_1soliton = _family("5kdv/PI-51.py", "5kdv/sw-51.py", "5kdv/sw51-adamw.py", "5kdv/bc-51.py")
kdv5_2soliton = _family("5kdv/PI-52.py", "5kdv/sw-52.py", "5kdv/sw52-adamw.py", "5kdv/bc-52.py")
kdv5_3soliton = _family("5kdv/PI-53.py", "5kdv/sw-53.py", "5kdv/sw53-adamW.py", "5kdv/bc-53.py")
burgers = _family("Burgers/Burgers_equation(PINN).py")
camassa_holm = _family("CH/PI-CH-1soliton.py")
camassa_holm_bc = _family(bc="CH/bc-CH-1soliton.py")
gnss = _family("NLS/PI-GNSS-torch.py", "NLS/sw-GNSS.py", "NLS/sw-GNSS-adamW.py", "NLS/bc-GNSS-torch.py")
nls = _family(sw="NLS/sw-PINN-rogue.py")
lpd = _family(bc="NLS/Ibc-test.py")
nls_reg = _family(sw="NLS/sw-test.py")

# flat aliases used in INTEGRATION.md
kdv7_3soliton_PI = kdv7_3soliton.PI
kdv7_3soliton_sw = kdv7_3soliton.sw
kdv7_3soliton_sw_adamW = kdv7_3soliton.sw_adamW
kdv7_3soliton_bc = kdv7_3soliton.bc
