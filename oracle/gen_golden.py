"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Generates tests/golden/*.npz by importing the reference's own scripts (CPU,
this container only: /root/reference does not exist on the GPU box) and
evaluating their `net_f` / `backward` / `predict` at shipped checkpoints and at
seeded xavier weights, in fp32 (reference behaviour) and fp64 (truth).

Only data is written: inputs, weights, expected outputs.  No reference source
is copied.  Run:  python oracle/gen_golden.py   (needs /root/reference)
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into the reference tree
import matplotlib

matplotlib.use("Agg")
import numpy as np
import scipy.io
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
N_PTS = 192

_stub = types.ModuleType("pyDOE")
_stub.lhs = lambda n, samples=None: np.random.rand(samples, n)
sys.modules["pyDOE"] = _stub


def load_script(rel):
    name = "ref_" + "".join(ch if ch.isalnum() else "_" for ch in rel)
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    cwd = os.getcwd()
    spec.loader.exec_module(mod)
    os.chdir(cwd)
    return mod


def _d(y, x):
    return torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), retain_graph=True, create_graph=True)[0]


def flat_grad(model):
    return torch.cat([p.grad.reshape(-1) for p in model.dnn.parameters()]).detach().numpy()


def flat_theta(model):
    return torch.cat([p.detach().reshape(-1) for p in model.dnn.parameters()]).numpy()


def set_dtype(model, dtype):
    model.dnn.to(dtype)
    sc = model.dnn[0]
    # Scale keeps float32 lb/ub as plain attributes; promote those exact values
    sc.lb = sc.lb.float().to(dtype)
    sc.ub = sc.ub.float().to(dtype)


def eval_case(mod, cls_name, layers, lb, ub, ckpt, seed, eq, K, KT, pts, extra, ctor_extra=()):
    """Returns a dict of arrays for one (script, weights) pair."""
    mod.lb, mod.ub = lb, ub  # PI scripts read module globals in set_training_data
    cls = getattr(mod, cls_name)
    if seed is not None:
        torch.manual_seed(seed)
    model = cls(layers, lb, ub, *ctor_extra)
    if ckpt is not None:
        sd = torch.load(os.path.join(REF, ckpt), map_location="cpu", weights_only=True)
        model.dnn.load_state_dict(sd)
    model.iter = 1
    model.logfilename = None
    theta32 = flat_theta(model).astype(np.float32)
    out = {"theta": theta32, "x": pts[:, 0].astype(np.float32), "t": pts[:, 1].astype(np.float32)}
    nout = layers[-1]
    for tag, dtype in (("64", torch.float64), ("32", torch.float32)):
        set_dtype(model, dtype)
        x = torch.tensor(out["x"][:, None].astype(np.float64)).to(dtype).requires_grad_(True)
        t = torch.tensor(out["t"][:, None].astype(np.float64)).to(dtype).requires_grad_(True)
        if eq == "burgers":
            model.nu = torch.tensor(0.01 / np.pi).float().to(dtype)
        # (1) every derivative on the path, via the reference's own dnn module
        uv = model.dnn(torch.cat([x, t], dim=1))
        der = np.zeros((len(out["x"]), nout, K + 1 + KT))
        for o in range(nout):
            dx = [uv[:, o:o + 1]]
            for _ in range(K):
                dx.append(_d(dx[-1], x))
            for k in range(K + 1):
                der[:, o, k] = dx[k].detach().numpy()[:, 0]
            for k in range(KT):
                der[:, o, K + 1 + k] = _d(dx[k], t).detach().numpy()[:, 0]
        out["derivs" + tag] = der.astype(np.float64 if tag == "64" else np.float32)
        # (2) the reference's net_f, loss_f and its gradient
        for p in model.dnn.parameters():
            p.grad = None
        f = model.net_f(x, t)
        fs = list(f) if isinstance(f, tuple) else [f]
        reg = None
        if eq == "nls_reg":      # NLS/sw-test.py:189 returns (fu, fv, reg_term); backward() adds reg_term to the loss (:147)
            reg = fs.pop()
        loss_f = sum(torch.mean(torch.square(fi)) for fi in fs)
        if reg is not None:
            loss_f = loss_f + reg
        loss_f.backward()
        out["f" + tag] = np.concatenate([fi.detach().numpy() for fi in fs], axis=1)
        if reg is not None:
            # the build carries the regulariser as a third residual row f2 with mean(f2^2) = reg_term; its golden
            # column is formed here from the reference's derivatives (columns 1, 2 = u_x, u_xx; 3 = u_t) and checked
            # against the reference's own reg_term
            dd = der[:, :, 1:4]
            f2 = np.sqrt(1e-9 * np.sum(dd * dd, axis=(1, 2)))
            assert abs(np.mean(f2 * f2) - reg.item()) <= 1e-5 * reg.item(), (np.mean(f2 * f2), reg.item())
            out["f" + tag] = np.concatenate([out["f" + tag], f2[:, None].astype(out["f" + tag].dtype)], axis=1)
            out["reg_term" + tag] = np.array(reg.item())
        out["loss_f" + tag] = np.array(loss_f.item())
        out["grad_f" + tag] = flat_grad(model)
        # (3) the reference's full backward() with its data terms
        if extra is not None:
            for p in model.dnn.parameters():
                p.grad = None
            for k, v in extra.items():
                if v is None:
                    setattr(model, k, None)
                elif k == "earliest_solutions":       # NLS/IBCtest.py:225: a list of (X_ref, h_ref) tensor pairs
                    setattr(model, k, [tuple(torch.tensor(np.asarray(a, dtype=np.float32).astype(np.float64)).to(dtype)
                                             for a in pair) for pair in v])
                else:
                    setattr(model, k, torch.tensor(np.asarray(v, dtype=np.float32).astype(np.float64)).to(dtype))
            model.xf, model.tf = x, t
            if eq == "burgers":
                model.nu = torch.tensor(0.01 / np.pi).float().to(dtype)
            loss = model.backward()
            out["loss_full" + tag] = np.array(loss.item())
            out["grad_full" + tag] = flat_grad(model)
    if extra is not None:
        for k, v in extra.items():
            if k == "earliest_solutions":
                for i, (X, h) in enumerate(v):
                    out["data_es_X%d" % i] = np.asarray(X, dtype=np.float32)
                    out["data_es_h%d" % i] = np.asarray(h, dtype=np.float32)
            elif v is not None:
                out["data_" + k] = np.asarray(v, dtype=np.float32)
    return out


def sample_points(lb, ub, n, rng):
    U = rng.random((n, 2))
    return (np.asarray(lb) + (np.asarray(ub) - np.asarray(lb)) * U).astype(np.float32)


def data_terms(rng, lb, ub, t0, nout, n0=24, nb=16, u_fn=None):
    x0 = (lb[0] + (ub[0] - lb[0]) * rng.random((n0, 1))).astype(np.float32)
    tb = (lb[1] + (ub[1] - lb[1]) * rng.random((nb, 1))).astype(np.float32)
    ex = {
        "x0": x0, "t0": np.full_like(x0, t0), "tb": tb,
        "x_lb": np.full_like(tb, lb[0]), "x_ub": np.full_like(tb, ub[0]),
        "u0": (2.0 / np.cosh(x0) ** 2).astype(np.float32),
    }
    if nout == 2:
        ex["v0"] = (0.5 * np.tanh(x0)).astype(np.float32)
    return ex


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1234)
    L50 = [2, 50, 50, 50, 50, 1]
    cases = []

    # ---- 7kdv 3-soliton PI (flagship) -------------------------------------
    m73 = load_script("7kdv/PI-7kdv3soliton.py")
    lb, ub = np.array([-15.0, -0.5]), np.array([15.0, 0.5])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -0.5, 1)
    cases.append(("kdv7_3sol_ckpt", "kdv7_3sol", L50, lb, ub,
                  eval_case(m73, "PhysicsInformedNN", L50, lb, ub, "7kdv/model/PI 73.pth", None,
                            "kdv7", 7, 1, pts, ex),
                  "7kdv/PI-7kdv3soliton.py + 7kdv/model/PI 73.pth"))
    cases.append(("kdv7_3sol_xavier", "kdv7_3sol", L50, lb, ub,
                  eval_case(m73, "PhysicsInformedNN", L50, lb, ub, None, 1234, "kdv7", 7, 1, pts, ex),
                  "7kdv/PI-7kdv3soliton.py + xavier(manual_seed 1234)"))

    # ---- 7kdv 1-soliton coefficient set ------------------------------------
    m71 = load_script("7kdv/PI-7kdv1soliton.py")
    lb, ub = np.array([-10.0, -2.0]), np.array([10.0, 2.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -2.0, 1)
    cases.append(("kdv7_1sol_ckpt", "kdv7_1sol", L50, lb, ub,
                  eval_case(m71, "PhysicsInformedNN", L50, lb, ub, "7kdv/model/PI 1t-2.pth", None,
                            "kdv7", 7, 1, pts, ex),
                  "7kdv/PI-7kdv1soliton.py + 7kdv/model/PI 1t-2.pth"))

    # ---- 7kdv sliding window class (loss_prev) -----------------------------
    msw = load_script("7kdv/sw-73.py")
    lb, ub = np.array([-15.0, -0.5]), np.array([15.0, 0.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -0.5, 1)
    Xp = sample_points(lb, ub, 20, rng)
    ex["X_prev"] = Xp
    ex["u_prev"] = (1.5 / np.cosh(Xp[:, 0:1]) ** 2).astype(np.float32)
    cases.append(("kdv7_sw_ckpt", "kdv7_3sol", L50, lb, ub,
                  eval_case(msw, "PhysicsInformedNN", L50, lb, ub, "7kdv/model/sw73_1.pth", None,
                            "kdv7", 7, 1, pts, ex),
                  "7kdv/sw-73.py + 7kdv/model/sw73_1.pth"))

    # ---- 7kdv bc class, 5 hidden layers (loss1) -----------------------------
    mbc = load_script("7kdv/bc-7kdv3soliton.py")
    L505 = [2, 50, 50, 50, 50, 50, 1]
    lb, ub = np.array([-15.0, -0.5]), np.array([15.0, 0.5])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -0.5, 1)
    Xh = sample_points(lb, ub, 40, rng)
    ex["x1"], ex["t1"] = Xh[:, 0:1], Xh[:, 1:2]
    ex["u1"] = (1.2 / np.cosh(Xh[:, 0:1] - Xh[:, 1:2]) ** 2).astype(np.float32)
    cases.append(("kdv7_bc_xavier", "kdv7_3sol", L505, lb, ub,
                  eval_case(mbc, "bcPhysicsInformedNN", L505, lb, ub, None, 4321, "kdv7", 7, 1, pts, ex),
                  "7kdv/bc-7kdv3soliton.py bcPhysicsInformedNN + xavier(manual_seed 4321)"))

    # ---- 5kdv ----------------------------------------------------------------
    m51 = load_script("5kdv/PI-51.py")
    lb, ub = np.array([-10.0, -2.0]), np.array([10.0, 2.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -2.0, 1)
    cases.append(("kdv5_ckpt", "kdv5", L50, lb, ub,
                  eval_case(m51, "PhysicsInformedNN", L50, lb, ub, "5kdv/model/PINN1t-2.pth", None,
                            "kdv5", 5, 1, pts, ex),
                  "5kdv/PI-51.py + 5kdv/model/PINN1t-2.pth"))
    cases.append(("kdv5_xavier", "kdv5", L50, lb, ub,
                  eval_case(m51, "PhysicsInformedNN", L50, lb, ub, None, 1234, "kdv5", 5, 1, pts, ex),
                  "5kdv/PI-51.py + xavier(manual_seed 1234)"))

    # ---- Burgers ---------------------------------------------------------------
    mb = load_script("Burgers/Burgers_equation(PINN).py")
    lb, ub = np.array([-1.0, 0.0]), np.array([1.0, 0.99])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, 0.0, 1)
    ex["u0"] = (-np.sin(np.pi * ex["x0"])).astype(np.float32)
    cases.append(("burgers_ckpt", "burgers", L50, lb, ub,
                  eval_case(mb, "PhysicsInformedNN", L50, lb, ub,
                            "Burgers/model/model_50,50,10000,4x50.pth", None, "burgers", 2, 1, pts, ex),
                  "Burgers/Burgers_equation(PINN).py + Burgers/model/model_50,50,10000,4x50.pth"))
    L20 = [2] + [20] * 8 + [1]   # BASELINE.json configs[0]: "9x20 tanh MLP" (9 Linear layers of width 20)
    cases.append(("burgers_9x20_xavier", "burgers", L20, lb, ub,
                  eval_case(mb, "PhysicsInformedNN", L20, lb, ub, None, 1234, "burgers", 2, 1, pts, ex),
                  "Burgers/Burgers_equation(PINN).py + [2,20x8,1] xavier(manual_seed 1234)"))

    # ---- CH (mixed derivative u_xxt) ------------------------------------------
    mch = load_script("CH/PI-CH-1soliton.py")
    lb, ub = np.array([-15.0, -1.0]), np.array([15.0, 1.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -1.0, 1)
    cases.append(("ch_ckpt", "ch", L50, lb, ub,
                  eval_case(mch, "PhysicsInformedNN", L50, lb, ub, "CH/model/PINN CH.pth", None,
                            "ch", 3, 3, pts, ex),
                  "CH/PI-CH-1soliton.py + CH/model/PINN CH.pth"))
    L64 = [2, 64, 64, 64, 64, 1]   # BASELINE.json configs[4]: width-64 MLP
    cases.append(("ch_w64_xavier", "ch", L64, lb, ub,
                  eval_case(mch, "PhysicsInformedNN", L64, lb, ub, None, 1234, "ch", 3, 3, pts, ex),
                  "CH/PI-CH-1soliton.py + [2,64x4,1] xavier(manual_seed 1234)"))
    mchbc = load_script("CH/bc-CH-1soliton.py")
    cases.append(("ch_bc_xavier", "ch_bc", L50, lb, ub,
                  eval_case(mchbc, "PhysicsInformedNN", L50, lb, ub, None, 99, "ch", 3, 3, pts, ex),
                  "CH/bc-CH-1soliton.py PhysicsInformedNN (+2u_x) + xavier(manual_seed 99)"))

    # ---- GNSS (two outputs) ------------------------------------------------------
    mg = load_script("NLS/sw-GNSS-adamW.py")
    L100 = [2, 100, 100, 100, 100, 2]
    lb, ub = np.array([-5.0, -2.0]), np.array([5.0, 0.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -2.0, 2)
    Xp = sample_points(lb, ub, 20, rng)
    ex["X_prev"] = Xp
    ex["u_prev"] = (0.7 * np.cos(Xp[:, 0:1])).astype(np.float32)
    ex["v_prev"] = (0.3 * np.sin(Xp[:, 1:2])).astype(np.float32)
    cases.append(("gnss_w100_ckpt", "gnss", L100, lb, ub,
                  eval_case(mg, "PhysicsInformedNN", L100, lb, ub,
                            "NLS/model/adamWsw(0.1+0)t-2_1.pth", None, "gnss", 3, 1, pts, ex),
                  "NLS/sw-GNSS-adamW.py + NLS/model/adamWsw(0.1+0)t-2_1.pth"))
    mgp = load_script("NLS/PI-GNSS-torch.py")
    L502 = [2, 50, 50, 50, 50, 2]
    lb, ub = np.array([-5.0, -2.0]), np.array([5.0, 2.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -2.0, 2)
    cases.append(("gnss_w50_ckpt", "gnss", L502, lb, ub,
                  eval_case(mgp, "PhysicsInformedNN", L502, lb, ub,
                            "NLS/model/PINN-(0.1+0)t-2.pth", None, "gnss", 3, 1, pts, ex),
                  "NLS/PI-GNSS-torch.py + NLS/model/PINN-(0.1+0)t-2.pth"))
    cases.append(("gnss_w50_xavier", "gnss", L502, lb, ub,
                  eval_case(mgp, "PhysicsInformedNN", L502, lb, ub, None, 1234, "gnss", 3, 1, pts, ex),
                  "NLS/PI-GNSS-torch.py + xavier(manual_seed 1234)"))

    # ---- two-output bc class (loss1 over u1 AND v1), NLS/bc-GNSS-torch.py:222-312 ------------------
    mgbc = load_script("NLS/bc-GNSS-torch.py")
    lb, ub = np.array([-5.0, -1.0]), np.array([5.0, 1.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -1.0, 2)
    Xh = sample_points(lb, ub, 40, rng)
    ex["x1"], ex["t1"] = Xh[:, 0:1], Xh[:, 1:2]
    ex["u1"] = (0.8 * np.cos(Xh[:, 0:1]) / np.cosh(Xh[:, 1:2])).astype(np.float32)
    ex["v1"] = (0.4 * np.sin(Xh[:, 0:1] + Xh[:, 1:2])).astype(np.float32)
    cases.append(("gnss_bc_xavier", "gnss", L502, lb, ub,
                  eval_case(mgbc, "bcPhysicsInformedNN", L502, lb, ub, None, 2468, "gnss", 3, 1, pts, ex),
                  "NLS/bc-GNSS-torch.py bcPhysicsInformedNN (two outputs, loss1 on u1 and v1) + xavier(manual_seed 2468)"))

    # ---- plain NLS (the GNSS body with c = 0), NLS/sw-PINN-rogue.py:160-181 ---------------------------
    mnls = load_script("NLS/sw-PINN-rogue.py")
    lb, ub = np.array([-5.0, -1.0]), np.array([5.0, 0.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -1.0, 2)
    Xp = sample_points(lb, ub, 20, rng)
    ex["X_prev"] = Xp
    ex["u_prev"] = (0.6 * np.cos(Xp[:, 0:1])).astype(np.float32)
    ex["v_prev"] = (0.2 * np.sin(Xp[:, 1:2])).astype(np.float32)
    cases.append(("nls_w100_xavier", "nls", L100, lb, ub,
                  eval_case(mnls, "PhysicsInformedNN", L100, lb, ub, None, 1357, "gnss", 3, 1, pts, ex),
                  "NLS/sw-PINN-rogue.py + xavier(manual_seed 1357)"))

    # ---- 4th-order NLS of the scratch scripts (optional id), NLS/Ibc-test.py:222-313 (bc class) ----------
    mlpd = load_script("NLS/Ibc-test.py")
    lb, ub = np.array([-5.0, -5.0]), np.array([5.0, 5.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -5.0, 2)
    Xh = sample_points(lb, ub, 40, rng)
    ex["x1"], ex["t1"] = Xh[:, 0:1], Xh[:, 1:2]
    ex["u1"] = (0.5 * np.cos(Xh[:, 0:1]) / np.cosh(0.3 * Xh[:, 1:2])).astype(np.float32)
    ex["v1"] = (0.3 * np.sin(Xh[:, 0:1] + 0.2 * Xh[:, 1:2])).astype(np.float32)
    cases.append(("lpd_w50_xavier", "lpd", L502, lb, ub,
                  eval_case(mlpd, "bcPhysicsInformedNN", L502, lb, ub, None, 8642, "lpd", 4, 1, pts, ex),
                  "NLS/Ibc-test.py bcPhysicsInformedNN (4th-order residual :311-312, loss1) + xavier(manual_seed 8642)"))

    # ---- list-of-reference-terms loss_s, NLS/IBCtest.py:203-291 (IbcPhysicsInformedNN, plain NLS residual) ------
    mibc = load_script("NLS/IBCtest.py")
    lb, ub = np.array([-5.0, -1.0]), np.array([5.0, 1.0])
    mibc.t0_val = -1.0
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, -1.0, 2)
    Xa, Xb = sample_points(lb, ub, 30, rng), sample_points(lb, ub, 17, rng)
    ex["earliest_solutions"] = [     # h_ref = |psi| has ONE column (:444-449): torch broadcasts it against (N, 2)
        (Xa, (0.9 / np.cosh(Xa[:, 0:1])).astype(np.float32)),
        (Xb, (0.4 + 0.1 * np.cos(Xb[:, 0:1] * Xb[:, 1:2])).astype(np.float32))]
    cases.append(("nls_ibc_xavier", "nls", L502, lb, ub,
                  eval_case(mibc, "IbcPhysicsInformedNN", L502, lb, ub, None, 9753, "gnss", 3, 1, pts, ex),
                  "NLS/IBCtest.py IbcPhysicsInformedNN (loss_s over a list of (X_ref, h_ref)) + xavier(manual_seed 9753)"))

    # ---- plain NLS + derivative regulariser, NLS/sw-test.py:159-189 (reg_term added to the loss, :147) ----------
    mreg = load_script("NLS/sw-test.py")
    lb, ub = np.array([-5.0, 0.0]), np.array([5.0, 1.0])
    pts = sample_points(lb, ub, N_PTS, rng)
    ex = data_terms(rng, lb, ub, 0.0, 2)
    Xp = sample_points(lb, ub, 20, rng)
    ex["X_prev"] = Xp
    ex["u_prev"] = (0.5 * np.cos(Xp[:, 0:1])).astype(np.float32)
    ex["v_prev"] = (0.25 * np.sin(Xp[:, 1:2])).astype(np.float32)
    cases.append(("nls_reg_xavier", "nls_reg", L100, lb, ub,
                  eval_case(mreg, "PhysicsInformedNN", L100, lb, ub, None, 1593, "nls_reg", 2, 1, pts, ex),
                  "NLS/sw-test.py (plain NLS + 1e-9 derivative regulariser) + xavier(manual_seed 1593)"))

    only = os.environ.get("GEN_GOLDEN_ONLY")          # e.g. "gnss_bc_xavier,nls_w100_xavier": write just these
    for name, eq, layers, lb, ub, arrs, src in cases:
        if only and name not in only.split(","):
            continue
        arrs.update(eq=np.array(eq), layers=np.array(layers), lb=lb.astype(np.float64),
                    ub=ub.astype(np.float64), source=np.array(src))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
        print("wrote", name, "loss_f64=%.6e loss_f32=%.6e" % (arrs["loss_f64"], arrs["loss_f32"]))

    # ---- end-to-end known answers: predict + rel-L2 vs the shipped exact solutions --
    # 7kdv: error.txt:101-102 records 3.743270e-01 for PI 73.pth.
    data = scipy.io.loadmat(os.path.join(REF, "7kdv/data/73[-0.5,0.5]201*512.mat"))
    tt = data["t"].T.flatten()[:, None]
    xx = data["x"].T.flatten()[:, None]
    Exact = np.real(data["uu"])
    X, T = np.meshgrid(xx, tt)
    X_star = np.hstack((X.flatten()[:, None], T.flatten()[:, None]))
    u_star = Exact.flatten()[:, None]
    lb, ub = X_star.min(0), X_star.max(0)
    m73.lb, m73.ub = lb, ub
    model = m73.PhysicsInformedNN(L50, lb, ub)
    model.dnn.load_state_dict(torch.load(os.path.join(REF, "7kdv/model/PI 73.pth"), map_location="cpu", weights_only=True))
    u_pred = model.predict(X_star)
    err = np.linalg.norm(u_star - u_pred, 2) / np.linalg.norm(u_star, 2)
    print("7kdv PI 73 rel-L2 = %e (error.txt:102 says 3.743270e-01)" % err)
    sel = np.arange(0, X_star.shape[0], 53)
    np.savez_compressed(os.path.join(OUT, "predict_kdv7_73.npz"),
                        theta=flat_theta(model).astype(np.float32), layers=np.array(L50), lb=lb, ub=ub,
                        t=tt, x=xx, uu=Exact, err_ref=np.array(err), sel=sel,
                        u_pred_sel=u_pred[sel, 0].astype(np.float32),
                        source=np.array("7kdv/PI-7kdv3soliton.py:253-258 + 7kdv/data/73[-0.5,0.5]201*512.mat"))

    # Burgers: checkpoint + burgers256*100.mat (key usol), SURVEY 8(c): 3.153112e-03
    data = scipy.io.loadmat(os.path.join(REF, "Burgers/data/burgers256*100.mat"))
    tt = data["t"].flatten()[:, None]
    xx = data["x"].flatten()[:, None]
    Exact = np.real(data["usol"]).T
    X, T = np.meshgrid(xx, tt)
    X_star = np.hstack((X.flatten()[:, None], T.flatten()[:, None]))
    u_star = Exact.flatten()[:, None]
    lb, ub = X_star.min(0), X_star.max(0)
    mb.lb, mb.ub = lb, ub
    model = mb.PhysicsInformedNN(L50, lb, ub)
    model.dnn.load_state_dict(torch.load(os.path.join(REF, "Burgers/model/model_50,50,10000,4x50.pth"), map_location="cpu", weights_only=True))
    u_pred = model.predict(X_star)
    err = np.linalg.norm(u_star - u_pred, 2) / np.linalg.norm(u_star, 2)
    print("Burgers rel-L2 = %e" % err)
    np.savez_compressed(os.path.join(OUT, "predict_burgers.npz"),
                        theta=flat_theta(model).astype(np.float32), layers=np.array(L50), lb=lb, ub=ub,
                        t=tt, x=xx, uu=Exact, err_ref=np.array(err),
                        u_pred=u_pred[:, 0].astype(np.float32),
                        source=np.array("Burgers/Burgers_equation(PINN).py:194-258 + Burgers/data/burgers256*100.mat"))


if __name__ == "__main__":
    main()
