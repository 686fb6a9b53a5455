"""CPU: host-side logic of pde_amd (no kernels): the drop-in class composes the
same loss as the reference's backward(), keeps the reference's state-dict keys,
and the C-ABI library exports what include/pinnjet.h declares.  Compute is
supplied by a TEST-ONLY oracle engine (tests/oracle_engine.py)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle_engine import install
from pde_amd import equations, scripts
from pde_amd._native import LIB_PATH, SYMBOLS


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def load_theta(model, theta):
    off = 0
    with torch.no_grad():
        for p in model.dnn.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(theta[off:off + n].reshape(tuple(p.shape))))
            off += n


def make_model(g, cls):
    model = install(cls)(g["layers"].tolist(), g["lb"], g["ub"])
    load_theta(model, g["theta"])
    return model


def flat_grad(model):
    return torch.cat([p.grad.reshape(-1) for p in model.dnn.parameters()]).numpy()


CASES = [
    ("kdv7_3sol_xavier", scripts.kdv7_3soliton.PI.PhysicsInformedNN, "pi"),
    ("kdv7_sw_ckpt", scripts.kdv7_3soliton.sw.PhysicsInformedNN, "sw"),
    ("kdv7_bc_xavier", scripts.kdv7_3soliton.bc.bcPhysicsInformedNN, "bc"),
    ("kdv5_xavier", scripts.kdv5_1soliton.PI.PhysicsInformedNN, "pi"),
    ("burgers_9x20_xavier", scripts.burgers.PI.PhysicsInformedNN, "burgers"),
    ("ch_w64_xavier", scripts.camassa_holm.PI.PhysicsInformedNN, "pi"),
    ("ch_bc_xavier", scripts.camassa_holm_bc.bc.PhysicsInformedNN, "pi"),
    ("gnss_w50_xavier", scripts.gnss.PI.PhysicsInformedNN, "gnss"),
    ("gnss_w100_ckpt", scripts.gnss.sw.PhysicsInformedNN, "gnss_sw"),
    ("gnss_bc_xavier", scripts.gnss.bc.bcPhysicsInformedNN, "gnss_bc"),      # two-output bc class, NLS/bc-GNSS-torch.py:222-312
    ("nls_w100_xavier", scripts.nls.sw.PhysicsInformedNN, "gnss_sw"),        # plain NLS (c = 0), NLS/sw-PINN-rogue.py
    ("lpd_w50_xavier", scripts.lpd.bc.bcPhysicsInformedNN, "gnss_bc"),       # 4th-order residual id, NLS/Ibc-test.py:222-313
    ("nls_ibc_xavier", scripts.nls.sw.PhysicsInformedNN, "gnss_es"),         # loss_s over a list of terms, NLS/IBCtest.py:252-258
    ("nls_reg_xavier", scripts.nls_reg.sw.PhysicsInformedNN, "gnss_sw"),     # derivative regulariser, NLS/sw-test.py:172-180
]


def feed(model, g, kind):
    Xf = np.stack([g["x"], g["t"]], axis=1).astype(np.float64)
    d = {k[5:]: g[k] for k in g.files if k.startswith("data_")}
    if kind == "pi":
        model.set_training_data(d["x0"], d["u0"], d["tb"], Xf)
    elif kind == "burgers":
        model.set_training_data(d["x0"], d["u0"], d["tb"], Xf, 0.01 / np.pi)
    elif kind == "sw":
        model.set_training_data(d["x0"], d["u0"], d["tb"], Xf, d["X_prev"], d["u_prev"])
    elif kind == "bc":
        Xh = np.concatenate([d["x1"], d["t1"]], axis=1)
        model.set_training_data(d["x0"], d["u0"], float(d["t0"][0, 0]), d["u1"], d["tb"], Xf, Xh)
    elif kind == "gnss_bc":
        Xh = np.concatenate([d["x1"], d["t1"]], axis=1)
        model.set_training_data(d["x0"], d["u0"], d["v0"], float(d["t0"][0, 0]), d["u1"], d["v1"], d["tb"], Xf, Xh)
    elif kind == "gnss":
        model.set_training_data(d["x0"], d["u0"], d["v0"], d["tb"], Xf)
    elif kind == "gnss_sw":
        model.set_training_data(d["x0"], d["u0"], d["v0"], d["tb"], Xf, d["X_prev"], d["u_prev"], d["v_prev"])
    elif kind == "gnss_es":
        es = [(d["es_X%d" % i], d["es_h%d" % i]) for i in range(sum(k.startswith("es_X") for k in d))]
        model.set_training_data(d["x0"], d["u0"], d["v0"], d["tb"], Xf, earliest_solutions=es)


@pytest.mark.parametrize("impl", ["direct", "autograd"])
@pytest.mark.parametrize("name,cls,kind", CASES)
def test_backward_matches_reference_backward(name, cls, kind, impl):
    """loss and flat gradient of the full backward() vs the reference's own
    backward() (fp64 golden), fp32 tolerance; both forms of the closure (PhysicsInformedNN.closure_impl)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    model = make_model(g, cls)
    model.closure_impl = impl
    feed(model, g, kind)
    model.iter = 1
    loss = model.backward()
    ref_gerr = rel(g["grad_full32"], g["grad_full64"])
    assert abs(loss.item() - g["loss_full64"]) <= max(1e-5, 4 * abs(g["loss_full32"] - g["loss_full64"]) / g["loss_full64"]) * g["loss_full64"]
    assert rel(flat_grad(model), g["grad_full64"]) <= max(1e-5, 4 * ref_gerr)
    assert model.iter == 2


def test_loss_s_log_line(capsys):
    """the log line gains ', Loss_s: %.5e' when reference-solution terms are present (NLS/IBCtest.py:262-263)"""
    g = np.load(os.path.join(GOLDEN, "nls_ibc_xavier.npz"))
    model = make_model(g, scripts.nls.sw.PhysicsInformedNN)
    feed(model, g, "gnss_es")
    model.iter = 0
    model.backward()
    out = capsys.readouterr().out
    assert re.search(r"Iter: 0, Loss: \S+, Loss0: \S+, Loss_b: \S+, Loss_f: \S+, Loss_prev: \S+, Loss_s: \d\.\d{5}e[-+]\d\d", out), out


def test_net_f_is_differentiable_like_the_reference():
    g = np.load(os.path.join(GOLDEN, "gnss_w50_xavier.npz"))
    model = make_model(g, scripts.gnss.PI.PhysicsInformedNN)
    x = torch.tensor(g["x"][:, None])
    t = torch.tensor(g["t"][:, None])
    fu, fv = model.net_f(x, t)
    assert fu.shape == (len(g["x"]), 1)
    loss = torch.mean(torch.square(fu)) + torch.mean(torch.square(fv))
    loss.backward()
    assert abs(loss.item() - g["loss_f64"]) < 1e-5 * g["loss_f64"]
    assert rel(flat_grad(model), g["grad_f64"]) < 1e-5
    f = np.concatenate([fu.detach().numpy(), fv.detach().numpy()], axis=1)
    assert rel(f, g["f64"]) < 1e-5


def test_state_dict_keys_and_roundtrip(tmp_path):
    g = np.load(os.path.join(GOLDEN, "kdv7_3sol_ckpt.npz"))
    cls = install(scripts.kdv7_3soliton.PI.PhysicsInformedNN)
    cls.base_dir = str(tmp_path)
    m = cls([2, 50, 50, 50, 50, 1], g["lb"], g["ub"])
    keys = list(m.dnn.state_dict().keys())
    assert keys == ["1.layers.layer_%d.%s" % (i, k) for i in range(5) for k in ("weight", "bias")]
    load_theta(m, g["theta"])
    m.save_model("a.pth")
    m2 = cls([2, 50, 50, 50, 50, 1], g["lb"], g["ub"])
    m2.load_model("a.pth")
    assert torch.equal(m2.dnn.flat_theta(), m.dnn.flat_theta())
    # parameters stay views into the flat buffer after load_state_dict
    assert m2.dnn.param_list[0].data_ptr() == m2.dnn.flat_theta().data_ptr()
    # sliding-window warm start copies the weights (7kdv/sw-73.py:83-85)
    sw = install(scripts.kdv7_3soliton.sw.PhysicsInformedNN)
    m3 = sw([2, 50, 50, 50, 50, 1], g["lb"], g["ub"], m)
    assert torch.equal(m3.dnn.flat_theta(), m.dnn.flat_theta())


def test_predict_known_answer(golden_dir):
    """Burgers checkpoint + burgers256*100.mat -> rel-L2 3.153112e-03."""
    g = np.load(os.path.join(golden_dir, "predict_burgers.npz"))
    m = install(scripts.burgers.PI.PhysicsInformedNN)(g["layers"].tolist(), g["lb"], g["ub"])
    load_theta(m, g["theta"])
    X, T = np.meshgrid(g["x"], g["t"])
    X_star = np.hstack((X.flatten()[:, None], T.flatten()[:, None]))
    u = m.predict(X_star)
    assert u.shape == (X_star.shape[0], 1)
    u_star = g["uu"].flatten()[:, None]
    err = np.linalg.norm(u_star - u, 2) / np.linalg.norm(u_star, 2)
    assert abs(err - float(g["err_ref"])) < 2e-6


def test_adam_step_runs_and_decreases_loss():
    g = np.load(os.path.join(GOLDEN, "burgers_9x20_xavier.npz"))
    m = make_model(g, scripts.burgers.PI.PhysicsInformedNN)
    feed(m, g, "burgers")
    m.train(3, LBFGS_max_iter=0)
    # three Adam iterations + the one closure evaluation of LBFGS.step, which the reference's train() always
    # calls (7kdv/PI-7kdv3soliton.py:161; with max_iter = 0 torch evaluates the closure once and returns)
    assert m.iter == 4


def test_script_literals_match_reference():
    """pde_amd/scripts.py vs the literals oracle/gen_script_literals.py read out of the reference's
    source text (tests/golden/script_literals.json): every class definition of every hot-path script
    has a drop-in with the same residual body, IC time t0, boundary-loss form, Adam/AdamW choice and
    L-BFGS settings (lr, max_eval, history_size, tolerance_grad) — e.g. 7kdv/sw-73.py:173-177,
    CH/PI-CH-1soliton.py:147-151, 5kdv/PI-52.py:150, NLS/bc-GNSS-torch.py:102."""
    import json
    lit = json.load(open(os.path.join(GOLDEN, "script_literals.json")))
    assert set(lit) == set(scripts.BY_SCRIPT), set(lit) ^ set(scripts.BY_SCRIPT)
    n = 0
    for path, ent in lit.items():
        ns = scripts.for_script(path)
        assert {c["class"] for c in ent["classes"]} == {k for k in vars(ns) if k.endswith("PhysicsInformedNN")}, path
        for c in ent["classes"]:
            v = getattr(ns, c["class"]).variant
            where = "%s:%s" % (path, c["class"])
            assert v.equation == c["equation"], where
            if "lambda_reg" in c:
                assert equations.get(v.equation).coef[2] == c["lambda_reg"], where
            assert v.bc == c["bc_form"], where
            assert v.optimizer == c["optimizer"], where
            if c["optimizer"] == "adamw":
                assert v.weight_decay == c["weight_decay"], where
            if c["t0"] == "argument":
                assert v.style == "bc", where
            else:
                assert v.t0 == c["t0"], where
            assert v.style == ("bc" if c["class"] == "bcPhysicsInformedNN" else
                               "sw" if "X_prev=None" in c["set_training_data"] else "pi"), where
            for k in ("lr", "max_eval", "history_size", "tolerance_grad"):
                assert float(v.lbfgs[k]) == c["lbfgs"][k], (where, k)
            n += 1
    assert n == 43


def test_train_hands_the_script_literals_to_lbfgs():
    """train() builds its L-BFGS from the variant: CH/PI-CH-1soliton.py:147-151 (lr 0.5, max_eval 75000,
    history 150, tolerance_grad 1e-7) and the reference's tolerance_change = np.finfo(float).eps."""
    g = np.load(os.path.join(GOLDEN, "ch_ckpt.npz"))
    m = make_model(g, scripts.for_script("CH/PI-CH-1soliton.py").PhysicsInformedNN)
    feed(m, g, "pi")
    type(m).lbfgs_impl = "torch"
    m.train(0, LBFGS_max_iter=0)
    grp = m.optimizer_LBFGS.param_groups[0]
    assert (grp["lr"], grp["max_eval"], grp["history_size"], grp["tolerance_grad"]) == (0.5, 75000, 150, 1e-7)
    assert grp["tolerance_change"] == np.finfo(float).eps and grp["line_search_fn"] == "strong_wolfe"


@pytest.mark.skipif(not os.path.exists(LIB_PATH), reason="libpinnjet.so not built")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "pinnjet.h")).read()
    declared = set(re.findall(r"\b(pinn_[a-z_]+)\s*\(", hdr))
    assert declared == set(SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    lib.pinn_error_string.restype = ctypes.c_char_p
    assert lib.pinn_error_string(0) == b"ok"
    # invalid descriptors are rejected without touching the GPU
    from pde_amd._native import PinnDesc
    d = PinnDesc()
    h = ctypes.c_void_p()
    assert lib.pinn_create(ctypes.byref(d), ctypes.byref(h)) == -1
    assert lib.pinn_create(None, ctypes.byref(h)) == -4


def test_poll_advance_decides_like_the_python_line_search():
    """pinn_lbfgs_poll_advance is host code: its decision (launch the next direction, or hand the numbers back) against the
    conditions of pde_amd/lbfgs.py on random and borderline inputs, without a GPU.  A NULL workspace makes the launch
    branch return PINN_ERR_NULL (-4) before anything touches a device: -4 = "would have launched", 0 = "caller decides",
    2 = "the ticket never came"."""
    import ctypes
    from pde_amd._native import load_library
    lib = load_library()
    rng = np.random.default_rng(11)
    host = np.zeros(16, dtype=np.float32)
    vals = (ctypes.c_float * 12)()
    hp = host.ctypes.data_as(ctypes.c_void_p)

    def call(ticket, f_prev, lr, tol_x, tol_g, allow, spins=10):
        return lib.pinn_lbfgs_poll_advance(hp, ticket, spins, f_prev, lr, tol_x, tol_g, allow, None, 100, 5, None, None, None,
                                           None, None, None, vals)

    host.view(np.int32)[12] = 7
    assert call(8, 1.0, 1.0, 1e-9, 1e-7, 1) == 2                      # wrong ticket: gives up after max_spins
    n_launch = n_back = 0
    for trial in range(4000):
        lr = float(rng.choice([1.0, 0.5]))
        tol_x, tol_g = float(rng.choice([1e-9, 2.2e-16, 1e-3])), float(rng.choice([1e-7, 1e-5, 1e-2]))
        f_prev = float(np.float32(rng.uniform(0.1, 2.0)))
        gtd = np.float32(-rng.uniform(0.0, 1.0) * (10.0 ** rng.integers(-12, 1)))
        if trial % 17 == 0:
            gtd = np.float32(abs(gtd))                                # not a descent direction
        dmax = np.float32(10.0 ** rng.uniform(-12, 0))
        # around the Armijo line and the curvature bound, on both sides
        fs = np.float32(f_prev + 1e-4 * lr * float(gtd) + rng.normal() * 10.0 ** rng.integers(-9, -1))
        gtds = np.float32(-0.9 * float(gtd) * rng.uniform(-1.3, 1.3))
        gmaxs = np.float32(10.0 ** rng.uniform(-8, 0))
        if trial % 29 == 0:
            fs = np.float32(np.nan)
        host[:12] = [gtd, 1.0, dmax, 0.5, 3.0, 1.0, 1.0, 0.0, fs, gtds, gmaxs, 2.0]
        host.view(np.int32)[12] = trial
        allow = trial % 13 != 0
        rc = call(trial, f_prev, lr, tol_x, tol_g, 1 if allow else 0)
        assert list(vals) == host[:12].tolist() or np.isnan(fs)
        g, d_, f_, gt, gm = float(gtd), float(dmax), float(fs), float(gtds), float(gmaxs)
        fast = (allow and not g > -tol_x and not f_ > f_prev + 1e-4 * lr * g and abs(gt) <= -0.9 * g
                and not gm <= tol_g and not d_ * abs(lr) <= tol_x and not abs(f_ - f_prev) < tol_x)
        assert rc == (-4 if fast else 0), (trial, rc, fast)
        n_launch += fast
        n_back += not fast
    assert n_launch > 300 and n_back > 300
