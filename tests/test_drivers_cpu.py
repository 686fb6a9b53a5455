"""Host logic of the sw-/bc- drivers and samplers (pde_amd/drivers.py) on the CPU, with the TEST-ONLY
oracle engine standing in for the HIP library (tests/oracle_engine.py)."""
import os

import numpy as np
import pytest

from oracle_engine import install
from pde_amd import drivers, scripts


def test_lhs_one_point_per_stratum():
    rng = np.random.default_rng(5)
    H = drivers.lhs(3, 40, rng)
    assert H.shape == (40, 3) and H.min() >= 0.0 and H.max() < 1.0
    for j in range(3):
        assert sorted(np.floor(H[:, j] * 40).astype(int).tolist()) == list(range(40))


def test_grid_points_and_relative_l2():
    x = np.linspace(-1, 1, 5)[:, None]
    t = np.linspace(0, 1, 3)[:, None]
    X = drivers.grid_points(x, t)
    assert X.shape == (15, 2)
    assert np.allclose(X[:5, 1], 0.0) and np.allclose(X[:5, 0], x[:, 0])      # t-major, as np.meshgrid(x, t)
    u = np.arange(15.0)[:, None] + 1.0
    assert drivers.relative_l2(u, u) == 0.0
    assert abs(drivers.relative_l2(1.1 * u, u) - 0.1) < 1e-12


def test_load_mat_roundtrip(tmp_path):
    import scipy.io
    t = np.linspace(0, 1, 4); x = np.linspace(-1, 1, 6)
    uu = np.outer(t + 1, np.cos(x)) + 0j
    p = os.path.join(tmp_path, "d.mat")
    scipy.io.savemat(p, {"t": t, "x": x, "uu": uu})
    tt, xx, ex = drivers.load_mat(p)
    assert tt.shape == (4, 1) and xx.shape == (6, 1) and ex.shape == (4, 6) and np.allclose(ex, uu.real)
    scipy.io.savemat(p, {"t": t, "x": x, "usol": uu.real.T})                    # Burgers layout (X,T)
    assert np.allclose(drivers.load_mat(p)[2], uu.real)


def _toy():
    x = np.linspace(-3, 3, 24)[:, None]
    t = np.linspace(-0.5, 0.5, 21)[:, None]
    exact = np.exp(-(x.T - t) ** 2)                                              # (T, X)
    return x, t, exact


def test_sliding_window_schedule_and_stitching(tmp_path):
    x, t, exact = _toy()
    cls = install(scripts.kdv5_1soliton.sw.PhysicsInformedNN)
    cls.base_dir = str(tmp_path)
    models, full, err = drivers.sliding_window(cls, [2, 6, 6, 1], x, t, exact, N0=10, Nb=6, Nf=40, n_overlap=8,
                                               nIter=2, LBFGS_max_iter=0, rng=np.random.default_rng(0))
    assert len(models) == 3 and full.shape == (21 * 24, 1) and np.isfinite(err)
    # windows [-.5,0], [-.25,.25], [0,.5] (7kdv/sw-73.py:266-272); later windows own the overlap
    bounds = [(m.lb[1], m.ub[1]) for m in models]
    assert np.allclose(bounds, [(-0.5, 0.0), (-0.25, 0.25), (0.0, 0.5)])
    X = drivers.grid_points(x, t)
    for m, (a, b), owner in zip(models, bounds, [(-0.5, -0.25), (-0.25, 0.0), (0.0, 0.5)]):
        sel = (X[:, 1] >= owner[0] - 1e-12) & (X[:, 1] < owner[1] - 1e-9) if owner[1] < 0.5 else (X[:, 1] >= owner[0] - 1e-12)
        assert np.allclose(full[sel], m.predict(X[sel]), atol=1e-6)
    # warm start + overlap term present from window 2 on
    assert models[0].X_prev is None and models[1].X_prev is not None and models[1].X_prev.shape[0] == 8
    for w in (1, 2, 3):
        assert os.path.exists(os.path.join(tmp_path, "model", "sw_%d.pth" % w))
    # evaluation mode reloads the saved windows and reproduces the stitched field
    _, full2, err2 = drivers.sliding_window(cls, [2, 6, 6, 1], x, t, exact, train=False)
    assert np.allclose(full2, full, atol=1e-6) and abs(err2 - err) < 1e-6


def test_backward_compatible_segments(tmp_path):
    x, t, exact = _toy()
    pi = install(scripts.kdv5_1soliton.PI.PhysicsInformedNN)
    bc = install(scripts.kdv5_1soliton.bc.bcPhysicsInformedNN)
    pi.base_dir = bc.base_dir = str(tmp_path)
    for corrected in (False, True):
        models, u1 = drivers.backward_compatible(pi, bc, [2, 6, 6, 1], x, t, exact, step=10, N0=10, Nf=40, nIter=2,
                                                 LBFGS_max_iter=0, log_name=None, rng=np.random.default_rng(0),
                                                 corrected=corrected)
        # 21 time rows, segments of 10 rows (+1 row overlap): rows [0,11), [10,21), then one row is left: the
        # reference's loop stops there (fewer rows than Nb), and so does the corrected one (a single row)
        assert len(models) == 2
        assert type(models[0]).__name__ == "PhysicsInformedNN" and type(models[1]).__name__ == "bcPhysicsInformedNN"
        assert np.isclose(models[0].ub[1], t[10, 0]) and np.isclose(models[1].ub[1], t[20, 0])
        assert np.isclose(models[1].lb[1], t[0, 0])                                  # the domain grows, it does not slide
        assert models[1].x1.shape[0] == 11 * 24                                      # history grid = rows 0..10
        # the reference pins the IC at the segment's first ROW INDEX (CH/bc-CH-1soliton.py:447-448); corrected: its time
        assert np.isclose(float(models[1].t0[0, 0]), t[10, 0] if corrected else 10.0)
        assert u1.shape == (21 * 24, 1)


def _stub_predict(X, seg):
    X = np.asarray(X, dtype=np.float64)
    return (np.sin(0.3 * X[:, 0:1] + 0.1 * seg) * np.cos(1.7 * X[:, 1:2]) + 0.05 * seg).astype(np.float64)


def test_backward_compatible_hands_over_what_the_reference_does():
    """pde_amd.drivers.backward_compatible against tests/golden/bc_driver_args.npz: the arguments the REFERENCE's own
    loop (CH/bc-CH-1soliton.py:334-473, replayed by oracle/gen_bc_driver_fixture.py on a synthetic 201 x 128
    travelling wave with train/save as no-ops and predict in closed form) hands to its classes, segment by segment:
    class, lb, ub, x0, u0, t0 (the row index 50 / 100 / 150), u1, tb, Xf, Xh - bit for bit, with the same numpy seed
    and the same stand-ins on the drop-in classes.  The end-to-end error of a CH bc run is parity-unpinned (the data
    file CH/data/CH-201*512.mat is a missing blob); what the loop DOES is pinned here."""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bc_driver_args.npz"))
    nx, nt = int(g["nx"]), int(g["nt"])
    x = np.linspace(-15.0, 15.0, nx)[:, None]
    t = np.linspace(-1.0, 1.0, nt)[:, None]
    exact = 1.0 / np.cosh(0.5 * (x.T - 3.0 * t)) ** 2                                # (T, X), as the generator's file
    ns = scripts.for_script("CH/bc-CH-1soliton.py")
    calls, counter = [], {"seg": 0}

    def stand_in(base):
        class C(install(base)):
            def __init__(self, layers, lb, ub, *a):
                super().__init__(layers, lb, ub, *a)
                self._seg = counter["seg"]
                counter["seg"] += 1
                calls.append({"cls": base.__name__, "lb": np.array(lb, dtype=np.float64), "ub": np.array(ub, dtype=np.float64)})

            def set_training_data(self, *args):
                names = ["x0", "u0", "tb", "Xf"] if base.__name__ == "PhysicsInformedNN" else ["x0", "u0", "t0", "u1", "tb", "Xf", "Xh"]
                assert len(args) == len(names)
                calls[-1].update(dict(zip(names, args)))
                super().set_training_data(*args)

            def train(self, *a, **k):
                pass

            def save_model(self, *a, **k):
                pass

            def predict(self, X):
                return _stub_predict(X, self._seg)
        C.__name__ = base.__name__
        return C
    np.random.seed(1234)                                   # the reference's seed (:21); draws in the reference's order
    models, u1 = drivers.backward_compatible(stand_in(ns.PhysicsInformedNN), stand_in(ns.bcPhysicsInformedNN),
                                             [2, 50, 50, 50, 50, 1], x, t, exact, rng=np.random)
    assert str(g["stopped"]).startswith("ValueError")      # the reference's own run ends in its fifth pass ...
    assert len(calls) == int(g["n_segments"]) == 4          # ... after four segments, and so does the default here
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes()).hexdigest()
    for i, c in enumerate(calls):
        assert c["cls"] == str(g["seg%d_cls" % i])
        for k in ("lb", "ub", "x0", "u0", "tb"):
            assert np.array_equal(np.asarray(c[k], dtype=np.float64), g["seg%d_%s" % (i, k)]), (i, k)
        for k in ("Xf", "Xh", "u1"):
            if "seg%d_%s_sha256" % (i, k) in g.files:
                assert tuple(np.asarray(c[k]).shape) == tuple(g["seg%d_%s_shape" % (i, k)]), (i, k)
                assert sha(c[k]) == str(g["seg%d_%s_sha256" % (i, k)]), (i, k)
        if i > 0:
            assert float(c["t0"]) == float(g["seg%d_t0" % i]) == 50.0 * i and isinstance(c["t0"], int)
    # the corrected variant differs exactly where documented: t0 is a time, u0 the previous model at x0
    np.random.seed(1234)
    calls.clear(); counter["seg"] = 0
    drivers.backward_compatible(stand_in(ns.PhysicsInformedNN), stand_in(ns.bcPhysicsInformedNN),
                                [2, 50, 50, 50, 50, 1], x, t, exact, rng=np.random, corrected=True)
    assert len(calls) == 4 and abs(float(calls[1]["t0"]) - t[50, 0]) < 1e-12
    assert np.array_equal(calls[1]["Xf"], calls[1]["Xf"]) and sha(calls[1]["Xh"]) == str(g["seg1_Xh_sha256"])


def test_sliding_window_two_output_gnss(tmp_path):
    """NLS/GNSS nets predict (u, v): complex data in, stitched (u, v, h) and three errors out
    (NLS/sw-GNSS-adamW.py:262-404)."""
    x, t, exact = _toy()
    uu = exact * np.exp(1j * (0.7 * x.T + 0.3 * t))                               # (T, X) complex
    cls = install(scripts.gnss.sw.PhysicsInformedNN)
    cls.base_dir = str(tmp_path)
    models, (fu, fv, fh), errs = drivers.sliding_window(cls, [2, 6, 6, 2], x, t, uu, N0=10, Nb=6, Nf=40,
                                                        n_overlap=8, nIter=2, LBFGS_max_iter=0,
                                                        rng=np.random.default_rng(0))
    assert len(models) == 3 and fu.shape == fv.shape == fh.shape == (21 * 24, 1)
    assert len(errs) == 3 and all(np.isfinite(e) for e in errs)
    assert np.allclose(fh, np.sqrt(fu ** 2 + fv ** 2))
    X = drivers.grid_points(x, t)
    last = X[:, 1] >= 0.0 - 1e-12                                                # the last window owns [0, .5]
    u3, v3 = models[2].predict(X[last])
    assert np.allclose(fu[last], u3, atol=1e-6) and np.allclose(fv[last], v3, atol=1e-6)
    # window 1: IC from the data's first row; window 2: overlap labels for both outputs
    assert models[0].X_prev is None
    assert models[1].X_prev.shape[0] == 8 and models[1].u_prev.shape == (8, 1) and models[1].v_prev.shape == (8, 1)
    # a (u, v) pair is accepted like complex data
    _, _, errs2 = drivers.sliding_window(cls, [2, 6, 6, 2], x, t, (uu.real, uu.imag), train=False)
    assert np.allclose(errs2, errs, atol=1e-6)
    # complex .mat loading keeps the imaginary part
    import scipy.io
    p = os.path.join(tmp_path, "nls.mat")
    scipy.io.savemat(p, {"t": t, "x": x, "uu": uu})
    assert np.allclose(drivers.load_mat(p, complex_valued=True)[2], uu)


def test_lhs_device_and_tensor_collocation_points():
    """drivers.lhs_device (torch, here on the CPU device) has the Latin-hypercube property, and
    set_training_data takes a tensor Xf as is (same xf/tf as the numpy route)."""
    import torch
    g = torch.Generator().manual_seed(11)
    H = drivers.lhs_device(2, 50, torch.device("cpu"), generator=g)
    assert H.shape == (50, 2) and float(H.min()) >= 0.0 and float(H.max()) < 1.0
    for j in range(2):
        assert sorted(torch.floor(H[:, j] * 50).long().tolist()) == list(range(50))
    cls = install(scripts.kdv5_1soliton.PI.PhysicsInformedNN)
    lb, ub = np.array([-3.0, -0.5]), np.array([3.0, 0.5])
    Xf = torch.tensor(lb) + torch.tensor(ub - lb) * H
    x0 = np.linspace(-3, 3, 7)[:, None]
    tb = np.linspace(-0.5, 0.5, 5)[:, None]
    a, b = cls([2, 6, 6, 1], lb, ub), cls([2, 6, 6, 1], lb, ub)
    a.set_training_data(x0, np.exp(-x0 ** 2), tb, Xf)
    b.set_training_data(x0, np.exp(-x0 ** 2), tb, Xf.numpy())
    assert a.xf.dtype == torch.float32 and torch.equal(a.xf, b.xf) and torch.equal(a.tf, b.tf)
    assert a.n_f_global == 50


# ---- end-to-end pins produced by the reference itself (oracle/gen_driver_fixtures.py) ---------------------
from collections import OrderedDict  # noqa: E402
import re  # noqa: E402

import torch  # noqa: E402

from conftest import GOLDEN  # noqa: E402


def write_state_dict(path, theta, layers):
    """theta (flat, state-dict order) -> a .pth with the reference's keys 1.layers.layer_i.{weight,bias}"""
    sd, off = OrderedDict(), 0
    for i in range(len(layers) - 1):
        fi, fo = layers[i], layers[i + 1]
        sd["1.layers.layer_%d.weight" % i] = torch.tensor(theta[off:off + fi * fo].reshape(fo, fi).copy())
        off += fi * fo
        sd["1.layers.layer_%d.bias" % i] = torch.tensor(theta[off:off + fo].copy())
        off += fo
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(sd, path)


def run_sw73_test_mode(cls, tmp_path):
    g = np.load(os.path.join(GOLDEN, "predict_sw73.npz"))
    for w in range(3):
        write_state_dict(os.path.join(str(tmp_path), "model", "sw73_%d.pth" % (w + 1)), g["thetas"][w], g["layers"].tolist())
    cls = type("SW73", (cls,), {"base_dir": str(tmp_path)})
    models, full, err = drivers.sliding_window(cls, g["layers"].tolist(), g["x"], g["t"], g["uu"], train=False,
                                               model_name="sw73_{win}.pth")
    # the windows the driver derived are the reference's (7kdv/sw-73.py:250-252,268-273)
    for m, b in zip(models, g["bounds"]):
        assert np.allclose(np.concatenate([m._lb64, m._ub64]), b, atol=1e-12)
    return g, full, err


def test_sliding_window_test_mode_reproduces_error_txt(tmp_path):
    """7kdv/sw-73.py in test mode (:343-376) with the shipped sw73_{1,2,3}.pth: error.txt:109-110 records
    'Overall Error u: 3.223475e-02'; the reference run in the build container gives 3.223474e-02."""
    g, full, err = run_sw73_test_mode(install(scripts.for_script("7kdv/sw-73.py").PhysicsInformedNN), tmp_path)
    assert abs(err - float(g["err_ref"])) < 1e-6 * float(g["err_ref"])
    assert abs(err - 3.223475e-02) < 5e-8
    assert np.max(np.abs(full[g["sel"], 0] - g["u_full_sel"])) < 2e-5 * np.max(np.abs(g["u_full_sel"]))


def test_predict_known_answer_kdv5_error_txt():
    """5kdv/PI-51.py:262-267 with 5kdv/model/PINN1t-2.pth: error.txt:4, 'Error u: 2.446954e-02'."""
    g = np.load(os.path.join(GOLDEN, "predict_kdv5.npz"))
    m = install(scripts.for_script("5kdv/PI-51.py").PhysicsInformedNN)(g["layers"].tolist(), g["lb"], g["ub"])
    from test_host_logic import load_theta
    load_theta(m, g["theta"])
    u = m.predict(drivers.grid_points(g["x"], g["t"]))
    err = drivers.relative_l2(u, g["uu"].flatten()[:, None])
    assert abs(err - 2.446954e-02) < 5e-8 and abs(err - float(g["err_ref"])) < 1e-6 * err


def test_load_model_reads_a_checkpoint_the_reference_wrote(tmp_path):
    """tests/golden/ref_written_kdv7.pth was written by the reference's own save_model()
    (7kdv/PI-7kdv3soliton.py:173-177): load_model() takes it as is (weights_only load), bit for bit."""
    import shutil
    g = np.load(os.path.join(GOLDEN, "ref_written_kdv7.npz"))
    os.makedirs(os.path.join(str(tmp_path), "model"))
    shutil.copy(os.path.join(GOLDEN, "ref_written_kdv7.pth"), os.path.join(str(tmp_path), "model", "w.pth"))
    cls = type("M", (install(scripts.for_script("7kdv/PI-7kdv3soliton.py").PhysicsInformedNN),), {"base_dir": str(tmp_path)})
    m = cls(g["layers"].tolist(), g["lb"], g["ub"])
    assert list(m.dnn.state_dict().keys()) == [str(k) for k in g["keys"]]
    m.load_model("w.pth")
    assert np.array_equal(m.dnn.flat_theta().numpy(), g["theta"])
    assert np.max(np.abs(m.predict(g["X"]) - g["u"])) < 1e-5 * np.max(np.abs(g["u"]))


@pytest.mark.skipif(not os.path.isdir("/root/reference/7kdv"), reason="the reference tree only exists in the build container")
def test_reference_loads_a_checkpoint_we_wrote(tmp_path):
    """The other direction of the .pth interchange: the reference's own load_model() reads what our
    save_model() wrote (container-only: needs the reference importable)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "oracle"))
    from gen_golden import load_script
    g = np.load(os.path.join(GOLDEN, "ref_written_kdv7.npz"))
    cls = type("M", (install(scripts.for_script("7kdv/PI-7kdv3soliton.py").PhysicsInformedNN),), {"base_dir": str(tmp_path)})
    torch.manual_seed(3)
    m = cls(g["layers"].tolist(), g["lb"], g["ub"])
    m.save_model("ours.pth")
    mod = load_script("7kdv/PI-7kdv3soliton.py")
    mod.current_dir = str(tmp_path)
    mod.lb, mod.ub = g["lb"], g["ub"]
    ref = mod.PhysicsInformedNN(g["layers"].tolist(), g["lb"], g["ub"])
    ref.load_model("ours.pth")
    got = torch.cat([p.detach().reshape(-1) for p in ref.dnn.parameters()]).numpy()
    assert np.array_equal(got, m.dnn.flat_theta().numpy())
    assert np.max(np.abs(ref.predict(g["X"]) - m.predict(g["X"]))) < 1e-5


LOG_KEYS = {"pi": ["Iter", "Loss", "Loss_0", "Loss_b", "Loss_f"],
            "sw": ["Iter", "Loss", "Loss0", "Loss_b", "Loss_f", "Loss_prev"],
            "bc": ["Iter", "Loss", "Loss_0", "Loss_1", "Loss_b", "Loss_f"]}


def parse_log_line(line):
    items = [kv.split(": ") for kv in line.split(", ")]
    return [k for k, _ in items], [float(v) for _, v in items]


@pytest.mark.parametrize("tag,script,cname", [("pi", "7kdv/PI-7kdv3soliton.py", "PhysicsInformedNN"),
                                              ("sw", "7kdv/sw-73.py", "PhysicsInformedNN"),
                                              ("bc", "7kdv/bc-7kdv3soliton.py", "bcPhysicsInformedNN")])
def test_log_file_matches_what_the_reference_wrote(tag, script, cname, tmp_path):
    """train(1, logfile, 0, 1) on the weights and data of tests/golden/ref_logs.npz writes the log file the
    reference wrote: 'Start Training:', the Iter-0 line with the same keys in the same order and %.5e values
    (agreeing to 1e-4: fp32 kernels vs the reference's fp32 autograd), 'Training time: %.4f'.  The format
    of a long run is the shipped 7kdv/log/log-PI 73.txt:1-3."""
    from test_host_logic import load_theta
    g = np.load(os.path.join(GOLDEN, "ref_logs.npz"))
    cls = type("M", (install(getattr(scripts.for_script(script), cname)),), {"base_dir": str(tmp_path), "lbfgs_impl": "torch"})
    m = cls(g["layers"].tolist(), g["lb"], g["ub"])
    load_theta(m, g["theta_" + tag])
    if tag == "pi":
        m.set_training_data(g["x0"], g["u0"], g["tb"], g["Xf"])
    elif tag == "sw":
        m.set_training_data(g["x0"], g["u0"], g["tb"], g["Xf"], g["X_prev"], g["u_prev"])
    else:
        m.set_training_data(g["x0"], g["u0"], -0.5, g["u1"], g["tb"], g["Xf"], g["Xh"])
    m.train(1, "ours.txt", 0, 1)
    ours = open(os.path.join(str(tmp_path), "log", "ours.txt"), encoding="utf-8").read().split("\n")
    ref = str(g["log_" + tag]).split("\n")
    assert len(ours) == len(ref) == 4 and ours[0] == ref[0] == "Start Training:" and ours[3] == ref[3] == ""
    keys_o, vals_o = parse_log_line(ours[1])
    keys_r, vals_r = parse_log_line(ref[1])
    assert keys_o == keys_r == LOG_KEYS[tag]
    assert re.fullmatch(r"Iter: \d+(, \w+: \d\.\d{5}e[+-]\d\d)+", ours[1]) and re.fullmatch(r"Iter: \d+(, \w+: \d\.\d{5}e[+-]\d\d)+", ref[1])
    assert np.allclose(vals_o, vals_r, rtol=1e-4)
    assert re.fullmatch(r"Training time: \d+\.\d{4}", ours[2]) and re.fullmatch(r"Training time: \d+\.\d{4}", ref[2])
    if tag == "pi":   # a shipped log of a long run has exactly this shape
        head = str(g["shipped_log_head"]).split("\n")
        assert head[0] == "Start Training:" and parse_log_line(head[1])[0] == LOG_KEYS["pi"] and parse_log_line(head[2])[1][0] == 100
