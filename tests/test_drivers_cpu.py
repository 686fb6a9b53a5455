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
    models, u1 = drivers.backward_compatible(pi, bc, [2, 6, 6, 1], x, t, exact, step=10, N0=10, Nf=40, nIter=2,
                                             LBFGS_max_iter=0, log_name=None, rng=np.random.default_rng(0))
    # 21 time rows, segments of 10 rows (+1 row overlap): rows [0,11), [10,21), then the grid is exhausted
    assert len(models) == 2
    assert type(models[0]).__name__ == "PhysicsInformedNN" and type(models[1]).__name__ == "bcPhysicsInformedNN"
    assert np.isclose(models[0].ub[1], t[10, 0]) and np.isclose(models[1].ub[1], t[20, 0])
    assert np.isclose(models[1].lb[1], t[0, 0])                                  # the domain grows, it does not slide
    assert models[1].x1.shape[0] == 11 * 24                                      # history grid = rows 0..10
    assert np.isclose(float(models[1].t0[0, 0]), t[10, 0])                       # IC pinned at the segment's left edge
    assert u1.shape == (21 * 24, 1)


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
