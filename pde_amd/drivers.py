"""The reference's sw- (sliding time window) and bc- (backward-compatible) training drivers as
library functions, with the small pieces of data preparation they need.

In the reference these loops live in each script's ``__main__`` (``7kdv/sw-73.py:264-380``,
``CH/bc-CH-1soliton.py:378-473``); here they are functions of (model classes, grid data, sizes), so a
script collapses to loading its ``.mat`` file and one call.  Sampling uses a caller-supplied
``numpy.random.Generator`` instead of the global numpy state (the reference seeds
``np.random.seed(1234)`` once per script).
"""
import numpy as np


def lhs(n, samples, rng=None):
    """Latin-hypercube sample of ``samples`` points in [0,1)^n: every coordinate has exactly one point
    in each of the ``samples`` equal strata, uniformly placed inside it, strata paired at random across
    coordinates.  Replaces ``pyDOE.lhs(n, samples)`` (default criterion) used for ``tb`` and ``Xf``
    (``7kdv/PI-7kdv3soliton.py:22,225``, ``7kdv/sw-73.py:322-327``)."""
    rng = np.random.default_rng() if rng is None else rng
    u = rng.random((samples, n))
    out = np.empty((samples, n))
    for j in range(n):
        out[:, j] = (rng.permutation(samples) + u[:, j]) / samples
    return out


def lhs_device(n, samples, device, generator=None, dtype=None):
    """`lhs` generated on `device` as a torch tensor (the 10M-point CH configuration would
    otherwise spend seconds in numpy permutations and ship 160 MB over PCIe per resampling).  Same
    construction: one point per stratum and coordinate, uniformly placed, strata paired at random.
    `set_training_data` accepts the result (after scaling to the domain) as `Xf` without a host copy."""
    import torch
    dtype = dtype or torch.float64
    u = torch.rand((samples, n), device=device, generator=generator, dtype=dtype)
    cols = [(torch.randperm(samples, device=device, generator=generator).to(dtype) + u[:, j]) / samples
            for j in range(n)]
    return torch.stack(cols, dim=1)


def load_mat(path, key=None, complex_valued=False):
    """``t, x, Exact_u`` of one of the reference's data files: ``t`` (T,1), ``x`` (X,1), real part of
    ``uu`` as (T,X) (``7kdv/sw-73.py:225-229``); Burgers' ``usol`` is stored (X,T) and transposed
    (``Burgers/Burgers_equation(PINN).py:194-201``).  ``complex_valued=True`` keeps ``uu`` complex
    (NLS/GNSS: u = Re, v = Im, ``NLS/sw-GNSS-adamW.py:241-244``)."""
    import scipy.io
    data = scipy.io.loadmat(path)
    t = data['t'].flatten()[:, None]
    x = data['x'].flatten()[:, None]
    if key is None:
        key = 'uu' if 'uu' in data else 'usol'
    exact = data[key] if complex_valued else np.real(data[key])
    if exact.shape == (x.shape[0], t.shape[0]) and x.shape[0] != t.shape[0]:
        exact = exact.T
    return t, x, exact


def grid_points(x, t):
    """``X_star`` (T*X, 2) in the reference's meshgrid order (t-major) — ``7kdv/sw-73.py:232-233``."""
    X, T = np.meshgrid(x, t)
    return np.hstack((X.flatten()[:, None], T.flatten()[:, None]))


def relative_l2(u_pred, u_star):
    """``||u* - u||_2 / ||u*||_2`` as written to error.txt (``7kdv/sw-73.py:375``)."""
    return float(np.linalg.norm(u_star - u_pred, 2) / np.linalg.norm(u_star, 2))


def sliding_window(model_cls, layers, x, t, exact_u, *, num_windows=3, window_frac=0.5, step_frac=0.5,
                   N0=100, Nb=100, Nf=20000, n_overlap=50, nIter=10000, LBFGS_max_iter=50000,
                   train=True, model_name="sw_{win}.pth", log_name="sw_{win}.txt", rng=None):
    """Sliding-time-window training / evaluation, ``7kdv/sw-73.py:247-376``.

    Windows ``[t_min, t_max]`` of length ``window_frac * (t_end - t_start)`` advance by
    ``step_frac`` of a window.  Window 1 takes its initial condition from the data row 0; later windows
    from the previous window's model at ``t_min``, warm-start from its weights (``prev_model``) and add
    an overlap term on ``n_overlap`` random points of the common time range labelled by the previous
    model.  Each window's model predicts the whole grid; its values are kept inside its own time range
    (later windows overwrite the overlap), points no window covers are filled by the last model.
    Returns ``(models, full_u_pred, relative L2 error)``.  With ``train=False`` the per-window models
    are loaded from ``<base_dir>/model/<model_name>`` instead.

    Two-output nets (NLS/GNSS, ``NLS/sw-GNSS-adamW.py:262-404``): pass ``exact_u`` complex (or a
    ``(u, v)`` pair); initial, overlap and stitched fields are then pairs, and the function returns
    ``(models, (full_u, full_v, full_h), (err_u, err_v, err_h))`` with ``h = sqrt(u^2 + v^2)``."""
    rng = np.random.default_rng(1234) if rng is None else rng
    X_star = grid_points(x, t)
    if isinstance(exact_u, (tuple, list)):
        fields = [np.asarray(exact_u[0]), np.asarray(exact_u[1])]
    elif np.iscomplexobj(exact_u):
        fields = [np.real(exact_u), np.imag(exact_u)]
    else:
        fields = [np.asarray(exact_u)]
    two = len(fields) == 2
    as_list = (lambda r: list(r)) if two else (lambda r: [r])
    stars = [f.flatten()[:, None] for f in fields]
    lb, ub = X_star.min(0), X_star.max(0)
    window = (ub[1] - lb[1]) * window_frac
    step = window * step_frac

    models, prev_model = [], None
    full = [np.zeros_like(s_) for s_ in stars]
    done = np.zeros(len(X_star), dtype=bool)
    for win in range(num_windows):
        t_min = lb[1] + win * step
        t_max = min(t_min + window, ub[1])
        win_lb, win_ub = [lb[0], t_min], [ub[0], t_max]
        name = model_name.format(win=win + 1)
        if train:
            idx = rng.choice(len(x), N0, replace=False)
            x0 = x[idx]
            prev_vals = []
            X_prev = None
            if win == 0:
                ic = [f[0, :].reshape(-1, 1)[idx] for f in fields]
            else:
                ic = [r[idx] for r in as_list(prev_model.predict(np.hstack((x, np.ones_like(x) * t_min))))]
                prev_t_min = lb[1] + (win - 1) * step
                prev_t_max = min(prev_t_min + window, ub[1])
                o_min, o_max = max(prev_t_min, t_min), min(prev_t_max, t_max)
                if o_max > o_min:
                    X_prev = np.zeros((n_overlap, 2))
                    X_prev[:, 0] = lb[0] + (ub[0] - lb[0]) * rng.random(n_overlap)
                    X_prev[:, 1] = o_min + (o_max - o_min) * rng.random(n_overlap)
                    prev_vals = as_list(prev_model.predict(X_prev))
            tb = t_min + (t_max - t_min) * lhs(1, Nb, rng)
            Xf = np.zeros((Nf, 2))
            Xf[:, 0] = win_lb[0] + (win_ub[0] - win_lb[0]) * lhs(1, Nf, rng).flatten()
            Xf[:, 1] = t_min + (t_max - t_min) * lhs(1, Nf, rng).flatten()
            model = model_cls(layers, win_lb, win_ub, prev_model)
            if X_prev is None:
                prev_vals = [None] * len(fields)
            model.set_training_data(x0, *ic, tb, Xf, X_prev, *prev_vals)
            model.train(nIter, log_name.format(win=win + 1) if log_name else None, LBFGS_max_iter=LBFGS_max_iter)
            model.save_model(name)
        else:
            model = model_cls(layers, win_lb, win_ub)
            model.load_model(name)
        prev_model = model
        models.append(model)
        preds = as_list(model.predict(X_star))
        mask = (X_star[:, 1] >= t_min) & (X_star[:, 1] <= t_max)
        for f_, p_ in zip(full, preds):
            f_[mask] = p_[mask]
        done[mask] = True
    if not np.all(done):
        for f_, p_ in zip(full, as_list(prev_model.predict(X_star[~done]))):
            f_[~done] = p_
    if not two:
        return models, full[0], relative_l2(full[0], stars[0])
    full_h = np.sqrt(full[0] ** 2 + full[1] ** 2)
    h_star = np.sqrt(stars[0] ** 2 + stars[1] ** 2)
    return models, (full[0], full[1], full_h), (relative_l2(full[0], stars[0]), relative_l2(full[1], stars[1]),
                                                 relative_l2(full_h, h_star))


def backward_compatible(pi_cls, bc_cls, layers, x, t, exact_u, *, step=50, N0=100, Nf=20000, nIter=5000,
                        LBFGS_max_iter=50000, model_name="model_segment_{seg}.pth",
                        log_name="log-bc{seg}.txt", rng=None, corrected=False):
    """Backward-compatible (bc-PINN) training over consecutive time segments,
    ``CH/bc-CH-1soliton.py:378-473`` (the same loop is commented out in ``7kdv/bc-7kdv*.py``, ``5kdv/bc-5*.py``).

    The time grid is cut into segments of ``step`` rows (one row of overlap).  Segment 0 trains the
    plain PI class on ``[t_0, t_seg_end]`` from the data's initial condition.  Every later segment
    trains the bc class on the domain grown to its right edge, with the retention term ``loss1`` over the
    whole history grid ``Xh`` (all rows up to the previous right edge) labelled by the previous model.
    Returns ``(models, u1)`` with ``u1`` the last model's prediction on the final history grid.

    By default every argument handed to the classes is what the reference's loop hands over
    (tests/golden/bc_driver_args.npz: a replay of ``:334-473`` on synthetic data; tests/test_drivers_cpu.py),
    including three things that look unintended:
      (i)   later segments get the segment's first ROW INDEX as ``t0`` (``:447-448``) although the argument is a
            time (50, 100, 150 on a grid whose times run from -1 to 1);
      (ii)  and ``u0 = u1[-N0:]`` (``:446``), the last N0 grid values of the history prediction, paired with the
            random ``x0`` of segment 0;
      (iii) the loop ends where the reference's run ends: at the first segment with fewer than ``step/2`` time rows
            (``np.random.choice(rows, Nb, replace=False)`` raises there: after four segments on a 201-row grid).
    ``corrected=True`` instead passes the TIME of the segment's first row as ``t0``, the previous model's values at
    ``x0`` on that row as ``u0``, and trains a short last segment on the rows it has.
    ``rng``: a ``numpy.random.Generator``, or the ``numpy.random`` module itself for the reference's global-state
    draws (after ``np.random.seed(1234)``, ``:21``); draws are made in the reference's order."""
    rng = np.random.default_rng(1234) if rng is None else rng
    nt = t.shape[0]
    steps_lb = np.arange(0, nt + step, step)
    steps_ub = 1 + steps_lb
    idx = rng.choice(x.shape[0], N0, replace=False)
    x0 = x[idx, :]
    u0 = exact_u[0:1, :].T[idx, :]
    Nb = int(step / 2)
    models, u1, Xh = [], None, None
    x_lo, x_hi = float(x.min()), float(x.max())
    t_start = float(t.min())
    for seg in range(steps_lb.size - 1):
        t1, t2 = int(steps_lb[seg]), int(steps_ub[seg + 1])
        temp_t = t[:t2, :]
        t_seg = t[t1:t2, :]
        rows = t_seg.shape[0]
        if rows < (2 if corrected else Nb):
            break
        tb = t_seg[rng.choice(rows, min(Nb, rows), replace=False), :]
        lb = np.array([x_lo, t_start])
        ub = np.array([x_hi, float(np.max(t_seg))])
        seg_lb = np.array([x_lo, float(np.min(t_seg))])
        seg_ub = np.array([x_hi, float(np.max(t_seg))])
        Xf = seg_lb + (seg_ub - seg_lb) * lhs(2, Nf, rng)
        if seg == 0:
            model = pi_cls(layers, lb, ub)
            model.set_training_data(x0, u0, tb, Xf)
        else:
            if corrected:
                u0 = u1[-x.shape[0]:][idx, :]      # previous model on the segment's first row (= last history row)
                t0 = float(t[t1, 0])
            else:
                u0, t0 = u1[-N0:], t1
            model = bc_cls(layers, lb, ub)
            model.set_training_data(x0, u0, t0, u1, tb, Xf, Xh)
        model.train(nIter, log_name.format(seg=seg) if log_name else None, LBFGS_max_iter=LBFGS_max_iter)
        Xh = grid_points(x, temp_t)
        u1 = model.predict(Xh)
        model.save_model(model_name.format(seg=seg))
        models.append(model)
    return models, u1
