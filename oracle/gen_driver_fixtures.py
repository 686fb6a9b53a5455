"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Container-only (needs /root/reference).

End-to-end pins for the drivers and the on-disk formats, produced by running the
reference's own classes (imported like oracle/gen_golden.py; only data is written):

  predict_sw73.npz   7kdv/sw-73.py test mode (:343-376): the three shipped window models
                     7kdv/model/sw73_{1,2,3}.pth stitched by time mask on the 201 x 512 grid
                     -> relative L2 error; error.txt:109-110 records 3.223475e-02
  predict_kdv5.npz   5kdv/PI-51.py (:262-267) with 5kdv/model/PINN1t-2.pth -> error.txt:4, 2.446954e-02
  ref_written_kdv7.pth  a checkpoint written by the reference's own save_model() (7kdv/PI-7kdv3soliton.py:173-177)
                     + ref_written_kdv7.npz with the same weights as a flat vector
  ref_logs.npz       what the reference's train(1, logfile, 0, 1) wrote to its log file for the PI, sw and bc
                     classes of the 7kdv scripts (text), with the weights and training data it ran on
Run:  python oracle/gen_driver_fixtures.py
"""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np
import scipy.io
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, REF, flat_theta, load_script  # noqa: E402

L50 = [2, 50, 50, 50, 50, 1]


def grid(data, key="uu", transpose=False):
    t = data["t"].flatten()[:, None]
    x = data["x"].flatten()[:, None]
    exact = np.real(data[key])
    if transpose:
        exact = exact.T
    X, T = np.meshgrid(x, t)
    X_star = np.hstack((X.flatten()[:, None], T.flatten()[:, None]))
    return t, x, exact, X_star, exact.flatten()[:, None]


def sw73():
    mod = load_script("7kdv/sw-73.py")
    data = scipy.io.loadmat(os.path.join(REF, "7kdv/data/73[-0.5,0.5]201*512.mat"))
    t, x, exact, X_star, u_star = grid(data)
    lb, ub = X_star.min(0), X_star.max(0)
    total = ub[1] - lb[1]
    window, step = total * 0.5, total * 0.25
    full = np.zeros_like(u_star)
    done = np.zeros(len(X_star), dtype=bool)
    thetas, bounds = [], []
    for win in range(3):
        t_min = lb[1] + win * step
        t_max = min(t_min + window, ub[1])
        model = mod.PhysicsInformedNN(L50, [lb[0], t_min], [ub[0], t_max])
        model.dnn.load_state_dict(torch.load(os.path.join(REF, "7kdv/model/sw73_%d.pth" % (win + 1)),
                                             map_location="cpu", weights_only=True))
        u_pred = model.predict(X_star)
        mask = (X_star[:, 1] >= t_min) & (X_star[:, 1] <= t_max)
        full[mask] = u_pred[mask]
        done[mask] = True
        thetas.append(flat_theta(model).astype(np.float32))
        bounds.append([lb[0], t_min, ub[0], t_max])
    assert done.all()
    err = np.linalg.norm(u_star - full, 2) / np.linalg.norm(u_star, 2)
    print("sw-73 stitched rel-L2 = %e (error.txt:110 says 3.223475e-02)" % err)
    sel = np.arange(0, X_star.shape[0], 97)
    np.savez_compressed(os.path.join(OUT, "predict_sw73.npz"), thetas=np.stack(thetas), bounds=np.array(bounds),
                        layers=np.array(L50), t=t, x=x, uu=exact, err_ref=np.array(err), sel=sel,
                        u_full_sel=full[sel, 0].astype(np.float32),
                        source=np.array("7kdv/sw-73.py:343-376 test mode + 7kdv/model/sw73_{1,2,3}.pth + 7kdv/data/73[-0.5,0.5]201*512.mat"))


def kdv5():
    mod = load_script("5kdv/PI-51.py")
    data = scipy.io.loadmat(os.path.join(REF, "5kdv/data/51[-2,2]201*512.mat"))
    t, x, exact, X_star, u_star = grid(data)
    lb, ub = X_star.min(0), X_star.max(0)
    mod.lb, mod.ub = lb, ub
    model = mod.PhysicsInformedNN(L50, lb, ub)
    model.dnn.load_state_dict(torch.load(os.path.join(REF, "5kdv/model/PINN1t-2.pth"), map_location="cpu", weights_only=True))
    u_pred = model.predict(X_star)
    err = np.linalg.norm(u_star - u_pred, 2) / np.linalg.norm(u_star, 2)
    print("5kdv PINN1t-2 rel-L2 = %e (error.txt:4 says 2.446954e-02)" % err)
    np.savez_compressed(os.path.join(OUT, "predict_kdv5.npz"), theta=flat_theta(model).astype(np.float32),
                        layers=np.array(L50), lb=lb, ub=ub, t=t, x=x, uu=exact, err_ref=np.array(err),
                        source=np.array("5kdv/PI-51.py:262-267 + 5kdv/model/PINN1t-2.pth + 5kdv/data/51[-2,2]201*512.mat"))


def written_checkpoint(tmp):
    mod = load_script("7kdv/PI-7kdv3soliton.py")
    mod.current_dir = tmp                       # the reference writes to <current_dir>/model/<name>
    os.makedirs(os.path.join(tmp, "model"), exist_ok=True)
    lb, ub = np.array([-15.0, -0.5]), np.array([15.0, 0.5])
    torch.manual_seed(777)
    model = mod.PhysicsInformedNN(L50, lb, ub)
    model.save_model("ref_written_kdv7.pth")
    shutil.copy(os.path.join(tmp, "model", "ref_written_kdv7.pth"), os.path.join(OUT, "ref_written_kdv7.pth"))
    X = np.stack([np.linspace(-15, 15, 33), np.linspace(-0.5, 0.5, 33)], axis=1)
    np.savez_compressed(os.path.join(OUT, "ref_written_kdv7.npz"), theta=flat_theta(model).astype(np.float32),
                        layers=np.array(L50), lb=lb, ub=ub, X=X, u=model.predict(X).astype(np.float32),
                        keys=np.array(list(model.dnn.state_dict().keys())),
                        source=np.array("7kdv/PI-7kdv3soliton.py:173-177 save_model(), xavier(manual_seed 777)"))


def logs(tmp):
    os.makedirs(os.path.join(tmp, "log"), exist_ok=True)
    rng = np.random.default_rng(31)
    out = {}
    lb, ub = np.array([-15.0, -0.5]), np.array([15.0, 0.5])
    n0, nb, nf = 24, 16, 64
    x0 = lb[0] + (ub[0] - lb[0]) * rng.random((n0, 1))
    u0 = 2.0 / np.cosh(x0) ** 2
    tb = lb[1] + (ub[1] - lb[1]) * rng.random((nb, 1))
    Xf = lb + (ub - lb) * rng.random((nf, 2))
    Xp = lb + (ub - lb) * rng.random((12, 2))
    up = 1.5 / np.cosh(Xp[:, 0:1]) ** 2
    Xh = lb + (ub - lb) * rng.random((20, 2))
    u1 = 1.2 / np.cosh(Xh[:, 0:1] - Xh[:, 1:2]) ** 2
    out.update(x0=x0, u0=u0, tb=tb, Xf=Xf, X_prev=Xp, u_prev=up, Xh=Xh, u1=u1, lb=lb, ub=ub, layers=np.array(L50))
    for tag, script, cname, args in (
            ("pi", "7kdv/PI-7kdv3soliton.py", "PhysicsInformedNN", (x0, u0, tb, Xf)),
            ("sw", "7kdv/sw-73.py", "PhysicsInformedNN", (x0, u0, tb, Xf, Xp, up)),
            ("bc", "7kdv/bc-7kdv3soliton.py", "bcPhysicsInformedNN", (x0, u0, -0.5, u1, tb, Xf, Xh))):
        mod = load_script(script)
        mod.current_dir = tmp
        mod.lb, mod.ub = lb, ub
        torch.manual_seed(5)
        model = getattr(mod, cname)(L50, lb, ub)
        out["theta_" + tag] = flat_theta(model).astype(np.float32)
        model.set_training_data(*args)
        model.train(1, "ref_%s.txt" % tag, 0, 1)          # one Adam iteration, one L-BFGS iteration
        out["log_" + tag] = np.array(open(os.path.join(tmp, "log", "ref_%s.txt" % tag), encoding="utf-8").read())
        print(tag, repr(str(out["log_" + tag])))
    # the first lines of a shipped log, for the format of a long run
    out["shipped_log_head"] = np.array("".join(open(os.path.join(REF, "7kdv/log/log-PI 73.txt"), encoding="utf-8").readlines()[:3]))
    np.savez_compressed(os.path.join(OUT, "ref_logs.npz"),
                        source=np.array("train(1, logfile, 0, 1) of 7kdv/PI-7kdv3soliton.py, 7kdv/sw-73.py, "
                                        "7kdv/bc-7kdv3soliton.py (bc class), xavier(manual_seed 5); 7kdv/log/log-PI 73.txt:1-3"),
                        **out)


if __name__ == "__main__":
    tmp = tempfile.mkdtemp()
    try:
        sw73()
        kdv5()
        written_checkpoint(tmp)
        logs(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
