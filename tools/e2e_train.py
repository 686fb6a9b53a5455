"""End-to-end: the reference's own training runs, with the reference's hyper-parameters, on the HIP path — how long do
they take on one MI355X and what relative L2 error do they reach, next to what the reference logged.

    python tools/e2e_train.py [case ...]        cases: burgers kdv5_pi kdv7_pi kdv7_sw   (default: all)

Each case follows its script's `__main__` (sampling with numpy's global state seeded 1234 as the scripts do, `lhs`
from pde_amd.drivers, N0 / Nb / N_f / layers / Adam iterations / L-BFGS settings from the script) on the data grid the
script loads; the grids travel as the `predict_*.npz` fixtures (tests/golden: x, t, uu of the reference's .mat files),
because /root/reference does not exist on the GPU box.  Weight initialisation is unseeded in the reference, so its
error.txt numbers are single draws; the comparison is "the same kind of accuracy", not digits.

    burgers   Burgers/Burgers_equation(PINN).py:187-258   [2,50x4,1]  N_f=10 000  50 000 Adam + L-BFGS
    kdv5_pi   5kdv/PI-51.py:195-262                       [2,50x4,1]  N_f=20 000  50 000 Adam + L-BFGS   error.txt:4   2.446954e-02
    kdv7_pi   7kdv/PI-7kdv3soliton.py:187-258             [2,50x4,1]  N_f=20 000   5 000 Adam + L-BFGS   error.txt:102 3.743270e-01 (4245.6 s)
    kdv7_sw   7kdv/sw-73.py:233-376                       3 windows x (10 000 Adam + L-BFGS)            error.txt:110 3.223475e-02 (~3 x 9 780 s)
One JSON line per case.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pde_amd import drivers, scripts  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {
    "burgers": dict(script="Burgers/Burgers_equation(PINN).py", data="predict_burgers", N0=50, Nb=50, Nf=10000, nIter=50000,
                    ref="no error.txt entry; the shipped checkpoint gives 3.153112e-03; log stops at iter 14 400"),
    "kdv5_pi": dict(script="5kdv/PI-51.py", data="predict_kdv5", N0=50, Nb=50, Nf=20000, nIter=50000,
                    ref="error.txt:2-5: 6.17e-02 / 1.59e-02 / 2.446954e-02 (t in [-2,2]) / 2.97e-02; 5kdv/log: 4 824.9 s for 50 000 closures"),
    "kdv7_pi": dict(script="7kdv/PI-7kdv3soliton.py", data="predict_kdv7_73", N0=50, Nb=50, Nf=20000, nIter=5000,
                    ref="error.txt:102: 3.743270e-01; 7kdv/log/log-PI 73.txt: 4 245.6 s"),
    "kdv7_sw": dict(script="7kdv/sw-73.py", data="predict_sw73", N0=100, Nb=100, Nf=20000, nIter=10000,
                    ref="error.txt:110: 3.223475e-02; 7kdv/log/sw73adamW_1.txt: 9 780 s for ONE window"),
}


def run(name, graph=True):
    c = CASES[name]
    g = np.load(os.path.join(GOLD, c["data"] + ".npz"))
    x, t, exact = g["x"], g["t"], g["uu"]
    layers = [2, 50, 50, 50, 50, 1]
    base = tempfile.mkdtemp(prefix="e2e_%s_" % name)
    ns = scripts.for_script(c["script"])
    cls = type("E2E", (ns.PhysicsInformedNN,), {"base_dir": base, "use_hip_graph": graph})
    np.random.seed(1234)                  # the scripts' np.random.seed(1234) (:21); torch is unseeded there
    torch.manual_seed(1234)
    t0 = time.time()
    if True:          # (the classes print the reference's log line every 100 closures: that is the progress output)
        if name == "kdv7_sw":
            models, full, err = drivers.sliding_window(cls, layers, x, t, exact, N0=c["N0"], Nb=c["Nb"], Nf=c["Nf"],
                                                       nIter=c["nIter"], rng=np.random)
            closures = sum(m.iter for m in models)
        else:
            X_star = drivers.grid_points(x, t)
            lb, ub = X_star.min(0), X_star.max(0)
            idx = np.random.choice(x.shape[0], c["N0"], replace=False)
            x0, u0 = x[idx, :], exact[0:1, :].T[idx, :]
            tb = t[np.random.choice(t.shape[0], c["Nb"], replace=False), :]
            Xf = lb + (ub - lb) * drivers.lhs(2, c["Nf"], np.random)
            m = cls(layers, lb, ub)
            if name == "burgers":
                m.set_training_data(x0, u0, tb, Xf, 0.01 / np.pi)
            else:
                m.set_training_data(x0, u0, tb, Xf)
            m.train(c["nIter"], "log.txt")
            err = drivers.relative_l2(m.predict(X_star), exact.flatten()[:, None])
            closures = m.iter
    torch.cuda.synchronize()
    wall = time.time() - t0
    lines = []
    logdir = os.path.join(base, "log")
    for fn in sorted(os.listdir(logdir)) if os.path.isdir(logdir) else []:
        lines += [ln.strip() for ln in open(os.path.join(logdir, fn)) if ln.startswith("Iter:")]
    return {"case": name, "script": c["script"], "hip_graph": graph, "relative_l2_error": err, "wall_s": wall,
            "closures": closures, "closures_per_s": closures / wall, "last_log_line": lines[-1] if lines else "",
            "reference": c["ref"]}


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CASES)):
        print(json.dumps(run(name)), flush=True)
