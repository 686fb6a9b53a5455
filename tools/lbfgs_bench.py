"""L-BFGS phase of train() at the scripts' own sizes: pde_amd.lbfgs.LBFGS (device) vs torch.optim.LBFGS,
same model, same data, same closure.  Prints closure evaluations per second for both.

    python tools/lbfgs_bench.py [--iters 200] [--n 20000]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import scripts  # noqa: E402

CASES = {
    "burgers": dict(family="burgers", layers=[2, 50, 50, 50, 50, 1], lb=[-1.0, 0.0], ub=[1.0, 0.99]),
    "burgers_9x20": dict(family="burgers", layers=[2] + [20] * 8 + [1], lb=[-1.0, 0.0], ub=[1.0, 0.99]),
    "kdv5": dict(family="kdv5_1soliton", layers=[2, 50, 50, 50, 50, 1], lb=[-10.0, -2.0], ub=[10.0, 2.0]),
    "kdv7": dict(family="kdv7_3soliton", layers=[2, 50, 50, 50, 50, 1], lb=[-15.0, -0.5], ub=[15.0, 0.5]),
    "gnss": dict(family="gnss", layers=[2, 100, 100, 100, 100, 2], lb=[-5.0, -2.0], ub=[5.0, 0.0]),
    # CH/PI-CH-1soliton.py:145-154: lr 0.5, history 150 (the direction kernels' panel path)
    "ch": dict(family="camassa_holm", layers=[2, 50, 50, 50, 50, 1], lb=[-15.0, -1.0], ub=[15.0, 1.0]),
}


def run(case, impl, n, iters, graph=False):
    c = CASES[case]
    torch.manual_seed(1234)
    base = getattr(scripts, c["family"]).PI.PhysicsInformedNN
    ov = {"0": False, "1": True}.get(os.environ.get("PINN_OVERLAP", ""), None)      # eager stream overlap: default automatic
    cls = type("M", (base,), {"lbfgs_impl": impl, "use_hip_graph": graph, "overlap_streams": ov})
    lb, ub = np.array(c["lb"]), np.array(c["ub"])
    m = cls(c["layers"], lb, ub)
    rng = np.random.default_rng(1)
    x0 = rng.uniform(lb[0], ub[0], (50, 1))
    u0 = 1.0 / np.cosh(x0) ** 2
    tb = rng.uniform(lb[1], ub[1], (50, 1))
    Xf = lb + (ub - lb) * rng.random((n, 2))
    if case == "gnss":
        m.set_training_data(x0, u0, 0.1 * u0, tb, Xf)
    elif case.startswith("burgers"):
        m.set_training_data(x0, -np.sin(np.pi * x0), tb, Xf, 0.01 / np.pi)
    else:
        m.set_training_data(x0, u0, tb, Xf)
    m.variant.lbfgs["max_eval"] = 10 * iters
    m.variant.lbfgs["tolerance_grad"] = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        m.train(0, LBFGS_max_iter=3)            # warm-up: workspaces, allocator
        torch.cuda.synchronize()
        m.iter = 1
        t0 = time.perf_counter()
        m.train(0, LBFGS_max_iter=iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    opt = m.optimizer_LBFGS
    st = opt.state[opt._params[0]]
    loss = float(m.backward().item())
    return dict(impl=impl, seconds=dt, evals=m.iter - 2, iters=st["n_iter"], evals_per_s=(m.iter - 2) / dt,
                iters_per_s=st["n_iter"] / dt, final_loss=loss)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--cases", default="burgers,burgers_9x20,kdv5,kdv7,gnss")
    ap.add_argument("--only", default="", help="run just this variant: torch | device | device+graph (profiling)")
    a = ap.parse_args()
    for case in a.cases.split(","):
        if a.only:
            r1 = run(case, "torch" if a.only == "torch" else "device", a.n, a.iters, graph=a.only == "device+graph")
            print(json.dumps({"case": case, "n_f": a.n, a.only: r1}), flush=True)
            continue
        r = {impl: run(case, impl, a.n, a.iters) for impl in ("torch", "device")}
        r["device+graph"] = run(case, "device", a.n, a.iters, graph=True)
        print(json.dumps({"case": case, "n_f": a.n, **r,
                          "speedup_evals_per_s": r["device"]["evals_per_s"] / r["torch"]["evals_per_s"],
                          "speedup_graph_evals_per_s": r["device+graph"]["evals_per_s"] / r["torch"]["evals_per_s"]}),
              flush=True)
