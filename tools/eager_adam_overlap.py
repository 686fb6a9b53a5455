"""Eager Adam steps/s (FusedAdam, direct closure, no graphs) with the collocation kernel on a second stream or not."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.lbfgs_bench as lb  # noqa: E402
from pde_amd import scripts  # noqa: E402

for case, n in (("burgers", 10000), ("burgers_9x20", 10000), ("kdv7", 20000), ("gnss", 20000)):
    row = []
    for ov in (False, True):
        c = lb.CASES[case]
        torch.manual_seed(1)
        base = getattr(scripts, c["family"]).PI.PhysicsInformedNN
        cls = type("M", (base,), {"overlap_streams": ov})
        lbv, ubv = np.array(c["lb"]), np.array(c["ub"])
        m = cls(c["layers"], lbv, ubv)
        rng = np.random.default_rng(1)
        x0 = rng.uniform(lbv[0], ubv[0], (50, 1)); u0 = 1.0 / np.cosh(x0) ** 2
        tb = rng.uniform(lbv[1], ubv[1], (50, 1)); Xf = lbv + (ubv - lbv) * rng.random((n, 2))
        if case == "gnss":
            m.set_training_data(x0, u0, 0.1 * u0, tb, Xf)
        elif case.startswith("burgers"):
            m.set_training_data(x0, -np.sin(np.pi * x0), tb, Xf, 0.01 / np.pi)
        else:
            m.set_training_data(x0, u0, tb, Xf)
        with contextlib.redirect_stdout(io.StringIO()):
            m.train(30, LBFGS_max_iter=0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.train(400, LBFGS_max_iter=0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        row.append("%s %.0f" % ("second stream" if ov else "one stream", 401 / dt))
    print(case, n, " | ".join(row), "eager Adam steps/s", flush=True)
