"""Fused loss+gradient kernel at the scripts' own collocation sizes (N_f = 20 000) and below: time per
launch vs N and vs the number of CUs a small launch leaves free (PINN_SPARE_CUS values as arguments; default 32),
7kdv [2,50x4,1] and GNSS [2,100x4,2]."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import equations  # noqa: E402
from pde_amd._native import NativeEngine  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [("kdv7_3sol", [2, 50, 50, 50, 50, 1], [-15.0, -0.5], [15.0, 0.5]),
         ("gnss", [2, 100, 100, 100, 100, 2], [-5.0, -2.0], [5.0, 0.0])]
for eq, layers, lb, ub in CASES:
    lb, ub = np.array(lb, dtype=np.float32), np.array(ub, dtype=np.float32)
    P = sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))
    th = torch.randn(P, device=dev) * 0.2
    for grid in (sys.argv[1:] or ["default"]):
        if grid != "default":
            os.environ["PINN_SPARE_CUS"] = grid
        else:
            os.environ.pop("PINN_SPARE_CUS", None)
        eng = NativeEngine(equations.get(eq), layers, lb, ub, dev, kernel=2)
        row = []
        for n in (16, 1024, 5000, 20000, 40000, 100000):
            X = torch.rand(n, 2, device=dev)
            x = (lb[0] + (ub[0] - lb[0]) * X[:, 0]).contiguous()
            t = (lb[1] + (ub[1] - lb[1]) * X[:, 1]).contiguous()
            buf = torch.zeros(P + eng.nf, device=dev)
            for _ in range(3):
                eng.loss_grad(th, x, t, 1.0 / n, buf[P:], buf[:P])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                eng.loss_grad(th, x, t, 1.0 / n, buf[P:], buf[:P])
            torch.cuda.synchronize()
            row.append("n=%d %.0f us" % (n, (time.perf_counter() - t0) / 30 * 1e6))
        print(eq, "spare CUs", grid, " | ".join(row), flush=True)
