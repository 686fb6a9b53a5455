"""Latency of the data-term calls (net_u forward / vjp at tens of points) for both kernel families."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import equations  # noqa: E402
from pde_amd._native import NativeEngine  # noqa: E402

dev = torch.device("cuda", 0)
layers = [2, 50, 50, 50, 50, 1]
lb, ub = np.array([-15.0, -0.5], dtype=np.float32), np.array([15.0, 0.5], dtype=np.float32)
P = sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))
th = torch.randn(P, device=dev) * 0.2
for kname, kernel in (("valu", 1), ("mfma", 2)):
    eng = NativeEngine(equations.get("none"), layers, lb, ub, dev, kernel=kernel)
    for n in (50, 128, 1024, 8192, 77312):
        X = torch.rand(n, 2, device=dev)
        x, t = X[:, 0].contiguous(), X[:, 1].contiguous()
        gbar = torch.randn(n, 1, device=dev)
        grad = torch.zeros(P, device=dev)
        for _ in range(3):
            eng.forward(th, x, t); eng.forward_vjp(th, x, t, gbar, grad)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.forward(th, x, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            eng.forward_vjp(th, x, t, gbar, grad)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s n=%6d  forward %.1f us  forward_vjp %.1f us" % (kname, n, (t1 - t0) / 50 * 1e6, (t2 - t1) / 50 * 1e6))
