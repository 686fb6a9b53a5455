"""Latency of the data-term launches (net_u forward and vjp on ~150 points: IC + two boundary blocks) for both kernel
families, back to back on one stream: microseconds per launch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import equations  # noqa: E402
from pde_amd._native import KERNEL_MFMA, KERNEL_VALU, NativeEngine  # noqa: E402

dev = torch.device("cuda", 0)
for layers in ([2, 50, 50, 50, 50, 1], [2, 100, 100, 100, 100, 2]):
    lb, ub = np.array([-1.0, 0.0], dtype=np.float32), np.array([1.0, 1.0], dtype=np.float32)
    P = sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))
    th = torch.randn(P, device=dev) * 0.2
    for n in (150, 300, 1000):
        X = torch.rand(n, 2, device=dev)
        row = []
        for name, k in (("mfma", KERNEL_MFMA), ("valu", KERNEL_VALU)):
            eng = NativeEngine(equations.get("none"), layers, lb, ub, dev, kernel=k)
            g = torch.zeros(P, device=dev)
            out = eng.forward(th, X, None)
            gbar = torch.ones_like(out)
            for _ in range(5):
                eng.forward(th, X, None)
                eng.forward_vjp(th, X, None, gbar, g)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                eng.forward(th, X, None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(200):
                eng.forward_vjp(th, X, None, gbar, g)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            row.append("%s fwd %.1f us vjp %.1f us" % (name, (t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
        print(layers[1], "n=%d:" % n, " | ".join(row), flush=True)
