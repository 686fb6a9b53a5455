"""Fused loss+gradient launch (kernel + slab reduction) at the scripts' own collocation sizes (N_f = 10 000 - 20 000)
and below: microseconds per launch vs N and the FP32-roofline fraction that corresponds to, for the nets of the
BASELINE configs.  Back-to-back launches on one stream, wall time / launches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FP32_PEAK_TFLOPS, WORKLOADS, algorithmic_flops  # noqa: E402
from pde_amd import equations  # noqa: E402
from pde_amd._native import NativeEngine  # noqa: E402

dev = torch.device("cuda", 0)
SIZES = (16, 1024, 5000, 10000, 20000, 40000, 100000)
print("%-14s " % "workload" + " ".join("%16s" % ("n=%d" % n) for n in SIZES) + "    (us per launch / fraction of the FP32 roofline)")
for name in ("burgers", "burgers_9x20", "kdv5", "kdv7", "ch", "gnss"):
    w = WORKLOADS[name]
    layers = w["layers"]
    lb, ub = np.array(w["lb"], dtype=np.float32), np.array(w["ub"], dtype=np.float32)
    P = sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))
    th = torch.randn(P, device=dev) * 0.2
    eng = NativeEngine(equations.get(w["eq"]), layers, lb, ub, dev, kernel=2)
    A = algorithmic_flops(layers, w["eq"])
    row = []
    for n in SIZES:
        X = torch.rand(n, 2, device=dev)
        x = (lb[0] + (ub[0] - lb[0]) * X[:, 0]).contiguous()
        t = (lb[1] + (ub[1] - lb[1]) * X[:, 1]).contiguous()
        buf = torch.zeros(P + eng.nf, device=dev)
        for _ in range(5):
            eng.loss_grad(th, x, t, 1.0 / n, buf[P:], buf[:P])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            eng.loss_grad(th, x, t, 1.0 / n, buf[P:], buf[:P])
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 40 * 1e6
        row.append("%8.0f / %5.3f" % (us, A * n / (us * 1e-6) / 1e12 / FP32_PEAK_TFLOPS))
    print("%-14s " % name + " ".join("%16s" % r for r in row), flush=True)
