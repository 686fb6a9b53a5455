"""Minimal profiling target: a few launches of the fused loss+grad kernel on the
bench workload (no optimizer, no CPU baseline).  Used under rocprofv3."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, checkpoint_theta  # noqa: E402
from pde_amd import equations  # noqa: E402
from pde_amd._native import NativeEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="kdv7")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--kernel", type=int, default=0)
ap.add_argument("--lib", default="", help="alternative libpinnjet.so (development builds)")
ap.add_argument("--check", action="store_true", help="compare loss/gradient with the VALU kernels of the same library")
args = ap.parse_args()
if args.lib:
    import pde_amd._native as _nat
    _nat.LIB_PATH = os.path.abspath(args.lib)
w = WORKLOADS[args.workload]
n = args.n or w["n"]
dev = torch.device("cuda", 0)
lb, ub = np.array(w["lb"], dtype=np.float32), np.array(w["ub"], dtype=np.float32)
eng = NativeEngine(equations.get(w["eq"]), w["layers"], lb, ub, dev, kernel=args.kernel)
rng = np.random.default_rng(1234)
X = (lb + (ub - lb) * rng.random((n, 2))).astype(np.float32)
layers = w["layers"]
theta = np.zeros(eng.P, dtype=np.float32)
off = 0
for i in range(len(layers) - 1):
    fi, fo = layers[i], layers[i + 1]
    theta[off:off + fi * fo] = rng.standard_normal(fi * fo) * np.sqrt(2.0 / (fi + fo))
    off += fi * fo + fo
ck = checkpoint_theta(w)          # the weights bench.py uses by default (the reference's shipped checkpoint, if any)
if ck is not None:
    theta = ck
th = torch.tensor(theta, device=dev)
x = torch.tensor(X[:, 0].copy(), device=dev)
t = torch.tensor(X[:, 1].copy(), device=dev)
buf = torch.zeros(eng.P + eng.nf, device=dev)
if args.check:
    nc = min(n, 20000)
    ref = NativeEngine(equations.get(w["eq"]), w["layers"], lb, ub, dev, kernel=1)
    b0, b1 = torch.zeros_like(buf), torch.zeros_like(buf)
    eng.loss_grad(th, x[:nc], t[:nc], 1.0 / nc, b0[eng.P:], b0[:eng.P])
    ref.loss_grad(th, x[:nc], t[:nc], 1.0 / nc, b1[eng.P:], b1[:eng.P])
    torch.cuda.synchronize()
    gerr = float((b0[:eng.P] - b1[:eng.P]).norm() / b1[:eng.P].norm())
    lerr = float(((b0[eng.P:] - b1[eng.P:]).abs() / b1[eng.P:].abs()).max())
    print("check vs valu kernels (n=%d): loss rel %.2e grad rel %.2e %s" % (nc, lerr, gerr, "OK" if gerr < 1e-4 and lerr < 1e-4 else "MISMATCH"))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(args.iters):
    ev0.record()
    eng.loss_grad(th, x, t, 1.0 / n, buf[eng.P:], buf[:eng.P])
    ev1.record()
    torch.cuda.synchronize()
    print("iter %d: %.3f ms  -> %.3e evals/s" % (it, ev0.elapsed_time(ev1), n / (ev0.elapsed_time(ev1) * 1e-3)))
