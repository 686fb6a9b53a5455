"""Where the host time of an L-BFGS iteration goes (device + graph, the scripts' sizes): inside the wait for the
statistics (poll_advance: the wait, the decision and the next direction's launch in C), in Python between its return and
the next closure's enqueue, and in that enqueue (graph replay + the statistics launch)."""
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.lbfgs_bench as lb  # noqa: E402
from pde_amd import lbfgs  # noqa: E402

acc = {"read": 0.0, "n_read": 0, "py": 0.0, "dir": 0.0, "closure": 0.0, "t_after_read": None}
_read2, _eval = lbfgs.LBFGS._read2, lbfgs.LBFGS._evaluate_async


def read2(self, a, b):
    t0 = time.perf_counter()
    r = _read2(self, a, b)
    t1 = time.perf_counter()
    acc["read"] += t1 - t0
    acc["n_read"] += 1
    acc["t_after_read"] = t1
    return r


def evaluate_async(self, closure, x0, t, moved=False):
    t0 = time.perf_counter()
    if acc["t_after_read"] is not None:
        acc["py"] += t0 - acc["t_after_read"]
        acc["t_after_read"] = None
    r = _eval(self, closure, x0, t, moved)
    acc["closure"] += time.perf_counter() - t0
    return r


lbfgs.LBFGS._read2, lbfgs.LBFGS._evaluate_async = read2, evaluate_async
from pde_amd import _native  # noqa: E402
_dir = _native.NativeLBFGSKernels.direction


def direction(self, *a, **k):
    t0 = time.perf_counter()
    if acc["t_after_read"] is not None:
        acc["py"] += t0 - acc["t_after_read"]
        acc["t_after_read"] = None
    r = _dir(self, *a, **k)
    acc["dir"] += time.perf_counter() - t0
    return r


_native.NativeLBFGSKernels.direction = direction
_pa = _native.NativeLBFGSKernels.poll_advance


def poll_advance(self, *a, **k):
    t0 = time.perf_counter()
    r = _pa(self, *a, **k)
    t1 = time.perf_counter()
    acc["read"] += t1 - t0
    acc["n_read"] += 1
    acc["t_after_read"] = t1 if r is not None and r[0] else None
    return r


_native.NativeLBFGSKernels.poll_advance = poll_advance
case = sys.argv[1] if len(sys.argv) > 1 else "kdv7"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
lb.run(case, "device", n, 30, graph=True)          # loads the code objects, warms the allocator
for key in ("read", "n_read", "py", "dir", "closure"):
    acc[key] = 0
_warm = lb.run


def run_timed():
    import numpy as np
    # (lb.run itself makes 3 warm-up iterations before its timed part: they are in the averages, 1 %)
    return _warm(case, "device", n, 300, graph=True)


r = run_timed()
k = acc["n_read"]
print("%s n=%d: %.0f evals/s; per iteration (us): wait + decision + next direction's launch (C) %.1f, python between that and "
      "the closure's enqueue %.1f, direction calls from python %.1f, closure + statistics enqueue %.1f" % (case, n, r["evals_per_s"], acc["read"] / k * 1e6,
                                                                   acc["py"] / k * 1e6, acc["dir"] / k * 1e6, acc["closure"] / k * 1e6))
