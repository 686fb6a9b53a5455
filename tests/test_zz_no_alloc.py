"""GPU: pinn_forward_vjp_static — the entry point that allocates nothing (include/pinnjet.h) — against
pinn_forward_vjp, through the C-ABI, on two calls in a row, the second with a smaller input (the
static scratch keeps the first call's contents).  Same kernels, same grid rule up to what fits the
static array, so the bar is the bar of the existing entry point's own parity test (1e-5 relative to
the fp64 oracle) plus agreement of the two entry points to float32 summation order (2e-6)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.pyoracle import Oracle
from pde_amd import equations
from pde_amd._native import KERNEL_MFMA, KERNEL_VALU, NativeEngine, NativeError

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def cu(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=torch.device("cuda", 0))


@pytest.mark.parametrize("name,kernel", [("kdv7_3sol_ckpt", KERNEL_MFMA), ("gnss_w100_ckpt", KERNEL_MFMA),
                                         ("burgers_9x20_xavier", KERNEL_MFMA), ("kdv7_3sol_ckpt", KERNEL_VALU)])
def test_static_entry_matches_forward_vjp(name, kernel):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    dev = torch.device("cuda", 0)
    eng = NativeEngine(equations.get("none"), g["layers"].tolist(), g["lb"].astype(np.float32),
                       g["ub"].astype(np.float32), dev, kernel=kernel)
    o = Oracle("none", g["layers"].tolist(), g["lb"], g["ub"])
    theta = cu(g["theta"])
    rng = np.random.default_rng(21)
    for n in (5000, 77):                         # two calls in a row, the second smaller
        X = (g["lb"] + (g["ub"] - g["lb"]) * rng.random((n, 2))).astype(np.float32)
        gb = rng.standard_normal((n, eng.nout)).astype(np.float32)
        Xd, gbd = cu(X), cu(gb)
        want = torch.zeros(eng.P, device=dev)
        eng.forward_vjp(theta, Xd, None, gbd, want)
        got = torch.full((eng.P,), 1.0, device=dev)          # accumulate semantics: += on top of what is there
        eng.forward_vjp_static(theta, Xd, None, gbd, got)
        torch.cuda.synchronize()
        got = got.cpu().numpy().astype(np.float64) - 1.0
        _, gref = o.loss_grad(g["theta"].astype(np.float64), X[:, 0], X[:, 1], gbar=gb.astype(np.float64))
        assert rel(got, gref) < 1e-5, n
        assert rel(got, want.cpu().numpy().astype(np.float64)) < 2e-6, n


def test_static_entry_rejects_what_does_not_fit():
    g = np.load(os.path.join(GOLDEN, "kdv7_3sol_ckpt.npz"))
    dev = torch.device("cuda", 0)
    eng = NativeEngine(equations.get("none"), g["layers"].tolist(), g["lb"].astype(np.float32),
                       g["ub"].astype(np.float32), dev)
    n = 65536 + 1
    X = torch.zeros((n, 2), device=dev)
    gb = torch.zeros((n, 1), device=dev)
    out = torch.zeros(eng.P, device=dev)
    with pytest.raises(NativeError, match="too large"):
        eng.forward_vjp_static(cu(g["theta"]), X, None, gb, out)
    assert float(out.abs().sum()) == 0.0
    # the largest accepted size runs (grid capped to the static array) and matches the allocating entry point
    n = 65536
    rng = np.random.default_rng(3)
    X = cu((g["lb"] + (g["ub"] - g["lb"]) * rng.random((n, 2))).astype(np.float32))
    gb = cu(rng.standard_normal((n, 1)).astype(np.float32))
    a = torch.zeros(eng.P, device=dev)
    b = torch.zeros(eng.P, device=dev)
    eng.forward_vjp(cu(g["theta"]), X, None, gb, a)
    eng.forward_vjp_static(cu(g["theta"]), X, None, gb, b)
    torch.cuda.synchronize()
    assert rel(b.cpu().numpy(), a.cpu().numpy().astype(np.float64)) < 2e-6
