"""GPU: the bc-PINN driver on the HIP path, and its retention term `loss1` at the reference's own size
(CH/bc-CH-1soliton.py:251-253,429-436: 512 x 151 = 77 312 history points) against the fp64 oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.pyoracle import Oracle
from oracle_engine import install
from pde_amd import drivers, equations, scripts
from pde_amd._native import KERNEL_MFMA, KERNEL_VALU, NativeEngine, data_loss, data_loss_cotangent

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("kname,kernel", [("mfma", KERNEL_MFMA), ("valu", KERNEL_VALU)])
def test_loss1_at_the_reference_size_against_oracle(kname, kernel):
    """loss1 = mean((net_u(Xh) - u1)^2) over the 77 312-point history grid of the last bc segment, as the closure
    computes it: pinn_forward -> pinn_data_loss -> pinn_data_loss_cotangent -> pinn_forward_vjp, each stage against
    the fp64 oracle at the shipped CH weights (1e-5)."""
    g = np.load(os.path.join(GOLDEN, "ch_ckpt.npz"))
    layers, lb, ub = g["layers"].tolist(), g["lb"], g["ub"]
    dev = torch.device("cuda", 0)
    x = np.linspace(-15.0, 15.0, 512)[:, None]
    t = np.linspace(-1.0, 1.0, 201)[:151, None]
    Xh = drivers.grid_points(x, t).astype(np.float32)
    assert Xh.shape == (77312, 2)
    u1 = (0.8 / np.cosh(0.7 * (Xh[:, 0:1].astype(np.float64) - 2.0 * Xh[:, 1:2]))).astype(np.float32)   # targets
    eng = NativeEngine(equations.get("none"), layers, lb.astype(np.float32), ub.astype(np.float32), dev, kernel=kernel)
    o = Oracle("none", layers, lb, ub)
    th64 = g["theta"].astype(np.float64)
    theta, Xd, tgt = torch.tensor(g["theta"], device=dev), torch.tensor(Xh, device=dev), torch.tensor(u1, device=dev)
    out = eng.forward(theta, Xd, None)
    der, _ = o.forward(th64, Xh[:, 0], Xh[:, 1])
    u64 = der[:, :, 0]
    assert rel(out.cpu().numpy(), u64) < 1e-5
    n = Xh.shape[0]
    losses = data_loss(out, tgt, n, 0, False)
    r64 = u64 - u1.astype(np.float64)
    l64 = float((r64 * r64).sum() / n)
    assert abs(losses[0].item() - l64) < 1e-5 * l64 and losses[1].item() == 0.0
    one = torch.ones((), device=dev)
    gbar = data_loss_cotangent(out, tgt, n, 0, False, one, one)
    assert rel(gbar.cpu().numpy(), 2.0 * r64 / n) < 1e-5
    grad = torch.zeros(eng.P, device=dev)
    eng.forward_vjp(theta, Xd, None, gbar, grad)
    _, g64 = o.loss_grad(th64, Xh[:, 0], Xh[:, 1], gbar=2.0 * r64 / n)
    torch.cuda.synchronize()
    assert rel(grad.cpu().numpy(), g64) < 1e-5


def test_backward_compatible_two_segments_on_the_hip_path(tmp_path):
    """drivers.backward_compatible with the drop-in classes of CH/bc-CH-1soliton.py on the GPU: two short segments
    (reference defaults: t0 = row index, u0 = u1[-N0:]), then the second segment's full closure (IC + periodic
    boundary + loss1 over the history grid + Camassa-Holm residual with kappa = 2) against the same class on the
    CPU oracle engine at the same weights and data."""
    ns = scripts.for_script("CH/bc-CH-1soliton.py")
    x = np.linspace(-15.0, 15.0, 64)[:, None]
    t = np.linspace(-1.0, 1.0, 21)[:, None]
    exact = 1.0 / np.cosh(0.5 * (x.T - 3.0 * t)) ** 2
    mk = lambda base: type(base.__name__, (base,), {"base_dir": str(tmp_path)})
    models, u1 = drivers.backward_compatible(mk(ns.PhysicsInformedNN), mk(ns.bcPhysicsInformedNN), [2, 50, 50, 50, 50, 1],
                                             x, t, exact, step=10, N0=20, Nf=3000, nIter=25, LBFGS_max_iter=4,
                                             log_name="bc{seg}.txt", rng=np.random.default_rng(3))
    assert len(models) == 2 and type(models[1]).__name__ == "bcPhysicsInformedNN"
    m = models[1]
    assert m.x1.shape[0] == 11 * 64 and float(m.t0[0, 0]) == 10.0 and u1.shape == (21 * 64, 1)
    assert np.isfinite(u1).all()
    for s in (0, 1):
        assert os.path.exists(os.path.join(str(tmp_path), "model", "model_segment_%d.pth" % s))
        log = open(os.path.join(str(tmp_path), "log", "bc%d.txt" % s)).read()
        assert "Iter: 0, Loss:" in log and "Training time:" in log and (("Loss_1:" in log) == (s == 1))
    # the closure of segment 1 against the oracle engine (fp32 C restatement) on the CPU
    ref = install(ns.bcPhysicsInformedNN)([2, 50, 50, 50, 50, 1], m._lb64, m._ub64)
    with torch.no_grad():
        ref.dnn.flat_theta().copy_(m.dnn.flat_theta().cpu())
    Xf = torch.cat([m.xf, m.tf], dim=1).cpu().numpy().astype(np.float64)
    Xh = torch.cat([m.x1, m.t1], dim=1).cpu().numpy().astype(np.float64)
    ref.set_training_data(m.x0.cpu().numpy(), m.u0.cpu().numpy(), 10, m.u1.cpu().numpy(), m.tb.cpu().numpy(), Xf, Xh)
    for mm in (m, ref):
        for p in mm.dnn.parameters():
            p.grad = None
        mm.iter = 1
    lg, lr = m.backward(), ref.backward()
    torch.cuda.synchronize()
    gg = torch.cat([p.grad.reshape(-1) for p in m.dnn.parameters()]).cpu().numpy()
    gr = torch.cat([p.grad.reshape(-1) for p in ref.dnn.parameters()]).numpy().astype(np.float64)
    assert abs(lg.item() - lr.item()) <= 2e-5 * abs(lr.item())
    assert rel(gg, gr) < 2e-4          # two float32 evaluations of a partly cancelling gradient
