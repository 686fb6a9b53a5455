"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Build-owned PyTorch restatement of the reference's *method*: a tanh MLP behind
a Scale layer, differentiated by nested torch.autograd.grad calls, written from
SURVEY.md 8(a) (not copied from the reference's files).  Two uses:

* second checker for the C oracle and the HIP path (it travels to the GPU box,
  the reference's own scripts cannot);
* bench.py's cpu_baseline leg: "the reference CPU PyTorch path" the north-star
  speed-up is quoted against, timed on the box's host cores.

Reference semantics followed:
  Scale        7kdv/PI-7kdv3soliton.py:34-41
  DNN          7kdv/PI-7kdv3soliton.py:43-61   (keys 1.layers.layer_i.{weight,bias})
  net_f 7kdv   7kdv/PI-7kdv3soliton.py:122-132, 7kdv/PI-7kdv1soliton.py:132
  net_f 5kdv   5kdv/PI-51.py:132-140
  net_f Burg.  Burgers/Burgers_equation(PINN).py:126-131
  net_f CH     CH/PI-CH-1soliton.py:131-138, CH/bc-CH-1soliton.py:270-278
  net_f GNSS   NLS/sw-GNSS-adamW.py:159-180
"""
from collections import OrderedDict

import numpy as np
import torch


class Scale(torch.nn.Module):
    def __init__(self, lb, ub, dtype=torch.float32):
        super().__init__()
        # the reference keeps lb/ub as float32; an fp64 evaluation promotes those
        self.lb = torch.tensor(np.asarray(lb, dtype=np.float32)).to(dtype)
        self.ub = torch.tensor(np.asarray(ub, dtype=np.float32)).to(dtype)

    def forward(self, X):
        return 2.0 * (X - self.lb) / (self.ub - self.lb) - 1.0


class DNN(torch.nn.Module):
    def __init__(self, layers):
        super().__init__()
        depth = len(layers) - 1
        mods = []
        for i in range(depth - 1):
            mods.append(("layer_%d" % i, torch.nn.Linear(layers[i], layers[i + 1])))
            mods.append(("activation_%d" % i, torch.nn.Tanh()))
        mods.append(("layer_%d" % (depth - 1), torch.nn.Linear(layers[-2], layers[-1])))
        self.layers = torch.nn.Sequential(OrderedDict(mods))

    def forward(self, x):
        return self.layers(x)


def make_dnn(layers, lb, ub, dtype=torch.float32, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    net = torch.nn.Sequential(Scale(lb, ub, dtype), DNN(layers)).to(dtype)
    for m in net.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_normal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    return net


def flat_theta(net):
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()])


def load_flat_theta(net, theta):
    theta = torch.as_tensor(theta)
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            n = p.numel()
            p.copy_(theta[off:off + n].reshape(p.shape).to(p.dtype))
            off += n
    assert off == theta.numel()


def _d(y, x):
    return torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), retain_graph=True, create_graph=True)[0]


def derivatives(net, x, t, K, mixed_t_orders=1):
    """u (N,nout) and lists: dx[k] = d^k u/dx^k per output, dt[k] = d/dt of dx[k]."""
    out = net(torch.cat([x, t], dim=1))
    res = []
    for o in range(out.shape[1]):
        u = out[:, o:o + 1]
        dx = [u]
        for _ in range(K):
            dx.append(_d(dx[-1], x))
        dt = [_d(dx[k], t) for k in range(mixed_t_orders)]
        res.append((dx, dt))
    return res


COEF = {
    "burgers": [float(np.float32(0.01 / np.pi))],  # nu is stored as a float32 tensor in the reference
    "kdv5": [180.0, 30.0, 30.0, 1.0],
    "kdv7_3sol": [5 / 98, 5 / 14, 10 / 7, 5 / 14, 5.0, 3.0, 1.0, 1.0],
    "kdv7_1sol": [140.0, 70.0, 280.0, 70.0, 70.0, 42.0, 14.0, 1.0],
    "ch": [0.0],
    "ch_bc": [2.0],
    "gnss": [0.5, 1.0, 0.1],
    "nls": [0.5, 1.0, 0.0],
    "lpd": [0.5, 1.0, 0.125, 0.0625, 6.0, 10.0],
    "nls_reg": [0.5, 1.0, 1e-9],
}
ORDERS = {"burgers": (2, 1), "kdv5": (5, 1), "kdv7_3sol": (7, 1), "kdv7_1sol": (7, 1),
          "ch": (3, 3), "ch_bc": (3, 3), "gnss": (3, 1), "nls": (3, 1), "lpd": (4, 1), "nls_reg": (2, 1)}


def net_f(eq, net, x, t, coef=None):
    """Residual(s) of equation `eq` with nested autograd; returns a list of (N,1) tensors."""
    c = COEF[eq] if coef is None else coef
    K, KT = ORDERS[eq]
    d = derivatives(net, x, t, K, KT)
    u, ut = d[0]
    if eq == "burgers":
        return [ut[0] + u[0] * u[1] - c[0] * u[2]]
    if eq == "kdv5":
        return [ut[0] + c[0] * u[0] * u[0] * u[1] + c[1] * u[1] * u[2] + c[2] * u[0] * u[3] + c[3] * u[5]]
    if eq in ("kdv7_3sol", "kdv7_1sol"):
        return [ut[0] + c[0] * u[0] ** 3 * u[1] + c[1] * u[1] ** 3 + c[2] * u[0] * u[1] * u[2]
                + c[3] * u[0] ** 2 * u[3] + c[4] * u[2] * u[3] + c[5] * u[1] * u[4]
                + c[6] * u[0] * u[5] + c[7] * u[7]]
    if eq in ("ch", "ch_bc"):
        return [ut[0] - ut[2] + 3 * u[0] * u[1] - 2 * u[2] * u[1] + c[0] * u[1] - u[3] * u[0]]
    if eq in ("gnss", "nls"):
        v, vt = d[1]
        rho = u[0] ** 2 + v[0] ** 2
        fu = ut[0] + c[0] * v[2] + c[1] * rho * v[0] - c[2] * (u[3] + 6 * rho * u[1])
        fv = vt[0] - c[0] * u[2] - c[1] * rho * u[0] - c[2] * (v[3] + 6 * rho * v[1])
        return [fu, fv]
    if eq == "lpd":   # NLS/Ibc-test.py:163-164 as coded
        v, vt = d[1]
        rho = u[0] ** 2 + v[0] ** 2
        a, b, c3, c4, pp, qq = c
        Gv = v[4] + 6 * v[0] * (v[1] ** 2 - u[1] ** 2) + 12 * u[0] * u[1] * v[1] + 4 * v[0] * (u[1] ** 2 + v[1] ** 2) \
            + pp * rho * v[2] + 6 * rho ** 2 * v[0]
        Gu = u[4] + 6 * u[0] * (u[1] ** 2 - v[1] ** 2) + 12 * v[0] * u[1] * v[1] + 4 * u[0] * (u[1] ** 2 + v[1] ** 2) \
            + qq * rho * u[2] + 6 * rho ** 2 * u[0]
        fu = ut[0] + a * v[2] + b * rho * v[0] + c4 * Gv - c3 * (u[3] + 6 * rho * u[1])
        fv = vt[0] - a * u[2] - b * rho * u[0] - c4 * Gu - c3 * (v[3] + 6 * rho * v[1])
        return [fu, fv]
    if eq == "nls_reg":   # NLS/sw-test.py:159-189; third row: mean(f2^2) = the script's reg_term
        v, vt = d[1]
        rho = u[0] ** 2 + v[0] ** 2
        fu = ut[0] + c[0] * v[2] + c[1] * rho * v[0]
        fv = vt[0] - c[0] * u[2] - c[1] * rho * u[0]
        f2 = torch.sqrt(c[2] * (u[1] ** 2 + u[2] ** 2 + ut[0] ** 2 + v[1] ** 2 + v[2] ** 2 + vt[0] ** 2))
        return [fu, fv, f2]
    raise ValueError(eq)


def loss_f_and_grad(eq, net, x, t, coef=None):
    """mean-square residual loss (sum over components of mean over points) and its
    flat gradient to the weights — the reference's loss_f + loss.backward()."""
    x = x.clone().requires_grad_(True)
    t = t.clone().requires_grad_(True)
    for p in net.parameters():
        p.grad = None
    fs = net_f(eq, net, x, t, coef)
    loss = sum(torch.mean(torch.square(f)) for f in fs)
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    return loss.detach(), g, [f.detach() for f in fs]
