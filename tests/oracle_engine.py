"""TEST-ONLY engine: the same interface as pde_amd._native.NativeEngine, computed by
the CPU oracle.  It exists so that the host logic (loss composition, parameter
flattening, sharding + all-reduce) can be tested without a GPU.  The product
never imports this file."""
import numpy as np
import torch

from oracle.pyoracle import Oracle


class OracleEngine:
    def __init__(self, equation, layers, lb, ub, device):
        self.o = Oracle(equation.eq_id, layers, lb, ub, coef=equation.coef)
        self.P, self.J, self.nf, self.nout = self.o.P, self.o.J, self.o.nf, self.o.nout

    @staticmethod
    def _xt(x, t):
        if t is None:
            X = x.detach().cpu().numpy()
            return X[:, 0], X[:, 1]
        return x.detach().cpu().numpy().reshape(-1), t.detach().cpu().numpy().reshape(-1)

    def forward(self, theta, x, t=None):
        xx, tt = self._xt(x, t)
        der, _ = self.o.forward(theta.detach().numpy(), xx, tt, dtype=np.float32)
        return torch.from_numpy(np.ascontiguousarray(der[:, :, 0]))

    def forward_vjp(self, theta, x, t, gbar, grad_accum):
        xx, tt = self._xt(x, t)
        _, g = self.o.loss_grad(theta.detach().numpy(), xx, tt, gbar=gbar.detach().numpy(), dtype=np.float32)
        grad_accum += torch.from_numpy(g)

    def residual_fwd(self, theta, x, t, want_derivs=False):
        xx, tt = self._xt(x, t)
        der, f = self.o.forward(theta.detach().numpy(), xx, tt, dtype=np.float32)
        f = torch.from_numpy(f)
        return (f, torch.from_numpy(der)) if want_derivs else f

    def residual_vjp(self, theta, x, t, gbar, grad_accum):
        self.forward_vjp(theta, x, t, gbar, grad_accum)

    def loss_grad(self, theta, x, t, scale, loss_sums_accum, grad_accum):
        xx, tt = self._xt(x, t)
        ls, g = self.o.loss_grad(theta.detach().numpy(), xx, tt, scale=scale, dtype=np.float32)
        loss_sums_accum += torch.from_numpy(ls.astype(np.float32))
        grad_accum += torch.from_numpy(g)


def install(cls):
    """Return a subclass of `cls` whose engines are OracleEngines on the CPU."""
    return type(cls.__name__, (cls,), {
        "engine_factory": staticmethod(lambda eq, layers, lb, ub, dev: OracleEngine(eq, layers, lb, ub, dev)),
        "default_device": torch.device("cpu"),
    })
