"""TEST-ONLY stand-in for the two L-BFGS launches of libpinnjet.so (csrc/lbfgs.hip), in plain torch on
the CPU, so that the host logic of pde_amd.lbfgs.LBFGS can be checked without a GPU.  Same contract as
pde_amd._native.NativeLBFGSKernels; never imported by the product."""
import torch


class CpuLBFGSKernels:
    def __init__(self, n, history, dtype=torch.float32):
        self.n, self.history, self.dtype = n, history, dtype
        self.S, self.Y, self.ro = [], [], []
        self.H = 1.0
        self.prev_g = None

    def direction(self, g, d, t, first, x=None, x0=None, t_next=0.0):
        r = self._direction(g, d, t, first)
        if x is not None:
            x0.copy_(x)
            x.add_(d, alpha=t_next)
        return r

    def _direction(self, g, d, t, first):
        g = g.to(self.dtype)
        if first:
            self.S, self.Y, self.ro, self.H = [], [], [], 1.0
        else:
            y = g - self.prev_g
            s = d.to(self.dtype) * t
            ys = float(y.dot(s))
            if ys > 1e-10:
                if len(self.S) == self.history:
                    self.S.pop(0); self.Y.pop(0); self.ro.pop(0)
                self.S.append(s); self.Y.append(y); self.ro.append(1.0 / ys)
                self.H = ys / float(y.dot(y))
        q = -g
        al = [0.0] * len(self.S)
        for i in reversed(range(len(self.S))):
            al[i] = float(self.S[i].dot(q)) * self.ro[i]
            q = q - al[i] * self.Y[i]
        r = q * self.H
        for i in range(len(self.S)):
            be = float(self.Y[i].dot(r)) * self.ro[i]
            r = r + (al[i] - be) * self.S[i]
        d.copy_(r.to(d.dtype))
        self.prev_g = g.clone()
        return torch.tensor([float(g.dot(r)), float(g.abs().sum()), float(r.abs().max()), float(g.abs().max()),
                             float(len(self.S)), self.H, 0.0, 0.0], dtype=torch.float64)

    def eval_stats(self, g, d, loss, g_copy=None):
        if g_copy is not None:
            g_copy.copy_(g)
        return torch.tensor([float(loss), float(g.to(self.dtype).dot(d.to(self.dtype))), float(g.abs().max()),
                             float(g.abs().sum())], dtype=torch.float64)
