"""`self.dnn` of the reference, backed by the native engine.

The reference builds `torch.nn.Sequential(Scale(lb, ub), DNN(layers))`
(7kdv/PI-7kdv3soliton.py:80) whose state-dict keys are
`1.layers.layer_{i}.{weight,bias}`.  NetU keeps exactly that module tree (so
shipped .pth checkpoints load and saved ones load back into the reference) but

  * every Linear weight/bias is a view into ONE flat float32 buffer `theta`
    in state-dict order, which is what the kernels read;
  * autograd sees ONE leaf (`grad_leaf()`, the same storage): the native ops hand
    it one flat gradient per backward pass, and a post-accumulate hook gives each
    parameter its slice as `.grad` (views, no copies) — instead of ten small
    accumulation kernels per op and pass;
  * calling the module runs the fused Scale+MLP HIP kernel through a
    torch.autograd.Function instead of a chain of addmm/tanh ATen ops.
"""
from collections import OrderedDict

import numpy as np
import torch


class Scale(torch.nn.Module):
    """Same role and attribute names as the reference's Scale (:34-41); the affine
    map itself is fused into the kernels, `forward` is kept for completeness."""

    def __init__(self, lb, ub, device):
        super().__init__()
        self.lb = torch.tensor(np.asarray(lb, dtype=np.float64)).float().to(device)
        self.ub = torch.tensor(np.asarray(ub, dtype=np.float64)).float().to(device)

    def forward(self, X):
        return 2.0 * (X - self.lb) / (self.ub - self.lb) - 1.0


class DNN(torch.nn.Module):
    """Module tree of the reference's DNN (:43-61): layers.layer_i / activation_i."""

    def __init__(self, layers):
        super().__init__()
        self.depth = len(layers) - 1
        mods = []
        for i in range(self.depth - 1):
            mods.append(("layer_%d" % i, torch.nn.Linear(layers[i], layers[i + 1])))
            mods.append(("activation_%d" % i, torch.nn.Tanh()))
        mods.append(("layer_%d" % (self.depth - 1), torch.nn.Linear(layers[-2], layers[-1])))
        self.layers = torch.nn.Sequential(OrderedDict(mods))

    def linears(self):
        return [m for m in self.layers if isinstance(m, torch.nn.Linear)]


def xavier(m):
    """reference :70-74"""
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            torch.nn.init.zeros_(m.bias)


def data_loss_values(out, tgt0, n0, nb, dirichlet, out_losses=None):
    """[loss0, loss_b] from out = net_u([IC | lower | upper rows]) (reference :99-105; Burgers :109).  GPU: one
    launch (pinn_data_loss); CPU tensors (the tests of the host logic): the torch expressions.
    `out_losses`: two consecutive floats to write them to (a slot of a sharded closure's packed buffer)."""
    if out.is_cuda:
        from ._native import data_loss
        return data_loss(out, tgt0, n0, nb, dirichlet, out_losses)
    r0 = out[:n0] - tgt0
    l0 = (r0 * r0).sum() / n0
    if nb == 0:
        res = torch.stack([l0, torch.zeros_like(l0)])
    else:
        rb = out[n0:] if dirichlet else out[n0:n0 + nb] - out[n0 + nb:]
        res = torch.stack([l0, (rb * rb).sum() / nb])
    if out_losses is not None:
        out_losses.copy_(res)
        return out_losses
    return res


def data_loss_cot(out, tgt0, n0, nb, dirichlet, g0, gb):
    """g0 * d loss0/d out + gb * d loss_b/d out (g0, gb: 0-dim tensors): the cotangent net_u's vjp takes."""
    if out.is_cuda:
        from ._native import data_loss_cotangent
        return data_loss_cotangent(out, tgt0, n0, nb, dirichlet, g0, gb)
    gbar = torch.empty_like(out)
    gbar[:n0] = (out[:n0] - tgt0) * (g0 * (2.0 / n0))
    if nb > 0:
        rb = out[n0:] if dirichlet else out[n0:n0 + nb] - out[n0 + nb:]
        tb = rb * (gb * (2.0 / nb))
        if dirichlet:
            gbar[n0:] = tb
        else:
            gbar[n0:n0 + nb] = tb
            gbar[n0 + nb:] = -tb
    return gbar


class _NetUFn(torch.autograd.Function):
    """u = dnn([x,t]) with a theta-gradient.  No gradient flows to the inputs: the
    reference differentiates w.r.t. x,t only inside net_f, which has its own op.
    `leaf` is NetU.grad_leaf(): the one autograd leaf standing for all parameters."""

    @staticmethod
    def forward(ctx, net, X, leaf):
        theta = net.flat_theta()
        ctx.net = net
        ctx.X = X
        return net.engine.forward(theta, X, None)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        theta = net.flat_theta()
        g = torch.zeros_like(theta)
        net.engine.forward_vjp(theta, ctx.X, None, gout, g)
        return None, None, g


class _DataLossFn(torch.autograd.Function):
    """loss_0 and loss_b of the reference's backward() (:99-105; Dirichlet form:
    Burgers/Burgers_equation(PINN).py:109) as ONE op over the batch X = [IC points | lower-boundary
    points | upper-boundary points]: one net_u forward, the two mean-square sums, and in backward one
    net_u vjp with the cotangent 2 r / n.  Same values as the chain of torch ops it replaces (the sums
    run over all columns at once), a dozen fewer kernels and autograd nodes per closure.

    On the GPU the sums and the cotangent are one launch each (pinn_data_loss / pinn_data_loss_cotangent:
    a closure at the scripts' sizes waits for the LENGTH of this chain, not for its arithmetic); the torch
    expressions below are what the CPU tests of the host logic run."""

    @staticmethod
    def forward(ctx, net, X, tgt0, n0, nb, dirichlet, leaf):
        theta = net.flat_theta()
        out = net.engine.forward(theta, X, None)
        ctx.net, ctx.X, ctx.out, ctx.tgt0 = net, X, out, tgt0
        ctx.n0, ctx.nb, ctx.dirichlet = n0, nb, dirichlet
        losses = data_loss_values(out, tgt0, n0, nb, dirichlet)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g0, gb):
        net, n0, nb, out, tgt0 = ctx.net, ctx.n0, ctx.nb, ctx.out, ctx.tgt0
        theta = net.flat_theta()
        gbar = data_loss_cot(out, tgt0, n0, nb, ctx.dirichlet, g0, gb)
        g = torch.zeros_like(theta)
        net.engine.forward_vjp(theta, ctx.X, None, gbar, g)
        return None, None, None, None, None, None, g


class _MseLossFn(torch.autograd.Function):
    """sum_c mean((net_u(X)[:, c] - tgt[:, c])^2): loss1 of the bc classes (7kdv/bc-7kdv3soliton.py:251-253) and
    loss_prev of the sliding-window classes (7kdv/sw-73.py:132-142) as one op: forward, one launch for the sum,
    one for the cotangent, the vjp (the same kernels as _DataLossFn with no boundary rows)."""

    @staticmethod
    def forward(ctx, net, X, tgt, leaf):
        theta = net.flat_theta()
        out = net.engine.forward(theta, X, None)
        ctx.net, ctx.X, ctx.out, ctx.tgt = net, X, out, tgt
        return data_loss_values(out, tgt, out.shape[0], 0, False)[0]

    @staticmethod
    def backward(ctx, g):
        net, out, tgt = ctx.net, ctx.out, ctx.tgt
        theta = net.flat_theta()
        gbar = data_loss_cot(out, tgt, out.shape[0], 0, False, g, g)
        gt = torch.zeros_like(theta)
        net.engine.forward_vjp(theta, ctx.X, None, gbar, gt)
        return None, None, None, gt


class _ResidualFn(torch.autograd.Function):
    """f = net_f(x, t) as an (N, nf) tensor whose backward reaches the weights."""

    @staticmethod
    def forward(ctx, net, engine, x, t, leaf):
        theta = net.flat_theta()
        ctx.net, ctx.engine, ctx.x, ctx.t = net, engine, x, t
        return engine.residual_fwd(theta, x, t)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        theta = net.flat_theta()
        g = torch.zeros_like(theta)
        ctx.engine.residual_vjp(theta, ctx.x, ctx.t, gout, g)
        return None, None, None, None, g


def all_reduce_sum(buf, group):
    """The closure's one collective: SUM over the ranks of `group`, in place.  On RCCL (backend "nccl")
    the device buffer goes straight in.  A gloo group (CPU rehearsals, several ranks sharing one GPU in
    the tests) gets it staged through the host: gloo reduces host memory anyway."""
    import torch.distributed as dist
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class _ResidualLossFn(torch.autograd.Function):
    """loss_f = sum_c mean_p f_c^2 in ONE fused kernel pass that also produces
    d loss_f / d theta (the reference's loss_f + its share of loss.backward(),
    :107-108,111).  `n_global` is the number of collocation points over ALL ranks: the value and
    the gradient are this rank's share of the global mean.  With `group` the shares are summed here
    ([grad (P floats), sum f_c^2 (nf floats)]: the collocation term alone, as bench.py times it);
    a sharded training closure passes group=None and all-reduces the WHOLE closure once instead
    (PhysicsInformedNN._loss_terms_sharded)."""

    @staticmethod
    def forward(ctx, net, engine, x, t, n_global, group, leaf):
        theta = net.flat_theta()
        buf = torch.zeros(engine.P + engine.nf, dtype=torch.float32, device=theta.device)
        if x.numel() > 0:
            engine.loss_grad(theta, x, t, 1.0 / float(n_global), buf[engine.P:], buf[:engine.P])
        if group is not None:
            all_reduce_sum(buf, group)
        ctx.net = net
        ctx.grad_flat = buf[:engine.P]
        engine.last_sums = buf[engine.P:]          # sum_p f_c^2 per residual row (this rank's, unless `group`)
        return buf[engine.P:].sum() / float(n_global)

    @staticmethod
    def backward(ctx, gout):
        return None, None, None, None, None, None, ctx.grad_flat * gout


class NetU(torch.nn.Sequential):
    """Sequential(Scale, DNN) with a flat parameter buffer and a native forward."""

    def __init__(self, layers, lb, ub, device, engine_factory):
        device = torch.device(device)
        super().__init__(Scale(lb, ub, device), DNN(layers))
        self.layer_sizes = list(layers)
        self._device = device
        self._collect = None
        self[1].apply(xavier)
        self._flatten_params()
        # engine for plain forward / data-loss terms (equation "none")
        self.engine = engine_factory(None)

    # -- flat parameter buffer -------------------------------------------------
    def _flatten_params(self):
        params = []
        for lin in self[1].linears():
            params += [lin.weight, lin.bias]
        flat = torch.cat([p.detach().reshape(-1).float() for p in params]).to(self._device).contiguous()
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self._theta = flat
        self.param_list = params

    def grad_leaf(self):
        """The autograd leaf a native op takes as its parameter input: a fresh alias of the flat buffer
        per call (no kernel), so that its gradient accumulation runs on the stream the op runs on —
        a cached leaf keeps the stream of its first use, which breaks HIP-graph capture on another one."""
        if self._collect is not None:
            return self._collect
        leaf = self.flat_theta().detach().requires_grad_(True)
        leaf.register_post_accumulate_grad_hook(self._distribute_grad)
        return leaf

    def collect_leaf(self):
        """Sharded closures (PhysicsInformedNN._loss_terms_sharded): ONE hook-free leaf that every native op
        of the closure takes until `end_collect()`, so that `torch.autograd.grad(loss, leaf)` yields the
        rank-local flat gradient of the whole closure — which is all-reduced before any parameter sees it."""
        self._collect = self.flat_theta().detach().requires_grad_(True)
        return self._collect

    def end_collect(self):
        self._collect = None

    def _distribute_grad(self, leaf):
        g = leaf.grad
        leaf.grad = None
        self.distribute_flat_grad(g)

    def flat_grad_view(self):
        """The flat gradient buffer when every `.grad` is a consecutive float32 view of ONE buffer, in parameter order
        (pde_amd.lbfgs.LBFGS and FusedAdam keep them so); None otherwise."""
        params = self.param_list
        first = params[0].grad
        if first is None or first.dtype != torch.float32:
            return None
        base, off = first.data_ptr(), 0
        for p in params:
            gp = p.grad
            if gp is None or gp.data_ptr() != base + 4 * off or not gp.is_contiguous():
                return None
            off += p.numel()
        return first.as_strided((off,), (1,))

    def distribute_flat_grad(self, g):
        """Runs once per native op and backward pass: every parameter gets its slice of the op's flat gradient.  `.grad` is None (after zero_grad(set_to_none=True)): the
        slices become the `.grad` tensors, no kernel at all.  `.grad` are consecutive views of one
        buffer (pde_amd.lbfgs.LBFGS keeps them so): one flat add.  Otherwise: one add per parameter,
        the usual accumulating semantics."""
        params = self.param_list
        first = params[0].grad
        if first is not None:
            base, off, flat_ok = first.data_ptr(), 0, first.dtype == torch.float32
            for p in params:
                gp = p.grad
                if gp is None or gp.data_ptr() != base + 4 * off or not gp.is_contiguous():
                    flat_ok = False
                    break
                off += p.numel()
            if flat_ok:
                first.as_strided((off,), (1,)).add_(g)
                return
        off = 0
        for p in params:
            n = p.numel()
            if p.requires_grad:
                v = g[off:off + n].view(p.shape)
                if p.grad is None:
                    p.grad = v
                else:
                    p.grad.add_(v)
            off += n

    def flat_theta(self):
        """The contiguous float32 parameter vector the kernels read.  Parameters are
        views into it; if something re-pointed them (e.g. `.to()`), re-flatten."""
        off = 0
        base = self._theta.data_ptr()
        for p in self.param_list:
            if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                self._flatten_params()
                break
            off += p.numel()
        return self._theta

    def forward(self, X):
        X = X.to(self._device, torch.float32)
        if X.dim() != 2 or X.shape[1] != 2:
            raise ValueError("dnn expects an (N,2) tensor of [x,t] rows")
        if X.shape[0] == 0:
            return torch.zeros((0, self.layer_sizes[-1]), device=self._device)
        return _NetUFn.apply(self, X.detach().contiguous(), self.grad_leaf())
