"""Drop-in `PhysicsInformedNN` / `bcPhysicsInformedNN` for the Emily-Qu/PDE scripts.

Same constructor, `set_training_data`, `net_u`/`dnn`, `net_f`, `backward`,
`train`, `predict`, `save_model`, `load_model` and log-line formats as the
per-script classes of the reference (canonical copy:
7kdv/PI-7kdv3soliton.py:76-184; sliding-window variant 7kdv/sw-73.py:76-230;
bc variant 7kdv/bc-7kdv3soliton.py:198-329; two-output GNSS variant
NLS/sw-GNSS-adamW.py:76-233; Burgers Burgers/Burgers_equation(PINN).py:82-183),
but `net_f`, the loss reduction and the weight gradient run in the HIP kernels
of libpinnjet.so instead of nested torch.autograd.grad graphs.

What differs per script in the reference (residual formula, IC time `t0`,
boundary-loss form, log format, AdamW vs Adam, L-BFGS settings) is a
`Variant`; `pde_amd.scripts` instantiates one class per reference script.
"""
import os
import time
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import equations
from ._native import KERNEL_AUTO, NativeEngine
from .nn import NetU, _DataLossFn, _MseLossFn, _ResidualFn, _ResidualLossFn, data_loss_cot, data_loss_values


_SIDE_STREAMS = {}


def _side_stream(device):
    """ONE second stream per GPU for all models of the process: HIP maps streams onto a few hardware queues in
    turn (GPU_MAX_HW_QUEUES = 4), so the stream of the n-th model of a process can land on the queue of the
    default stream, where nothing overlaps (measured: the closures of a fourth model ran 25 % slower)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _SIDE_STREAMS[idx]


class FileWriter:
    """reference :63-68 — append one line, re-opening the file each time."""

    def __init__(self, logfilename, base_dir):
        self.file_path = os.path.join(base_dir, "log", logfilename)

    def write(self, content, mode="a"):
        os.makedirs(os.path.dirname(self.file_path), exist_ok=True)
        with open(self.file_path, mode, encoding="utf-8") as f:
            f.write(content + "\n")


@dataclass
class Variant:
    equation: str = "kdv7_3sol"
    t0: float = -0.5                 # IC time literal
    bc: str = "periodic"             # "periodic" (:105) or "dirichlet" (Burgers :109)
    style: str = "pi"                # "pi" | "sw" | "bc"  (set_training_data signature, extra loss term)
    optimizer: str = "adam"          # "adam" | "adamw"
    weight_decay: float = 1e-2       # AdamW (7kdv/sw73-adamW.py:170)
    lbfgs: dict = field(default_factory=lambda: dict(
        lr=1.0, max_eval=5000, history_size=50, tolerance_grad=1e-5))


def _as_col(a):
    a = np.asarray(a, dtype=np.float64)
    return a.reshape(-1, 1) if a.ndim < 2 else a


class PhysicsInformedNN:
    variant = Variant()
    base_dir = os.getcwd()           # where model/ and log/ live (<script dir> in the reference)
    default_device = None            # None -> current CUDA device
    kernel = KERNEL_AUTO
    use_hip_graph = False            # train(): replay the Adam phase / the L-BFGS closure from a captured HIP graph
    # eager backward(): the collocation kernel on a second stream beside the data-term chain.  None = automatic: when
    # that kernel is long enough to pay for the host work of two streams, N_f x P >= 1e8 (round 3, eager Adam steps/s
    # with / without: 7kdv at 20 000 points 3 877 / 3 194, GNSS-100 1 841 / 1 365, but Burgers 4x50 at 10 000 points
    # 4 940 / 5 202 and Burgers 8x20 4 733 / 7 018: closures that short are host-bound)
    overlap_streams = None
    closure_impl = "direct"          # backward(): "direct" (no autograd graph, see _loss_terms_direct) or "autograd"
    hip_graph_fallback = False       # a failed capture raises; True: report it and continue with the eager loop
    adam_impl = "fused"              # "fused": pde_amd.optim.FusedAdam (one launch per step); "torch": torch.optim.Adam/AdamW
    lbfgs_impl = "device"            # "device": pde_amd.lbfgs.LBFGS (three launches per direction, csrc/lbfgs.hip); "torch": torch.optim.LBFGS
    shard_loss1_min_rows = 4096      # sharded runs: a bc class's loss1 grid of at least this many rows is sharded over the ranks
    lbfgs_kernels_factory = None     # test hook: callable(n, history) -> stand-in for the L-BFGS launches
    engine_factory = None            # test hook: callable(equation|None, layers, lb, ub, device) -> engine

    def __init__(self, layers, lb, ub, prev_model=None, process_group=None):
        v = self.variant
        dev = self.default_device
        if dev is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.device = torch.device(dev) if dev is not None else None
        self.layers = list(layers)
        self._lb64 = np.asarray(lb, dtype=np.float64).reshape(-1)
        self._ub64 = np.asarray(ub, dtype=np.float64).reshape(-1)
        lb32 = self._lb64.astype(np.float32)
        ub32 = self._ub64.astype(np.float32)
        self.equation = equations.get(v.equation)
        if self.equation.nout != layers[-1]:
            raise ValueError("equation %s needs %d network outputs" % (v.equation, self.equation.nout))

        def factory(eq):
            e = eq if eq is not None else equations.get("none")
            if type(self).engine_factory is not None:
                return type(self).engine_factory(e, self.layers, lb32, ub32, self.device)
            if self.device is None:
                raise RuntimeError("pde_amd needs an AMD GPU; there is no CPU fallback")
            return NativeEngine(e, self.layers, lb32, ub32, self.device, kernel=self.kernel)

        if self.device is None and type(self).engine_factory is None:
            raise RuntimeError("pde_amd needs an AMD GPU; there is no CPU fallback")
        dd = self.device if self.device is not None else torch.device("cpu")
        self.device = dd
        self.lb = torch.tensor(self._lb64).float().to(dd)
        self.ub = torch.tensor(self._ub64).float().to(dd)
        self.dnn = NetU(self.layers, lb32, ub32, dd, factory)
        self.engine_f = factory(self.equation)
        if prev_model is not None:   # sliding-window warm start, 7kdv/sw-73.py:83-85
            self.dnn.load_state_dict(prev_model.dnn.state_dict())
            print("Loaded weights from previous model")
        self.group = process_group
        self.iter = 0
        self.logfilename = None
        self.X_prev = None
        self.nu = None

    # ------------------------------------------------------------------ data
    def _t(self, a):
        return torch.tensor(_as_col(a)).float().to(self.device)

    def _shard(self, Xf):
        """Contiguous equal shards of the collocation points over the ranks of the
        process group; remainder goes to the low ranks."""
        if not torch.is_tensor(Xf):
            Xf = np.asarray(Xf)
        self.n_f_global = Xf.shape[0]
        if self.group is None:
            return Xf
        import torch.distributed as dist
        R, r = dist.get_world_size(self.group), dist.get_rank(self.group)
        q, rem = divmod(Xf.shape[0], R)
        lo = r * q + min(r, rem)
        hi = lo + q + (1 if r < rem else 0)
        return Xf[lo:hi]

    def set_training_data(self, x0, u0, *args, **kw):
        """Signatures of the reference, by variant:
          pi      (x0, u0, tb, Xf)                       7kdv/PI-7kdv3soliton.py:83
          burgers (x0, u0, tb, Xf, nu)                   Burgers/Burgers_equation(PINN).py:89
          sw      (x0, u0, tb, Xf, X_prev=None, u_prev=None)              7kdv/sw-73.py:89
          gnss    (x0, u0, v0, tb, Xf[, X_prev, u_prev, v_prev])          NLS/sw-GNSS-adamW.py:89
          bc      (x0, u0, t0, u1, tb, Xf, Xh)           7kdv/bc-7kdv3soliton.py:205
        numpy float64 arrays in, float32 device tensors kept.
        Any variant also takes `earliest_solutions=[(X_ref, h_ref), ...]` (keyword): the Python list of
        reference-solution terms of NLS/IBCtest.py:211,225, summed into the loss as `loss_s` (:252-258)."""
        v = self.variant
        two = self.equation.nout == 2
        args = list(args)
        es = kw.pop("earliest_solutions", None)
        self.earliest_solutions = [(torch.as_tensor(np.asarray(X) if not torch.is_tensor(X) else X).float().to(self.device),
                                    torch.as_tensor(np.asarray(h) if not torch.is_tensor(h) else h).float().to(self.device))
                                   for X, h in es] if es else []
        self.x0 = self._t(x0)
        self.u0 = self._t(u0)
        t0 = v.t0
        self.u1 = self.x1 = self.t1 = None
        if v.style == "bc":
            names = ["t0", "u1", "tb", "Xf", "Xh"] if not two else ["v0", "t0", "u1", "v1", "tb", "Xf", "Xh"]
            vals = dict(zip(names, args))
            vals.update(kw)
            t0 = float(vals["t0"])
            if two:
                self.v0 = self._t(vals["v0"])
                self.v1 = self._t(vals["v1"])
            self.u1 = self._t(vals["u1"])
            Xh = np.asarray(vals["Xh"], dtype=np.float64)
            self.x1, self.t1 = self._t(Xh[:, 0:1]), self._t(Xh[:, 1:2])
            tb, Xf = vals["tb"], vals["Xf"]
        else:
            names = (["v0"] if two else []) + ["tb", "Xf"]
            if self.equation.name == "burgers":
                names += ["nu"]
            names += ["X_prev", "u_prev"] + (["v_prev"] if two else [])
            vals = dict(zip(names, args))
            vals.update(kw)
            if two:
                self.v0 = self._t(vals["v0"])
            tb, Xf = vals["tb"], vals["Xf"]
            if vals.get("nu") is not None:
                self.set_nu(vals["nu"])
            if vals.get("X_prev") is not None:
                self.X_prev = torch.tensor(np.asarray(vals["X_prev"], dtype=np.float64)).float().to(self.device)
                self.u_prev = self._t(vals["u_prev"])
                if two:
                    self.v_prev = self._t(vals["v_prev"])
            else:
                self.X_prev = None
        self.tb = self._t(tb)
        # the extra mean-square terms as one (points, targets) pair each
        if self.u1 is not None:
            self._X1 = torch.cat([self.x1, self.t1], dim=1).contiguous()
            self._tgt1 = torch.cat([self.u1, self.v1], dim=1).contiguous() if two else self.u1
        if self.X_prev is not None:
            self.X_prev = self.X_prev.contiguous()
            self._tgt_prev = torch.cat([self.u_prev, self.v_prev], dim=1).contiguous() if two else self.u_prev
        if torch.is_tensor(Xf):
            # collocation points sampled on the device (drivers.lhs_device): no host round trip
            Xf = self._shard(Xf.detach())
            self.xf = Xf[:, 0:1].to(self.device, torch.float32).contiguous()
            self.tf = Xf[:, 1:2].to(self.device, torch.float32).contiguous()
        else:
            Xf = self._shard(np.asarray(Xf, dtype=np.float64))
            self.xf = torch.tensor(Xf[:, 0:1]).float().to(self.device).contiguous()
            self.tf = torch.tensor(Xf[:, 1:2]).float().to(self.device).contiguous()
        # the reference reads module globals lb/ub here (:92-93); the sw variants use self.lb
        self.x_lb = torch.full_like(self.tb, float(self.lb[0]))
        self.x_ub = torch.full_like(self.tb, float(self.ub[0]))
        self.t0 = torch.full_like(self.x0, t0)
        # IC, lower-boundary and upper-boundary points as ONE (n0 + 2 nb, 2) batch, built once
        self._X_data = torch.cat([torch.cat([self.x0, self.t0], dim=1),
                                  torch.cat([self.x_lb, self.tb], dim=1),
                                  torch.cat([self.x_ub, self.tb], dim=1)], dim=0).contiguous()
        self._tgt0 = torch.cat([self.u0, self.v0], dim=1).contiguous() if two else self.u0
        # size the engines' workspaces once: no closure allocates (or synchronises) afterwards, so the
        # first one is as capturable into a HIP graph as any other (include/pinnjet.h, pinn_reserve)
        n_data = max([self._X_data.shape[0], self.x1.shape[0] if self.x1 is not None else 0,
                      self.X_prev.shape[0] if self.X_prev is not None else 0]
                     + [X.shape[0] for X, _ in self.earliest_solutions])
        if hasattr(self.engine_f, "reserve"):
            self.engine_f.reserve(self.xf.shape[0])
        if hasattr(self.dnn.engine, "reserve"):
            self.dnn.engine.reserve(n_data)

    def set_nu(self, nu):
        """Burgers viscosity: a float32 tensor in the reference (:96); a new value needs
        a new engine because the coefficient is baked into the descriptor."""
        nu32 = float(np.float32(nu))
        self.nu = torch.tensor(nu32).float().to(self.device)
        if abs(nu32 - self.equation.coef[0]) > 0:
            self.equation = equations.get(self.equation.name, [nu32])
            lb32, ub32 = self._lb64.astype(np.float32), self._ub64.astype(np.float32)
            if type(self).engine_factory is not None:
                self.engine_f = type(self).engine_factory(self.equation, self.layers, lb32, ub32, self.device)
            else:
                self.engine_f = NativeEngine(self.equation, self.layers, lb32, ub32, self.device, kernel=self.kernel)

    # ------------------------------------------------------------------ model
    def net_u(self, x, t):
        return self.dnn(torch.cat([x, t], dim=1))

    def net_f(self, x, t):
        """Residual at (x,t): (N,1) tensor ((f_u, f_v) tuple for GNSS), differentiable
        w.r.t. the weights.  Reference: nested autograd chain, :122-132."""
        x = x.detach().to(self.device, torch.float32)
        t = t.detach().to(self.device, torch.float32)
        f = _ResidualFn.apply(self.dnn, self.engine_f, x, t, self.dnn.grad_leaf())
        if self.equation.nf == 2:
            return f[:, 0:1], f[:, 1:2]
        if self.equation.nf == 3:      # NLS/sw-test.py:189 returns (fu, fv, reg_term): the third row carries the
            return f[:, 0:1], f[:, 1:2], torch.mean(torch.square(f[:, 2:3]))      # regulariser (equations.py)
        return f

    def loss_f(self):
        """mean(square(f)) (summed over components) over ALL ranks' collocation points,
        fused with its weight gradient; one all-reduce when sharded."""
        return _ResidualLossFn.apply(self.dnn, self.engine_f, self.xf, self.tf, self.n_f_global, self.group,
                                     self.dnn.grad_leaf())

    def _loss_terms(self, overlap=False):
        """Loss composition of the reference's backward() (:97-120 and variants) + loss.backward().
        Pure device work (no host sync): this is what train() captures into a HIP graph.

        `overlap` (the captured steps use it): the collocation term's fused kernel runs on a second
        stream while the data terms (forward, small reductions, vjp) run on the current one — at the
        scripts' 20 000 points that kernel leaves ~100 of the 256 CUs idle and the data-term chain is a
        third of a closure's GPU time.  The two branches are back-propagated separately (data first,
        then the collocation term, ordered by stream waits because both add into the same flat
        gradient); the returned `loss` is then a detached sum."""
        if self.group is not None:
            return self._loss_terms_sharded()
        if self.closure_impl == "direct":
            return self._loss_terms_direct(overlap)
        v = self.variant
        two = self.equation.nout == 2
        side = cur = None
        if overlap and torch.device(self.device).type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            side = _side_stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                loss_f = self.loss_f()
        # the three small data-term batches (IC, lower and upper boundary) go through net_u as ONE
        # batch: one forward and one vjp launch instead of three each (same values per point)
        n0, nb = self.x0.shape[0], self.x_lb.shape[0]
        loss0, loss_b = _DataLossFn.apply(self.dnn, self._X_data, self._tgt0, n0, nb, v.bc == "dirichlet",
                                          self.dnn.grad_leaf())

        if side is None:
            loss_f = self.loss_f()

        loss1, loss_prev, loss_s = self._extra_data_terms()

        # (terms that are absent are the reference's Python 0 / None: adding them would be a launch each)
        extra = [v for v in (loss1, loss_prev, loss_s) if torch.is_tensor(v)]
        if side is None:
            loss = loss0 + loss_b + loss_f
            for v in extra:
                loss = loss + v
            loss.backward()
        else:
            loss_d = loss0 + loss_b
            for v in extra:
                loss_d = loss_d + v
            loss_d.backward()                  # data-term vjp: concurrent with the kernel on `side`
            side.wait_stream(cur)              # ... whose gradient is added after the data terms' one
            with torch.cuda.stream(side):
                loss_f.backward()
            cur.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                loss_f.record_stream(cur)
            loss = loss_d.detach() + loss_f.detach()
        return {"loss": loss, "loss0": loss0, "loss_b": loss_b, "loss_f": loss_f, "loss1": loss1, "loss_prev": loss_prev,
                "loss_s": loss_s, "loss_reg": self._reg_share()}

    @torch.no_grad()
    def _loss_terms_direct(self, overlap=False, assign_grad=False):
        """The same closure without autograd.  Every term of the reference's loss enters with weight 1 and every
        native op here knows its own gradient, so `loss.backward()` is: the collocation pass (value + gradient in
        one kernel), one cotangent launch + one net_u vjp per data term, and ONE launch (pinn_closure_sum) that
        adds the terms up in the reference's order and the two flat gradients into `.grad` — no graph, no ones /
        multiply / accumulate launches, a third of the host time of the autograd form (`closure_impl = "autograd"`
        selects that one; both are held to the reference's backward() by the same tests).  The list-of-terms
        loss_s (broadcasting targets) stays on autograd.
        `assign_grad` (the captured L-BFGS closure): `.grad` is overwritten instead of added to, which saves the
        zeroing launch in front of it."""
        v = self.variant
        dnn, eng = self.dnn, self.dnn.engine
        theta = dnn.flat_theta()
        P, nf, N = theta.numel(), self.engine_f.nf, float(self.n_f_global)
        one = getattr(self, "_one", None)
        if one is None or one.device != theta.device:
            one = self._one = torch.ones((), dtype=torch.float32, device=theta.device)

        # (the derivative regulariser's share of loss_f is read from the sums again for the log line: not with nf = 3)
        persist = assign_grad and nf != 3

        def collocation():
            if persist:
                # the captured closure: ONE buffer for the collocation pass to accumulate into, zeroed here once and by
                # pinn_closure_sum after every use (flag `clear`) - no fill launch in front of the kernel
                buf = getattr(self, "_colloc_buf", None)
                if buf is None or buf.numel() != P + nf + 3 or buf.device != theta.device:
                    buf = self._colloc_buf = torch.zeros(P + nf + 3, dtype=torch.float32, device=theta.device)
                buf = buf[:P + nf]
            else:
                buf = torch.zeros(P + nf, dtype=torch.float32, device=theta.device)
            self.engine_f.loss_grad(theta, self.xf, self.tf, 1.0 / N, buf[P:], buf[:P])
            self.engine_f.last_sums = buf[P:]
            return buf

        side = cur = None
        if overlap and torch.device(self.device).type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            side = _side_stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                buf = collocation()
        g = torch.zeros(P, dtype=torch.float32, device=theta.device)
        # the data terms' values, in the order the reference adds them: loss0, loss_b, then loss1 | loss_prev | loss_s
        # (a mean-square term writes its value and a 0 behind it, which the next term overwrites)
        terms = torch.empty(8, dtype=torch.float32, device=theta.device)
        n0, nb = self.x0.shape[0], self.x_lb.shape[0]
        dirichlet = v.bc == "dirichlet"
        out = eng.forward(theta, self._X_data, None)
        data_loss_values(out, self._tgt0, n0, nb, dirichlet, out_losses=terms[0:2])
        eng.forward_vjp(theta, self._X_data, None, data_loss_cot(out, self._tgt0, n0, nb, dirichlet, one, one), g)
        loss0, loss_b = terms[0], terms[1]
        n_terms = 2

        def mse_term(X, tgt):
            o = eng.forward(theta, X, None)
            data_loss_values(o, tgt, o.shape[0], 0, False, out_losses=terms[n_terms:n_terms + 2])
            eng.forward_vjp(theta, X, None, data_loss_cot(o, tgt, o.shape[0], 0, False, one, one), g)
            return terms[n_terms]

        loss1, loss_prev, loss_s = None, 0, 0
        if v.style == "bc" and self.u1 is not None:
            loss1 = mse_term(self._X1, self._tgt1)
            n_terms += 1
        if v.style == "sw" and self.X_prev is not None and self.X_prev.numel() > 0:
            loss_prev = mse_term(self.X_prev, self._tgt_prev)
            n_terms += 1
        if getattr(self, "earliest_solutions", None):
            if assign_grad:
                raise RuntimeError("assign_grad: the autograd term loss_s accumulates into .grad on its own")
            with torch.enable_grad():
                for X_ref, h_ref in self.earliest_solutions:
                    loss_s = loss_s + torch.mean(torch.square(dnn(X_ref) - h_ref))
                loss_s.backward()
            loss_s = loss_s.detach()
            terms[n_terms].copy_(loss_s)
            n_terms += 1
        if side is None:
            buf = collocation()
        else:
            cur.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                buf.record_stream(cur)
        flat = dnn.flat_grad_view() if theta.is_cuda else None
        if flat is not None:
            from ._native import closure_sum
            loss2 = closure_sum(terms, n_terms, buf[P:], N, g, buf[:P], flat, accumulate=not assign_grad, clear=persist)
            loss, loss_f = loss2[0], loss2[1]
        else:
            if assign_grad:
                raise RuntimeError("assign_grad needs `.grad` to be consecutive views of one flat buffer")
            loss_f = (buf[P] if nf == 1 else buf[P:].sum()) / N
            g.add_(buf[:P])
            dnn.distribute_flat_grad(g)
            loss = loss0 + loss_b + loss_f
            for k in range(2, n_terms):
                loss = loss + terms[k]
        return {"loss": loss, "loss0": loss0, "loss_b": loss_b, "loss_f": loss_f, "loss1": loss1, "loss_prev": loss_prev,
                "loss_s": loss_s, "loss_reg": self._reg_share()}

    def _reg_share(self):
        """This rank's share of the derivative regulariser (NLS/sw-test.py:173-180), which the fused kernel
        has already summed into loss_f as a third residual row: only the log line needs it apart."""
        if self.equation.nf != 3:
            return 0
        return self.engine_f.last_sums[2].detach() / float(self.n_f_global)

    def _extra_data_terms(self):
        """loss1 of the bc classes (7kdv/bc-7kdv3soliton.py:251-253), loss_prev of the sw classes
        (7kdv/sw-73.py:132-142) and the list-of-terms loss_s (NLS/IBCtest.py:252-256: mean over ALL elements
        of square(dnn(X_ref) - h_ref), broadcasting as torch does when h_ref has one column):
        (None | tensor, 0 | tensor, 0 | tensor) like the reference's locals."""
        v = self.variant
        two = self.equation.nout == 2
        loss1 = None
        loss_prev = 0
        loss_s = 0
        for X_ref, h_ref in getattr(self, "earliest_solutions", ()):
            loss_s = loss_s + torch.mean(torch.square(self.dnn(X_ref) - h_ref))
        if v.style == "bc" and self.u1 is not None:
            loss1 = _MseLossFn.apply(self.dnn, self._X1, self._tgt1, self.dnn.grad_leaf())
        if v.style == "sw" and self.X_prev is not None and self.X_prev.numel() > 0:
            loss_prev = _MseLossFn.apply(self.dnn, self.X_prev, self._tgt_prev, self.dnn.grad_leaf())
        return loss1, loss_prev, loss_s

    # layout of the tail of a sharded closure's packed buffer (floats after the P gradient entries)
    _T_SUMS, _T_L0, _T_L1, _T_LPREV, _T_LS, _T_SIZE = 0, 3, 5, 7, 9, 10

    def _loss_terms_sharded(self, data_terms=True):
        """The closure of a sharded run: every rank evaluates the collocation term on ITS
        shard (gradient already scaled by 1/N_global), rank 0 alone evaluates the data terms
        (IC, boundary, loss1, loss_prev, loss_s: at most a few thousand points), and ONE all-reduce(SUM) of
        [flat gradient (P) | sum_p f_c^2 (3) | loss0, loss_b | loss1 | loss_prev | loss_s] gives every rank the
        same bits for everything an optimizer reads: the float32 sums inside the kernels depend on wave arrival
        order (LDS / memory-side float adds), so nothing a rank computed by itself may reach `.grad` or the
        returned loss.  Adam/AdamW (elementwise) and pde_amd.lbfgs.LBFGS (fixed-order reductions) then
        make identical updates and identical line-search decisions: the replicas stay bit-identical
        without a broadcast, and the closure counts agree (no mismatched collectives).

        `closure_impl = "direct"` (default): no autograd graph — the kernels' accumulate outputs ARE the packed
        buffer (the collocation kernel adds its gradient and its sums into it, the data-loss launches write
        their slots, every net_u vjp adds into the same P entries), so the closure is the launches, one
        all-reduce on the buffer and a handful of scalar ops; nothing in it synchronises or allocates after
        the first call, which makes it capturable where the backend's all-reduce is (RCCL).
        `data_terms=False`: the collocation term and the collective alone (what bench.py times on N > 1 GPUs)."""
        if self.closure_impl != "direct":
            return self._loss_terms_sharded_autograd()
        import torch.distributed as dist
        from .nn import all_reduce_sum
        v = self.variant
        dnn, eng = self.dnn, self.dnn.engine
        theta = dnn.flat_theta()
        P, nf, N = theta.numel(), self.engine_f.nf, float(self.n_f_global)
        T = self
        with torch.no_grad():
            buf = torch.zeros(P + T._T_SIZE, dtype=torch.float32, device=theta.device)
            g, tail = buf[:P], buf[P:]
            if self.xf.shape[0] > 0:
                self.engine_f.loss_grad(theta, self.xf, self.tf, 1.0 / N, tail[T._T_SUMS:T._T_SUMS + nf], g)
            has1 = v.style == "bc" and self.u1 is not None
            hasp = v.style == "sw" and self.X_prev is not None and self.X_prev.numel() > 0
            hass = bool(getattr(self, "earliest_solutions", ()))
            one = getattr(self, "_one", None)
            if one is None or one.device != theta.device:
                one = self._one = torch.ones((), dtype=torch.float32, device=theta.device)
            # loss1 of the bc classes runs over the whole history grid (77 312 points in CH/bc-CH-1soliton.py): large
            # enough to shard like the collocation points (each rank its contiguous rows, value and cotangent scaled
            # by rows_local / rows) instead of leaving seven ranks waiting for rank 0
            shard1 = data_terms and has1 and self._X1.shape[0] >= self.shard_loss1_min_rows
            if shard1:
                R, r = dist.get_world_size(self.group), dist.get_rank(self.group)
                n1 = self._X1.shape[0]
                q1, rem1 = divmod(n1, R)
                lo1 = r * q1 + min(r, rem1)
                hi1 = lo1 + q1 + (1 if r < rem1 else 0)
                if hi1 > lo1:
                    frac = torch.full((), (hi1 - lo1) / float(n1), dtype=torch.float32, device=theta.device)
                    X1s, t1s = self._X1[lo1:hi1], self._tgt1[lo1:hi1]
                    o = eng.forward(theta, X1s, None)
                    slot = tail[T._T_L1:T._T_L1 + 2]
                    data_loss_values(o, t1s, o.shape[0], 0, False, out_losses=slot)
                    slot.mul_(frac)
                    eng.forward_vjp(theta, X1s, None, data_loss_cot(o, t1s, o.shape[0], 0, False, frac, frac), g)
            if data_terms and dist.get_rank(self.group) == 0:
                n0, nb = self.x0.shape[0], self.x_lb.shape[0]
                dirichlet = v.bc == "dirichlet"
                out = eng.forward(theta, self._X_data, None)
                data_loss_values(out, self._tgt0, n0, nb, dirichlet, out_losses=tail[T._T_L0:T._T_L0 + 2])
                eng.forward_vjp(theta, self._X_data, None, data_loss_cot(out, self._tgt0, n0, nb, dirichlet, one, one), g)

                def mse_term(X, tgt, slot):
                    o = eng.forward(theta, X, None)
                    data_loss_values(o, tgt, o.shape[0], 0, False, out_losses=tail[slot:slot + 2])
                    eng.forward_vjp(theta, X, None, data_loss_cot(o, tgt, o.shape[0], 0, False, one, one), g)

                if has1 and not shard1:
                    mse_term(self._X1, self._tgt1, T._T_L1)
                if hasp:
                    mse_term(self.X_prev, self._tgt_prev, T._T_LPREV)
                if hass:      # the list-of-terms loss_s (broadcasting targets) stays on autograd, through one leaf
                    leaf = dnn.collect_leaf()
                    try:
                        with torch.enable_grad():
                            ls = 0
                            for X_ref, h_ref in self.earliest_solutions:
                                ls = ls + torch.mean(torch.square(dnn(X_ref) - h_ref))
                            gs, = torch.autograd.grad(ls, leaf)
                    finally:
                        dnn.end_collect()
                    g.add_(gs)
                    tail[T._T_LS] = ls.detach()
            all_reduce_sum(buf, self.group)
            dnn.distribute_flat_grad(g)
            sums = tail[T._T_SUMS:T._T_SUMS + nf]
            self.engine_f.last_sums = sums
            loss_f = (sums[0] if nf == 1 else sums.sum()) / N
            loss = tail[T._T_L0] + tail[T._T_L0 + 1] + loss_f
            if has1:
                loss = loss + tail[T._T_L1]
            if hasp:
                loss = loss + tail[T._T_LPREV]
            if hass:
                loss = loss + tail[T._T_LS]
        return {"loss": loss, "loss0": tail[T._T_L0], "loss_b": tail[T._T_L0 + 1], "loss_f": loss_f,
                "loss1": tail[T._T_L1] if has1 else None, "loss_prev": tail[T._T_LPREV] if hasp else 0,
                "loss_s": tail[T._T_LS] if hass else 0, "loss_reg": self._reg_share()}

    def _loss_terms_sharded_autograd(self):
        """`closure_impl = "autograd"` form of the sharded closure: the same terms as autograd ops through one
        hook-free leaf, `torch.autograd.grad` for the rank-local flat gradient, then the one all-reduce."""
        import torch.distributed as dist
        from .nn import all_reduce_sum
        v = self.variant
        P = self.engine_f.P
        leaf = self.dnn.collect_leaf()
        try:
            loss_f = _ResidualLossFn.apply(self.dnn, self.engine_f, self.xf, self.tf, self.n_f_global, None, leaf)
            zero = torch.zeros((), dtype=torch.float32, device=self.device)
            loss0 = loss_b = zero
            loss1, loss_prev, loss_s = None, 0, 0
            if dist.get_rank(self.group) == 0:
                n0, nb = self.x0.shape[0], self.x_lb.shape[0]
                loss0, loss_b = _DataLossFn.apply(self.dnn, self._X_data, self._tgt0, n0, nb, v.bc == "dirichlet", leaf)
                loss1, loss_prev, loss_s = self._extra_data_terms()
            has1 = v.style == "bc" and self.u1 is not None
            hasp = v.style == "sw" and self.X_prev is not None and self.X_prev.numel() > 0
            hass = bool(getattr(self, "earliest_solutions", ()))
            local = loss0 + loss_b + loss_f
            if loss1 is not None:
                local = local + loss1
            if torch.is_tensor(loss_prev):
                local = local + loss_prev
            if torch.is_tensor(loss_s):
                local = local + loss_s
            g, = torch.autograd.grad(local, leaf)
        finally:
            self.dnn.end_collect()
        reg = self._reg_share()
        buf = torch.empty(P + 7, dtype=torch.float32, device=self.device)
        buf[:P] = g
        buf[P:] = torch.stack([loss0.detach(), loss_b.detach(), loss_f.detach(),
                               loss1.detach() if loss1 is not None else zero,
                               loss_prev.detach() if torch.is_tensor(loss_prev) else zero,
                               loss_s.detach() if torch.is_tensor(loss_s) else zero,
                               reg if torch.is_tensor(reg) else zero])
        all_reduce_sum(buf, self.group)
        self.dnn.distribute_flat_grad(buf[:P])
        t = buf[P:]
        loss = t[0] + t[1] + t[2]
        if has1:
            loss = loss + t[3]
        if hasp:
            loss = loss + t[4]
        if hass:
            loss = loss + t[5]
        return {"loss": loss, "loss0": t[0], "loss_b": t[1], "loss_f": t[2], "loss1": t[3] if has1 else None,
                "loss_prev": t[4] if hasp else 0, "loss_s": t[5] if hass else 0,
                "loss_reg": t[6] if torch.is_tensor(reg) else 0}

    def check_replicas(self):
        """Sharded runs keep the weights identical on every rank WITHOUT a broadcast: every rank applies
        the same optimizer update to the same all-reduced closure result (_loss_terms_sharded).  This verifies it:
        the element-wise MIN and MAX over ranks of [sum theta, sum theta^2] must coincide (ONE all-reduce:
        MAX of [sig, -sig]).  Called with the every-100-iterations log line; raises on every rank if the replicas
        have drifted apart."""
        if self.group is None:
            return
        import torch.distributed as dist
        th = self.dnn.flat_theta().detach().double()
        sig = torch.stack([th.sum(), (th * th).sum()])
        both = torch.cat([sig, -sig])
        if both.is_cuda and dist.get_backend(self.group) == "gloo":
            host = both.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            both = host
        else:
            dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
        hi, lo = both[:2], -both[2:]
        if not torch.equal(lo, hi):
            raise RuntimeError("weights differ across ranks (checksums min %s max %s): the replicas no longer "
                               "follow the same trajectory" % (lo.tolist(), hi.tolist()))

    def _log_terms(self, terms):
        """The reference's every-100-iterations log line (the only host syncs of a closure)."""
        if self.iter % 100 != 0:
            return
        self.check_replicas()
        v = self.variant
        loss, loss0, loss_b, loss_f = terms["loss"], terms["loss0"], terms["loss_b"], terms["loss_f"]
        if torch.is_tensor(terms.get("loss_reg", 0)):     # the script logs Loss_f without its reg_term (sw-test.py:133,151)
            loss_f = loss_f.detach() - terms["loss_reg"]
        loss1, loss_prev = terms["loss1"], terms["loss_prev"]
        if v.style == "sw" and torch.is_tensor(loss_prev):
            print(f"Overlap loss: {loss_prev.item():.5e}")
        if v.style == "sw":
            message = 'Iter: %d, Loss: %.5e, Loss0: %.5e, Loss_b: %.5e, Loss_f: %.5e, Loss_prev: %.5e' % (
                self.iter, loss.item(), loss0.item(), loss_b.item(), loss_f.item(),
                loss_prev.item() if torch.is_tensor(loss_prev) else float(loss_prev))
        elif loss1 is not None:
            message = 'Iter: %d, Loss: %.5e, Loss_0: %.5e, Loss_1: %.5e, Loss_b: %.5e, Loss_f: %.5e' % (
                self.iter, loss.item(), loss0.item(), loss1.item(), loss_b.item(), loss_f.item())
        else:
            message = 'Iter: %d, Loss: %.5e, Loss_0: %.5e, Loss_b: %.5e, Loss_f: %.5e' % (
                self.iter, loss.item(), loss0.item(), loss_b.item(), loss_f.item())
        if torch.is_tensor(terms.get("loss_s", 0)):       # NLS/IBCtest.py:262-263
            message += ', Loss_s: %.5e' % terms["loss_s"].item()
        if self.logfilename is not None:
            self.writer.write(message)
        if self._is_rank0():
            print(message)

    def backward(self):
        """Same loss composition as the reference's backward() (:97-120 and variants);
        returns the 0-dim loss with `.grad` accumulated on dnn.parameters()."""
        ov = self.overlap_streams
        if ov is None:
            ov = self.xf.shape[0] * self.engine_f.P >= 1e8
        terms = self._loss_terms(overlap=ov)
        self._log_terms(terms)
        self.iter += 1
        return terms["loss"]

    def _is_rank0(self):
        if self.group is None:
            return True
        import torch.distributed as dist
        return dist.get_rank(self.group) == 0

    # ------------------------------------------------------------------ train
    def train(self, nIter, logfilename=None, init_iter=0, LBFGS_max_iter=50000):
        """Adam/AdamW for nIter steps, then one L-BFGS(strong_wolfe) step — :135-165.
        The optimizers are torch.optim (third-party in the reference too)."""
        v = self.variant
        self.dnn.train()
        self.iter = init_iter
        params = list(self.dnn.parameters())
        self._fused_adam = self.adam_impl == "fused" and torch.device(self.device).type == "cuda"
        if self._fused_adam:
            # same update rule and defaults as torch.optim.Adam / AdamW(weight_decay=1e-2), one launch per step
            from .optim import FusedAdam
            self.optimizer_Adam = FusedAdam(params, flat_params=self.dnn.flat_theta(), decoupled=v.optimizer == "adamw",
                                            weight_decay=v.weight_decay if v.optimizer == "adamw" else 0.0)
        elif v.optimizer == "adamw":
            self.optimizer_Adam = torch.optim.AdamW(params, weight_decay=v.weight_decay)
        else:
            self.optimizer_Adam = torch.optim.Adam(params)
        lbfgs_kw = dict(lr=v.lbfgs.get("lr", 1.0), max_iter=LBFGS_max_iter,
                        max_eval=v.lbfgs.get("max_eval", 5000), history_size=v.lbfgs.get("history_size", 50),
                        tolerance_grad=v.lbfgs.get("tolerance_grad", 1e-5),
                        tolerance_change=1.0 * np.finfo(float).eps, line_search_fn="strong_wolfe")
        kfac = type(self).lbfgs_kernels_factory
        if self.lbfgs_impl == "device" and (torch.device(self.device).type == "cuda" or kfac is not None):
            from .lbfgs import LBFGS
            theta = self.dnn.flat_theta()
            self.optimizer_LBFGS = LBFGS(params, flat_params=theta,
                                         kernels=kfac(theta.numel(), lbfgs_kw["history_size"]) if kfac else None,
                                         **lbfgs_kw)
        else:
            self.optimizer_LBFGS = torch.optim.LBFGS(params, **lbfgs_kw)
        self.logfilename = logfilename if self._is_rank0() else None
        if self.logfilename is not None:
            self.writer = FileWriter(logfilename, self.base_dir)
            self.writer.write("Start Training:", mode='w')
        start_time = time.time()
        done = 0
        if self.use_hip_graph and nIter > 16 and self._graph_capturable():
            done = self._train_adam_graph(nIter, params)
        for _ in range(nIter - done):
            self.optimizer_Adam.zero_grad()
            self.backward()
            self.optimizer_Adam.step()

        def LBFGS_callback():
            self.optimizer_LBFGS.zero_grad()
            return self.backward()
        # (the reference always makes this call, :161 - with LBFGS_max_iter = 0 torch evaluates the closure once)
        closure = LBFGS_callback
        if LBFGS_max_iter > 0 and self.use_hip_graph and self._graph_capturable() and hasattr(self.optimizer_LBFGS, "flat_grad"):
            closure = self._graph_closure() or LBFGS_callback
        if hasattr(self.optimizer_LBFGS, "on_discard"):
            def _undo_iteration():      # a speculative evaluation pde_amd.lbfgs.LBFGS made in vain: not an iteration
                self.iter -= 1
            self.optimizer_LBFGS.on_discard = _undo_iteration
        self.optimizer_LBFGS.step(closure)
        elapsed = time.time() - start_time
        if self.logfilename is not None:
            self.writer.write('Training time: %.4f' % (elapsed))
        if self._is_rank0():
            print('Training time: %.4f' % (elapsed))

    def _graph_capturable(self):
        """Can a closure of this model be captured into a HIP graph?  A single-process run always; a sharded run
        when its closure is the direct form (no autograd graph) and its collective runs on RCCL, whose all-reduce
        is captured like any kernel (tests/test_gpu_rccl.py).  A gloo group (host-staged buffer) cannot."""
        if torch.device(self.device).type != "cuda":
            return False
        if self.group is None:
            return True
        import torch.distributed as dist
        return self.closure_impl == "direct" and dist.get_backend(self.group) == "nccl"

    def _train_adam_graph(self, nIter, params):
        """the Adam phase without per-iteration host work.  One whole iteration
        (zero_grad, the loss terms, loss.backward(), Adam(capturable).step()) is captured into a HIP
        graph after a few eager iterations and replayed; the host only touches the device for the
        reference's every-100-iterations log line.  Same kernels in the same order as the eager loop
        (Adam's bias corrections are evaluated on the device in fp32 instead of in Python doubles).
        Returns the number of iterations done.  A capture failure raises (a broken capture must not
        turn into a silent slow run); with `hip_graph_fallback = True` it is reported and the caller
        continues eagerly with the SAME optimizer (capturable Adam also steps eagerly), so the moments
        and step count of the warm-up iterations are kept."""
        v = self.variant
        kw = dict(capturable=True)
        if self._fused_adam:
            opt = self.optimizer_Adam            # its step count lives on the device: capturable as it is
        elif v.optimizer == "adamw":
            opt = torch.optim.AdamW(params, weight_decay=v.weight_decay, **kw)
        else:
            opt = torch.optim.Adam(params, **kw)
        self.optimizer_Adam = opt
        done = 0
        try:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(3):                      # eager: allocates workspaces, initialises Adam state
                    opt.zero_grad(set_to_none=True)
                    terms = self._loss_terms(overlap=True)
                    self._log_terms(terms)
                    self.iter += 1
                    opt.step()
                    done += 1
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            opt.zero_grad(set_to_none=True)

            # the direct closure writes the flat gradient in its last launch: no zeroing launches in front of the kernels
            assign = (self._fused_adam and self.group is None and self.closure_impl == "direct"
                      and not getattr(self, "earliest_solutions", None) and self.dnn.flat_grad_view() is not None
                      and self.dnn.flat_grad_view().data_ptr() == opt.flat_grad().data_ptr())

            def body():
                if assign:
                    terms = self._loss_terms_direct(overlap=True, assign_grad=True)
                else:
                    if self._fused_adam:
                        opt.flat_grad().zero_()         # the flat gradient is a static buffer: cleared inside the graph
                    terms = self._loss_terms(overlap=True)
                opt.step()
                return terms
            graph, terms = self._capture(body)
        except Exception as ex:                         # noqa: BLE001
            torch.cuda.synchronize(self.device)
            if not self.hip_graph_fallback:
                raise RuntimeError("HIP-graph capture of the training step failed (set hip_graph_fallback = True "
                                   "to continue eagerly instead): %s" % (ex,)) from ex
            print("HIP-graph capture of the training step failed (%s); continuing eagerly" % (ex,))
            return done
        t_rep, n_rep = time.time(), nIter - done
        while done < nIter:
            graph.replay()
            self._log_terms(terms)
            self.iter += 1
            done += 1
        torch.cuda.synchronize(self.device)
        self.graph_stats = {"replays": n_rep, "seconds": time.time() - t_rep}
        self._adam_graph = graph                        # keep the captured buffers alive with the model
        return done

    def _capture(self, body):
        """`body()` captured into a HIP graph: (graph, what body returned).  A capture can be invalidated by work that
        is not its own (hipErrorStreamCaptureInvalidated was seen once in this round's GPU test runs, in a process where
        models and their graphs come and go; not reproduced, cause not established).  Two precautions: the cyclic
        garbage collector is paused for the duration, so that no older graph is destroyed and no device memory freed
        inside the capture, and an invalidated capture is attempted once more before the error is raised — nothing
        that was captured has run."""
        import gc
        last = None
        for _ in range(2):
            graph = torch.cuda.CUDAGraph()
            was_enabled = gc.isenabled()              # (torch.cuda.graph's __enter__ runs a full collection itself)
            gc.disable()
            try:
                with torch.cuda.graph(graph):
                    out = body()
                return graph, out
            except Exception as ex:                     # noqa: BLE001
                last = ex
                torch.cuda.synchronize(self.device)
            finally:
                if was_enabled:
                    gc.enable()
        raise last

    def _graph_closure(self):
        """The L-BFGS closure (zero the flat gradient, the loss terms, loss.backward()) captured once
        into a HIP graph and replayed per evaluation: the line search moves the weights in place, so
        every address the graph holds stays valid.  Needs pde_amd.lbfgs.LBFGS (its `.grad` are views
        of one static buffer).  A capture failure raises unless `hip_graph_fallback` is set (then: None)."""
        opt = self.optimizer_LBFGS
        try:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):                      # eager: allocates the kernels' workspaces
                    opt.zero_grad()
                    self._loss_terms(overlap=True)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            opt.zero_grad()
            # the direct closure writes `.grad` in its last launch: no zeroing launch in front of the kernels
            assign = (self.group is None and self.closure_impl == "direct"
                      and not getattr(self, "earliest_solutions", None) and self.dnn.flat_grad_view() is not None)

            def body():
                if assign:
                    return self._loss_terms_direct(overlap=True, assign_grad=True)
                opt.flat_grad().zero_()
                return self._loss_terms(overlap=True)
            graph, terms = self._capture(body)
        except Exception as ex:                         # noqa: BLE001
            torch.cuda.synchronize(self.device)
            if not self.hip_graph_fallback:
                raise RuntimeError("HIP-graph capture of the L-BFGS closure failed (set hip_graph_fallback = True "
                                   "to continue eagerly instead): %s" % (ex,)) from ex
            print("HIP-graph capture of the L-BFGS closure failed (%s); continuing eagerly" % (ex,))
            return None
        self._lbfgs_graph = graph

        def closure():
            graph.replay()
            self._log_terms(terms)
            self.iter += 1
            return terms["loss"]
        return closure

    # ------------------------------------------------------------------ io
    def load_model(self, filename):
        if filename is not None:
            load_path = os.path.join(self.base_dir, 'model', filename)
            sd = torch.load(load_path, map_location=self.device, weights_only=True)
            self.dnn.load_state_dict(sd)
            print(f"Model parameters loaded from {load_path}.")

    def save_model(self, filename):
        if filename is not None:
            save_path = os.path.join(self.base_dir, 'model', filename)
            os.makedirs(os.path.dirname(save_path), exist_ok=True)
            sd = OrderedDictClone(self.dnn.state_dict())
            torch.save(sd, save_path)
            print(f"Model parameters saved to {save_path}.")

    def predict(self, X_star):
        """u on X_star (M,2) as numpy (M,1); (u, v) tuple for two-output nets — :179-184."""
        X = torch.tensor(np.asarray(X_star, dtype=np.float64)).float().to(self.device)
        self.dnn.eval()
        with torch.no_grad():
            out = self.dnn(X)
        out = out.detach().cpu().numpy()
        if self.equation.nout == 2:
            return out[:, 0:1], out[:, 1:2]
        return out


def OrderedDictClone(sd):
    from collections import OrderedDict
    return OrderedDict((k, v.detach().clone()) for k, v in sd.items())


class bcPhysicsInformedNN(PhysicsInformedNN):
    """Segments 1..3 of the bc-PINN drivers: extra loss1 over all earlier grid points
    (7kdv/bc-7kdv3soliton.py:198-262, CH/bc-CH-1soliton.py:204-262)."""
    variant = Variant(style="bc")


def make_class(name, base=PhysicsInformedNN, **variant_kw):
    """Build a per-script class: `make_class("PI73", equation="kdv7_3sol", t0=-0.5)`."""
    style = variant_kw.get("style", base.variant.style)
    v = Variant(**{**dict(style=style), **variant_kw})
    return type(name, (base,), {"variant": v})
