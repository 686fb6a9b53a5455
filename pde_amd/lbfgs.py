"""L-BFGS with strong-Wolfe line search whose vector work stays on the GPU.

Every reference script ends `train()` with ONE `torch.optim.LBFGS(...).step(closure)` of up to
50 000 closure evaluations (7kdv/PI-7kdv3soliton.py:138-161) — at the scripts' own sizes that is
where most of the wall time goes.  torch's optimizer issues several tiny tensor ops per history
pair and blocks on the device for every scalar it compares.  This class is the same algorithm
(same constructor arguments, same `step(closure)` protocol and stopping rules, same history
update / two-loop recursion / bracketing / zoom / cubic interpolation, so a run follows the
trajectory of `torch.optim.LBFGS` up to float32 summation order) arranged differently:

  * parameters and gradients are ONE flat buffer each (`p.grad` are views, so the closure's
    `loss.backward()` accumulates straight into it — no gather/scatter ops);
  * the direction update is one call of three launches (`pinn_lbfgs_direction_advance`, csrc/lbfgs.hip), the
    last of which also moves the weights to the line search's first trial point;
  * each closure evaluation is followed by one launch that packs [loss, g.d, max|g|] and keeps a copy of the
    gradient, and ONE device->host copy; all line-search arithmetic runs on Python floats.

There is no CPU fallback in the product: without the HIP library the constructor raises.
(`kernels=` is the seam tests/ uses to drive the host logic on the CPU.)
"""
import math

import torch


def _cubic_min(x1, f1, g1, x2, f2, g2, bounds=None):
    """Minimiser of the cubic through (x1,f1,g1), (x2,f2,g2), clipped to `bounds`
    (default: the interval between the two points); midpoint when the cubic has no minimum."""
    if bounds is not None:
        lo, hi = bounds
    else:
        lo, hi = (x1, x2) if x1 <= x2 else (x2, x1)
    if x1 == x2:
        return min(max(x1, lo), hi)
    d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2)
    disc = d1 * d1 - g1 * g2
    if disc >= 0.0:
        d2 = math.sqrt(disc)
        den = (g2 - g1 if x1 <= x2 else g1 - g2) + 2.0 * d2
        if den != 0.0:
            if x1 <= x2:
                pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / den)
            else:
                pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / den)
            if pos == pos:
                return min(max(pos, lo), hi)
    return (lo + hi) / 2.0


class _Point:
    """One evaluated step length of the line search."""
    __slots__ = ("t", "f", "g", "gtd", "gmax")

    def __init__(self, t, f, g, gtd, gmax):
        self.t, self.f, self.g, self.gtd, self.gmax = t, f, g, gtd, gmax


class LBFGS(torch.optim.Optimizer):
    """Drop-in for `torch.optim.LBFGS` (one parameter group, dense float32 parameters on one
    device).  `flat_params`: the contiguous buffer the parameters are views of, in order
    (NetU.flat_theta()); when omitted the parameters are re-pointed into a new flat buffer."""

    speculate = True      # step(): first line-search evaluation enqueued before the direction's statistics are read
    on_discard = None     # callable: a speculative closure evaluation was made in vain and has been undone

    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9,
                 history_size=100, line_search_fn=None, flat_params=None, kernels=None):
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        if line_search_fn not in (None, "strong_wolfe"):
            raise RuntimeError("only 'strong_wolfe' is supported")
        defaults = dict(lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                        tolerance_change=tolerance_change, history_size=history_size, line_search_fn=line_search_fn)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("LBFGS doesn't support per-parameter options (parameter groups)")
        self._params = self.param_groups[0]["params"]
        n = sum(p.numel() for p in self._params)
        dev = self._params[0].device
        if flat_params is None:
            flat_params = torch.cat([p.detach().reshape(-1).float() for p in self._params]).contiguous()
            off = 0
            for p in self._params:
                p.data = flat_params[off:off + p.numel()].view(p.shape)
                off += p.numel()
        else:
            off = 0
            for p in self._params:
                if p.data_ptr() != flat_params.data_ptr() + 4 * off or p.dtype != torch.float32:
                    raise ValueError("flat_params is not the buffer the parameters are views of")
                off += p.numel()
            if off != flat_params.numel():
                raise ValueError("flat_params has %d elements, the parameters %d" % (flat_params.numel(), off))
        self._x = flat_params
        self._g = torch.zeros(n, dtype=torch.float32, device=dev)
        self._d = torch.zeros(n, dtype=torch.float32, device=dev)
        if kernels is None:
            from ._native import NativeLBFGSKernels
            kernels = NativeLBFGSKernels(n, history_size, dev)
        self._k = kernels
        self._host = torch.empty(8, dtype=torch.float32, pin_memory=True) if dev.type == "cuda" else None
        self._host2 = torch.empty(16, dtype=torch.float32, pin_memory=True) if dev.type == "cuda" else None
        self._attach_grads()
        st = self.state[self._params[0]]
        st.setdefault("func_evals", 0)
        st.setdefault("n_iter", 0)
        self.host_syncs = 0
        self._x_at = None
        self._x0 = None                     # the line search's starting point when the direction launch saves it

    # ---- flat gradient -------------------------------------------------------------------
    def _intact(self):
        """every `.grad` is still the view this optimizer attached (identity checks: the per-evaluation host path)"""
        views = getattr(self, "_views", None)
        return views is not None and all(p.grad is v for p, v in zip(self._params, views))

    def _attach_grads(self):
        if self._intact():
            return
        off = 0
        for p in self._params:
            n = p.numel()
            want = self._g.data_ptr() + 4 * off
            if p.grad is None or p.grad.data_ptr() != want:
                p.grad = self._g[off:off + n].view(p.shape)
            off += n
        self._views = [p.grad for p in self._params]

    def zero_grad(self, set_to_none=True):
        """Zeroes the flat gradient; `.grad` stay views into it (never set to None)."""
        self._g.zero_()
        self._attach_grads()

    def flat_grad(self):
        """The static flat gradient buffer `.grad` are views of (what a captured closure zeroes)."""
        return self._g

    def _gather(self):
        """Flat gradient after a closure; copies only what a foreign zero_grad()/backward re-pointed."""
        if self._intact():
            return self._g
        off = 0
        for p in self._params:
            n = p.numel()
            seg = self._g[off:off + n]
            if p.grad is None:
                seg.zero_()
            elif p.grad.data_ptr() != seg.data_ptr():
                seg.copy_(p.grad.reshape(-1))
            off += n
        return self._g

    def _fetch(self, ev):
        """(stats8, out4) through the kernels' pinned-host mirror when `ev` is what their last statistics launch wrote"""
        fetch = getattr(self._k, "fetch", None)
        if fetch is None or ev is not getattr(self._k, "out", None):
            return None
        return fetch()

    def _read(self, dev_vec):
        """device floats -> Python floats: the one blocking read per evaluation."""
        self.host_syncs += 1
        got = self._fetch(dev_vec)
        if got is not None:
            return got[1]
        if self._host is None:
            return [float(v) for v in dev_vec.tolist()]
        k = dev_vec.numel()
        self._host[:k].copy_(dev_vec, non_blocking=True)
        torch.cuda.current_stream(dev_vec.device).synchronize()
        return self._host[:k].tolist()

    def _read2(self, a, b):
        """two device vectors, ONE blocking read (and one copy when they lie side by side, as the native kernels'
        statistics do)"""
        self.host_syncs += 1
        if self._host is None:
            return [float(v) for v in a.tolist()], [float(v) for v in b.tolist()]
        if a is getattr(self._k, "stats", None):
            got = self._fetch(b)
            if got is not None:
                return got
        ka, kb = a.numel(), b.numel()
        both = getattr(self._k, "both", None)
        if both is not None and both.numel() == ka + kb and a.data_ptr() == both.data_ptr() \
                and b.data_ptr() == both.data_ptr() + 4 * ka:
            self._host2[:ka + kb].copy_(both, non_blocking=True)
        else:
            self._host2[:ka].copy_(a, non_blocking=True)
            self._host2[ka:ka + kb].copy_(b, non_blocking=True)
        torch.cuda.current_stream(a.device).synchronize()
        v = self._host2[:ka + kb].tolist()
        return v[:ka], v[ka:]

    # ---- one closure evaluation at x0 + t d ------------------------------------------------
    def _evaluate_async(self, closure, x0, t, moved=False):
        """enqueue the evaluation; returns (flat gradient copy, device [f, g.d, max|g|, sum|g|]) without reading.
        `moved`: the parameters already hold x0 + t d (the direction launch put them there)."""
        if not moved:
            torch.add(x0, self._d, alpha=t, out=self._x)
        self._x_at = t                      # the parameters now hold x0 + t d
        loss = closure()
        g = self._gather()
        keep = torch.empty_like(g)          # (no launch: the statistics launch fills it)
        return keep, self._k.eval_stats(g, self._d, loss, keep)

    def _evaluate(self, closure, x0, t):
        g, ev = self._evaluate_async(closure, x0, t)
        f, gtd, gmax, _ = self._read(ev)
        return _Point(t, f, g, gtd, gmax)

    def _strong_wolfe(self, closure, x0, t, f0, g0, gtd0, gmax0, d_norm, c1=1e-4, c2=0.9, tolerance_change=1e-9,
                      max_ls=25, first_point=None):
        """Bracketing + zoom phases of the strong-Wolfe search (Nocedal & Wright alg. 3.5/3.6 as
        arranged in torch.optim.lbfgs._strong_wolfe).  Returns (accepted point, evaluations).
        `first_point`: the evaluation at x0 + t d when step() has already made it."""
        start = _Point(0.0, f0, g0, gtd0, gmax0)
        new = first_point if first_point is not None else self._evaluate(closure, x0, t)
        evals, prev, it = 1, start, 0
        done, bracket = False, None
        while it < max_ls:
            if new.f > f0 + c1 * new.t * gtd0 or (it > 1 and new.f >= prev.f):
                bracket = [prev, new]
                break
            if abs(new.gtd) <= -c2 * gtd0:
                bracket, done = [new], True
                break
            if new.gtd >= 0:
                bracket = [prev, new]
                break
            lo = new.t + 0.01 * (new.t - prev.t)
            hi = new.t * 10
            t_next = _cubic_min(prev.t, prev.f, prev.gtd, new.t, new.f, new.gtd, bounds=(lo, hi))
            prev = new
            new = self._evaluate(closure, x0, t_next)
            evals += 1
            it += 1
        if it == max_ls:
            bracket = [start, new]
        # zoom: shrink the bracket until a point satisfies both conditions or it collapses
        insuf = False
        low, high = (0, 1) if bracket[0].f <= bracket[-1].f else (1, 0)
        while not done and it < max_ls:
            if abs(bracket[1].t - bracket[0].t) * d_norm < tolerance_change:
                break
            t_new = _cubic_min(bracket[0].t, bracket[0].f, bracket[0].gtd, bracket[1].t, bracket[1].f, bracket[1].gtd)
            b_hi, b_lo = max(bracket[0].t, bracket[1].t), min(bracket[0].t, bracket[1].t)
            eps = 0.1 * (b_hi - b_lo)
            if min(b_hi - t_new, t_new - b_lo) < eps:
                if insuf or t_new >= b_hi or t_new <= b_lo:
                    t_new = b_hi - eps if abs(t_new - b_hi) < abs(t_new - b_lo) else b_lo + eps
                    insuf = False
                else:
                    insuf = True
            else:
                insuf = False
            new = self._evaluate(closure, x0, t_new)
            evals += 1
            it += 1
            if new.f > f0 + c1 * new.t * gtd0 or new.f >= bracket[low].f:
                bracket[high] = new
                low, high = (0, 1) if bracket[0].f <= bracket[1].f else (1, 0)
            else:
                if abs(new.gtd) <= -c2 * gtd0:
                    done = True
                elif new.gtd * (bracket[high].t - bracket[low].t) >= 0:
                    bracket[high] = bracket[low]
                bracket[low] = new
        return (bracket[low] if len(bracket) > 1 else bracket[0]), evals

    @torch.no_grad()
    def step(self, closure):
        """Up to `max_iter` L-BFGS iterations / `max_eval` closure evaluations; returns the first
        loss, like torch.optim.LBFGS.step."""
        grp = self.param_groups[0]
        lr, max_iter, max_eval = grp["lr"], grp["max_iter"], grp["max_eval"]
        tol_g, tol_x = grp["tolerance_grad"], grp["tolerance_change"]
        wolfe = grp["line_search_fn"] == "strong_wolfe"
        closure = torch.enable_grad()(closure)
        st = self.state[self._params[0]]

        orig_loss = closure()
        g = self._gather()
        f, _, gmax, _ = self._read(self._k.eval_stats(g, self._d, orig_loss))
        evals = 1
        st["func_evals"] += 1
        if gmax <= tol_g:
            return orig_loss
        g = g.clone()
        t = st.get("t", lr)
        n_iter = 0
        launched = False                    # the direction of the coming iteration is already on its way (poll_advance)
        poll_advance = getattr(self._k, "poll_advance", None) if self.speculate else None
        while n_iter < max_iter:
            n_iter += 1
            st["n_iter"] += 1
            first = st["n_iter"] == 1
            spec = x0 = None
            speculating = wolfe and not first and self.speculate
            if launched:
                launched = False
                dstats = self._k.stats
            elif speculating:
                if self._x0 is None:
                    self._x0 = torch.empty_like(self._x)
                dstats = self._k.direction(g, self._d, t, first, x=self._x, x0=self._x0, t_next=lr)
            else:
                dstats = self._k.direction(g, self._d, t, first)
            if speculating:
                # From the second iteration on the line search starts at t = lr whatever the direction's
                # statistics say: that evaluation is enqueued before they are read, and both come back in ONE
                # blocking read instead of two.  Decisions and evaluation counts are those of the plain order;
                # only when the search direction turns out not to descend (the break below, i.e. the last
                # iteration of a run) has one evaluation been made in vain.  It is undone: weights and flat
                # gradient are put back, it is not counted in func_evals, and `on_discard` (if set) lets the
                # closure's owner roll back its own side effects (PhysicsInformedNN: the iteration counter).
                x0 = self._x0
                gs, ev = self._evaluate_async(closure, x0, lr, moved=True)
                got = None
                if poll_advance is not None:
                    # the wait, the usual decision (first trial point accepted, no stopping rule fires) and the next
                    # direction's launch happen in C, back to back: the interpreter is not between the statistics'
                    # arrival and the GPU's next work.  Same tests, same arithmetic as below.
                    got = poll_advance(f, lr, tol_x, tol_g, n_iter != max_iter and evals + 1 < max_eval, gs, self._d,
                                       self._x, self._x0)
                if got is not None:
                    self.host_syncs += 1
                    launched, (gtd, gsum, dmax, _gm, _cnt, _hd, _, _), (fs, gtds, gmaxs, _) = got
                    if launched:
                        t, f, g, gmax = lr, fs, gs, gmaxs
                        evals += 1
                        st["func_evals"] += 1
                        continue
                else:
                    (gtd, gsum, dmax, _gm, _cnt, _hd, _, _), (fs, gtds, gmaxs, _) = self._read2(dstats, ev)
                spec = True
            else:
                gtd, gsum, dmax, _gm, _cnt, _hd, _, _ = self._read(dstats)
            prev_f = f
            t = min(1.0, 1.0 / gsum) * lr if first else lr
            if gtd > -tol_x:
                if spec is not None:
                    self._x.copy_(x0)
                    self._g.copy_(g)
                    if self.on_discard is not None:
                        self.on_discard()
                break
            if spec is not None and not fs > f + 1e-4 * lr * gtd and abs(gtds) <= -0.9 * gtd:
                # the usual case, without the search's bookkeeping: the first trial point satisfies both Wolfe
                # conditions (the first two tests of _strong_wolfe with c1 = 1e-4, c2 = 0.9)
                f, g, gmax, ls_evals = fs, gs, gmaxs, 1
            elif wolfe:
                if x0 is None:
                    x0 = self._x.clone()
                pt, ls_evals = self._strong_wolfe(closure, x0, t, f, g, gtd, gmax, dmax,
                                                  first_point=_Point(lr, fs, gs, gtds, gmaxs) if spec is not None else None)
                t, f, g, gmax = pt.t, pt.f, pt.g, pt.gmax
                if self._x_at != t:         # (the accepted point is usually the last one evaluated: already there)
                    torch.add(x0, self._d, alpha=t, out=self._x)
                    self._x_at = t
            else:
                self._x.add_(self._d, alpha=t)
                ls_evals = 0
                if n_iter != max_iter:
                    loss = closure()
                    gv = self._gather()
                    f, _, gmax, _ = self._read(self._k.eval_stats(gv, self._d, loss))
                    g = gv.clone()
                    ls_evals = 1
            evals += ls_evals
            st["func_evals"] += ls_evals
            if n_iter == max_iter or evals >= max_eval:
                break
            if gmax <= tol_g:
                break
            if dmax * abs(t) <= tol_x:
                break
            if abs(f - prev_f) < tol_x:
                break
        st["t"] = t
        return orig_loss
