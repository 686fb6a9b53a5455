"""ctypes binding of libpinnjet.so (include/pinnjet.h) and the Engine that
PhysicsInformedNN drives.  There is no CPU fallback: if the HIP library is
missing or no GPU is present, constructing a NativeEngine raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpinnjet.so")

KERNEL_AUTO, KERNEL_VALU, KERNEL_MFMA = 0, 1, 2
MAX_LAYERS = 16


class PinnDesc(ctypes.Structure):
    _fields_ = [
        ("eq", ctypes.c_int32),
        ("n_lin", ctypes.c_int32),
        ("widths", ctypes.c_int32 * (MAX_LAYERS + 1)),
        ("lb", ctypes.c_float * 2),
        ("ub", ctypes.c_float * 2),
        ("coef", ctypes.c_float * 8),
        ("kernel", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 3),
    ]


class NativeError(RuntimeError):
    pass


_lib = None

# every symbol include/pinnjet.h declares
SYMBOLS = [
    "pinn_create", "pinn_destroy", "pinn_param_count", "pinn_jet_count", "pinn_nf", "pinn_nout",
    "pinn_kernel_kind", "pinn_forward", "pinn_forward_vjp", "pinn_residual_fwd", "pinn_residual_vjp",
    "pinn_loss_grad", "pinn_error_string", "pinn_reserve", "pinn_forward_vjp_static", "pinn_residual_from_derivs", "pinn_adam_step",
    "pinn_data_loss", "pinn_data_loss_cotangent", "pinn_closure_sum",
    "pinn_lbfgs_workspace_floats", "pinn_lbfgs_direction", "pinn_lbfgs_eval_stats",
    "pinn_lbfgs_direction_advance", "pinn_lbfgs_eval_stats_keep", "pinn_host_device_pointer",
    "pinn_lbfgs_poll_advance",
]


def load_library():
    """dlopen the in-tree HIP library; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "pde_amd/libpinnjet.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    lib.pinn_create.argtypes = [ctypes.POINTER(PinnDesc), ctypes.POINTER(vp)]
    lib.pinn_destroy.argtypes = [vp]
    for name in ("pinn_param_count", "pinn_jet_count", "pinn_nf", "pinn_nout", "pinn_kernel_kind"):
        getattr(lib, name).argtypes = [vp]
    lib.pinn_forward.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp]
    lib.pinn_forward_vjp.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp]
    lib.pinn_forward_vjp_static.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp]
    lib.pinn_reserve.argtypes = [vp, i64]
    lib.pinn_residual_from_derivs.argtypes = [vp, vp, i64, vp, vp]
    lib.pinn_residual_fwd.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp]
    lib.pinn_residual_vjp.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, vp]
    lib.pinn_loss_grad.argtypes = [vp, vp, vp, vp, i64, i64, f32, vp, vp, vp]
    lib.pinn_lbfgs_workspace_floats.argtypes = [i64, ctypes.c_int32]
    lib.pinn_lbfgs_workspace_floats.restype = i64
    lib.pinn_lbfgs_direction.argtypes = [vp, i64, ctypes.c_int32, vp, vp, f32, ctypes.c_int32, vp, vp]
    lib.pinn_lbfgs_eval_stats.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.pinn_lbfgs_direction_advance.argtypes = [vp, i64, ctypes.c_int32, vp, vp, f32, ctypes.c_int32, vp, vp, vp, f32, vp]
    lib.pinn_lbfgs_eval_stats_keep.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, ctypes.c_int32, vp]
    lib.pinn_host_device_pointer.argtypes = [vp, ctypes.POINTER(vp)]
    f64 = ctypes.c_double
    lib.pinn_lbfgs_poll_advance.argtypes = [vp, ctypes.c_int32, i64, f64, f64, f64, f64, ctypes.c_int32, vp, i64, ctypes.c_int32,
                                            vp, vp, vp, vp, vp, vp, vp]
    lib.pinn_adam_step.argtypes = [vp, vp, vp, vp, vp, i64, f32, ctypes.c_double, ctypes.c_double, f32, f32, ctypes.c_int32, vp]
    i32 = ctypes.c_int32
    lib.pinn_data_loss.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.pinn_data_loss_cotangent.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.pinn_closure_sum.argtypes = [vp, i32, vp, i32, f32, vp, vp, vp, vp, i64, i32, vp]
    lib.pinn_error_string.argtypes = [ctypes.c_int]
    lib.pinn_error_string.restype = ctypes.c_char_p
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().pinn_error_string(rc).decode()
        raise NativeError("%s failed: %s (%d)" % (what, msg, rc))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


def _on(device):
    """Context in which `device` is the current HIP device (the library's entry points launch on the current device).
    Nothing when it already is: torch.cuda.device() costs several microseconds per call, on every launch of a closure."""
    idx = device.index
    if idx is None or torch.cuda.current_device() == idx:
        return _NULL_CTX
    return torch.cuda.device(device)


def _split_xt(x, t):
    """(x, t) -> (x_ptr_tensor, t_ptr_tensor, stride, n).  `t is None` means x is an
    (N,2) row-major [x,t] array (predict's X_star, 7kdv/PI-7kdv3soliton.py:179-184)."""
    if t is None:
        X = x
        assert X.dim() == 2 and X.shape[1] == 2
        X = X.contiguous()
        return X, X, X.data_ptr(), X.data_ptr() + 4, 2, X.shape[0]
    x = x.reshape(-1).contiguous()
    t = t.reshape(-1).contiguous()
    assert x.numel() == t.numel()
    return x, t, x.data_ptr(), t.data_ptr(), 1, x.numel()


def _cur_stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def data_loss(out, tgt0, n0, nb, dirichlet, losses=None):
    """[loss0, loss_b] of the reference's backward() from out = net_u([IC | lower | upper rows]) in one launch
    (include/pinnjet.h, pinn_data_loss).  nb = 0: a plain mean-square term (loss1, loss_prev).
    `losses`: where to write the two floats (default: a new tensor)."""
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == n0 + 2 * nb
    tgt0 = tgt0.contiguous()
    assert tgt0.shape == (n0, out.shape[1]) and tgt0.dtype == torch.float32
    assert tgt0.device == out.device
    if losses is None:
        losses = torch.empty(2, dtype=torch.float32, device=out.device)
    assert losses.dtype == torch.float32 and losses.numel() == 2 and losses.is_contiguous() and losses.device == out.device
    with _on(out.device):
        _check(load_library().pinn_data_loss(out.data_ptr(), tgt0.data_ptr(), n0, nb, out.shape[1], int(bool(dirichlet)),
                                             losses.data_ptr(), _cur_stream(out)), "pinn_data_loss")
    return losses


def data_loss_cotangent(out, tgt0, n0, nb, dirichlet, g0, gb):
    """gbar = g0 * d loss0/d out + gb * d loss_b/d out (g0, gb: 0-dim device tensors), one launch."""
    tgt0 = tgt0.contiguous()
    g0 = g0.reshape(1).float().contiguous()
    gb = gb.reshape(1).float().contiguous()
    assert tgt0.device == out.device and g0.device == out.device and gb.device == out.device
    gbar = torch.empty_like(out)
    with _on(out.device):
        _check(load_library().pinn_data_loss_cotangent(out.data_ptr(), tgt0.data_ptr(), n0, nb, out.shape[1], int(bool(dirichlet)),
                                                       g0.data_ptr(), gb.data_ptr(), gbar.data_ptr(), _cur_stream(out)),
               "pinn_data_loss_cotangent")
    return gbar


def closure_sum(terms, n_terms, sums, n_total, ga, gb, grad, accumulate, clear=False):
    """The tail of backward() in one launch (include/pinnjet.h, pinn_closure_sum): returns loss2 = [loss, loss_f] with
    loss_f = sum(sums) / n_total and loss = terms[0] + terms[1] + loss_f + terms[2] + ..., and writes (or adds, when
    `accumulate`) ga + gb into the flat gradient `grad`.  `clear`: gb and sums are zeroed once read."""
    P = grad.numel()
    for v in (terms, sums, ga, gb, grad):
        assert v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() and v.device == grad.device
    assert ga.numel() == P and gb.numel() == P and terms.numel() >= n_terms and 0 <= sums.numel() <= 3
    loss2 = torch.empty(2, dtype=torch.float32, device=grad.device)
    with _on(grad.device):
        _check(load_library().pinn_closure_sum(terms.data_ptr(), int(n_terms), sums.data_ptr(), sums.numel(), float(n_total),
                                               loss2.data_ptr(), ga.data_ptr(), gb.data_ptr(), grad.data_ptr(), P,
                                               (1 if accumulate else 0) | (2 if clear else 0), _cur_stream(grad)), "pinn_closure_sum")
    return loss2


class NativeEngine:
    """One (network shape, equation) on one GPU.  All tensors are float32 CUDA
    tensors on `device`; kernels run on torch's current stream."""

    def __init__(self, equation, layers, lb, ub, device, kernel=KERNEL_AUTO):
        if not torch.cuda.is_available():
            raise NativeError("pde_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.lib = load_library()
        self.device = torch.device(device)
        assert self.device.type == "cuda"
        self.equation = equation
        d = PinnDesc()
        d.eq = equation.eq_id
        d.n_lin = len(layers) - 1
        if d.n_lin > MAX_LAYERS:
            raise NativeError("too many layers")
        for i, w in enumerate(layers):
            d.widths[i] = int(w)
        for i in range(2):
            d.lb[i] = float(lb[i])
            d.ub[i] = float(ub[i])
        for i, c in enumerate(equation.coef):
            d.coef[i] = float(c)
        d.kernel = kernel
        self._h = ctypes.c_void_p()
        with _on(self.device):
            _check(self.lib.pinn_create(ctypes.byref(d), ctypes.byref(self._h)), "pinn_create")
        self.P = self.lib.pinn_param_count(self._h)
        self.J = self.lib.pinn_jet_count(self._h)
        self.nf = self.lib.pinn_nf(self._h)
        self.nout = self.lib.pinn_nout(self._h)
        self.kernel_kind = self.lib.pinn_kernel_kind(self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self.lib.pinn_destroy(h)
            except Exception:
                pass
            self._h = None

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk_theta(self, theta):
        assert theta.dtype == torch.float32 and theta.is_contiguous() and theta.numel() == self.P
        assert theta.device == self.device

    # ---- net_u ----
    def forward(self, theta, x, t=None):
        self._chk_theta(theta)
        kx, kt, px, pt, stride, n = _split_xt(x, t)
        out = torch.empty((n, self.nout), dtype=torch.float32, device=self.device)
        with _on(self.device):
            _check(self.lib.pinn_forward(self._h, theta.data_ptr(), px, pt, stride, n, out.data_ptr(),
                                         self._stream()), "pinn_forward")
        return out

    def forward_vjp(self, theta, x, t, gbar, grad_accum):
        self._chk_theta(theta)
        kx, kt, px, pt, stride, n = _split_xt(x, t)
        gbar = gbar.reshape(n, self.nout).contiguous().float()
        with _on(self.device):
            _check(self.lib.pinn_forward_vjp(self._h, theta.data_ptr(), px, pt, stride, n, gbar.data_ptr(),
                                             grad_accum.data_ptr(), self._stream()), "pinn_forward_vjp")

    def forward_vjp_static(self, theta, x, t, gbar, grad_accum):
        """pinn_forward_vjp_static: same result, scratch in the library's static device array (no allocation)."""
        self._chk_theta(theta)
        kx, kt, px, pt, stride, n = _split_xt(x, t)
        gbar = gbar.reshape(n, self.nout).contiguous().float()
        with _on(self.device):
            _check(self.lib.pinn_forward_vjp_static(self._h, theta.data_ptr(), px, pt, stride, n, gbar.data_ptr(),
                                                    grad_accum.data_ptr(), self._stream()), "pinn_forward_vjp_static")

    def reserve(self, n_max):
        """Size the handle's workspace for calls of up to n_max points: no later call allocates."""
        with _on(self.device):
            _check(self.lib.pinn_reserve(self._h, int(n_max)), "pinn_reserve")

    # ---- net_f ----
    def residual_fwd(self, theta, x, t, want_derivs=False):
        self._chk_theta(theta)
        kx, kt, px, pt, stride, n = _split_xt(x, t)
        f = torch.empty((n, self.nf), dtype=torch.float32, device=self.device)
        der = torch.empty((n, self.nout, self.J), dtype=torch.float32, device=self.device) if want_derivs else None
        with _on(self.device):
            _check(self.lib.pinn_residual_fwd(self._h, theta.data_ptr(), px, pt, stride, n, f.data_ptr(),
                                              der.data_ptr() if want_derivs else None, self._stream()),
                   "pinn_residual_fwd")
        return (f, der) if want_derivs else f

    def residual_from_derivs(self, derivs):
        """the residual polynomial alone on true derivatives (n, nout, J) -> f (n, nf)"""
        derivs = derivs.reshape(-1, self.nout, self.J).contiguous().float()
        n = derivs.shape[0]
        f = torch.empty((n, self.nf), dtype=torch.float32, device=self.device)
        with _on(self.device):
            _check(self.lib.pinn_residual_from_derivs(self._h, derivs.data_ptr(), n, f.data_ptr(), self._stream()),
                   "pinn_residual_from_derivs")
        return f

    def residual_vjp(self, theta, x, t, gbar, grad_accum):
        self._chk_theta(theta)
        kx, kt, px, pt, stride, n = _split_xt(x, t)
        gbar = gbar.reshape(n, self.nf).contiguous().float()
        with _on(self.device):
            _check(self.lib.pinn_residual_vjp(self._h, theta.data_ptr(), px, pt, stride, n, gbar.data_ptr(),
                                              grad_accum.data_ptr(), self._stream()), "pinn_residual_vjp")

    def loss_grad(self, theta, x, t, scale, loss_sums_accum, grad_accum):
        """loss_sums_accum[nf] += sum f_c^2 ; grad_accum[P] += scale * d/dtheta sum f^2."""
        self._chk_theta(theta)
        kx, kt, px, pt, stride, n = _split_xt(x, t)
        with _on(self.device):
            _check(self.lib.pinn_loss_grad(self._h, theta.data_ptr(), px, pt, stride, n, float(scale),
                                           loss_sums_accum.data_ptr(), grad_accum.data_ptr(), self._stream()),
                   "pinn_loss_grad")


class NativeLBFGSKernels:
    """Device side of pde_amd.lbfgs.LBFGS: the history ring, the statistics buffers (and their pinned-host mirror) and
    include/pinnjet.h's pinn_lbfgs_* entry points, on torch's current stream."""

    def __init__(self, n, history, device):
        if not torch.cuda.is_available():
            raise NativeError("pde_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.lib = load_library()
        self.device = torch.device(device)
        self.n, self.history = int(n), int(history)
        floats = self.lib.pinn_lbfgs_workspace_floats(self.n, self.history)
        if floats <= 0:
            raise NativeError("pinn_lbfgs_workspace_floats(%d, %d) failed (%d)" % (n, history, floats))
        self.ws = torch.zeros(floats, dtype=torch.float32, device=self.device)
        # what the host reads per iteration, side by side: ONE device->host copy fetches both
        self.both = torch.zeros(12, dtype=torch.float32, device=self.device)
        self.stats, self.out = self.both[:8], self.both[8:]
        # ... and their mirror in pinned host memory, which the statistics launch writes itself (ticket last): the
        # host polls the ticket instead of enqueuing a copy and sleeping on the stream
        self.mirror = torch.zeros(16, dtype=torch.float32, pin_memory=True)
        dptr = ctypes.c_void_p()
        with _on(self.device):
            rc = self.lib.pinn_host_device_pointer(self.mirror.data_ptr(), ctypes.byref(dptr))
        self.mirror_dev = dptr.value if rc == 0 else None
        self._mirror_f = self.mirror.numpy()
        self._mirror_i = self._mirror_f.view("int32")
        self.ticket = 0
        self._ws_ptr, self._stats_ptr, self._out_ptr = self.ws.data_ptr(), self.stats.data_ptr(), self.out.data_ptr()
        self._mirror_ptr = self.mirror.data_ptr()
        self._vals = (ctypes.c_float * 12)()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, v):
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == self.n and v.device == self.device

    def direction(self, g, d, t, first, x=None, x0=None, t_next=0.0):
        """stats8 of the new direction; with `x`: also x0 = x and x += t_next * d (the line search's first trial point)"""
        self._chk(g); self._chk(d)
        if x is not None:
            self._chk(x); self._chk(x0)
        with _on(self.device):
            _check(self.lib.pinn_lbfgs_direction_advance(self._ws_ptr, self.n, self.history, g.data_ptr(), d.data_ptr(),
                                                         float(t), 1 if first else 0, self._stats_ptr,
                                                         x.data_ptr() if x is not None else None,
                                                         x0.data_ptr() if x is not None else None, float(t_next),
                                                         self._stream()), "pinn_lbfgs_direction_advance")
        return self.stats

    def eval_stats(self, g, d, loss, g_copy=None):
        """out4 of one line-search evaluation; with `g_copy`: also g_copy = g (the gradient the search keeps for the
        point).  The launch also mirrors [stats8 | out4] into pinned host memory: `fetch()` returns them."""
        self._chk(g); self._chk(d)
        if g_copy is not None:
            self._chk(g_copy)
        if loss.dtype != torch.float32 or loss.requires_grad:
            loss = loss.detach().float()
        assert loss.device == self.device and loss.numel() == 1
        self.ticket = (self.ticket + 1) & 0x3fffffff
        with _on(self.device):
            _check(self.lib.pinn_lbfgs_eval_stats_keep(g.data_ptr(), d.data_ptr(), self.n, loss.data_ptr(), self._out_ptr,
                                                       g_copy.data_ptr() if g_copy is not None else None,
                                                       self._stats_ptr, self.mirror_dev, self.ticket, self._stream()),
                   "pinn_lbfgs_eval_stats_keep")
        return self.out

    def poll_advance(self, f_prev, lr, tol_x, tol_g, allow, g_keep, d, x, x0):
        """Wait for the last eval_stats call's statistics and, in the usual case (first trial point accepted, no stopping
        rule fires, `allow`), launch the next direction update from C at once (include/pinnjet.h,
        pinn_lbfgs_poll_advance).  Returns (launched, stats8, out4), or None when the host mirror is not available."""
        if self.mirror_dev is None:
            return None
        self._chk(g_keep); self._chk(d); self._chk(x); self._chk(x0)
        args = (self._mirror_ptr, self.ticket, 4000000, float(f_prev), float(lr), float(tol_x), float(tol_g),
                1 if allow else 0, self._ws_ptr, self.n, self.history, g_keep.data_ptr(), d.data_ptr(), self._stats_ptr,
                x.data_ptr(), x0.data_ptr())
        with _on(self.device):
            rc = self.lib.pinn_lbfgs_poll_advance(*args, self._stream(), self._vals)
            if rc == 2:
                torch.cuda.current_stream(self.device).synchronize()
                rc = self.lib.pinn_lbfgs_poll_advance(*args, self._stream(), self._vals)
                if rc == 2:
                    raise NativeError("the statistics of a line-search evaluation never reached the host")
        if rc < 0:
            _check(rc, "pinn_lbfgs_poll_advance")
        v = list(self._vals)
        return rc == 1, v[:8], v[8:]

    def fetch(self):
        """(stats8, out4) of the last eval_stats call as Python floats, or None when the host mirror is not available.
        Polls the ticket word the launch writes last; falls back to a stream wait if it takes seconds."""
        if self.mirror_dev is None:
            return None
        mi, tk, spins = self._mirror_i, self.ticket, 0
        while mi[12] != tk:
            spins += 1
            if spins > 4000000:
                torch.cuda.current_stream(self.device).synchronize()
                if mi[12] != tk:
                    raise NativeError("the statistics of a line-search evaluation never reached the host")
        v = self._mirror_f[:12].tolist()
        return v[:8], v[8:]
