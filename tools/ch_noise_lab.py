"""CPU lab (numpy, no GPU): where does the fp32 noise of the Taylor-jet formulation come from at a trained
checkpoint, stage by stage?  Emulates the kernels' forward arithmetic (jet_math.hpp) in float32 with
per-stage precision switches and prints the residual noise (RMS over the batches of a
tests/golden/noise_*.npz fixture) next to the reference's own fp32 noise stored in the fixture.

    python tools/ch_noise_lab.py [case] [variant ...]

Test infrastructure / development tool: uses oracle/ as the fp64 truth.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, ORDERS, COEF  # noqa: E402

f32 = np.float32


def fma(a, b, c):
    """fmaf on float32 arrays (through float64: exact product, one rounding of the sum up to double rounding)"""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def tanh_sech2_gpu(x):
    """tanh_sech2 of jet_math.hpp, instruction by instruction in float32"""
    x = x.astype(f32)
    ax = np.abs(x)
    ax = np.minimum(ax, f32(22.0))
    a = f32(-2.0) * ax
    nf = np.rint(a * f32(1.44269504)).astype(f32)
    r = fma(nf, f32(-0.693145752), a)
    r = fma(nf, f32(-1.42860677e-6), r)
    q = np.full_like(r, f32(1.0 / 40320.0))
    for c in (1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5):
        q = fma(q, r, f32(c))
    p = fma(r * r, q, r)
    s = np.ldexp(f32(1.0), nf.astype(np.int32)).astype(f32)
    e4 = fma(f32(4.0) * s, p, f32(4.0) * s)
    num = fma(-s, p, f32(1.0) - s)
    den = fma(s, p, f32(1.0) + s)
    rc = (f32(1.0) / den).astype(f32)
    y = (num * rc).astype(f32)
    y = fma(fma(-den, y, num), rc, y)
    w0 = ((e4 * rc) * rc).astype(f32)
    return np.copysign(y, x), w0


def tanh_libm(x):
    y = np.tanh(x.astype(np.float64)).astype(f32)       # correctly rounded tanhf
    return y, (f32(1.0) - y * y).astype(f32)


def tanh_exact_pair(x):
    x64 = x.astype(np.float64)
    return np.tanh(x64).astype(f32), (1.0 / np.cosh(x64) ** 2).astype(f32)


class Cfg:
    lin = "f32"          # "f32": sequential fmaf dot products; "f64": exact dot products rounded once
    tanh = "gpu"         # "gpu" | "libm" | "exact"
    rec = "f32"          # tanh recurrences: "f32" | "f64"
    out = "f32"          # output layer + residual polynomial
    ch_fused = False     # u_t - u_xxt as one dot product of Wout with the differenced hidden jets
    dt = f32

    def __init__(self, **kw):
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def linear(W, b, a, mode):
    """a [n, nin, J] -> z [n, nout, J]; bias on component 0"""
    n, nin, J = a.shape
    nout = W.shape[0]
    if mode == "f64":
        z = np.einsum("oi,nij->noj", W.astype(np.float64), a.astype(np.float64))
        z[:, :, 0] += b.astype(np.float64)
        return z.astype(f32)
    z = np.zeros((n, nout, J), f32)
    z[:, :, 0] = b[None, :]
    for i in range(nin):
        z = fma(W[None, :, i, None], a[:, i, None, :], z)
    return z


def tanh_jet(z, K, KT, cfg):
    """jet_math.hpp tanh_jet_fwd on z [n, W, J] -> y"""
    J = K + 1 + KT
    if cfg.rec == "f64":
        z = z.astype(np.float64)
        dt = np.float64
    else:
        dt = f32
    y = np.zeros(z.shape, dt)
    w = np.zeros(z.shape[:2] + (K + 1,), dt)
    if cfg.tanh == "gpu":
        y0, w0 = tanh_sech2_gpu(z[:, :, 0].astype(f32))
    elif cfg.tanh == "libm":
        y0, w0 = tanh_libm(z[:, :, 0])
    else:
        y0, w0 = tanh_exact_pair(z[:, :, 0])
    if cfg.rec == "f64" and cfg.tanh == "exact":
        y0, w0 = np.tanh(z[:, :, 0]), 1.0 / np.cosh(z[:, :, 0]) ** 2
    y[:, :, 0], w[:, :, 0] = y0, w0
    F = fma if dt == f32 else (lambda a, b, c: a * b + c)
    zs = [None] + [dt(j) * z[:, :, j] for j in range(1, K + 1)]
    for k in range(1, K + 1):
        s = zs[1] * w[:, :, k - 1]
        for j in range(2, k + 1):
            s = F(zs[j], w[:, :, k - j], s)
        y[:, :, k] = s * dt(1.0 / k)
        # w_k = -sum_i y_i y_{k-i}, half the terms with pre-doubled factors
        qv = (dt(-2.0) * y[:, :, 0]) * y[:, :, k]
        i = 1
        while 2 * i < k:
            qv = F(dt(-2.0) * y[:, :, i], y[:, :, k - i], qv)
            i += 1
        if k % 2 == 0:
            qv = F(-y[:, :, k // 2], y[:, :, k // 2], qv)
        w[:, :, k] = qv
    for k in range(KT):
        s = np.zeros(z.shape[:2], dt)
        for j in range(k + 1):
            s = F(w[:, :, j], z[:, :, K + 1 + k - j], s)
        y[:, :, K + 1 + k] = s
    return y.astype(f32) if cfg.rec != "f64" else y.astype(f32)


def forward(eq, layers, lb, ub, theta, X, cfg):
    eq_id, coef = COEF[eq]
    K, KT = ORDERS[eq_id]
    J = K + 1 + KT
    lb32, ub32 = np.asarray(lb, f32), np.asarray(ub, f32)
    n = X.shape[0]
    x, t = X[:, 0].astype(f32), X[:, 1].astype(f32)
    hx = (f32(2.0) * (x - lb32[0]) / (ub32[0] - lb32[0]) - f32(1.0)).astype(f32)
    ht = (f32(2.0) * (t - lb32[1]) / (ub32[1] - lb32[1]) - f32(1.0)).astype(f32)
    sx, st = f32(2.0) / (ub32[0] - lb32[0]), f32(2.0) / (ub32[1] - lb32[1])
    Ws, bs, off = [], [], 0
    for l in range(len(layers) - 1):
        fi, fo = layers[l], layers[l + 1]
        Ws.append(theta[off:off + fi * fo].reshape(fo, fi).astype(f32)); off += fi * fo
        bs.append(theta[off:off + fo].astype(f32)); off += fo
    W0, b0 = Ws[0], bs[0]
    z = np.zeros((n, layers[1], J), f32)
    z[:, :, 0] = fma(W0[None, :, 0], hx[:, None], fma(W0[None, :, 1], ht[:, None], b0[None, :]))
    z[:, :, 1] = W0[None, :, 0] * sx
    z[:, :, K + 1] = W0[None, :, 1] * st
    y = tanh_jet(z, K, KT, cfg)
    for l in range(1, len(layers) - 2):
        z = linear(Ws[l], bs[l], y, cfg.lin)
        y = tanh_jet(z, K, KT, cfg)
    fact = np.array([1, 1, 2, 6, 24, 120, 720, 5040], np.float64)
    Wout, bout = Ws[-1], bs[-1]
    if cfg.out == "f64":
        u = np.einsum("oi,nij->noj", Wout.astype(np.float64), y.astype(np.float64))
        u[:, :, 0] += bout
    else:
        u = linear(Wout, bout, y, "f32").astype(np.float64 if cfg.out == "f64" else f32)
    dt = np.float64 if cfg.out == "f64" else f32
    D = np.stack([dt(fact[k]) * u[:, 0, k] for k in range(K + 1)], axis=1).astype(dt)
    T = np.stack([dt(fact[k]) * u[:, 0, K + 1 + k] for k in range(KT)], axis=1).astype(dt)
    if eq in ("ch", "ch_bc"):
        c0 = dt(coef[0])
        if cfg.ch_fused:
            # (u_t - u_xxt) from ONE dot product over the hidden layer of yt_0 - 2 yt_2
            dh = (y[:, :, K + 1] - f32(2.0) * y[:, :, K + 3]).astype(f32)
            mt = linear(Wout, np.zeros_like(bout), dh[:, :, None], "f32")[:, 0, 0].astype(dt)
        else:
            mt = (T[:, 0] - T[:, 2]).astype(dt)
        f = mt + dt(3.0) * D[:, 0] * D[:, 1] - dt(2.0) * D[:, 2] * D[:, 1] + c0 * D[:, 1] - D[:, 3] * D[:, 0]
        return np.asarray(f, np.float64)[:, None], D, T
    raise NotImplementedError(eq)


def rms(v):
    return float(np.sqrt(np.mean(np.square(v))))


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "ch_ckpt"
    g = np.load(os.path.join(ROOT, "tests", "golden", "noise_%s.npz" % case))
    eq, layers = str(g["eq"]), g["layers"].tolist()
    o = Oracle(eq, layers, g["lb"], g["ub"])
    th64 = g["theta"].astype(np.float64)
    NB = g["X"].shape[0]
    ref_f = rms(g["ref_f_err"] / g["ref_f_norm"])
    truth = [o.forward(th64, g["X"][b][:, 0], g["X"][b][:, 1]) for b in range(NB)]
    c32 = [o.forward(g["theta"], g["X"][b][:, 0], g["X"][b][:, 1], dtype=np.float32) for b in range(NB)]
    e = [np.linalg.norm(c32[b][1] - truth[b][1]) / np.linalg.norm(truth[b][1]) for b in range(NB)]
    print("reference fp32 noise (fixture)      f %.3e" % ref_f)
    print("C oracle fp32                       f %.3e = %.2fx" % (rms(e), rms(e) / ref_f))
    variants = {
        "gpu-like (f32 everything, tanh_sech2)": Cfg(),
        "libm tanh + w0 = 1 - y^2": Cfg(tanh="libm"),
        "exact tanh/sech2 pair": Cfg(tanh="exact"),
        "linear layers exact": Cfg(lin="f64"),
        "linear exact + libm": Cfg(lin="f64", tanh="libm"),
        "recurrences f64": Cfg(rec="f64"),
        "output layer + residual f64": Cfg(out="f64"),
        "linear + output exact": Cfg(lin="f64", out="f64"),
        "linear + output + rec exact (tanh gpu)": Cfg(lin="f64", out="f64", rec="f64"),
        "ch fused m_t": Cfg(ch_fused=True),
    }
    for name, cfg in variants.items():
        e = []
        for b in range(NB):
            f, D, T = forward(eq, layers, g["lb"], g["ub"], g["theta"], g["X"][b], cfg)
            e.append(np.linalg.norm(f - truth[b][1]) / np.linalg.norm(truth[b][1]))
        print("%-42s f %.3e = %.2fx" % (name, rms(e), rms(e) / ref_f))


if __name__ == "__main__":
    main()
