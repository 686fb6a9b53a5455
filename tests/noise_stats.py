"""Helper of the GPU parity tests and tools/noise_report.py (not collected by pytest): fp32 noise of the HIP
path at one of the reference's checkpoints, as RMS over the batches of a tests/golden/noise_*.npz fixture
(oracle/gen_noise_fixtures.py), next to the reference's own fp32 noise on the same batches.  Truth = the
fp64 C oracle, cross-checked against the fixture's reference-fp64 residual (`oracle_vs_ref64`)."""
import numpy as np
import torch

from oracle.pyoracle import Oracle
from pde_amd import equations
from pde_amd._native import NativeEngine


def noise_stats(g, kernel, dev):
    eq = str(g["eq"])
    layers = g["layers"].tolist()
    eng = NativeEngine(equations.get(eq), layers, g["lb"].astype(np.float32), g["ub"].astype(np.float32), dev, kernel=kernel)
    o = Oracle(eq, layers, g["lb"], g["ub"])
    theta = torch.tensor(g["theta"], device=dev)
    th64 = g["theta"].astype(np.float64)
    NB, B = g["X"].shape[:2]
    fe, ge, le, chk = np.zeros(NB), np.zeros(NB), np.zeros(NB), 0.0
    for b in range(NB):
        X = g["X"][b]
        x, t = torch.tensor(X[:, 0].copy(), device=dev), torch.tensor(X[:, 1].copy(), device=dev)
        f = eng.residual_fwd(theta, x, t).cpu().numpy().astype(np.float64)
        buf = torch.zeros(eng.P + eng.nf, device=dev)
        eng.loss_grad(theta, x, t, 1.0 / B, buf[eng.P:], buf[:eng.P])
        torch.cuda.synchronize()
        got = buf.cpu().numpy().astype(np.float64)
        _, f64 = o.forward(th64, X[:, 0], X[:, 1])
        ls64, g64 = o.loss_grad(th64, X[:, 0], X[:, 1], scale=1.0 / B)
        chk = max(chk, np.linalg.norm(f64 - g["f64"][b]) / np.linalg.norm(g["f64"][b]))
        fe[b] = np.linalg.norm(f - f64) / np.linalg.norm(f64)
        ge[b] = np.linalg.norm(got[:eng.P] - g64) / np.linalg.norm(g64)
        le[b] = abs(got[eng.P:].sum() - ls64.sum()) / ls64.sum()
    rms = lambda v: float(np.sqrt(np.mean(np.square(v))))
    ref_f = rms(g["ref_f_err"] / g["ref_f_norm"])
    ref_g = rms(g["ref_g_err"] / g["ref_g_norm"])
    ref_l = rms((g["ref_loss32"] - g["ref_loss64"]) / g["ref_loss64"])
    return dict(f=rms(fe), g=rms(ge), l=rms(le), ref_f=ref_f, ref_g=ref_g, ref_l=ref_l, oracle_vs_ref64=chk)
