"""Print the parity margins of every golden case for both kernel families (GPU)."""
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pde_amd import equations  # noqa: E402
from pde_amd._native import NativeEngine, NativeError  # noqa: E402

dev = torch.device("cuda", 0)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))
for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "*.npz"))):
    name = os.path.basename(path)[:-4]
    if name.startswith("predict"):
        continue
    g = np.load(path)
    for kname, kernel in (("valu", 1), ("mfma", 2)):
        try:
            eng = NativeEngine(equations.get(str(g["eq"])), g["layers"].tolist(), g["lb"].astype(np.float32),
                               g["ub"].astype(np.float32), dev, kernel=kernel)
        except NativeError:
            continue
        cu = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
        theta, x, t = cu(g["theta"]), cu(g["x"]), cu(g["t"])
        n = len(g["x"])
        f = eng.residual_fwd(theta, x, t)
        ls = torch.zeros(eng.nf, device=dev)
        grad = torch.zeros(eng.P, device=dev)
        eng.loss_grad(theta, x, t, 1.0 / n, ls, grad)
        torch.cuda.synchronize()
        loss = ls.sum().item() / n
        print("%-22s %s  f %.2e (ref32 %.2e)  loss %.2e (ref32 %.2e)  grad %.2e (ref32 %.2e)  mean(f-f64)/rms %.2e (ref32 %.2e)" % (
            name, kname, rel(f.cpu().numpy().reshape(g["f64"].shape), g["f64"]), rel(g["f32"], g["f64"]),
            abs(loss - float(g["loss_f64"])) / float(g["loss_f64"]), abs(float(g["loss_f32"]) - float(g["loss_f64"])) / float(g["loss_f64"]),
            rel(grad.cpu().numpy(), g["grad_f64"]), rel(g["grad_f32"], g["grad_f64"]),
            float(np.mean(f.cpu().numpy().reshape(g["f64"].shape) - g["f64"]) / np.sqrt(np.mean(g["f64"] ** 2))),
            float(np.mean(g["f32"] - g["f64"]) / np.sqrt(np.mean(g["f64"] ** 2)))))
