"""End-to-end stability run on the GPU: 5kdv drop-in class, 20 000 collocation points, 2000 Adam + 600 L-BFGS iterations,
eager and with use_hip_graph; prints the final loss of both (they follow different float32 roundings)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pde_amd import scripts
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/kdv5_ckpt.npz"))
d = {k[5:]: g[k] for k in g.files if k.startswith("data_")}
rng = np.random.default_rng(0)
lb, ub = g["lb"], g["ub"]
Xf = lb + (ub - lb) * rng.random((20000, 2))
for graph in (False, True):
    torch.manual_seed(0)
    cls = scripts.for_script("5kdv/PI-51.py").PhysicsInformedNN
    cls.use_hip_graph = graph
    m = cls(g["layers"].tolist(), lb, ub)
    m.set_training_data(d["x0"], d["u0"], d["tb"], Xf)
    t0 = time.time()
    m.train(2000, LBFGS_max_iter=600)
    torch.cuda.synchronize()
    m.optimizer_LBFGS.zero_grad()
    l = m.backward().item()
    print("graph", graph, "iters", m.iter, "final loss %.4e" % l, "wall %.2f s" % (time.time() - t0), "finite", bool(torch.isfinite(m.dnn.flat_theta()).all()), flush=True)
