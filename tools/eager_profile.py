"""Host-side profile (cProfile) of the eager Adam loop at a script-sized problem: where the Python time of a closure goes.
usage: python tools/eager_profile.py [burgers|kdv7]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pde_amd import scripts  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "burgers"
name, script, n = {"burgers": ("burgers_9x20_xavier", "Burgers/Burgers_equation(PINN).py", 10000),
                   "kdv7": ("kdv7_3sol_xavier", "7kdv/PI-7kdv3soliton.py", 20000)}[case]
g = np.load(os.path.join(ROOT, "tests/golden/%s.npz" % name))
d = {k[5:]: g[k] for k in g.files if k.startswith("data_")}
rng = np.random.default_rng(0)
Xf = g["lb"] + (g["ub"] - g["lb"]) * rng.random((n, 2))
m = scripts.for_script(script).PhysicsInformedNN(g["layers"].tolist(), g["lb"], g["ub"])
args = (d["x0"], d["u0"], d["tb"], Xf) + ((0.01 / np.pi,) if case == "burgers" else ())
m.set_training_data(*args)
m.train(50, LBFGS_max_iter=0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
m.train(300, LBFGS_max_iter=0, init_iter=1)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
