"""`PhysicsInformedNN.train` against the REFERENCE's own training runs.

tests/golden/train_trajectories.npz (oracle/gen_train_golden.py) holds, for three reference scripts, what the reference's
`train(12, None, 0, L)` did to seeded weights on a small seeded data set: the weights after the Adam / AdamW phase (L = 0)
and after Adam + 6 L-BFGS iterations (L = 6), each from a float32 run (the reference's arithmetic) and a float64 run
(truth).  The drop-in classes start from the same weights and data and must land where the reference lands:

  * Adam phase: distance to the float64 trajectory, relative to the distance travelled, at most 3x the reference's own
    float32 run's (which is 3e-6 ... 1.2e-5), and the same closure count;
  * Adam + L-BFGS: the same closure count as the reference (the strong-Wolfe line search took the same branches), the
    final loss within 2e-4 of the reference's float64 run, and the weights within 3x the reference float32 runs' distance
    to float64.  "Runs": the fixture also holds the reference's float32 run with the collocation points in another order
    (the same loss, other rounding) and one from initial weights moved by about one rounding (relative 1e-7), because
    the line search amplifies rounding and one run is one draw: 5kdv 3.8e-6 / 3.8e-6 / 5.4e-6 and GNSS 1.2e-5 / 1.2e-5 /
    1.4e-5 of the distance travelled, but Camassa-Holm 4.1e-6 / 1.07e-4 / 5.9e-4.  Measured here: 5kdv 3.8e-6, GNSS
    1.2e-5, CH 3.2e-4 ... 5.9e-4 depending on the order of the float32 loss sum (loss 3.9e-5 ... 7.6e-5 off).

CPU (not gpu): the classes on the oracle engine with torch's optimizers and with pde_amd.lbfgs.LBFGS on its CPU stand-in
kernels — this pins the host logic of train().  GPU: the product path (FusedAdam + device L-BFGS, eager and HIP graph;
torch optimizers as the cross-check), through the C-ABI.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from pde_amd import scripts

G = np.load(os.path.join(GOLDEN, "train_trajectories.npz"))
CASES = ["kdv5_pi", "ch_pi", "gnss_sw_adamw"]
N_ADAM, N_LBFGS = int(G["n_adam"]), int(G["n_lbfgs"])


def run(cls, case, n_lbfgs, tmp_path, **attrs):
    attrs = dict(attrs, base_dir=str(tmp_path))
    cls = type(cls.__name__, (cls,), attrs)
    layers, lb, ub = G[case + "_layers"].tolist(), G[case + "_lb"], G[case + "_ub"]
    m = cls(layers, lb, ub)
    with torch.no_grad():
        m.dnn.flat_theta().copy_(torch.tensor(G[case + "_theta0"]))
    d = {k: G["%s_data_%s" % (case, k)] for k in ("x0", "u0", "tb", "Xf")}
    if layers[-1] == 2:
        m.set_training_data(d["x0"], d["u0"], G[case + "_data_v0"], d["tb"], d["Xf"])
    else:
        m.set_training_data(d["x0"], d["u0"], d["tb"], d["Xf"])
    m.train(N_ADAM, None, 0, n_lbfgs)
    theta = m.dnn.flat_theta().detach().double().cpu().numpy()
    if n_lbfgs == 0:
        return m, theta, None
    closures = m.iter
    m.iter = 1
    m.optimizer_LBFGS.zero_grad()
    return m, theta, (closures, float(m.backward().item()))


def check_adam(case, m, theta):
    th0, t64, t32 = G[case + "_theta0"].astype(np.float64), G[case + "_theta_adam64"], G[case + "_theta_adam32"]
    travelled = np.linalg.norm(t64 - th0)
    ref_noise = np.linalg.norm(t32 - t64) / travelled
    ours = np.linalg.norm(theta - t64) / travelled
    print("train golden %s Adam phase: ours %.2e, reference float32 %.2e of the distance travelled" % (case, ours, ref_noise))
    assert m.iter == int(G[case + "_iter_adam"])
    assert ours <= max(3.0 * ref_noise, 1e-6), (case, ours, ref_noise)
    return ours, ref_noise


def check_end(case, theta, closures, loss):
    th0, t64, t32 = G[case + "_theta0"].astype(np.float64), G[case + "_theta_end64"], G[case + "_theta_end32"]
    travelled = np.linalg.norm(t64 - th0)
    ref_noise = max(np.linalg.norm(t - t64) for t in (t32, G[case + "_theta_end32p"], G[case + "_theta_end32q"])) / travelled
    ours = np.linalg.norm(theta - t64) / travelled
    l64, l0 = float(G[case + "_loss_end64"]), float(G[case + "_loss_start64"])
    print("train golden %s Adam + L-BFGS: ours %.2e, reference float32 runs %.2e of the distance travelled; loss %.6e vs %.6e; "
          "%d closures" % (case, ours, ref_noise, loss, l64, closures))
    assert closures == int(G[case + "_iter_end64"]) == int(G[case + "_iter_end32"]), (case, closures)
    assert l64 < 0.2 * l0                      # the fixture is a run that trains
    assert abs(loss - l64) <= 2e-4 * l64, (case, loss, l64)
    assert ours <= max(3.0 * ref_noise, 1e-5), (case, ours, ref_noise)
    return ours, ref_noise


# ---------------------------------------------------------------------------------------------- CPU: host logic
@pytest.mark.parametrize("case", CASES)
def test_adam_phase_on_the_oracle_engine(case, tmp_path):
    from oracle_engine import install
    ns = scripts.for_script(str(G[case + "_script"]))
    m, theta, _ = run(install(ns.PhysicsInformedNN), case, 0, tmp_path)
    assert type(m.optimizer_Adam).__name__ == ("AdamW" if case.endswith("adamw") else "Adam")
    check_adam(case, m, theta)


@pytest.mark.parametrize("lbfgs", ["torch", "device"])
@pytest.mark.parametrize("case", CASES)
def test_adam_then_lbfgs_on_the_oracle_engine(case, lbfgs, tmp_path):
    from lbfgs_cpu_kernels import CpuLBFGSKernels
    from oracle_engine import install
    ns = scripts.for_script(str(G[case + "_script"]))
    extra = {"lbfgs_impl": lbfgs}
    if lbfgs == "device":
        extra["lbfgs_kernels_factory"] = staticmethod(lambda n, h: CpuLBFGSKernels(n, h))
    m, theta, (closures, loss) = run(install(ns.PhysicsInformedNN), case, N_LBFGS, tmp_path, **extra)
    assert type(m.optimizer_LBFGS).__module__.startswith("pde_amd" if lbfgs == "device" else "torch")
    check_end(case, theta, closures, loss)


# ---------------------------------------------------------------------------------------------- GPU: the product path
@pytest.mark.gpu
@pytest.mark.parametrize("adam", ["fused", "torch"])
@pytest.mark.parametrize("case", CASES)
def test_adam_phase_on_the_hip_path(case, adam, tmp_path):
    ns = scripts.for_script(str(G[case + "_script"]))
    m, theta, _ = run(ns.PhysicsInformedNN, case, 0, tmp_path, adam_impl=adam)
    assert (type(m.optimizer_Adam).__name__ == "FusedAdam") == (adam == "fused")
    check_adam(case, m, theta)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["device", "device+graph", "torch"])
@pytest.mark.parametrize("case", CASES)
def test_adam_then_lbfgs_on_the_hip_path(case, impl, tmp_path):
    ns = scripts.for_script(str(G[case + "_script"]))
    attrs = dict(lbfgs_impl=impl.split("+")[0], adam_impl="torch" if impl == "torch" else "fused",
                 use_hip_graph=impl.endswith("graph"))
    m, theta, (closures, loss) = run(ns.PhysicsInformedNN, case, N_LBFGS, tmp_path, **attrs)
    assert type(m.optimizer_LBFGS).__module__.startswith("torch" if impl == "torch" else "pde_amd")
    check_end(case, theta, closures, loss)
