"""Worker of tests/test_gpu_multirank.py (not collected by pytest): ONE rank of a sharded training run
with the real HIP engine.  Several ranks share cuda:0 and talk over gloo (the closure's all-reduce is
staged through the host, pde_amd.nn.all_reduce_sum), which is what a one-GPU box can rehearse; on a
node the same code runs one rank per GPU over RCCL.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python multirank_worker.py <case> <out.json>
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # family, layers, lb, ub, points, Adam iterations, L-BFGS iterations
    "kdv5_w50": ("kdv5_1soliton", [2, 50, 50, 50, 50, 1], [-10.0, -2.0], [10.0, 2.0], 3001, 130, 12),
    "gnss_w100": ("gnss", [2, 100, 100, 100, 100, 2], [-5.0, -2.0], [5.0, 0.0], 2003, 130, 8),
}


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pde_amd import scripts
    fam, layers, lb, ub, n, n_adam, n_lbfgs = CASES[case]
    lb, ub = np.array(lb), np.array(ub)
    style = "sw" if case.startswith("gnss") else "PI"
    cls = getattr(getattr(scripts, fam), style).PhysicsInformedNN
    torch.manual_seed(1234)                       # identical xavier weights on every rank
    m = cls(layers, lb, ub, process_group=dist.group.WORLD)
    assert m.engine_f.kernel_kind == 2            # the MFMA kernels: LDS / memory-side float adds in arrival order
    rng = np.random.default_rng(5)                # every rank draws the SAME data, the class shards Xf
    x0 = rng.uniform(lb[0], ub[0], (50, 1))
    u0 = 1.0 / np.cosh(x0) ** 2
    tb = rng.uniform(lb[1], ub[1], (50, 1))
    Xf = lb + (ub - lb) * rng.random((n, 2))
    if case.startswith("gnss"):
        Xp = lb + (ub - lb) * rng.random((40, 2))
        m.set_training_data(x0, u0, 0.3 * u0, tb, Xf, Xp, 0.1 * np.ones((40, 1)), np.zeros((40, 1)))
    else:
        m.set_training_data(x0, u0, tb, Xf)
    n_local = int(m.xf.shape[0])
    m.train(n_adam, LBFGS_max_iter=0)             # passes iteration 100: check_replicas() runs inside
    m.check_replicas()
    it_adam = m.iter
    th_adam = m.dnn.flat_theta().detach().cpu().numpy().copy()
    m.train(0, init_iter=it_adam, LBFGS_max_iter=n_lbfgs)
    m.check_replicas()
    opt = m.optimizer_LBFGS
    th = m.dnn.flat_theta().detach().cpu().numpy()
    res = {"rank": rank, "n_local": n_local, "iter_adam": it_adam, "iter": m.iter,
           "lbfgs_module": type(opt).__module__, "func_evals": opt.state[opt._params[0]]["func_evals"],
           "theta_adam_hex": th_adam.tobytes().hex()[:4096], "theta_adam_sum": float(th_adam.astype(np.float64).sum()),
           "theta_hex": th.tobytes().hex()[:4096], "theta_sum": float(th.astype(np.float64).sum()),
           "finite": bool(np.isfinite(th).all())}
    with open(out_path, "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
