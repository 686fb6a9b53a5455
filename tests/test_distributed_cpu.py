"""CPU, world_size 2 (and 3, 4) on gloo: sharding of the collocation points + the one
all-reduce per closure give every rank the same loss/gradient as a single
process on all points.  Compute comes from the TEST-ONLY oracle
engine; this tests the host logic of pde_amd/pinn.py and nn.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# golden case -> (drop-in class, feed kind): a one-output PI class, and a two-output class with the
# list-of-terms loss_s (NLS/IBCtest.py:252-258) that only rank 0 evaluates
CASES = {"kdv5_xavier": ("5kdv/PI-51.py", "pi"), "nls_ibc_xavier": ("NLS/sw-PINN-rogue.py", "gnss_es"),
         # a bc class: its loss1 history grid is sharded over the ranks like the collocation points
         "kdv7_bc_xavier": ("7kdv/bc-7kdv3soliton.py", "bc")}


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle_engine import install
    from pde_amd import scripts
    from test_host_logic import feed, flat_grad, load_theta
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    script, kind = CASES[name]
    ns = scripts.for_script(script)
    cls = install(ns.bcPhysicsInformedNN if kind == "bc" else ns.PhysicsInformedNN)
    cls.shard_loss1_min_rows = 8
    m = cls(g["layers"].tolist(), g["lb"], g["ub"], process_group=dist.group.WORLD)
    load_theta(m, g["theta"])
    feed(m, g, kind)
    assert m.n_f_global == len(g["x"])
    n_local = m.xf.shape[0]
    m.iter = 1
    loss = m.backward()
    out[rank] = (n_local, float(loss), flat_grad(m).copy())
    # the autograd form of the sharded closure (closure_impl = "autograd") hands out the same numbers
    for p in m.dnn.parameters():
        p.grad = None
    m.closure_impl = "autograd"
    loss_a = m.backward()
    ga = flat_grad(m).copy()
    m.closure_impl = "direct"
    assert abs(float(loss_a) - out[rank][1]) <= 2e-6 * abs(out[rank][1])
    assert np.linalg.norm(ga - out[rank][2]) <= 2e-6 * np.linalg.norm(out[rank][2])
    for p in m.dnn.parameters():
        p.grad = None
    # two Adam steps keep the replicas identical without any broadcast
    m.train(2, LBFGS_max_iter=0)
    th = m.dnn.flat_theta().clone()
    ref = th.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(th, ref)
    m.check_replicas()                       # identical replicas pass the checksum
    # the L-BFGS phase (pde_amd.lbfgs.LBFGS, CPU stand-in kernels) makes the same decisions on both ranks
    from lbfgs_cpu_kernels import CpuLBFGSKernels
    type(m).lbfgs_kernels_factory = staticmethod(lambda n, h: CpuLBFGSKernels(n, h))
    m.train(0, LBFGS_max_iter=3)
    assert type(m.optimizer_LBFGS).__module__ == "pde_amd.lbfgs"
    out["lbfgs_iter_%d" % rank] = m.iter
    m.check_replicas()
    if rank == 1:                            # one drifting replica is detected on EVERY rank
        with torch.no_grad():
            m.dnn.flat_theta()[3] += 1e-3
    try:
        m.check_replicas()
        out["drift_%d" % rank] = "not detected"
    except RuntimeError as ex:
        out["drift_%d" % rank] = "detected" if "differ across ranks" in str(ex) else str(ex)
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [(n, 2) for n in sorted(CASES)] + [("kdv7_bc_xavier", 4), ("kdv5_xavier", 3)])
def test_sharding_matches_single_process(name, world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = len(g["x"])
    shards = [out[r][0] for r in range(world)]
    assert sum(shards) == n and max(shards) - min(shards) <= 1
    # identical on every rank (all-reduce result), and equal to the reference's full backward()
    for r in range(1, world):
        assert out[0][1] == out[r][1]
        assert np.array_equal(out[0][2], out[r][2])
    assert abs(out[0][1] - g["loss_full64"]) <= 1e-5 * g["loss_full64"]
    err = np.linalg.norm(out[0][2].astype(np.float64) - g["grad_full64"]) / np.linalg.norm(g["grad_full64"])
    assert err < 1e-5
    for r in range(world):
        assert out["drift_%d" % r] == "detected"
        assert out["lbfgs_iter_%d" % r] == out["lbfgs_iter_0"] > 3
