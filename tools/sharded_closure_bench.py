"""Host time and device time of ONE sharded training closure (pde_amd/pinn.py::_loss_terms_sharded) at the
per-rank sizes of BASELINE's 8-GPU configs: 7kdv 1M/8 = 125 000 points and CH 10M/8 = 1 250 000 points per rank.

Run as ONE process with backend nccl (RCCL) and world size 1: the collective executes on the device buffer like
on a node, minus the wire (a label, not a scaling measurement).  Per closure form ("direct" / "autograd"):
  host_us    host time to ENQUEUE one closure (loop of closures with no synchronisation, clock stopped before the
             final synchronize): what the CPU has to spend per closure
  wall_us    wall time per closure of the same loop including the final synchronize: max(host, device) pipeline
  kernel_us  the collocation kernel alone (HIP events)
  allreduce_us  the packed buffer's all-reduce alone, enqueued back to back
One JSON line per (workload, points, form).
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from pde_amd import scripts
    for wname, n in (("kdv7", 125_000), ("kdv7", 1_000_000), ("ch", 1_250_000), ("gnss", 250_000)):
        w = WORKLOADS[wname]
        lb, ub = np.array(w["lb"]), np.array(w["ub"])
        cls = getattr(scripts, w["family"]).PI.PhysicsInformedNN
        torch.manual_seed(1234)
        m = cls(w["layers"], lb, ub, process_group=dist.group.WORLD)
        rng = np.random.default_rng(1)
        Xf = (lb + (ub - lb) * rng.random((n, 2))).astype(np.float32)
        x0 = lb[0] + (ub[0] - lb[0]) * rng.random((50, 1))
        tb = lb[1] + (ub[1] - lb[1]) * rng.random((50, 1))
        u0 = 2.0 / np.cosh(x0) ** 2
        if w["eq"] == "gnss":
            m.set_training_data(x0, u0, 0.1 * u0, tb, Xf)
        else:
            m.set_training_data(x0, u0, tb, Xf)
        eng, theta = m.engine_f, m.dnn.flat_theta()
        buf = torch.zeros(eng.P + eng.nf, device=dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kt = []
        for i in range(8):
            ev0.record()
            eng.loss_grad(theta, m.xf, m.tf, 1.0 / n, buf[eng.P:], buf[:eng.P])
            ev1.record()
            ev1.synchronize()
            if i >= 3:
                kt.append(ev0.elapsed_time(ev1) * 1e3)
        msg = torch.zeros(eng.P + 10, device=dev)
        for _ in range(5):
            dist.all_reduce(msg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(msg)
        torch.cuda.synchronize()
        ar_us = 1e6 * (time.perf_counter() - t0) / 50
        for impl in ("direct", "autograd"):
            m.closure_impl = impl
            reps = 30
            for _ in range(5):
                for p in m.dnn.parameters():
                    p.grad = None
                m._loss_terms()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                for p in m.dnn.parameters():
                    p.grad = None
                m._loss_terms()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_wall = time.perf_counter() - t0
            print(json.dumps({"workload": wname, "points_per_rank": n, "closure": impl, "backend": "nccl, world size 1 (one process)",
                              "host_us": 1e6 * t_host / reps, "wall_us": 1e6 * t_wall / reps,
                              "kernel_us": float(np.mean(kt)), "allreduce_us": ar_us}), flush=True)
        del m
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
