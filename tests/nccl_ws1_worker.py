"""Worker of tests/test_gpu_rccl.py (not collected by pytest): ONE process, world size 1, backend "nccl" (= RCCL):
the sharded closure's all-reduce runs on the DEVICE buffer exactly as it does with one rank per GPU on a node
(no host staging), which is all of the RCCL path a one-GPU box can execute.

    MASTER_ADDR=127.0.0.1 MASTER_PORT=p python nccl_ws1_worker.py <case> <out.json>
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
    # family, style, layers, lb, ub, points, Adam iterations, L-BFGS iterations
    "kdv5_w50": ("kdv5_1soliton", "PI", [2, 50, 50, 50, 50, 1], [-10.0, -2.0], [10.0, 2.0], 3001, 30, 8),
    "gnss_w100": ("gnss", "sw", [2, 100, 100, 100, 100, 2], [-5.0, -2.0], [5.0, 0.0], 2003, 30, 6),
}


def flat_grad(m):
    return torch.cat([p.grad.reshape(-1) for p in m.dnn.parameters()]).detach().cpu().double().numpy()


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import pde_amd.nn as pnn
    from pde_amd import scripts
    fam, style, layers, lb, ub, n, n_adam, n_lbfgs = CASES[case]
    lb, ub = np.array(lb), np.array(ub)
    cls = getattr(getattr(scripts, fam), style).PhysicsInformedNN
    rng = np.random.default_rng(5)
    x0 = rng.uniform(lb[0], ub[0], (50, 1))
    u0 = 1.0 / np.cosh(x0) ** 2
    tb = rng.uniform(lb[1], ub[1], (50, 1))
    Xf = lb + (ub - lb) * rng.random((n, 2))
    Xp = lb + (ub - lb) * rng.random((40, 2))

    def make(group):
        torch.manual_seed(1234)
        m = cls(layers, lb, ub, process_group=group)
        if case.startswith("gnss"):
            m.set_training_data(x0, u0, 0.3 * u0, tb, Xf, Xp, 0.1 * np.ones((40, 1)), np.zeros((40, 1)))
        else:
            m.set_training_data(x0, u0, tb, Xf)
        return m

    res = {"backend": dist.get_backend(), "case": case}
    # the collective is seen in flight: the buffer it is handed lives on the device, and at world size 1 it
    # must come back bit for bit
    seen = []
    real = pnn.all_reduce_sum

    def spy(buf, group):
        before = buf.clone()
        out = real(buf, group)
        seen.append((buf.is_cuda, dist.get_backend(group), bool(torch.equal(before, buf)), buf.numel()))
        return out
    pnn.all_reduce_sum = spy

    ms, mu = make(dist.group.WORLD), make(None)
    assert ms.engine_f.kernel_kind == 2
    for impl in ("direct", "autograd"):
        ms.closure_impl = mu.closure_impl = impl
        for m in (ms, mu):
            for p in m.dnn.parameters():
                p.grad = None
            m.iter = 1
        ls, lu = ms.backward(), mu.backward()
        torch.cuda.synchronize()
        gs, gu = flat_grad(ms), flat_grad(mu)
        res["loss_rel_" + impl] = abs(ls.item() - lu.item()) / abs(lu.item())
        res["grad_rel_" + impl] = float(np.linalg.norm(gs - gu) / np.linalg.norm(gu))
    res["collectives"] = [list(s) for s in seen]
    pnn.all_reduce_sum = real
    ms.closure_impl = mu.closure_impl = "direct"

    # (c) the closure with its all-reduce inside a captured HIP graph
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in ms.dnn.parameters():
                    p.grad = None
                ms._loss_terms_sharded()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        for p in ms.dnn.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            terms = ms._loss_terms_sharded()
        graph.replay()
        torch.cuda.synchronize()
        g1 = flat_grad(ms)
        res["capture"] = "ok"
        res["capture_loss_rel"] = abs(terms["loss"].item() - lu.item()) / abs(lu.item())
        res["capture_grad_rel"] = float(np.linalg.norm(g1 - gu) / np.linalg.norm(gu))
        del graph
    except Exception as ex:                      # noqa: BLE001 - the text goes into DESIGN.md
        torch.cuda.synchronize()
        res["capture"] = "failed: %s: %s" % (type(ex).__name__, str(ex)[:600])

    # training: Adam then L-BFGS on the sharded model (RCCL all-reduce per closure) next to the unsharded one
    for m in (ms, mu):
        for p in m.dnn.parameters():
            p.grad = None
    out = {}
    for name, m in (("sharded", ms), ("single", mu)):
        m.train(n_adam, LBFGS_max_iter=0)
        th_adam = m.dnn.flat_theta().detach().cpu().double().numpy().copy()
        l_adam = m._loss_terms()["loss"].item()
        m.train(0, init_iter=m.iter, LBFGS_max_iter=n_lbfgs)
        opt = m.optimizer_LBFGS
        for p in m.dnn.parameters():
            p.grad = None
        out[name] = dict(th_adam=th_adam, l_adam=l_adam, l_end=m._loss_terms()["loss"].item(),
                         evals=opt.state[opt._params[0]]["func_evals"], module=type(opt).__module__,
                         finite=bool(torch.isfinite(m.dnn.flat_theta()).all()))
    ms.check_replicas()
    # the same training from captured HIP graphs: the Adam iteration and the L-BFGS closure, each with the RCCL
    # all-reduce INSIDE the graph
    mg = make(dist.group.WORLD)
    mg.use_hip_graph = True
    mg.train(n_adam, LBFGS_max_iter=0)
    res["graph_replays"] = getattr(mg, "graph_stats", {}).get("replays", 0)
    th_g = mg.dnn.flat_theta().detach().cpu().double().numpy().copy()
    res["adam_graph_theta_rel"] = float(np.linalg.norm(th_g - out["sharded"]["th_adam"]) / np.linalg.norm(th_g))
    mg.train(0, init_iter=mg.iter, LBFGS_max_iter=n_lbfgs)
    res["lbfgs_graph_captured"] = hasattr(mg, "_lbfgs_graph")
    for p in mg.dnn.parameters():
        p.grad = None
    res["loss_end_graph"] = mg._loss_terms()["loss"].item()
    mg.check_replicas()
    a, b = out["sharded"], out["single"]
    res["adam_theta_rel"] = float(np.linalg.norm(a["th_adam"] - b["th_adam"]) / np.linalg.norm(b["th_adam"]))
    res["loss_adam"] = [a["l_adam"], b["l_adam"]]
    res["loss_end"] = [a["l_end"], b["l_end"]]
    res["lbfgs_evals"] = [a["evals"], b["evals"]]
    res["lbfgs_module"] = a["module"]
    res["finite"] = a["finite"] and b["finite"]
    with open(out_path, "w") as fh:
        json.dump(res, fh)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
