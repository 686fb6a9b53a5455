"""Benchmark of the PINN collocation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): collocation residual+grad evals/sec.  One "step" = one
closure-equivalent of the PDE term: forward jets + residual + sum f^2 + gradient to
the weights over the rank's shard of collocation points, plus (N>1) the
all-reduce of [grad, sum f^2] that keeps L-BFGS consistent across ranks.
Workload = BASELINE.json configs[2]: 7th-order KdV 3-soliton, net [2,50x4,1],
1M synthetic collocation points PER GPU (weak scaling), weights of the reference's
shipped checkpoint 7kdv/model/PI 73.pth (`--weights xavier`:
xavier-normal, seed 1234), points uniform in [-15,15]x[-0.5,0.5].

Besides the contract fields the JSON line carries
  roofline      FP32 compute roofline of the dominant kernel (HIP-event timed)
  cpu_baseline  the reference's method (nested torch.autograd.grad) timed on the
                host cores on a bounded sample (rank 0, N=1 only)
  train_steps_per_s   full backward()+Adam.step() iterations/s (IC/BC terms included)
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# jet orders (K, KT) per equation
ORDERS = {"kdv7_3sol": (7, 1), "kdv5": (5, 1), "burgers": (2, 1), "gnss": (3, 1), "ch": (3, 3)}


def algorithmic_flops(layers, eq):
    """Algorithmic flops per point per closure, SURVEY 8(d): minimal Taylor-mode count, FMA = 2, tanh = 1.
    M = MACs of one plain forward, H = tanh neurons, J = K+1+KT jet components, c(K) = 2K^2 + 5K (order-K tanh
    recurrence), t-tangent cost tau = 1 (KT = 1) or 2 c(KT-1) + 2; forward F = 2 M J + H (c(K) + 1 + tau);
    reverse to the weights = 2F; A = 3F.  Gives SURVEY's figures: 7kdv 4.94e5, 5kdv 3.68e5, Burgers 4x50 1.96e5,
    GNSS-100 9.5e5, CH-64 5.8e5 (and 7.8e4 for the 8x20 Burgers net).  No recomputation, padding or unused
    derivatives are counted."""
    K, KT = ORDERS[eq]
    c = lambda k: 2 * k * k + 5 * k
    M = sum(layers[i] * layers[i + 1] for i in range(len(layers) - 1))
    H = sum(layers[1:-1])
    J = K + 1 + KT
    tau = 1 if KT == 1 else 2 * c(KT - 1) + 2
    return 3.0 * (2 * M * J + H * (c(K) + 1 + tau))


FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector == FP32-input MFMA

# `ckpt`: golden fixture holding the weights of the reference's shipped checkpoint for this config (SURVEY 8(d):
# C3 = 7kdv/model/PI 73.pth, C2 = 5kdv/model/PINN1t-2.pth, C4 = NLS/model/adamWsw(0.1+0)t-2_1.pth); configs without a
# checkpoint of their shape (Burgers 8x20, CH width 64) use xavier-normal weights, seed 1234.
WORKLOADS = {
    "kdv7": dict(eq="kdv7_3sol", layers=[2, 50, 50, 50, 50, 1], lb=[-15.0, -0.5], ub=[15.0, 0.5], ckpt="kdv7_3sol_ckpt",
                 n=1_000_000, family="kdv7_3soliton", name="7kdv PI-7kdv3soliton, [2,50x4,1], N_f=1M per GPU"),
    "kdv5": dict(eq="kdv5", layers=[2, 50, 50, 50, 50, 1], lb=[-10.0, -2.0], ub=[10.0, 2.0], ckpt="kdv5_ckpt",
                 n=100_000, family="kdv5_1soliton", name="5kdv PI-51, [2,50x4,1], N_f=100k"),
    "gnss": dict(eq="gnss", layers=[2, 100, 100, 100, 100, 2], lb=[-5.0, -2.0], ub=[5.0, 0.0], ckpt="gnss_w100_ckpt",
                 n=2_000_000, family="gnss", name="NLS sw-GNSS-adamW, [2,100x4,2], N_f=2M"),
    "ch": dict(eq="ch", layers=[2, 64, 64, 64, 64, 1], lb=[-15.0, -1.0], ub=[15.0, 1.0],
               n=1_250_000, family="camassa_holm", name="CH PI, [2,64x4,1], N_f=1.25M per GPU (10M over 8)"),
    # the script as committed (Burgers/Burgers_equation(PINN).py:209-210: [2,50x4,1], N_f = 10 000)
    # with its shipped checkpoint; BASELINE.json words the config as a "9x20 tanh MLP": that net is burgers_9x20
    "burgers": dict(eq="burgers", layers=[2, 50, 50, 50, 50, 1], lb=[-1.0, 0.0], ub=[1.0, 0.99], ckpt="burgers_ckpt",
                    n=10_000, family="burgers", name="Burgers_equation(PINN).py, [2,50x4,1], N_f=10k"),
    "burgers_9x20": dict(eq="burgers", layers=[2] + [20] * 8 + [1], lb=[-1.0, 0.0], ub=[1.0, 0.99],
                         n=10_000, family="burgers", name="Burgers, [2,20x8,1], N_f=10k"),
}


def checkpoint_theta(w):
    """flat weights of the reference's shipped checkpoint for this workload (tests/golden fixture), or None"""
    name = w.get("ckpt")
    if not name:
        return None
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    return g["theta"].astype(np.float32) if g["layers"].tolist() == w["layers"] else None


def cpu_baseline(w, theta=None, budget_s=25.0):
    """Reference method on the host: nested autograd (oracle/torch_ref.py), fp32: median of 8 closures after
    2 warm-ups (fewer only if the time budget runs out), fastest and slowest reported."""
    from oracle import torch_ref
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    net = torch_ref.make_dnn(w["layers"], w["lb"], w["ub"], seed=1234)
    if theta is not None:
        torch_ref.load_flat_theta(net, theta)
    rng = np.random.default_rng(1234)
    lb, ub = np.array(w["lb"]), np.array(w["ub"])
    n = 20_000
    X = torch.tensor((lb + (ub - lb) * rng.random((n, 2))).astype(np.float32))
    x, t = X[:, 0:1].contiguous(), X[:, 1:2].contiguous()
    torch_ref.loss_f_and_grad(w["eq"], net, x[:2000], t[:2000])  # warm-ups
    torch_ref.loss_f_and_grad(w["eq"], net, x, t)
    times = []
    t_start = time.time()
    while len(times) < 8 and (len(times) < 3 or (time.time() - t_start) < budget_s):
        t0 = time.time()
        torch_ref.loss_f_and_grad(w["eq"], net, x, t)
        times.append(time.time() - t0)
    per = float(np.median(times))
    out = {"value": n / per, "unit": "collocation evals/s", "cores": cores, "kind": "port",
           "sample": "median of %d closures (after 2 warm-ups) of nested-autograd net_f+backward at N=%d (oracle/torch_ref.py, fp32, "
                     "torch CPU; fastest %.3e, slowest %.3e evals/s)" % (len(times), n, n / min(times), n / max(times))}
    # the C Taylor-jet oracle (same algorithm as the kernels) on the same cores, for context
    try:
        from oracle.pyoracle import Oracle
        o = Oracle(w["eq"], w["layers"], w["lb"], w["ub"])
        th = torch_ref.flat_theta(net).numpy()
        Xn = X.numpy()
        n2 = 100_000
        X2 = (lb + (ub - lb) * rng.random((n2, 2))).astype(np.float32)
        o.loss_grad(th, Xn[:1000, 0], Xn[:1000, 1], dtype=np.float32)
        tj = []
        for _ in range(3):
            t0 = time.time()
            o.loss_grad(th, X2[:, 0], X2[:, 1], scale=1.0 / n2, dtype=np.float32)
            tj.append(time.time() - t0)
        out["jet_c_port_value"] = n2 / float(np.median(tj))
        out["jet_c_port_sample"] = "median of 3 closures of the C Taylor-jet oracle (OpenMP, fp32) at N=%d" % n2
    except Exception as ex:  # the oracle is optional for the bench
        out["jet_c_port_value"] = None
        out["jet_c_port_sample"] = "unavailable: %s" % ex
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kdv7", choices=sorted(WORKLOADS))
    ap.add_argument("--points", type=int, default=0, help="override points per GPU")
    ap.add_argument("--kernel", default="auto", choices=["auto", "valu", "mfma"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weights", default="ckpt", choices=["ckpt", "xavier"],
                    help="ckpt: the reference's shipped checkpoint where one of the workload's shape exists "
                         "(tests/golden fixture; SURVEY 8(d)), else xavier; xavier: xavier-normal, seed 1234")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's points PER GPU; strong: the workload's points IN TOTAL "
                         "(BASELINE: 7kdv N_f=1M on 1 and 8 GPUs; CH 10M over 8), sharded over the ranks")
    args = ap.parse_args()

    verbose = os.environ.get("PINN_BENCH_VERBOSE") == "1"
    if verbose:      # where is a stuck rank?  Python stacks of all threads, every 45 s
        import faulthandler
        faulthandler.dump_traceback_later(45, repeat=True, file=sys.stderr)

    def say(msg):
        if verbose:
            print("[rank %s] %.2f %s" % (os.environ.get("RANK", "0"), time.time() % 1000, msg), file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # one rank per GPU; PINN_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks
    # (ranks then share devices and the all-reduce goes through the host) - never a measurement
    backend = os.environ.get("PINN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    from pde_amd import scripts
    from pde_amd import _native

    w = dict(WORKLOADS[args.workload])
    if args.scaling == "strong":
        # fixed total work: BASELINE's totals (CH is quoted at 10M over the node; the others at their N_f)
        n_global = args.points or {"ch": 10_000_000}.get(args.workload, w["n"])
        q, rem = divmod(n_global, world)
        n_local = q + (1 if rank < rem else 0)
        w["name"] = w["name"].replace(" per GPU", "").replace("N_f=1.25M (10M over 8)", "N_f=10M") \
            + " [strong scaling: %d points in total over %d GPU(s)]" % (n_global, world)
    else:
        if args.points:
            w["n"] = args.points
            w["name"] += " [points per GPU overridden: %d]" % args.points
        n_local = w["n"]
        n_global = n_local * world
    fam = getattr(scripts, w["family"])
    cls = fam.PI.PhysicsInformedNN
    cls = type("BenchPINN", (cls,), {"kernel": {"auto": 0, "valu": 1, "mfma": 2}[args.kernel]})
    torch.manual_seed(1234)            # identical xavier weights on every rank
    model = cls(w["layers"], np.array(w["lb"]), np.array(w["ub"]), process_group=group)
    theta_ck = checkpoint_theta(w) if args.weights == "ckpt" else None
    weights_desc = "xavier-normal weights seed 1234"
    if theta_ck is not None:
        with torch.no_grad():
            model.dnn.flat_theta().copy_(torch.from_numpy(theta_ck))
        weights_desc = "weights of the reference's shipped checkpoint (tests/golden/%s.npz)" % w["ckpt"]
    rng = np.random.default_rng(1234 + rank)
    lb, ub = np.array(w["lb"]), np.array(w["ub"])
    Xf = (lb + (ub - lb) * rng.random((n_local, 2))).astype(np.float32)
    n0 = nb = 50
    rs = np.random.default_rng(99)
    x0 = lb[0] + (ub[0] - lb[0]) * rs.random((n0, 1))
    tb = lb[1] + (ub[1] - lb[1]) * rs.random((nb, 1))
    u0 = 2.0 / np.cosh(x0) ** 2
    # every rank holds its own shard already: bypass the splitter
    model.group = None
    if w["eq"] == "gnss":
        model.set_training_data(x0, u0, 0.1 * u0, tb, Xf)
    elif w["eq"] == "burgers":
        model.set_training_data(x0, -np.sin(np.pi * x0), tb, Xf, 0.01 / np.pi)
    else:
        model.set_training_data(x0, u0, tb, Xf)
    model.group = group
    model.n_f_global = n_global

    def step():
        for p in model.dnn.parameters():
            p.grad = None
        if world > 1:
            # the sharded training closure itself (pde_amd/pinn.py::_loss_terms_sharded), minus the data terms:
            # fused kernel pass into the packed buffer, ONE all-reduce on RCCL, gradient handed to .grad
            return model._loss_terms_sharded(data_terms=False)["loss_f"]
        loss_f = model.loss_f()       # fused kernel pass
        loss_f.backward()             # hands the gradient to .grad (P-vector ops)
        return loss_f

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    say("model and data ready")
    for _ in range(args.warmup):
        step()
    say("warm-up launched")
    sync()
    say("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    say("timed steps done")
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_global * args.steps / elapsed

    # the closure's one collective alone: [grad (P), sum f^2 (nf)] floats, SURVEY 8(e)
    allreduce_us = None
    if world > 1:
        eng0 = model.engine_f
        msg = torch.zeros(eng0.P + model._T_SIZE, device=dev)     # the closure's packed buffer
        from pde_amd.nn import all_reduce_sum       # RCCL: the device buffer goes straight in; gloo rehearsal: staged through the host
        for _ in range(5):
            all_reduce_sum(msg, group)
        sync()
        t_ar = time.perf_counter()
        for _ in range(50):
            all_reduce_sum(msg, group)
        torch.cuda.synchronize()
        allreduce_us = 1e6 * (time.perf_counter() - t_ar) / 50

    say("all-reduce timed")
    # ---- dominant kernel alone, HIP events on the launch stream (torch's current stream) ----
    eng = model.engine_f
    theta = model.dnn.flat_theta()
    buf = torch.zeros(eng.P + eng.nf, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kt = []
    for _ in range(max(3, min(args.steps, 10))):
        ev0.record()
        eng.loss_grad(theta, model.xf, model.tf, 1.0 / n_global, buf[eng.P:], buf[:eng.P])
        ev1.record()
        ev1.synchronize()
        kt.append(ev0.elapsed_time(ev1) * 1e-3)
    k_avg = float(np.mean(kt))
    flops_pp = algorithmic_flops(w["layers"], w["eq"])
    flops = flops_pp * n_local
    achieved = flops / k_avg / 1e12
    # HBM bytes per launch of that kernel: from the committed rocprofv3 PMC passes (profiles/traffic.json,
    # FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE), when this exact workload / size / kernel was profiled
    traffic = None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as fh:
            ent = json.load(fh).get("%s|%d|%s" % (args.workload, n_local, {1: "valu", 2: "mfma"}[eng.kernel_kind]))
        traffic = ent["bytes"] if ent else None
    except (OSError, ValueError, KeyError):
        traffic = None

    say("kernel timed")
    # ---- full training step (IC/BC terms + Adam) ----
    train_sps = None
    graph_info = None
    if args.train_steps > 0:
        model.iter = 1
        opt = torch.optim.Adam(model.dnn.parameters())
        for _ in range(2):
            opt.zero_grad(); model.backward(); opt.step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            opt.zero_grad(); model.backward(); opt.step()
        sync()
        train_sps = args.train_steps / (time.perf_counter() - t1)
        # same iterations replayed from a captured HIP graph (single process only)
        if world == 1:
            model.use_hip_graph = True
            model.iter = 1
            with contextlib.redirect_stdout(sys.stderr):   # train() prints the reference's log lines
                model.train(max(args.train_steps, 20) + 3, LBFGS_max_iter=0)   # 3 eager iterations + capture + replays
            sync()
            gs = getattr(model, "graph_stats", None)
            graph_info = {"replays": gs["replays"], "train_steps_per_s": gs["replays"] / gs["seconds"]} if gs else None

    say("training steps timed")
    if rank == 0:
        line = {
            "metric": "collocation residual+grad evals/sec",
            "value": value,
            "unit": "collocation evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (uniform collocation points, %s)" % weights_desc
                    + ("" if backend == "nccl" else " - REHEARSAL over %s, not a measurement" % backend),
            "config": {"workload": w["name"], "equation": w["eq"], "layers": w["layers"],
                       "points_per_gpu": n_local, "points_total": n_global, "scaling": args.scaling,
                       "parallelism": "dp%d over collocation points" % world,
                       "weights": weights_desc, "kernel": {1: "valu", 2: "mfma"}[eng.kernel_kind]},
            "allreduce_us": allreduce_us,
            "train_steps_per_s": train_sps,
            "train_hip_graph": graph_info,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic,
                         "kernel_ms": 1e3 * k_avg, "kernel_ms_min": 1e3 * min(kt), "kernel_ms_max": 1e3 * max(kt),
                         "kernel_launches_timed": len(kt),
                         "algorithmic_flops_per_point": flops_pp,
                         "note": "FP32 compute roofline (vector == f32-input MFMA peak); "
                                 "algorithmic HBM traffic is 8 B/point (x, t in); `traffic` = HBM bytes per launch "
                                 "looked up in profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                 "passes of this workload, not measured in this run; null if this size was not "
                                 "profiled), almost all of it the per-wave jet stash (DESIGN.md)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, theta_ck)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
