"""GPU: the RCCL code path of a sharded run, executed (north_star: "each closure call all-reduces the
small parameter-gradient vector and the loss on RCCL over xGMI").

A one-GPU box cannot run two ranks on RCCL, but it can run ONE: a fresh child process initialises
`backend="nccl"` (= RCCL on ROCm) with world size 1 and the sharded closure's collective then runs on the device
buffer exactly as it does with one rank per GPU (tests/test_gpu_multirank.py rehearses two ranks, but stages the
buffer through the host over gloo).  Checked: the collective was handed a device buffer on the nccl backend and
returned it bit for bit; the sharded closure (both forms) equals the unsharded one to float32 summation order;
Adam and L-BFGS training on the sharded model follows the unsharded trajectory; and whether the closure with its
all-reduce inside can be captured into a HIP graph (recorded)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("case", ["kdv5_w50", "gnss_w100"])
def test_sharded_closure_on_rccl_world_size_one(case, tmp_path):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = str(tmp_path / "out.json")
    log_path = str(tmp_path / "worker.log")
    with open(log_path, "w") as log:
        p = subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "nccl_ws1_worker.py"), case, out],
                             env=env, stdout=log, stderr=subprocess.STDOUT)
        try:
            code = p.wait(timeout=300)
        finally:
            if p.poll() is None:
                p.kill()                       # exactly the child started here
    tail = open(log_path).read()[-3000:]
    assert code == 0, tail
    r = json.load(open(out))
    print("RCCL world-size-1 run:", json.dumps({k: v for k, v in r.items() if k != "collectives"}))
    assert r["backend"] == "nccl"
    # every collective of the closures: device buffer, nccl backend, bit-identical result at world size 1
    assert len(r["collectives"]) == 2 and all(c[0] and c[1] == "nccl" and c[2] for c in r["collectives"]), r["collectives"]
    for impl in ("direct", "autograd"):
        assert r["loss_rel_" + impl] < 2e-6 and r["grad_rel_" + impl] < 2e-6, (impl, r)
    # RCCL's all-reduce is captured like a kernel: the closure replayed from a HIP graph gives the eager numbers, and
    # train(use_hip_graph=True) on the sharded model replays the Adam iteration and the L-BFGS closure from graphs
    assert r["capture"] == "ok", r["capture"]
    assert r["capture_loss_rel"] < 2e-6 and r["capture_grad_rel"] < 2e-6, r
    assert r["graph_replays"] > 10 and r["adam_graph_theta_rel"] < 1e-4 and r["lbfgs_graph_captured"], r
    assert r["loss_end_graph"] < r["loss_adam"][0], r
    assert r["finite"] and r["lbfgs_module"] == "pde_amd.lbfgs"
    assert r["adam_theta_rel"] < 1e-4, r
    assert r["loss_end"][0] < r["loss_adam"][0] and r["loss_end"][1] < r["loss_adam"][1], r
    assert min(r["lbfgs_evals"]) > 1
