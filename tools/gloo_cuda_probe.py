"""Probe: does a gloo all-reduce of a CUDA tensor complete when two ranks share one GPU?  (rehearsal mode of
bench.py depends on it.)  Prints progress lines; run under a short timeout."""
import os
import sys
import time
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
log = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                        "gloo_probe_rank%d.log" % rank), "w")


def say(msg):
    log.write("%.1f %s\n" % (time.time() % 1000, msg)); log.flush()


torch.cuda.set_device(0)
say("init")
dist.init_process_group("gloo")
say("barrier")
dist.barrier()
say("cpu all_reduce")
c = torch.ones(4)
dist.all_reduce(c)
say("cpu ok %s" % c.tolist())
g = torch.ones(4, device="cuda")
torch.cuda.synchronize()
say("cuda all_reduce")
dist.all_reduce(g)
torch.cuda.synchronize()
say("cuda ok %s" % g.tolist())
dist.destroy_process_group()
say("done")
