#!/usr/bin/env python3
"""World-size-1 RCCL smoke on a 1-GPU box: every torch.distributed call bench.py makes with the nccl backend, one by one, with a
progress line before each (so a crash in the runtime names its call)."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from forgex_amd import dist as fxdist


def say(msg):
    print("[rccl_smoke]", msg, file=sys.stderr, flush=True)


dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
say("init_process_group")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
say("barrier")
dist.barrier()
say("all_reduce")
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
say("gather_results L=128")
n = 100000
f = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
a = (torch.arange(n, device=dev) % 100 + 1).to(torch.int32) * f
res = fxdist.gather_results(f, a, a, n, 128)
assert torch.equal(res[0], f) and torch.equal(res[1], a)
say("gather_results L=256")
res = fxdist.gather_results(f, a, a, n, 256)
assert torch.equal(res[0], f) and torch.equal(res[1], a)
say("barrier + destroy")
dist.barrier()
dist.destroy_process_group()
say("OK")
