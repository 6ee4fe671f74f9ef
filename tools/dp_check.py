"""Two-rank data-parallel step on ONE GPU (gloo), used by tests/test_gpu_trainer.py: every rank runs Trainer.step on its
shard with the asynchronous gradient exchange; rank 0 writes the exchanged (mean over ranks) flat gradients to argv[1].
    python -m torch.distributed.run --nproc-per-node 2 tools/dp_check.py out.pt <exact 0|1> <mode allreduce|zero1>"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
S_ = importlib.import_module("robust-dynrf_amd.step")
P = importlib.import_module("robust-dynrf_amd.parallel")
out, exact, mode = sys.argv[1], sys.argv[2] == "1", sys.argv[3]
rank, local, world = P.init_distributed("gloo")
dev = torch.device("cuda", 0)
from test_gpu_trainer import dp_check_cfg   # the one small scene both sides use
cfg = dp_check_cfg(S_)
torch.manual_seed(0)
tr = S_.Trainer(cfg, dev, dp_mode=mode, dp_exact_stats=exact)
tr.it = 9000
tr.step((rank, world))
flats = []
for i, st in enumerate(tr.opt.state):
    g, lo, n = tr.opt.ex.grads(i, st["g"])      # waits for the exchange of buffer i: the SUM over ranks (of this rank's slice)
    full = torch.zeros_like(st["g"])
    full[lo: lo + n] = g / world
    if mode == "zero1":
        import torch.distributed as dist
        dist.all_reduce(full)                   # assemble the slices for the comparison
    flats.append(full.cpu())
if rank == 0:
    torch.save(flats, out)
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
