"""Which torch ops (and which source lines) issue the small launches of one training step: torch.profiler
over 3 steps, grouped by op and by the innermost repo stack frame."""
import collections
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
cfg = S_.scene_config("nvidia", "stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
for _ in range(3):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
NS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(NS):
        tr.step(); tr.finish_step()
    torch.cuda.synchronize()
ev = prof.events()
by_op = collections.Counter()
by_line = collections.Counter()
kern = 0
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern += 1
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    n = len(e.kernels)
    by_op[e.name] += n
    frame = next((f for f in (e.stack or []) if "robust-dynrf_amd" in f or "bench.py" in f), "(autograd / other)")
    by_line[(e.name, frame.split("/")[-1][:70])] += n
print("device kernels per step:", kern / NS)
print("--- launches per step by op")
for k, v in by_op.most_common(30):
    print(f"{v / NS:8.1f}  {k}")
print("--- launches per step by (op, source line)")
for k, v in by_line.most_common(60):
    print(f"{v / NS:8.1f}  {k[0]:40s} {k[1]}")
