"""Which source lines of the package issue the small ATen launches of one training step (fills, cats, adds, copies, index):
a TorchFunctionMode counts the calls of allocation / elementwise ops by the innermost frame inside robust-dynrf_amd (the
autograd engine's own accumulations happen in C++ and are listed by torch_ops.py, not here).
    python tools/aten_sites.py [config] [stage]"""
import collections
import importlib
import os
import sys
import traceback

import torch
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
cfg = S_.scene_config(sys.argv[1] if len(sys.argv) > 1 else "nvidia", sys.argv[2] if len(sys.argv) > 2 else "stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
for _ in range(3):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
WATCH = ("zeros", "zeros_like", "new_zeros", "zero_", "fill_", "cat", "stack", "add", "add_", "sub", "mul", "clone", "contiguous", "copy_",
         "__getitem__", "index_select", "clamp", "neg", "where", "sum", "mean", "full", "ones", "ones_like", "to", "float", "__add__", "__sub__",
         "__mul__", "__radd__", "__neg__", "__iadd__", "expand", "repeat", "arange")
sites = collections.Counter()


class Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name in WATCH:
            fr = None
            for f in reversed(traceback.extract_stack()[:-1]):
                if "robust-dynrf_amd" in f.filename:
                    fr = f"{os.path.basename(f.filename)}:{f.lineno}"
                    break
            sites[(name, fr)] += 1
        return func(*args, **(kwargs or {}))


N = 2
with Mode():
    for _ in range(N):
        tr.step(); tr.finish_step()
torch.cuda.synchronize()
for (name, fr), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / N:6.1f}  {name:14s} {fr}")
