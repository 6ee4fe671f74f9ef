"""HIP-event time of named launch ranges over a few training steps: python tools/prof_names.py name1 name2 ... [--stage final]"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
L = importlib.import_module("robust-dynrf_amd._lib")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
stage = "final" if "--stage=final" in sys.argv else "stage0"
cfg = S_.scene_config("nvidia", stage)
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
for _ in range(3):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
L.lib.rdrf_prof_enable(1)
L.lib.rdrf_prof_reset()
NP = 5
for _ in range(NP):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
for n in args:
    ms, k = C.c_double(), C.c_int()
    L.lib.rdrf_prof_get(n.encode(), C.byref(ms), C.byref(k))
    print(f"{n:28s} {ms.value / NP:8.3f} ms/step  {k.value / NP:5.1f} launches/step  {ms.value / max(k.value, 1) * 1e3:8.1f} us each")
L.lib.rdrf_prof_enable(0)
