"""Host-side cost of one training step: wall time of enqueueing a step without waiting for the GPU (the queue is
drained before and after), and a cProfile of where the Python time goes."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
cfg = S_.scene_config(sys.argv[1] if len(sys.argv) > 1 else "nvidia", sys.argv[2] if len(sys.argv) > 2 else "stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
for _ in range(5):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(); tr.finish_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms/step (median):", sorted(t[0] for t in ts)[5] * 1e3, " enqueue + drain:", sorted(t[1] for t in ts)[5] * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step(); tr.finish_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
