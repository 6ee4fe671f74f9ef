"""Soak run of the trainer (python tools/long_run.py [steps]): the benchmark configuration for `steps` iterations with a
grid upsample (+ Adam rebuild / lr reset, train.py:2582-2606) a third of the way in: loss, finiteness of every parameter,
throughput per window and peak memory -- a leak, a drift of the step time or a NaN shows up here, not in a 200-step
bench."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S_ = importlib.import_module("robust-dynrf_amd.step")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = S_.scene_config("nvidia", "stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
up_at, win = steps // 3, max(1, steps // 15)
t0 = tw = time.perf_counter()
for i in range(steps):
    if i == up_at:
        g = [int(v * 1.26) for v in cfg["grid"]]
        tr.upsample(g, int(cfg["n_samples"] * 1.26))
        print(f"-- step {i}: upsampled to {g}, {tr.cfg['n_samples']} samples per ray", flush=True)
    loss = tr.step()
    tr.finish_step()
    if (i + 1) % win == 0 or i == steps - 1:
        torch.cuda.synchronize()
        now = time.perf_counter()
        fin = all(bool(torch.isfinite(p).all()) for m in (tr.st, tr.dy) for p in m.parameters())
        print(f"step {i + 1:5d} loss {float(loss):.5f} finite {fin} {win / (now - tw) * cfg['batch_size'] / 1e3:7.1f} k rays/s "
              f"mem {torch.cuda.memory_allocated() / 2 ** 30:.2f} GB (peak {torch.cuda.max_memory_allocated() / 2 ** 30:.2f})", flush=True)
        assert fin
        tw = now
print(f"total {time.perf_counter() - t0:.1f} s")
