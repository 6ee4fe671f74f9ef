"""Per-step trace of a FRESH trainer process (python tools/step_trace.py [steps]): for every iteration the host enqueue
time, the wall time with the queue drained, the caching allocator's device-malloc count and -- every 10 steps -- the
appearance-mask fractions of both fields (the workload of the appearance kernels follows them).  Separates the three
things that can make the first steps of a run slower than its steady state: host enqueue, allocator growth, and a
workload that changes as the weights train."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sync_each = os.environ.get("TRACE_SYNC", "1") == "1"
dev = torch.device("cuda", 0)
cfg = S_.scene_config("nvidia", "stage0")
tr = S_.Trainer(cfg, dev, dead_work=True)


def fractions():
    with torch.no_grad():
        ids = tr.data.batch(0, cfg["batch_size"], 0)
        rays = tr.rays_for(ids).detach()
        ts = tr.data.ts_of(ids)
        o_s, o_d, _, _ = S_.ray_pass(tr.st, tr.dy, rays, ts, cfg["n_samples"], cfg["ray_type"], S_.StepRng(), is_train=False)
        return float((o_s[4] > 1e-4).float().mean()), float((o_d[4] > 1e-4).float().mean())


torch.cuda.synchronize()
tw = time.perf_counter()
for i in range(steps):
    if i % 10 == 0:
        fs, fd = fractions()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step()
    tr.finish_step()
    t1 = time.perf_counter()
    if sync_each or i % 10 == 9:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    if sync_each or i % 10 == 9:
        print(f"step {i:4d} enqueue {1e3 * (t1 - t0):7.2f} ms  wall {1e3 * (t2 - (t0 if sync_each else tw)) / (1 if sync_each else 10):7.2f} ms  "
              f"dev_mallocs {st.get('num_device_alloc', -1)} alloc {torch.cuda.memory_allocated() / 2 ** 30:.2f} GB "
              f"reserved {torch.cuda.memory_reserved() / 2 ** 30:.2f} GB  app_mask static {fs:.3f} dynamic {fd:.3f}", flush=True)
        tw = time.perf_counter()
