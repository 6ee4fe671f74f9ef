"""captured-vs-eager gradient distance next to the eager-vs-eager distance (two runs of ONE path: the order of the fp32
atomics), per gradient buffer: the bound of tests/test_gpu_graph.py.  python tools/graph/noise.py <config> [steps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
S_ = importlib.import_module("robust-dynrf_amd.step")
from test_gpu_graph import _sync_eager_to   # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
cfg = S_.scene_config(name, "stage0")
tr_g = S_.Trainer(dict(cfg), dev, graph=True)
twins = []
for _ in range(2):
    t = S_.Trainer(dict(cfg), dev)
    t.rng = S_.GraphRng(dev)
    t.rng.frozen = True
    twins.append(t)
tr_g.it = 30000
names = ["static", "dynamic", "poses", "fov"]


def grads(t):
    g = [x.detach().clone() for x in t.grad_flats]
    if t.optimize_poses:
        g += [t.poses.grad.clone(), t.fov.grad.clone()]
    return g


for k in range(steps):
    tr_g.step()
    gg = grads(tr_g)
    ge = []
    for t in twins:
        _sync_eager_to(t, tr_g)
        t._forward_backward(t.data.make_batch(t.it, cfg["batch_size"]), tv_between=True)
        ge.append(grads(t))
    row = []
    for i, (a, b, c) in enumerate(zip(gg, ge[0], ge[1])):
        row.append(f"{names[i]} g-e {float((a - b).norm() / b.norm()):.2e} e-e {float((b - c).norm() / c.norm()):.2e}")
    print(name, k, "graphs", len(tr_g._graphs), " | ".join(row), flush=True)
    tr_g.finish_step()
