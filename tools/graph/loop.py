"""Trainer(graph=True) for N iterations with a synchronisation after each: where does a replay go wrong?
python tools/graph/loop.py <config> <n> [nofinish|noposeadam|frozen|eager]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
name, n = sys.argv[1], int(sys.argv[2])
flags = set(a for a in sys.argv[3:] if "=" not in a)
cfg = S_.scene_config(name, "stage0")
for a in sys.argv[3:]:
    if "=" in a:
        k, v = a.split("=")
        cfg[k] = [int(x) for x in v.split(",")] if "," in v else (float(v) if "." in v else int(v))
if "tv_off" in flags:
    cfg["tv_density"] = cfg["tv_app"] = 0.0
dev = torch.device("cuda", 0)
tr = S_.Trainer(cfg, dev, graph="eager" not in flags, dead_work=True)
for k in range(n):
    if "frozen" in flags and k == 3:
        tr.rng.frozen = True
    loss = tr.step()
    if "nofinish" in flags:
        tr.it += 1
    elif "noposeadam" in flags:
        tr.opt.step()
        tr.it += 1
    else:
        tr.finish_step()
    torch.cuda.synchronize()
    print(k, float(loss), len(tr._graphs), flush=True)
print("done", flush=True)
