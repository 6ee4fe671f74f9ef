"""Which part of a training iteration survives HIP-graph capture (torch.cuda.graph)?  python tools/graph/probe.py <what> [config] [stage]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
what = sys.argv[1]
cfg = S_.scene_config(sys.argv[2] if len(sys.argv) > 2 else "nvidia_no_poses", sys.argv[3] if len(sys.argv) > 3 else "stage0")
if os.environ.get("PROBE_SMALL"):
    cfg.update(n_samples=13, grid=[17, 19, 11])
dev = torch.device("cuda", 0)
tr = S_.Trainer(cfg, dev)
tr.it = 30000
tr.rng = S_.GraphRng(dev)
tr.rng.frozen = True
bs = cfg["batch_size"]
ids = (tr.data.batch(tr.it, bs, 0).clone(), tr.data.batch(tr.it, bs, 1).clone())


def body():
    if not torch.cuda.is_current_stream_capturing():
        tr.rng.begin()
    else:
        tr.rng._cur, tr.rng._ncoin = 64, 0
    b = tr.data.make_batch(tr.it, bs, ids=ids)
    if what == "batch":
        return b["rgb"].sum()
    if what == "rays":
        return tr.rays_for(b["ids"]).sum()
    if what == "sample":
        rays = tr.rays_for(b["ids"]).detach()
        j, jo = tr.rng.jitter(cfg["n_samples"], cfg["ray_type"], dev)
        return S_.sampleXYZ(tr.dy, rays, cfg["n_samples"], ray_type=cfg["ray_type"], is_train=True, jitter=j, jitter_outer=jo)[0].sum()
    if what in ("static", "dynamic", "pass"):
        rays = tr.rays_for(b["ids"]).detach()
        j, jo = tr.rng.jitter(cfg["n_samples"], cfg["ray_type"], dev)
        xyz, z, valid = S_.sampleXYZ(tr.dy, rays, cfg["n_samples"], ray_type=cfg["ray_type"], is_train=True, jitter=j, jitter_outer=jo)
        if what == "pass":
            return S_.ray_pass(tr.st, tr.dy, rays, b["ts"], cfg["n_samples"], cfg["ray_type"], tr.rng)[2][0].sum()
        with torch.no_grad():
            f = tr.st if what == "static" else tr.dy
            return f(rays, b["ts"], None, xyz, z, valid, is_train=True, ray_type=cfg["ray_type"])[7].sum()
    if what == "losses":
        ld, ls = tr.losses(b)
        return (ld + ls).detach()
    if what == "bwd_s":
        ld, ls = tr.losses(b)
        tr.opt.zero_grad()
        ls.backward()
        return ls.detach()
    if what == "bwd_d":
        ld, ls = tr.losses(b)
        tr.opt.zero_grad()
        ld.backward()
        return ld.detach()
    if what == "full":
        ld, ls = tr._forward_backward(b, tv_between=False)
        return (ld + ls).detach()
    raise SystemExit(what)


for _ in range(2):
    r = body()
torch.cuda.synchronize()
print(what, "eager", float(r), flush=True)
tr.terms, tr.last, r = None, {}, None
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream(dev)
with torch.cuda.graph(g, stream=st):
    r = body()
print(what, "captured", flush=True)
g.replay()
torch.cuda.synchronize()
print(what, "replayed", float(r), flush=True)
tr.rng.frozen = False
for k in range(int(os.environ.get("PROBE_REPLAYS", 0))):
    tr.it += 1
    ids[0].copy_(tr.data.batch(tr.it, bs, 0))
    ids[1].copy_(tr.data.batch(tr.it, bs, 1))
    tr.rng.begin()
    g.replay()
    torch.cuda.synchronize()
    print(what, "replay", k, float(r), flush=True)
