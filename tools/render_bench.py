"""render Mpix/s (whole frame and 512-ray chunks) + the per-kernel forward times of a no-grad pass: python tools/render_bench.py"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S_ = importlib.import_module("robust-dynrf_amd.step")
R = importlib.import_module("robust-dynrf_amd.renderer")
L = importlib.import_module("robust-dynrf_amd._lib")
stage = "final" if "--stage=final" in sys.argv else "stage0"
cfg = S_.scene_config("nvidia", stage)
dev = torch.device("cuda", 0)
tr = S_.Trainer(cfg, dev)
H, W = cfg["H"], cfg["W"]
ids = torch.arange(H * W, device=dev)
with torch.no_grad():
    rays_f = tr.rays_for(ids + 3 * H * W).detach()
ts_f = tr.data.ts_of(ids + 3 * H * W)


def frame(chunk):
    for c0 in range(0, H * W, chunk):
        R.render_rays(tr.st, tr.dy, rays_f[c0:c0 + chunk], ts_f[c0:c0 + chunk], N_samples=cfg["n_samples"], ray_type=cfg["ray_type"])


which = [a for a in sys.argv[1:] if not a.startswith("--")]
chunks = {"whole": (H * W,), "chunk512": (512,)}.get(which[0] if which else "", (H * W, 4096, 512))
for chunk in chunks:
    frame(chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5 if chunk > 512 else 2
    for _ in range(n):
        frame(chunk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"chunk {chunk:6d}: {H * W / dt / 1e6:6.3f} Mpix/s  {dt * 1e3:7.3f} ms/frame")
if 512 in chunks:   # the native chunk loop (rdrf_render_chunks_fwd), one stream and four
    for ns in (1, 4, 8, 16):
        R.render_chunks(tr.st, tr.dy, rays_f, ts_f, 512, N_samples=cfg["n_samples"], ray_type=cfg["ray_type"], streams=ns)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            R.render_chunks(tr.st, tr.dy, rays_f, ts_f, 512, N_samples=cfg["n_samples"], ray_type=cfg["ray_type"], streams=ns)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"native chunk loop, 512-ray chunks, {ns} stream(s): {H * W / dt / 1e6:6.3f} Mpix/s  {dt * 1e3:7.3f} ms/frame")
L.lib.rdrf_prof_enable(1)
L.lib.rdrf_prof_reset()
for chunk in chunks:
    frame(chunk)
torch.cuda.synchronize()
for k in ("sample_ndc", "static_density", "static_app", "time_branch", "dyn_density", "ray_scan", "dyn_app", "composite", "render_fused"):
    ms, c = C.c_double(), C.c_int()
    L.lib.rdrf_prof_get(k.encode(), C.byref(ms), C.byref(c))
    if c.value:
        print(f"  {k:16s} {ms.value:7.3f} ms ({c.value} launches, {ms.value / c.value * 1e3:7.1f} us each)")
L.lib.rdrf_prof_enable(0)
