"""Quick forward timing on the GPU box (not the bench): per-kernel HIP-event times."""
import ctypes as C
import importlib
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import rodynrf
from _gpu_util import COMMON, make_rays
L = importlib.import_module("robust-dynrf_amd._lib")

N, S, grid = int(sys.argv[1]), int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")]
torch.manual_seed(20211202)
aabb = torch.tensor([[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
kw = dict(COMMON, near_far=[0.0, 1.0], density_shift=-10.0, fea2denseAct="relu")
st = rodynrf.TensorVMSplit(aabb, grid, 12, "cuda", shadingMode="MLP_Fea", fea_pe=2, **kw)
dy = rodynrf.TensorVMSplit_TimeEmbedding(aabb, grid, 12, "cuda", shadingMode="MLP_Fea_late_view", fea_pe=0, **kw)
rays, ts = make_rays(N, 7)
rays, ts = rays.cuda(), ts.cuda()

def one():
    with torch.no_grad():
        xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type="ndc", is_train=True)
        o_s = st(rays, ts, None, xyz, z, valid, ray_type="ndc")
        o_d = dy(rays, ts, None, xyz, z, valid, ray_type="ndc")
        outs = rodynrf.raw2outputs(o_s[6], o_s[7], o_d[6], o_d[7], o_d[9], o_d[2], o_d[8], rays, is_train=True, ray_type="ndc")
        sf = dy.get_forward_backward_scene_flow(xyz, ts)
    return o_s, o_d

for _ in range(3):
    o_s, o_d = one()
torch.cuda.synchronize()
print("app frac s", float((o_s[4] > 1e-4).float().mean()), "d", float((o_d[4] > 1e-4).float().mean()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = 10
for _ in range(K):
    one()
e1.record(); torch.cuda.synchronize()
print(f"eval ray-pass: {e0.elapsed_time(e1)/K:.3f} ms  -> {N/(e0.elapsed_time(e1)/K)*1e3:.0f} rays/s")
L.lib.rdrf_prof_enable(1); L.lib.rdrf_prof_reset()
for _ in range(5):
    one()
torch.cuda.synchronize()
for k in ["pack", "sample_ndc", "static_density", "static_app", "time_branch", "dyn_density", "dyn_app", "composite", "scene_flow"]:
    ms, n = C.c_double(), C.c_int()
    L.lib.rdrf_prof_get(k.encode(), C.byref(ms), C.byref(n))
    if n.value:
        print(f"  {k:16s} {ms.value/n.value*1e3:9.1f} us x{n.value}")
L.lib.rdrf_prof_enable(0)
