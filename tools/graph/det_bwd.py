"""is d(loss)/d(xyz_sampled) of the dynamic field bit-reproducible?  (per-sample outputs of the backward-data kernels: no atomics)"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rodynrf
from _gpu_util import fields_from_case, make_rays
case, N, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g, st, dy, _ = fields_from_case(case)
rt = str(g["meta.ray_type"])
rays, ts = (t.cuda() for t in make_rays(N, 3, rt))
xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type=rt, is_train=False)
gen = torch.Generator().manual_seed(1)
gs, gb = torch.randn(N, S, generator=gen).cuda(), torch.randn(N, S, generator=gen).cuda()
sig = collections.Counter()
for rep in range(int(os.environ.get("REPS", 30))):
    x = xyz.clone().requires_grad_(True)
    o = dy(rays, ts, None, x, z, valid, is_train=True, ray_type=rt)
    ((o[7] * gs).sum() + (o[2] * gb).sum()).backward()
    sig[int(x.grad.view(torch.int32).long().sum())] += 1
print(case, N, S, "d/d xyz over", sum(sig.values()), "runs, distinct:", sorted(sig.values(), reverse=True))
