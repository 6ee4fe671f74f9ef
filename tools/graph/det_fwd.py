"""is the dynamic field's forward deterministic?  inference (flat tiles), training (wave per ray), feature mode"""
import os, sys
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
names = ["xyz", "pts_ref", "blending", "xyz_prime", "weight", "3", "rgb", "sigma", "z", "dists"]
for mode in ("inference", "training"):
    res = []
    for rep in range(int(os.environ.get("REPS", 6))):
        with torch.set_grad_enabled(mode == "training"):
            o = dy(rays, ts, None, xyz, z, valid, is_train=True, ray_type=rt)
        res.append([None if t is None else t.detach().clone() for t in o])
    import collections
    sig = collections.Counter()
    for r in res:
        sig[tuple(int(t.view(torch.int32).long().sum()) for t in (r[4], r[7], r[2]))] += 1
    print(case, N, S, mode, "runs", len(res), "distinct results (weight, sigma, blending):", sorted(sig.values(), reverse=True))
    print(case, N, S, mode, "checked")
