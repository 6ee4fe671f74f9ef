"""is d(loss)/d(xyz_sampled) through the APPEARANCE phase bit-reproducible?  loss = <rgb, g>: the gradient runs through every
backward-data layer of k_static_app_bwd / k_dyn_app_bwd (bf16 x 3 with split storage, odd block counts) into the per-sample
coordinate gradients.   python tools/graph/det_bwd_app.py <case> <N> <S>   (REPS=30)
Under RDRF_DETERMINISTIC=1 (librodynrf_det.so) the parameter gradients (flat buffer, fixed-point accumulation) are compared too."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
import rodynrf
L = importlib.import_module("robust-dynrf_amd._lib")
from _gpu_util import fields_from_case, make_rays
case, N, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g, st, dy, _ = fields_from_case(case)
rt = str(g["meta.ray_type"])
rays, ts = (t.cuda() for t in make_rays(N, 3, rt))
xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type=rt, is_train=False)
gen = torch.Generator().manual_seed(1)
gr = torch.randn(N, S, 3, generator=gen).cuda()
for name, f in (("static", st), ("dynamic", dy)):
    sig, sigp = collections.Counter(), collections.Counter()
    nz = 0
    if L.DETERMINISTIC:
        f.fused_grad = True
    for rep in range(int(os.environ.get("REPS", 30))):
        if L.DETERMINISTIC:
            f.zero_grad_fused()
        x = xyz.clone().requires_grad_(True)
        o = f(rays, ts, None, x, z, valid, is_train=True, ray_type=rt)
        (o[6] * gr).sum().backward()
        nz = int((x.grad != 0).sum())
        sig[int(x.grad.view(torch.int32).long().sum())] += 1
        if L.DETERMINISTIC:
            f.det_fold_()
            assert float(f._gflat.abs().max()) > 0
            sigp[int(f._gflat.view(torch.int32).long().sum())] += 1
    print(case, N, S, name, "d/d xyz through rgb over", sum(sig.values()), "runs, nonzero", nz, "distinct:", sorted(sig.values(), reverse=True),
          *(["parameter gradients distinct:", sorted(sigp.values(), reverse=True)] if sigp else []))
