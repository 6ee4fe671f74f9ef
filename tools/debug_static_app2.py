import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import rodynrf
from _gpu_util import COMMON, make_rays, oracle_cfg, oracle_sd
from oracle import rodynrf_oracle as O

def run(N, S, grid, seed):
    torch.manual_seed(seed)
    aabb = torch.tensor([[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
    kw = dict(COMMON, near_far=[0.0, 1.0], density_shift=-10.0, fea2denseAct="relu")
    st = rodynrf.TensorVMSplit(aabb, grid, 12, "cuda", shadingMode="MLP_Fea", fea_pe=2, **kw)
    rays, ts = make_rays(N, 11)
    jit = torch.rand(S, generator=torch.Generator().manual_seed(4))
    sd_s = oracle_sd(st)
    cfg_s = oracle_cfg(st)
    xyz, z, valid = O.sampleXYZ(rays, aabb, [0.0, 1.0], S, "ndc", jit)
    xyz_o = xyz.clone().requires_grad_(True)
    r_s = O.field_forward(sd_s, cfg_s, rays, ts, xyz_o, z, valid, "ndc", dynamic=False)
    g = torch.Generator().manual_seed(1)
    wr = torch.randn(N, S, 3, generator=g)
    (r_s[6] * wr).sum().backward()
    dev = "cuda"
    xyz_g = xyz.to(dev).requires_grad_(True)
    o_s = st(rays.to(dev), ts.to(dev), None, xyz_g, z.to(dev), valid.to(dev), ray_type="ndc")
    (o_s[6] * wr.to(dev)).sum().backward()
    a = xyz_g.grad.cpu(); b = xyz_o.grad
    err = (a - b).abs().amax(-1)
    scale = b.abs().max()
    bad = (err > 1e-4 * scale).nonzero()
    mask = r_s[4] > 1e-4
    print(f"grid={grid} nmask={int(mask.sum())} bad samples={len(bad)} scale={float(scale):.3e}")
    xn = O.normalize_coord(xyz, aabb)
    for n, j in bad[:12].tolist():
        print(f"   ray {n} sample {j} w={float(r_s[4][n,j]):.3e} mask={bool(mask[n,j])} xn={[round(float(v),4) for v in xn[n,j]]} gpu={[f'{float(v):.3e}' for v in a[n,j]]} ref={[f'{float(v):.3e}' for v in b[n,j]]}")
    # position in the compacted order is unknown, but report per-ray counts
    if len(bad):
        rays_bad = sorted(set(n for n, _ in bad.tolist()))
        print("   bad rays:", rays_bad[:40])
        js = sorted(set(j for _, j in bad.tolist()))
        print("   bad sample idx:", js[:70])

run(96, 70, [40, 44, 26], 5)
run(96, 70, [18, 19, 11], 5)
run(96, 70, [30, 33, 20], 5)
