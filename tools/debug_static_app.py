import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import rodynrf
from _gpu_util import COMMON, make_rays, oracle_cfg, oracle_sd
from oracle import rodynrf_oracle as O

def run(N, S, grid, seed, lossmode):
    torch.manual_seed(seed)
    aabb = torch.tensor([[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
    kw = dict(COMMON, near_far=[0.0, 1.0], density_shift=-10.0, fea2denseAct="relu")
    st = rodynrf.TensorVMSplit(aabb, grid, 12, "cuda", shadingMode="MLP_Fea", fea_pe=2, **kw)
    rays, ts = make_rays(N, 11)
    jit = torch.rand(S, generator=torch.Generator().manual_seed(4))
    sd_s = oracle_sd(st)
    for v in sd_s.values(): v.requires_grad_(True)
    cfg_s = oracle_cfg(st)
    xyz, z, valid = O.sampleXYZ(rays, aabb, [0.0, 1.0], S, "ndc", jit)
    r_s = O.field_forward(sd_s, cfg_s, rays, ts, xyz, z, valid, "ndc", dynamic=False)
    g = torch.Generator().manual_seed(1)
    wr = torch.randn(N, S, 3, generator=g)
    if lossmode == "lin":
        Lr = (r_s[6] * wr).sum()
    else:
        Lr = ((r_s[6] * r_s[4][..., None]).sum(1) ** 2).mean()
    ks = list(sd_s.keys())
    gref = torch.autograd.grad(Lr, [sd_s[k] for k in ks], allow_unused=True)
    dev = "cuda"
    o_s = st(rays.to(dev), ts.to(dev), None, xyz.to(dev), z.to(dev), valid.to(dev), ray_type="ndc")
    if lossmode == "lin":
        Lg = (o_s[6] * wr.to(dev)).sum()
    else:
        Lg = ((o_s[6] * o_s[4][..., None]).sum(1) ** 2).mean()
    Lg.backward()
    mm = int(((o_s[4].cpu() > 1e-4) != (r_s[4] > 1e-4)).sum())
    print(f"N={N} S={S} grid={grid} loss={lossmode} mask mismatches={mm} nmask={int((r_s[4]>1e-4).sum())} fwd rgb err={float((o_s[6].detach().cpu()-r_s[6]).abs().max()):.2e}")
    own = dict(st.named_parameters())
    for k, gr in zip(ks, gref):
        if gr is None: continue
        a = own[k].grad.detach().cpu().double(); b = gr.double()
        print(f"   {k:28s} rel {float((a-b).abs().max()/b.abs().max()):.2e}")

run(96, 70, [40, 44, 26], 5, "lin")
run(96, 70, [40, 44, 26], 5, "sq")
run(32, 13, [18, 19, 11], 5, "sq")
run(96, 70, [18, 19, 11], 5, "sq")
