"""Helpers for the -m gpu parity tests: build the HIP-backed fields from a golden case / seed."""
import numpy as np
import torch

from _util import load_case

COMMON = dict(density_n_comp=[16, 4, 4], appearance_n_comp=[48, 12, 12], app_dim=27,
              alphaMask_thres=1e-4, distance_scale=25, pos_pe=6, view_pe=0, featureC=128,
              step_ratio=2.0)


def fields_from_case(name, device="cuda"):
    import rodynrf
    g, sd_s, cfg_s, sd_d, cfg_d = load_case(name)
    grid = [int(v) for v in g["meta.grid"]]
    nf = [float(v) for v in g["meta.near_far"]]
    kw = dict(COMMON, near_far=nf, density_shift=float(g["meta.density_shift"]),
              fea2denseAct=str(g["meta.act"]))
    aabb = torch.from_numpy(g["aabb"])
    st = rodynrf.TensorVMSplit(aabb, grid, 12, device, shadingMode=str(g["meta.static_head"]),
                               fea_pe=2, **kw)
    dy = rodynrf.TensorVMSplit_TimeEmbedding(aabb, grid, 12, device, shadingMode="MLP_Fea_late_view",
                                             fea_pe=0, **kw)
    st.load_state_dict(sd_s)
    dy.load_state_dict(sd_d)
    return g, st, dy, (sd_s, cfg_s, sd_d, cfg_d)


def oracle_sd(module):
    """state_dict of a HIP-backed field as CPU NCHW-contiguous tensors for the oracle."""
    return {k: v.detach().cpu().contiguous().clone() for k, v in module.state_dict().items()}


def oracle_cfg(module, head=None):
    return dict(aabb=module.aabb.detach().cpu(), act=module.fea2denseAct,
                density_shift=float(module.density_shift), distance_scale=float(module.distance_scale),
                weight_thres=float(module.rayMarch_weight_thres), view_pe=0, fea_pe=module.fea_pe,
                head=head or module.shadingMode)


def make_rays(N, seed, ray_type="ndc"):
    g = torch.Generator().manual_seed(seed)
    if ray_type == "ndc":
        o = torch.stack([torch.empty(N).uniform_(-1.45, 1.45, generator=g),
                         torch.empty(N).uniform_(-1.6, 1.6, generator=g), -torch.ones(N)], -1)
        d = torch.stack([torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1,
                         2 * torch.ones(N)], -1)
    else:
        o = torch.randn(N, 3, generator=g) * 0.3
        d = torch.randn(N, 3, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
    ts = torch.randint(0, 12, (N,), generator=g).float() * 2 / 11 - 1
    return torch.cat([o, d], -1), ts
