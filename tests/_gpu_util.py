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


# element-wise gradient tolerance next to the max-norm one: |err| <= 2e-3 |ref| + 4e-5 max|ref| for every
# entry (fp32 sums of ~1e5 atomically accumulated terms carry an absolute noise floor tied to the tensor's
# scale; above it every entry must be right to 0.2 %)
ELEM = (2e-3, 4e-5)


def kink_free_rays(O, sd_s, cfg_s, sd_d, cfg_d, rays, ts, xyz, z, valid, rt, r_s, r_d, outs, eps=2e-6):
    """Rays none of whose samples sits on a non-differentiable point of the path, so that the gradient
    comparison is deterministic: a hidden pre-activation of any MLP within eps x (the layer's mean
    magnitude) of 0 (a 1-ulp difference between the GPU and the CPU flips that relu), a relu density
    feature likewise, a weight within
    1e-6 of the app-mask threshold, a clamped / relu'd compositor output within eps of its kink.
    r_s / r_d: the oracle's field_forward tuples, outs: its raw2outputs tuple.  Returns bool [N]."""
    import torch
    import torch.nn.functional as F
    with torch.no_grad():
        sd_s = {k: v.detach() for k, v in sd_s.items()}
        sd_d = {k: v.detach() for k, v in sd_d.items()}
        aabb = cfg_d["aabb"]
        N, S = z.shape
        xn = O.normalize_coord(xyz.detach(), aabb).reshape(-1, 3)
        tt = ts[:, None].expand(N, S).reshape(-1)
        vflat = valid.reshape(-1)
        # margins are RELATIVE to each layer's own scale (the initialiser's pre-activations are O(0.1), and so
        # is the GPU / CPU rounding difference: ~3e-7 of that scale): |pre| < 2e-6 mean|pre|
        near0 = lambda pre: (pre.abs() < eps * pre.abs().mean()).any(-1)
        lin = lambda x, sd, name: F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))
        pe = O.positional_encoding
        risk = torch.zeros(N * S, dtype=torch.bool)
        # warp MLP (time branch + xyz branch), evaluated on every sample
        h1 = lin(torch.cat([tt[:, None], pe(tt[:, None], 8)], -1), sd_d, "layer1")
        risk |= near0(h1)
        tout = lin(F.relu(h1), sd_d, "layer2")
        h3 = lin(torch.cat([xn, pe(xn, 10), tout], -1), sd_d, "layer3")
        h4 = lin(F.relu(h3), sd_d, "layer4")
        risk |= near0(h3) | near0(h4)
        xw = O.normalize_coord(O.unnormalize_coord(xn, aabb) + lin(F.relu(h4), sd_d, "layer5"), aabb)
        tail = [xn, pe(xn, 10), tt[:, None], pe(tt[:, None], 8)]
        for prefix in ("density", "blending"):
            feats = O.vm_features(*O._planes(sd_d, prefix), xw, (1, 2, 4))
            hd = lin(torch.cat([feats] + tail, -1), sd_d, prefix + "_layer1")
            risk |= near0(hd) & vflat
            if prefix == "density" and cfg_d["act"] == "relu":
                fd = lin(F.relu(hd), sd_d, "density_layer2")[..., 0]
                risk |= (fd.abs() < eps * fd.abs().mean()) & vflat
        for r in (r_s, r_d):
            risk |= ((r[4].detach() - cfg_d["weight_thres"]).abs() < 1e-6).reshape(-1)
        am_d = (r_d[4].detach() > cfg_d["weight_thres"]).reshape(-1)
        af = F.linear(O.vm_features(*O._planes(sd_d, "app"), xw, (1, 2, 4)), sd_d["basis_mat.weight"])
        g1 = lin(torch.cat([af] + tail, -1), sd_d, "renderModule.mlp.0")
        g2 = lin(F.relu(g1), sd_d, "renderModule.mlp.2")
        risk |= (near0(g1) | near0(g2)) & am_d
        # static field
        fs = O.vm_features(*O._planes(sd_s, "density"), xn).sum(-1)
        if cfg_s["act"] == "relu":
            risk |= (fs.abs() < eps * fs.abs().mean()) & vflat
        am_s = (r_s[4].detach() > cfg_s["weight_thres"]).reshape(-1)
        afs = F.linear(O.vm_features(*O._planes(sd_s, "app"), xn), sd_s["basis_mat.weight"])
        _, vd = O._dists_viewdirs(rays.detach(), z, rt)
        vdf = vd.view(-1, 1, 3).expand(N, S, 3).reshape(-1, 3)
        ins = [afs, vdf, pe(afs, 2)] if cfg_s["head"] == "MLP_Fea" else [afs, pe(afs, 2)]
        s1 = lin(torch.cat(ins, -1), sd_s, "renderModule.mlp.0")
        s2 = lin(F.relu(s1), sd_s, "renderModule.mlp.2")
        risk |= (near0(s1) | near0(s2)) & am_s
        # scene flow MLP on the sample points
        x = torch.cat([xn, pe(xn, 4), tt[:, None], pe(tt[:, None], 4)], -1)
        for i in (0, 2, 4):
            pre = lin(x, sd_d, f"scene_flow_mlp.{i}")
            risk |= near0(pre)
            x = F.relu(pre)
        ok = ~risk.view(N, S).any(1)
        kink_free_rays.sample_risk = float(risk.float().mean())   # per-SAMPLE rate (S-independent), for the guards
        for k in (0, 4, 8):       # clamp(rgb_map, 0, 1)
            v = outs[k].detach()
            # (exactly 0 / exactly 1 are not kinks for this purpose: clamp's backward is inclusive on both sides)
            ok &= ~((((v.abs() < 4e-6) & (v != 0.0)) | (((v - 1.0).abs() < 4e-6) & (v != 1.0))).any(-1))
        r1 = 1.0 - outs[2].detach()
        ok &= ~((r1.abs() < 4e-6) & (r1 != 0.0))         # relu(1 - acc_map_full)
    return ok
