"""Ray generation for the training path (train.py:96-103,1062-1077; dataLoader/ray_utils.py:53-140;
camera.py:8-15) as one HIP kernel: flat ray ids + 6-D poses + focal -> NDC rays[N,6]."""
import ctypes as C

import torch

from . import _lib as L


class _RayGenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, poses9, focal, H, W, ndc, near, uv, view_shift):
        L.require_device(ids, poses9, focal)
        ids = ids.contiguous().long()
        poses9 = L.f32c(poses9)
        ctx.focal_shape = focal.shape
        focal = L.f32c(focal.reshape(1))
        if uv is not None:
            L.require_device(uv)
            uv = L.f32c(uv)
            if uv.shape != (ids.shape[0], 2):
                raise L.RdrfError("generate_rays: uv must be [N,2] pixel coordinates")
        N, T = ids.shape[0], poses9.shape[0]
        rays = torch.empty(N, 6, device=ids.device)
        L.check(L.lib.rdrf_generate_rays_uv(L.ptr(ids), L.ptr(uv), int(view_shift), L.ptr(poses9), L.ptr(focal), N, T,
                                            H, W, int(ndc), C.c_float(near), L.ptr(rays), L.stream_of(rays)),
                "rdrf_generate_rays")
        ctx.meta = (H, W, int(ndc), float(near), int(view_shift))
        ctx.save_for_backward(ids, poses9, focal, uv)
        return rays

    @staticmethod
    def backward(ctx, g_rays):
        ids, poses9, focal, uv = ctx.saved_tensors
        H, W, ndc, near, shift = ctx.meta
        N, T = ids.shape[0], poses9.shape[0]
        g_rays = L.f32c(g_rays)
        gp = torch.zeros_like(poses9)
        gf = torch.zeros_like(focal)
        L.check(L.lib.rdrf_generate_rays_uv_bwd(L.ptr(ids), L.ptr(uv), shift, L.ptr(poses9), L.ptr(focal), N, T, H, W,
                                                ndc, C.c_float(near), L.ptr(g_rays), L.ptr(gp), L.ptr(gf),
                                                L.stream_of(g_rays)), "rdrf_generate_rays_bwd")
        return None, gp, gf.reshape(ctx.focal_shape), None, None, None, None, None, None


def generate_rays(ray_idx, poses9, focal, H, W, ndc=True, near=1.0, uv=None, view_shift=0):
    """rays[N,6] of the flat ray ids (train.py:1062-1077).  `uv` [N,2] replaces the pixel centres by
    (column + 0.5 + flow_x, row + 0.5 + flow_y) and `view_shift` = +1 / -1 selects the next / previous
    frame's camera (clamped), i.e. the flow-displaced rays of train.py:1433-1460, 1968-1990.  Differentiable
    wrt poses9 [T,9] and focal (scalar or [1])."""
    focal = torch.as_tensor(focal, dtype=torch.float32, device=ray_idx.device)
    return _RayGenFn.apply(ray_idx, poses9, focal, int(H), int(W), bool(ndc), float(near), uv, int(view_shift))


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx):
        L.require_device(table, idx)
        idx = idx.contiguous().long()
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return table.index_select(0, idx)

    @staticmethod
    def backward(ctx, g_rows):
        (idx,) = ctx.saved_tensors
        R = ctx.shape[0]
        Cc = 1
        for d in ctx.shape[1:]:
            Cc *= d
        g_rows = L.f32c(g_rows)
        g_table = torch.zeros(ctx.shape, device=g_rows.device)
        L.check(L.lib.rdrf_rows_scatter_add(L.ptr(idx), L.ptr(g_rows), idx.numel(), R, Cc, L.ptr(g_table), L.stream_of(g_rows)),
                "rdrf_rows_scatter_add")
        return g_table, None


def gather_rows(table, idx):
    """table[idx] for a small per-frame table (the [T,3,4] camera matrices of the neighbour frames,
    `allposes_refine[view +- 1]`, train.py:1895-1948) with a backward that accumulates per workgroup in LDS
    (rdrf_rows_scatter_add) instead of torch's sort-based index backward; idx [N] int64 in [0, T)."""
    if not table.requires_grad or not torch.is_grad_enabled():
        return table[idx]
    return _GatherRowsFn.apply(table, idx)


def ids2pixel(W, H, ids):
    return ids % W, (ids // W) % H, ids // (W * H)


def pose_to_mtx(pose9):
    """camera.py:8-15: 6-D rotation (two 3-vectors, Gram-Schmidt) + translation -> c2w [.,3,4] with
    columns (b1, b2, b1 x b2, t).  A [T,9] per-frame table, not per-ray work: plain torch."""
    a1, a2, t = pose9[..., 0:3], pose9[..., 3:6], pose9[..., 6:9]
    b1 = a1 / a1.norm(dim=-1, keepdim=True)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = b2 / b2.norm(dim=-1, keepdim=True)
    return torch.stack([b1, b2, torch.linalg.cross(b1, b2), t], dim=-1)
