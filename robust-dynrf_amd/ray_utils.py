"""Ray generation for the training path (train.py:96-103,1062-1077; dataLoader/ray_utils.py:53-140;
camera.py:8-15) as one HIP kernel: flat ray ids + 6-D poses + focal -> NDC rays[N,6]."""
import ctypes as C

import torch

from . import _lib as L


class _RayGenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, poses9, focal, H, W, ndc, near):
        L.require_device(ids, poses9, focal)
        ids = ids.contiguous().long()
        poses9 = L.f32c(poses9)
        focal = L.f32c(focal.reshape(1))
        N, T = ids.shape[0], poses9.shape[0]
        rays = torch.empty(N, 6, device=ids.device)
        L.check(L.lib.rdrf_generate_rays(L.ptr(ids), L.ptr(poses9), L.ptr(focal), N, T, H, W,
                                         int(ndc), C.c_float(near), L.ptr(rays), L.stream_of(rays)),
                "rdrf_generate_rays")
        ctx.meta = (H, W, int(ndc), float(near))
        ctx.save_for_backward(ids, poses9, focal)
        return rays

    @staticmethod
    def backward(ctx, g_rays):
        ids, poses9, focal = ctx.saved_tensors
        H, W, ndc, near = ctx.meta
        N, T = ids.shape[0], poses9.shape[0]
        g_rays = L.f32c(g_rays)
        gp = torch.zeros_like(poses9)
        gf = torch.zeros_like(focal)
        L.check(L.lib.rdrf_generate_rays_bwd(L.ptr(ids), L.ptr(poses9), L.ptr(focal), N, T, H, W, ndc,
                                             C.c_float(near), L.ptr(g_rays), L.ptr(gp), L.ptr(gf),
                                             L.stream_of(g_rays)), "rdrf_generate_rays_bwd")
        return None, gp, gf.reshape(()), None, None, None, None


def generate_rays(ray_idx, poses9, focal, H, W, ndc=True, near=1.0):
    focal = torch.as_tensor(focal, dtype=torch.float32, device=ray_idx.device)
    return _RayGenFn.apply(ray_idx, poses9, focal, int(H), int(W), bool(ndc), float(near))


def ids2pixel(W, H, ids):
    return ids % W, (ids // W) % H, ids // (W * H)
