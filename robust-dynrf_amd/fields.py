"""Host-side mirror of the reference field objects for the ray-batch hot path.

``TensorVMSplit`` / ``TensorVMSplit_TimeEmbedding`` keep the reference's constructor kwargs,
attribute names, ``state_dict`` keys/shapes, ``get_optparam_groups`` and the
``forward(rays_chunk, ts_chunk, timeembeddings_chunk, xyz_sampled, z_vals, ray_valid, ...)``
10-tuple (/root/reference/models/tensorBase.py:281-339, 704-850; models/tensoRF.py:11-61,
277-376), but every per-sample computation runs in the HIP kernels behind the C ABI
(include/rodynrf.h).  The nn.Linear / nn.Sequential objects here are parameter containers only.

VM factors are stored channel-last: a plane parameter has the reference's logical shape
(1,C,H,W) with strides of an [H][W][C] array, so state_dicts interchange with the reference while
one bilinear tap of 4 components is a single 16-byte load on the GPU.
"""
import ctypes as C

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .regularizers import dense_l1, tv_family, vector_diffs

MAT_MODE = [[0, 1], [0, 2], [1, 2]]
VEC_MODE = [2, 1, 0]


# --------------------------------------------------------------------------------------------
# parameter containers (same attribute / key structure as models/tensorBase.py:81-183)
# --------------------------------------------------------------------------------------------
class _Head(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("the RGB head is fused into the HIP appearance kernel; call the field")


class MLPRender_Fea(_Head):
    def __init__(self, inChanel, viewpe=6, feape=6, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * viewpe * 3 + 2 * feape * inChanel + 3 + inChanel
        self.viewpe, self.feape = viewpe, feape
        self.mlp = nn.Sequential(nn.Linear(self.in_mlpC, featureC), nn.ReLU(inplace=True),
                                 nn.Linear(featureC, featureC), nn.ReLU(inplace=True),
                                 nn.Linear(featureC, 3))
        nn.init.constant_(self.mlp[-1].bias, 0)


class MLPRender_Fea_TimeEmbedding(_Head):
    def __init__(self, inChanel, viewpe=6, feape=6, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * feape * inChanel + inChanel
        self.in_view = 2 * viewpe * 3 + 3
        self.viewpe, self.feape = viewpe, feape
        layer1 = nn.Linear(self.in_mlpC, featureC)
        layer2 = nn.Linear(featureC, featureC)
        layer3 = nn.Linear(featureC + self.in_view, 3)
        self.mlp = nn.Sequential(layer1, nn.ReLU(inplace=True), layer2, nn.ReLU(inplace=True))
        self.mlp_view = nn.Sequential(layer3)
        nn.init.constant_(self.mlp_view[-1].bias, 0)


class MLPRender_Fea_late_view(_Head):
    def __init__(self, inChanel, viewpe=6, feape=6, featureC=128):
        super().__init__()
        self.in_mlpC = 2 * feape * inChanel + inChanel + 2 * 10 * 3 + 3 + 2 * 8 * 1 + 1
        self.in_view = 2 * viewpe * 3 + 3
        self.viewpe, self.feape = viewpe, feape
        layer1 = nn.Linear(self.in_mlpC, featureC)
        layer2 = nn.Linear(featureC, featureC)
        layer3 = nn.Linear(featureC + self.in_view, 3)
        self.mlp = nn.Sequential(layer1, nn.ReLU(inplace=True), layer2, nn.ReLU(inplace=True))
        self.mlp_view = nn.Sequential(layer3)
        nn.init.constant_(self.mlp_view[-1].bias, 0)


# storage order of the XZ / YZ planes: False = [z][x|y][C] (x fastest: the two bilinear columns of a
# tap are 16-32 bytes apart and the scatter's column-split atomics merge into one L2 request each);
# True = [x|y][z][C].  The kernels take explicit strides, either works.
Z_FAST = os.environ.get("RDRF_Z_FAST", "0") == "1"


def channel_last_(t, h_fast=False):
    """Re-stride a (1,C,H,W) tensor with the component axis contiguous (values preserved):
    [H][W][C] by default, [W][H][C] when `h_fast` (used for the XZ / YZ planes, whose H axis is z:
    consecutive samples of a forward-facing ray then read / update adjacent texels)."""
    _, c, h, w = t.shape
    strides = (c * h * w, 1, c, h * c) if h_fast else (c * h * w, 1, w * c, c)
    out = torch.empty_strided(t.shape, strides, dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def _is_channel_last(t):
    return t.stride(1) == 1 or t.numel() == 0


# --------------------------------------------------------------------------------------------
# autograd bridges
# --------------------------------------------------------------------------------------------
def _vm_struct(planes, lines):
    vm = L.RdrfVM()
    for i in range(3):
        p, l = planes[i], lines[i]
        # the kernels read components with stride 1 (16-byte quads) and lines as [L][C]: a factor that
        # lost its channel-last storage (user-assigned parameter, .contiguous(), external resampling)
        # would be read with the wrong layout -- fail loudly instead
        if p.stride(1) != 1 or l.stride(1) != 1 or l.stride(2) != l.shape[1] or p.shape[1] % 4 != 0:
            raise L.RdrfError(f"VM factor {i} is not channel-last (plane strides {tuple(p.stride())}, line strides "
                              f"{tuple(l.stride())}): use fields.channel_last_()")
        vm.plane[i] = p.data_ptr()
        vm.line[i] = l.data_ptr()
        vm.C[i] = p.shape[1]
        vm.H[i] = p.shape[2]
        vm.W[i] = p.shape[3]
        vm.L[i] = l.shape[2]
        vm.sH[i] = p.stride(2)
        vm.sW[i] = p.stride(3)
    return vm


def _cfg_struct(field, ray_type, weight_thres=None):
    c = L.RdrfFieldCfg()
    ab = field._aabb_host
    for i in range(6):
        c.aabb[i] = ab[i]
    c.distance_scale = float(field.distance_scale)
    c.weight_thres = float(field.rayMarch_weight_thres if weight_thres is None else weight_thres)
    c.density_shift = float(field.density_shift)
    c.act = L.ACTS[field.fea2denseAct]
    c.ray_type = L.RAY_TYPES.get(ray_type, 2)
    c.static_head = L.HEADS.get(field.shadingMode, 0)
    return c


STATIC_KEYS = (["density_plane.%d" % i for i in range(3)] + ["density_line.%d" % i for i in range(3)]
               + ["app_plane.%d" % i for i in range(3)] + ["app_line.%d" % i for i in range(3)]
               + ["basis_mat.weight"])
DYN_VM = (["density_plane.%d" % i for i in range(3)] + ["density_line.%d" % i for i in range(3)]
          + ["blending_plane.%d" % i for i in range(3)] + ["blending_line.%d" % i for i in range(3)]
          + ["app_plane.%d" % i for i in range(3)] + ["app_line.%d" % i for i in range(3)])


def _static_struct(t):
    """t: list of tensors in TensorVMSplit._param_list() order."""
    P = L.RdrfStaticParams()
    P.density = _vm_struct(t[0:3], t[3:6])
    P.app = _vm_struct(t[6:9], t[9:12])
    for name, ten in zip(("basis", "w1", "b1", "w2", "b2", "w3", "b3"), t[12:19]):
        setattr(P, name, ten.data_ptr())
    return P


def _dynamic_struct(t):
    P = L.RdrfDynamicParams()
    P.density = _vm_struct(t[0:3], t[3:6])
    P.blending = _vm_struct(t[6:9], t[9:12])
    P.app = _vm_struct(t[12:15], t[15:18])
    names = ("basis", "rw1", "rb1", "rw2", "rb2", "rwv", "rbv", "l1w", "l1b", "l2w", "l2b", "l3w",
             "l3b", "l4w", "l4b", "l5w", "l5b", "dw1", "db1", "dw2", "db2", "bw1", "bb1", "bw2", "bb2")
    for name, ten in zip(names, t[18:43]):
        setattr(P, name, ten.data_ptr())
    for i in range(4):
        P.sfw[i] = t[43 + 2 * i].data_ptr()
        P.sfb[i] = t[44 + 2 * i].data_ptr()
    return P


# Packed weight images (rdrf_static_pack / rdrf_dynamic_pack): the MLP weights of a field do not change between the
# passes of an iteration or the chunks of a render, so the image is packed once per (weights, stream) and handed to
# the entry points through packed_fwd / packed_bwd instead of being re-packed by every call (~40 launches of ~6 us
# per training iteration).  The key is every MLP tensor's (data_ptr, torch version counter) plus the field's
# `_pack_epoch`, which optim.FlatAdam bumps because rdrf_adam_step writes the parameters through raw pointers.
PACK_CACHE = os.environ.get("RDRF_PACK_CACHE", "1") == "1"


def _attach_packed(field, P, params, backward, dynamic):
    if not PACK_CACHE or field is None:
        return
    first = 18 if dynamic else 12
    stream = torch.cuda.current_stream(params[0].device).cuda_stream
    key = (tuple((p.data_ptr(), p._version) for p in params[first:]), field._pack_epoch)
    cache = field.__dict__.setdefault("_pack_cache", {})
    # one image per (direction, stream): a re-pack into an image is ordered behind that stream's earlier readers;
    # another stream gets its own image instead of racing with kernels still reading this one
    slot = cache.get((backward, stream))
    if slot is None or slot[0] != key:
        img = slot[1] if slot is not None else torch.empty(int(L.lib.rdrf_pack_floats()), device=params[0].device)
        if dynamic:
            L.check(L.lib.rdrf_dynamic_pack(C.byref(P), int(backward), L.ptr(img), C.c_void_p(stream)), "rdrf_dynamic_pack")
        else:
            L.check(L.lib.rdrf_static_pack(C.byref(P), L.HEADS.get(field.shadingMode, 0), int(backward), L.ptr(img),
                                           C.c_void_p(stream)), "rdrf_static_pack")
        cache[(backward, stream)] = slot = (key, img)
    if backward:
        P.packed_bwd = slot[1].data_ptr()
    else:
        P.packed_fwd = slot[1].data_ptr()


def _alloc_saved(ctx, kind, N, S, dev):
    """training mode: a per-call buffer the forward fills with the activations its backward needs
    (inference -- no input requires grad -- keeps nothing)."""
    if not any(ctx.needs_input_grad):
        return None, 0
    nbytes = int(L.lib.rdrf_saved_bytes(kind, N, S))
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def _prep_inputs(rays, ts, xyz, z, valid):
    L.require_device(rays, ts, xyz, z, valid)
    rays, ts, xyz, z = L.f32c(rays), L.f32c(ts), L.f32c(xyz), L.f32c(z)
    valid = valid.contiguous()
    if valid.dtype == torch.bool:
        valid = valid.view(torch.uint8)
    return rays, ts, xyz, z, valid


class _StaticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, field, ray_type, rays, ts, xyz, z, valid, *params):
        ctx.set_materialize_grads(False)   # unused outputs arrive as None: their branch is skipped
        rays, ts, xyz, z, valid = _prep_inputs(rays, ts, xyz, z, valid)
        N, S = z.shape
        dev = z.device
        # ray_type "ndc" / "contract"; a trailing "-norgb" (forward(..., rgb=False)): the colours are not wanted
        want_rgb = not ray_type.endswith("-norgb")
        ray_type = ray_type.replace("-norgb", "")
        rgb = torch.empty(N, S, 3, device=dev) if want_rgb else None
        sigma = torch.empty(N, S, device=dev)
        weight = torch.empty(N, S, device=dev)
        dists = torch.empty(N, S, device=dev)
        saved, sbytes = _alloc_saved(ctx, 0, N, S, dev)
        ws = L.workspace(dev, (L.lib.rdrf_workspace_bytes if saved is not None else L.lib.rdrf_forward_workspace_bytes)(N, S))
        P = _static_struct(params)
        _attach_packed(field, P, params, False, False)
        cfg = _cfg_struct(field, ray_type)
        L.check(L.lib.rdrf_static_fwd(C.byref(P), C.byref(cfg), L.ptr(rays), L.ptr(ts), L.ptr(xyz),
                                      L.ptr(z), L.ptr(valid), N, S, L.ptr(rgb), L.ptr(sigma),
                                      L.ptr(weight), L.ptr(dists), L.ptr(saved), C.c_size_t(sbytes),
                                      L.ptr(ws), C.c_size_t(ws.numel()), L.stream_of(z)),
                "rdrf_static_fwd")
        ctx.field, ctx.ray_type, ctx.saved, ctx.want_rgb = field, ray_type, saved, want_rgb
        ctx.save_for_backward(rays, ts, xyz, z, valid, *params)
        return rgb, sigma, weight, dists

    @staticmethod
    def backward(ctx, g_rgb, g_sigma, g_weight, g_dists):
        rays, ts, xyz, z, valid, *params = ctx.saved_tensors
        if g_rgb is not None and not ctx.want_rgb:
            raise L.RdrfError("forward(..., rgb=False) did not run the appearance phase: no gradient can flow through rgb")
        need = ctx.needs_input_grad
        if g_rgb is None and g_sigma is None and g_weight is None and not (need[2] or need[5]):
            # only `dists` is consumed downstream: it depends on z_vals and |d| alone, no parameter
            # (and here neither rays nor z_vals want a gradient) -- nothing to differentiate, exactly
            # as in the reference where dists has no parameter ancestry
            ctx.saved = None
            return (None,) * (7 + len(params))
        if ctx.saved is None:
            raise L.RdrfError("TensorVMSplit.forward: backward called twice (the saved activations are released "
                              "after the first backward; retain_graph is not supported)")
        N, S = z.shape
        dev = z.device
        fused = ctx.field.fused_grad
        ctx.field._det_bind()
        grads = ctx.field.fused_grads() if fused else [torch.zeros_like(p) for p in params]
        G = _static_struct(grads)
        P = _static_struct(params)
        _attach_packed(ctx.field, P, params, True, False)
        cfg = _cfg_struct(ctx.field, ctx.ray_type)
        need = ctx.needs_input_grad
        g_rays, g_xyz, g_z = L.zeros_like_many([rays, xyz, z], [need[2], need[4], need[5]])
        cont = lambda g: None if g is None else L.f32c(g)
        g_rgb, g_sigma, g_weight, g_dists = cont(g_rgb), cont(g_sigma), cont(g_weight), cont(g_dists)
        ws = L.workspace(dev, L.lib.rdrf_workspace_bytes(N, S))
        L.check(L.lib.rdrf_static_bwd(C.byref(P), C.byref(cfg), L.ptr(rays), L.ptr(ts), L.ptr(xyz),
                                      L.ptr(z), L.ptr(valid), N, S, L.ptr(g_rgb), L.ptr(g_sigma),
                                      L.ptr(g_weight), L.ptr(g_dists), C.byref(G), L.ptr(g_xyz),
                                      L.ptr(g_z), L.ptr(g_rays), L.ptr(ctx.saved),
                                      C.c_size_t(ctx.saved.numel()), L.ptr(ws), C.c_size_t(ws.numel()),
                                      L.stream_of(z)), "rdrf_static_bwd")
        ctx.saved = None
        if fused:   # already accumulated into p.grad (views of the field's flat buffer)
            grads = [None] * len(params)
        return (None, None, g_rays, None, g_xyz, g_z, None, *grads)


class _DynamicFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, field, ray_type, rays, ts, xyz, z, valid, *params):
        ctx.set_materialize_grads(False)   # unused outputs arrive as None: their branch is skipped
        rays, ts, xyz, z, valid = _prep_inputs(rays, ts, xyz, z, valid)
        N, S = z.shape
        dev = z.device
        want_rgb = not ray_type.endswith("-norgb")   # see _StaticFn
        ray_type = ray_type.replace("-norgb", "")
        rgb = torch.empty(N, S, 3, device=dev) if want_rgb else None
        xyz_prime = torch.empty(N, S, 3, device=dev)
        sigma, weight, dists, blending = (torch.empty(N, S, device=dev) for _ in range(4))
        saved, sbytes = _alloc_saved(ctx, 1, N, S, dev)
        ws = L.workspace(dev, (L.lib.rdrf_workspace_bytes if saved is not None else L.lib.rdrf_forward_workspace_bytes)(N, S))
        P = _dynamic_struct(params)
        _attach_packed(field, P, params, False, True)
        cfg = _cfg_struct(field, ray_type)
        L.check(L.lib.rdrf_dynamic_fwd(C.byref(P), C.byref(cfg), L.ptr(rays), L.ptr(ts), L.ptr(xyz),
                                       L.ptr(z), L.ptr(valid), N, S, L.ptr(blending), L.ptr(weight),
                                       L.ptr(xyz_prime), L.ptr(rgb), L.ptr(sigma), L.ptr(dists),
                                       L.ptr(saved), C.c_size_t(sbytes), L.ptr(ws),
                                       C.c_size_t(ws.numel()), L.stream_of(z)), "rdrf_dynamic_fwd")
        ctx.field, ctx.ray_type, ctx.saved, ctx.want_rgb = field, ray_type, saved, want_rgb
        ctx.save_for_backward(rays, ts, xyz, z, valid, *params)
        return blending, weight, xyz_prime, rgb, sigma, dists

    @staticmethod
    def backward(ctx, g_blending, g_weight, g_xyz_prime, g_rgb, g_sigma, g_dists):
        rays, ts, xyz, z, valid, *params = ctx.saved_tensors
        if g_rgb is not None and not ctx.want_rgb:
            raise L.RdrfError("forward(..., rgb=False) did not run the appearance phase: no gradient can flow through rgb")
        need = ctx.needs_input_grad
        if all(g is None for g in (g_blending, g_weight, g_xyz_prime, g_rgb, g_sigma)) and not (need[2] or need[5]):
            # only `dists` is consumed (pass E of the trainer feeds the dynamic field's dists to the
            # compositor): no parameter ancestry, nothing to differentiate
            ctx.saved = None
            return (None,) * (7 + len(params))
        if ctx.saved is None:
            raise L.RdrfError("TensorVMSplit_TimeEmbedding.forward: backward called twice (the saved activations "
                              "are released after the first backward; retain_graph is not supported)")
        N, S = z.shape
        dev = z.device
        fused = ctx.field.fused_grad
        ctx.field._det_bind()
        grads = ctx.field.fused_grads() if fused else [torch.zeros_like(p) for p in params]
        G = _dynamic_struct(grads)
        P = _dynamic_struct(params)
        _attach_packed(ctx.field, P, params, True, True)
        cfg = _cfg_struct(ctx.field, ctx.ray_type)
        need = ctx.needs_input_grad
        g_rays, g_xyz, g_z = L.zeros_like_many([rays, xyz, z], [need[2], need[4], need[5]])
        cont = lambda g: None if g is None else L.f32c(g)
        gb, gw, gxp, gr, gs, gd = (cont(g) for g in (g_blending, g_weight, g_xyz_prime, g_rgb,
                                                      g_sigma, g_dists))
        ws = L.workspace(dev, L.lib.rdrf_workspace_bytes(N, S))
        L.check(L.lib.rdrf_dynamic_bwd(C.byref(P), C.byref(cfg), L.ptr(rays), L.ptr(ts), L.ptr(xyz),
                                       L.ptr(z), L.ptr(valid), N, S, L.ptr(gb), L.ptr(gw),
                                       L.ptr(gxp), L.ptr(gr), L.ptr(gs), L.ptr(gd), C.byref(G),
                                       L.ptr(g_xyz), L.ptr(g_z), L.ptr(g_rays), L.ptr(ctx.saved),
                                       C.c_size_t(ctx.saved.numel()), L.ptr(ws),
                                       C.c_size_t(ws.numel()), L.stream_of(z)), "rdrf_dynamic_bwd")
        ctx.saved = None
        if fused:   # already accumulated into p.grad (views of the field's flat buffer)
            grads = [None] * len(params)
        return (None, None, g_rays, None, g_xyz, g_z, None, *grads)


class _SceneFlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, field, pts, ts, *params):
        ctx.set_materialize_grads(False)
        L.require_device(pts, ts)
        pts, ts = L.f32c(pts), L.f32c(ts)
        N, S, _ = pts.shape
        sf_f = torch.empty(N, S, 3, device=pts.device)
        sf_b = torch.empty(N, S, 3, device=pts.device)
        ws = L.workspace(pts.device, L.lib.rdrf_workspace_bytes(N, S))
        saved, sbytes = _alloc_saved(ctx, 2, N, S, pts.device)
        P = _dynamic_struct(params)
        _attach_packed(field, P, params, False, True)
        cfg = _cfg_struct(field, "ndc")
        L.check(L.lib.rdrf_scene_flow_fwd(C.byref(P), C.byref(cfg), L.ptr(pts), L.ptr(ts), N, S,
                                          L.ptr(sf_f), L.ptr(sf_b), L.ptr(saved), C.c_size_t(sbytes),
                                          L.ptr(ws), C.c_size_t(ws.numel()), L.stream_of(pts)),
                "rdrf_scene_flow_fwd")
        ctx.field, ctx.saved = field, saved
        ctx.save_for_backward(pts, ts, *params)
        return sf_f, sf_b

    @staticmethod
    def backward(ctx, g_f, g_b):
        pts, ts, *params = ctx.saved_tensors
        if ctx.saved is None:
            raise L.RdrfError("get_forward_backward_scene_flow: backward called twice (retain_graph is not supported)")
        N, S, _ = pts.shape
        fused = ctx.field.fused_grad
        ctx.field._det_bind()
        grads = ctx.field.fused_grads() if fused else [torch.zeros_like(p) for p in params]
        G = _dynamic_struct(grads)
        P = _dynamic_struct(params)
        _attach_packed(ctx.field, P, params, True, True)
        cfg = _cfg_struct(ctx.field, "ndc")
        g_pts = torch.zeros_like(pts) if ctx.needs_input_grad[1] else None
        g_f = None if g_f is None else L.f32c(g_f)
        g_b = None if g_b is None else L.f32c(g_b)
        ws = L.workspace(pts.device, L.lib.rdrf_workspace_bytes(N, S))
        L.check(L.lib.rdrf_scene_flow_bwd(C.byref(P), C.byref(cfg), L.ptr(pts), L.ptr(ts), N, S,
                                          L.ptr(g_f), L.ptr(g_b), C.byref(G), L.ptr(g_pts),
                                          L.ptr(ctx.saved), C.c_size_t(ctx.saved.numel()), L.ptr(ws),
                                          C.c_size_t(ws.numel()), L.stream_of(pts)),
                "rdrf_scene_flow_bwd")
        ctx.saved = None
        if fused:
            grads = [None] * len(params)
        return (None, g_pts, None, *grads)


# --------------------------------------------------------------------------------------------
# compute_* / warp_coordinate: the per-point building blocks (rdrf_*_features_fwd/bwd)
# --------------------------------------------------------------------------------------------
def _feat_saved(ctx, dynamic, M, dev):
    if not any(ctx.needs_input_grad):
        return None, 0
    nbytes = int(L.lib.rdrf_features_saved_bytes(int(dynamic), M))
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


class _StaticFeatFn(torch.autograd.Function):
    """(density feature [M], appearance feature [M,27]) of the static field at normalised points"""

    @staticmethod
    def forward(ctx, field, want_density, want_app, xn, *params):
        ctx.set_materialize_grads(False)
        L.require_device(xn)
        xn = L.f32c(xn)
        M, dev = xn.shape[0], xn.device
        dens = torch.empty(M, device=dev) if want_density else None
        app = torch.empty(M, 27, device=dev) if want_app else None
        ws = L.workspace(dev, L.lib.rdrf_features_workspace_bytes(M))
        saved, sbytes = _feat_saved(ctx, 0, M, dev)
        P = _static_struct(params)
        _attach_packed(field, P, params, False, False)
        cfg = _cfg_struct(field, "ndc")
        L.check(L.lib.rdrf_static_features_fwd(C.byref(P), C.byref(cfg), L.ptr(xn), M, L.ptr(dens), L.ptr(app),
                                               L.ptr(saved), C.c_size_t(sbytes), L.ptr(ws),
                                               C.c_size_t(ws.numel()), L.stream_of(xn)),
                "rdrf_static_features_fwd")
        ctx.field, ctx.saved = field, saved
        ctx.save_for_backward(xn, *params)
        return dens, app

    @staticmethod
    def backward(ctx, g_dens, g_app):
        xn, *params = ctx.saved_tensors
        if ctx.saved is None:
            raise L.RdrfError("compute_*: backward called twice (the saved activations were released)")
        M, dev = xn.shape[0], xn.device
        fused = ctx.field.fused_grad
        ctx.field._det_bind()
        grads = ctx.field.fused_grads() if fused else [torch.zeros_like(p) for p in params]
        if g_dens is None and g_app is None:
            return (None,) * (4 + len(params))
        G, P = _static_struct(grads), _static_struct(params)
        _attach_packed(ctx.field, P, params, True, False)
        cfg = _cfg_struct(ctx.field, "ndc")
        g_xn = torch.zeros_like(xn) if ctx.needs_input_grad[3] else None
        g_dens = None if g_dens is None else L.f32c(g_dens)
        g_app = None if g_app is None else L.f32c(g_app)
        ws = L.workspace(dev, L.lib.rdrf_features_bwd_workspace_bytes(M))
        L.check(L.lib.rdrf_static_features_bwd(C.byref(P), C.byref(cfg), L.ptr(xn), M, L.ptr(g_dens),
                                               L.ptr(g_app), C.byref(G), L.ptr(g_xn), L.ptr(ctx.saved),
                                               C.c_size_t(ctx.saved.numel()), L.ptr(ws), C.c_size_t(ws.numel()),
                                               L.stream_of(xn)), "rdrf_static_features_bwd")
        ctx.saved = None
        if fused:
            grads = [None] * len(params)
        return (None, None, None, g_xn, *grads)


class _DynFeatFn(torch.autograd.Function):
    """(density [M], blending [M], app [M,27], xyz_prime [M,3]) of the dynamic field at M points with
    per-point times; `x` normalised (compute_*) or un-normalised (warp_coordinate)"""

    @staticmethod
    def forward(ctx, field, want, x_is_normalized, x, t, *params):
        ctx.set_materialize_grads(False)
        L.require_device(x, t)
        x, t = L.f32c(x), L.f32c(t)
        M, dev = x.shape[0], x.device
        dens = torch.empty(M, device=dev) if "density" in want else None
        blend = torch.empty(M, device=dev) if "blending" in want else None
        app = torch.empty(M, 27, device=dev) if "app" in want else None
        xp = torch.empty(M, 3, device=dev) if "warp" in want else None
        ws = L.workspace(dev, L.lib.rdrf_features_workspace_bytes(M))
        saved, sbytes = _feat_saved(ctx, 1, M, dev)
        P = _dynamic_struct(params)
        _attach_packed(field, P, params, False, True)
        cfg = _cfg_struct(field, "ndc")
        L.check(L.lib.rdrf_dynamic_features_fwd(C.byref(P), C.byref(cfg), L.ptr(x), L.ptr(t), M,
                                                int(x_is_normalized), L.ptr(dens), L.ptr(blend), L.ptr(app),
                                                L.ptr(xp), L.ptr(saved), C.c_size_t(sbytes), L.ptr(ws),
                                                C.c_size_t(ws.numel()), L.stream_of(x)),
                "rdrf_dynamic_features_fwd")
        ctx.field, ctx.saved, ctx.norm = field, saved, int(x_is_normalized)
        ctx.save_for_backward(x, t, *params)
        return dens, blend, app, xp

    @staticmethod
    def backward(ctx, g_dens, g_blend, g_app, g_xp):
        x, t, *params = ctx.saved_tensors
        if all(g is None for g in (g_dens, g_blend, g_app, g_xp)):
            return (None,) * (5 + len(params))
        if ctx.saved is None:
            raise L.RdrfError("compute_*: backward called twice (the saved activations were released)")
        M, dev = x.shape[0], x.device
        fused = ctx.field.fused_grad
        ctx.field._det_bind()
        grads = ctx.field.fused_grads() if fused else [torch.zeros_like(p) for p in params]
        G, P = _dynamic_struct(grads), _dynamic_struct(params)
        _attach_packed(ctx.field, P, params, True, True)
        cfg = _cfg_struct(ctx.field, "ndc")
        g_x = torch.zeros_like(x) if ctx.needs_input_grad[3] else None
        c = lambda g: None if g is None else L.f32c(g)
        g_dens, g_blend, g_app, g_xp = c(g_dens), c(g_blend), c(g_app), c(g_xp)
        ws = L.workspace(dev, L.lib.rdrf_features_bwd_workspace_bytes(M))
        L.check(L.lib.rdrf_dynamic_features_bwd(C.byref(P), C.byref(cfg), L.ptr(x), L.ptr(t), M, ctx.norm,
                                                L.ptr(g_dens), L.ptr(g_blend), L.ptr(g_app), L.ptr(g_xp),
                                                C.byref(G), L.ptr(g_x), L.ptr(ctx.saved),
                                                C.c_size_t(ctx.saved.numel()), L.ptr(ws), C.c_size_t(ws.numel()),
                                                L.stream_of(x)), "rdrf_dynamic_features_bwd")
        ctx.saved = None
        if fused:
            grads = [None] * len(params)
        return (None, None, None, g_x, None, *grads)


# --------------------------------------------------------------------------------------------
# TensorBase (models/tensorBase.py:281-559)
# --------------------------------------------------------------------------------------------
class TensorBase(nn.Module):
    def __init__(self, aabb, gridSize, tSize, device, density_n_comp=8, appearance_n_comp=24,
                 app_dim=27, shadingMode="MLP_PE", alphaMask=None, near_far=[2.0, 6.0],
                 density_shift=-10, alphaMask_thres=0.001, distance_scale=25,
                 rayMarch_weight_thres=0.0001, pos_pe=6, view_pe=6, fea_pe=6, featureC=128,
                 step_ratio=2.0, fea2denseAct="softplus"):
        super().__init__()
        self.density_n_comp = list(density_n_comp)
        self.app_n_comp = list(appearance_n_comp)
        self.app_dim = app_dim
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).to(device)
        self._aabb_host = [float(v) for v in self.aabb.detach().cpu().reshape(-1)]
        self.alphaMask = alphaMask
        self.device = device
        self.density_shift = density_shift
        self.alphaMask_thres = alphaMask_thres
        self.distance_scale = distance_scale
        self.rayMarch_weight_thres = rayMarch_weight_thres
        self.fea2denseAct = fea2denseAct
        self.near_far = near_far
        self.step_ratio = step_ratio
        self.tSizeFixed = tSize
        self.update_stepSize(gridSize, tSize)
        self.matMode = MAT_MODE
        self.vecMode = VEC_MODE
        self.comp_w = [1, 1, 1]
        if self.density_n_comp != [16, 4, 4] or self.app_n_comp != [48, 12, 12] or app_dim != 27 \
                or featureC != 128 or view_pe != 0:
            raise NotImplementedError(
                "the HIP kernels are built for the shipped configs: density comps [16,4,4], "
                "appearance comps [48,12,12], app_dim 27, featureC 128, view_pe 0")
        self.init_svd_volume(gridSize[0], device)
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = (
            shadingMode, pos_pe, view_pe, fea_pe, featureC)
        self.init_render_func(shadingMode, pos_pe, view_pe, fea_pe, featureC, device)

    # ---- construction ------------------------------------------------------------------
    def init_render_func(self, shadingMode, pos_pe, view_pe, fea_pe, featureC, device):
        if shadingMode == "MLP_Fea":
            self.renderModule = MLPRender_Fea(self.app_dim, view_pe, fea_pe, featureC).to(device)
        elif shadingMode == "MLP_Fea_TimeEmbedding":
            self.renderModule = MLPRender_Fea_TimeEmbedding(self.app_dim, view_pe, fea_pe,
                                                            featureC).to(device)
        elif shadingMode == "MLP_Fea_late_view":
            self.renderModule = MLPRender_Fea_late_view(self.app_dim, view_pe, fea_pe,
                                                        featureC).to(device)
        else:
            raise NotImplementedError(f"shadingMode {shadingMode} is not reachable from the shipped "
                                      "configs and is not built")

    def update_stepSize(self, gridSize, tSize):
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.gridSize = torch.LongTensor(list(gridSize)).to(self.device)
        self.units = self.aabbSize / (self.gridSize - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio
        self.aabbDiag = torch.sqrt(torch.sum(torch.square(self.aabbSize)))
        self.nSamples = int((self.aabbDiag / self.stepSize).item()) + 1
        self.tSize = torch.unsqueeze(torch.tensor(tSize), 0).to(self.device)

    def init_one_svd(self, n_component, gridSize, scale, device):
        plane_coef, line_coef = [], []
        gs = [int(g) for g in gridSize]
        for i in range(3):
            vec_id = VEC_MODE[i]
            m0, m1 = MAT_MODE[i]
            p = scale * torch.randn((1, n_component[i], gs[m1], gs[m0]))
            l = scale * torch.randn((1, n_component[i], gs[vec_id], 1))
            plane_coef.append(nn.Parameter(channel_last_(p.to(device), h_fast=Z_FAST and i > 0)))
            line_coef.append(nn.Parameter(channel_last_(l.to(device))))
        return nn.ParameterList(plane_coef), nn.ParameterList(line_coef)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # accept planes at a different resolution (checkpoints saved after upsampling)
        for name, p in list(self.named_parameters(recurse=True)):
            key = prefix + name
            if key in state_dict and ("_plane." in name or "_line." in name) \
                    and state_dict[key].shape != p.shape:
                mod, attr = name.split(".")
                new = channel_last_(state_dict[key].to(p.device).float(),
                                    h_fast=Z_FAST and "_plane" in mod and int(attr) > 0)
                getattr(self, mod)[int(attr)] = nn.Parameter(new)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ---- helpers kept from the reference API ---------------------------------------------
    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invaabbSize - 1

    def unnormalize_coord(self, xyz_sampled):
        return (xyz_sampled + 1) / self.invaabbSize + self.aabb[0]

    def feature2density(self, density_features):
        if self.fea2denseAct == "softplus":
            return F.softplus(density_features + self.density_shift)
        return F.relu(density_features)

    def get_kwargs(self):
        return {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), "tSize": self.tSize.item(),
                "density_n_comp": self.density_n_comp, "appearance_n_comp": self.app_n_comp,
                "app_dim": self.app_dim, "density_shift": self.density_shift,
                "alphaMask_thres": self.alphaMask_thres, "distance_scale": self.distance_scale,
                "rayMarch_weight_thres": self.rayMarch_weight_thres,
                "fea2denseAct": self.fea2denseAct, "near_far": self.near_far,
                "step_ratio": self.step_ratio, "shadingMode": self.shadingMode,
                "pos_pe": self.pos_pe, "view_pe": self.view_pe, "fea_pe": self.fea_pe,
                "featureC": self.featureC}

    def save(self, se3_poses, focal_ratio_refine, path):
        kwargs = self.get_kwargs()
        kwargs["se3_poses"] = se3_poses
        kwargs["focal_ratio_refine"] = focal_ratio_refine
        torch.save({"kwargs": kwargs, "state_dict": self.state_dict()}, path)

    def load(self, ckpt):
        """models/tensorBase.py:472-485.  The alpha-mask machinery is dead in the reference (compute_alpha calls
        compute_densityfeature with the wrong arity) and is not built here: a checkpoint
        that carries one is refused instead of silently dropping it."""
        if any(k.startswith("alphaMask") for k in ckpt) or any(k.startswith("alphaMask") for k in ckpt["state_dict"]):
            raise NotImplementedError("checkpoint carries an alphaMask: AlphaGridMask is not built (dead code in "
                                      "the reference, SURVEY.md section 0)")
        self.load_state_dict(ckpt["state_dict"])

    # ---- samplers (models/tensorBase.py:487-559): device kernels, RNG stays in torch ----------
    def sample_ray_ndc(self, rays_o, rays_d, is_train=True, N_samples=-1):
        from .renderer import sample_rays
        N_samples = N_samples if N_samples > 0 else self.nSamples
        xyz, z, valid = sample_rays(self, torch.cat([rays_o, rays_d], -1), N_samples, "ndc", is_train)
        return xyz, z[:1], valid

    def sample_ray_contracted(self, rays_o, rays_d, is_train=True, N_samples=-1):
        from .renderer import sample_rays
        N_samples = N_samples if N_samples > 0 else self.nSamples
        xyz, z, valid = sample_rays(self, torch.cat([rays_o, rays_d], -1), N_samples, "contract",
                                    is_train)
        return xyz, z[:1], valid

    @torch.no_grad()
    def up_sampling_VM(self, plane_coef, line_coef, res_target):
        """models/tensoRF.py:199-221 / 814-836: bilinear (align_corners) resampling of a factor family to
        the new grid -- rdrf_upsample_bilinear, one launch for the six tensors, channel-last in and out."""
        from .regularizers import _tensor4
        srcs, dsts = [], []
        for i in range(3):
            vec_id = VEC_MODE[i]
            m0, m1 = MAT_MODE[i]
            p, l = plane_coef[i].data, line_coef[i].data
            L.require_device(p, l)
            newp = torch.empty((1, p.shape[1], int(res_target[m1]), int(res_target[m0])), device=p.device)
            newl = torch.empty((1, l.shape[1], int(res_target[vec_id]), 1), device=l.device)
            newp, newl = channel_last_(newp, h_fast=Z_FAST and i > 0), channel_last_(newl)
            srcs += [p, l]
            dsts += [newp, newl]
        sa = (L.RdrfTensor4 * 6)(*[_tensor4(t) for t in srcs])
        da = (L.RdrfTensor4 * 6)(*[_tensor4(t) for t in dsts])
        L.check(L.lib.rdrf_upsample_bilinear(sa, da, 6, L.stream_of(srcs[0])), "rdrf_upsample_bilinear")
        for i in range(3):
            plane_coef[i] = nn.Parameter(dsts[2 * i])
            line_coef[i] = nn.Parameter(dsts[2 * i + 1])
        return plane_coef, line_coef

    # ---- fused gradient accumulation ----------------------------------------------------
    # The C ABI accumulates (+=) into caller-owned gradient buffers.  With `fused_grad` on, every
    # backward pass of this field adds straight into p.grad, and all p.grad are views of ONE flat
    # fp32 buffer: a step costs one memset instead of ~40 zeros_like + ~40 autograd adds per pass
    # (rocprofv3: 413 fill + 230 add launches, 2.3 ms, per 5-pass step), and the data-parallel
    # exchange all-reduces the flat buffer in place.  Off by default: plain autograd semantics
    # (torch.autograd.grad, retain_graph, ...) need freshly returned gradients.
    fused_grad = False
    _pack_epoch = 0    # bumped by whoever writes the parameters through raw pointers (optim.FlatAdam): see _attach_packed

    def invalidate_packed(self):
        """Drop the cached packed weight images.  Needed only after an MLP weight was edited in a way torch's version
        counters do not see: through `.data`, or through raw pointers (optim.FlatAdam does it itself)."""
        self._pack_epoch += 1

    FLAT_ALIGN = 4096   # floats: the flat buffers split evenly over 1, 2, 4, 8 ... ranks in 16-byte units

    def _flat_layout(self):
        """(offsets, total, split): float offset of every parameter of _param_list() inside the flat
        buffers (64-float aligned), the padded total, and the offset where the VM factors end and the
        networks begin (the two learning rates of get_optparam_groups)."""
        params = self._param_list()
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        n_vm = sum(1 for n, _ in self.named_parameters() if "_plane." in n or "_line." in n)
        split = offs[n_vm] if n_vm < len(offs) else total
        total = (total + self.FLAT_ALIGN - 1) // self.FLAT_ALIGN * self.FLAT_ALIGN
        return offs, total, split

    def fused_grads(self):
        params = self._param_list()
        views = getattr(self, "_gviews", None)
        ok = views is not None and len(views) == len(params) and all(
            p.grad is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == v.stride()
            and p.shape == v.shape for p, v in zip(params, views))
        if not ok:
            offs, total, _ = self._flat_layout()
            flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            views = [torch.as_strided(flat, p.size(), p.stride(), o) for p, o in zip(params, offs)]
            with torch.no_grad():
                for p, v in zip(params, views):
                    if p.grad is not None and p.grad.shape == v.shape:
                        v.copy_(p.grad)
                    p.grad = v
            self._gflat, self._gviews = flat, views
            self._det_shadow = torch.zeros(total, dtype=torch.int64, device=flat.device) if L.DETERMINISTIC else None
        return views

    # ---- deterministic debugging build (RDRF_DETERMINISTIC=1, librodynrf_det.so) -----------------------------------------
    def _det_bind(self):
        """bind this field's flat gradient buffer to its fixed-point shadow for the backward about to run"""
        if not L.DETERMINISTIC:
            return
        if not self.fused_grad:
            raise L.RdrfError("RDRF_DETERMINISTIC=1 needs the fused gradient buffers (field.fused_grad = True): only additions "
                              "into a bound flat buffer are order-independent")
        self.fused_grads()
        L.check(L.lib.rdrf_det_bind(1 if isinstance(self, TensorVMSplit_TimeEmbedding) else 0, L.ptr(self._gflat),
                                    C.c_size_t(self._gflat.numel()), L.ptr(self._det_shadow), L.stream_of(self._gflat)),
                "rdrf_det_bind")

    def det_fold_(self):
        """fold the fixed-point shadow into the fp32 gradients (call before anything reads .grad); no-op otherwise"""
        if L.DETERMINISTIC and getattr(self, "_det_shadow", None) is not None:
            self._det_bind()
            L.check(L.lib.rdrf_det_finish(1 if isinstance(self, TensorVMSplit_TimeEmbedding) else 0,
                                          L.stream_of(self._gflat)), "rdrf_det_finish")

    def flatten_params_(self):
        """Make every parameter a view of ONE flat fp32 buffer (same layout as the fused gradient buffer):
        the Adam kernel then updates a field with one launch over (p, g, m, v) flat ranges, and the
        sharded data-parallel step reduce-scatters / all-gathers the flat buffers in place.  Values,
        shapes, strides and state_dict() are unchanged.  Returns the flat buffer (idempotent)."""
        params = self._param_list()
        offs, total, _ = self._flat_layout()
        flat = getattr(self, "_pflat", None)
        ok = flat is not None and flat.numel() == total and all(
            p.data_ptr() == flat.data_ptr() + 4 * o for p, o in zip(params, offs))
        if not ok:
            flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            with torch.no_grad():
                for p, o in zip(params, offs):
                    v = torch.as_strided(flat, p.size(), p.stride(), o)
                    v.copy_(p.data)
                    p.data = v
            self._pflat = flat
        return flat

    def zero_grad_fused(self):
        """one memset for every gradient of this field (replaces optimizer.zero_grad())"""
        self.fused_grads()
        self._gflat.zero_()
        if getattr(self, "_det_shadow", None) is not None:
            self._det_shadow.zero_()
        return self._gflat

    def _check_layout(self):
        for n, p in self.named_parameters():
            if ("_plane." in n or "_line." in n) and not _is_channel_last(p):
                raise L.RdrfError(f"{n} lost its channel-last layout")


class TensorVMSplit(TensorBase):
    """Static field (models/tensoRF.py:11-274)."""

    def init_svd_volume(self, res, device):
        self.density_plane, self.density_line = self.init_one_svd(self.density_n_comp, self.gridSize,
                                                                  0.1, device)
        self.app_plane, self.app_line = self.init_one_svd(self.app_n_comp, self.gridSize, 0.1, device)
        self.basis_mat = nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        return [{"params": self.density_line, "lr": lr_init_spatialxyz},
                {"params": self.density_plane, "lr": lr_init_spatialxyz},
                {"params": self.app_line, "lr": lr_init_spatialxyz},
                {"params": self.app_plane, "lr": lr_init_spatialxyz},
                {"params": self.basis_mat.parameters(), "lr": lr_init_network},
                {"params": self.renderModule.parameters(), "lr": lr_init_network}]

    def _param_list(self):
        rm = self.renderModule
        if self.shadingMode == "MLP_Fea":
            if self.fea_pe != 2:
                raise NotImplementedError("static MLP_Fea head is built for fea_pe=2 (train.py:889)")
            last = rm.mlp[4]
        elif self.shadingMode == "MLP_Fea_TimeEmbedding":
            if self.fea_pe != 2:
                raise NotImplementedError("static head is built for fea_pe=2 (train.py:889)")
            last = rm.mlp_view[0]
        else:
            raise NotImplementedError("static field heads: MLP_Fea | MLP_Fea_TimeEmbedding")
        return (list(self.density_plane) + list(self.density_line) + list(self.app_plane)
                + list(self.app_line) + [self.basis_mat.weight, rm.mlp[0].weight, rm.mlp[0].bias,
                                         rm.mlp[2].weight, rm.mlp[2].bias, last.weight, last.bias])

    def warp_coordinate(self, xyz_sampled, t_sampled):
        return None

    # ---- models/tensoRF.py:118-196: the per-point building blocks of forward() -------------------
    def compute_densityfeature(self, xyz_sampled, t_sampled=None, time_embedding_sampled=None):
        """xyz_sampled [M,3] NORMALISED coordinates -> sigma feature [M] (sum of the 24 VM products,
        before feature2density).  t_sampled / time_embedding_sampled are accepted and unused, as in
        the reference.  Differentiable wrt the density factors and the coordinates."""
        return _StaticFeatFn.apply(self, True, False, xyz_sampled.reshape(-1, 3), *self._param_list())[0]

    def compute_appfeature(self, xyz_sampled, t_sampled=None, time_embedding_sampled=None):
        """xyz_sampled [M,3] normalised -> basis_mat(72 VM products) [M,27]"""
        return _StaticFeatFn.apply(self, False, True, xyz_sampled.reshape(-1, 3), *self._param_list())[1]

    # models/tensoRF.py:63-98
    def vectorDiffs(self, vector_comps):
        return vector_diffs(vector_comps)

    def vector_comp_diffs(self):
        return vector_diffs(self.density_line) + vector_diffs(self.app_line)

    def density_L1(self):
        return dense_l1(self, self.density_plane, self.density_line)

    # models/tensoRF.py:100-116
    def TV_loss_density(self, reg):
        return tv_family(self, reg, self.density_plane, self.density_line)

    def TV_loss_app(self, reg):
        return tv_family(self, reg, self.app_plane, self.app_line)

    def forward(self, rays_chunk, ts_chunk, timeembeddings_chunk, xyz_sampled, z_vals, ray_valid,
                white_bg=True, is_train=False, ray_type="ndc", N_samples=-1, rgb=True):
        """rgb=False (extension): the caller does not consume the colours (entry 6 of the tuple is None) and the
        appearance phase -- gather, basis, RGB head: ~70 % of this field's forward work -- is not run."""
        if timeembeddings_chunk is not None:
            raise NotImplementedError("timeembeddings_chunk is None at every reference call site")
        rgb, sigma, weight, dists = _StaticFn.apply(self, ray_type if rgb else ray_type + "-norgb", rays_chunk, ts_chunk,
                                                    xyz_sampled, z_vals, ray_valid, *self._param_list())
        return (None, None, None, xyz_sampled, weight, None, rgb, sigma, z_vals, dists)

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        self.app_plane, self.app_line = self.up_sampling_VM(self.app_plane, self.app_line, res_target)
        self.density_plane, self.density_line = self.up_sampling_VM(self.density_plane,
                                                                    self.density_line, res_target)
        self.update_stepSize(res_target, 1)


class TensorVMSplit_TimeEmbedding(TensorBase):
    """Dynamic field (models/tensoRF.py:277-892)."""

    def __init__(self, aabb, gridSize, tSize, device, **kargs):
        super().__init__(aabb, gridSize, tSize, device, **kargs)
        self.layer1 = nn.Linear(1 + 8 * 2 * 1, 64).to(device)
        self.layer2 = nn.Linear(64, 30).to(device)
        self.layer3 = nn.Linear((3 + 10 * 2 * 3) + 30, 64).to(device)
        self.layer4 = nn.Linear(64, 64).to(device)
        self.layer5 = nn.Linear(64, 3).to(device)
        nin = sum(self.density_n_comp) * 3 + 3 + 10 * 2 * 3 + 1 + 8 * 2 * 1
        self.density_layer1 = nn.Linear(nin, 64).to(device)
        self.density_layer2 = nn.Linear(64, 1).to(device)
        self.blending_layer1 = nn.Linear(nin, 64).to(device)
        self.blending_layer2 = nn.Linear(64, 1).to(device)
        self.scene_flow_mlp = nn.Sequential(
            nn.Linear(4 * 2 * 4 + 4, 64), nn.ReLU(inplace=True), nn.Linear(64, 64),
            nn.ReLU(inplace=True), nn.Linear(64, 64), nn.ReLU(inplace=True), nn.Linear(64, 6),
        ).to(device)
        if self.shadingMode != "MLP_Fea_late_view" or self.fea_pe != 0:
            raise NotImplementedError("dynamic head is built for MLP_Fea_late_view, fea_pe=0 "
                                      "(configs/*.txt, train.py:918)")

    def init_svd_volume(self, res, device):
        self.blending_plane, self.blending_line = self.init_one_svd(self.density_n_comp,
                                                                    self.gridSize, 0.1, device)
        self.density_plane, self.density_line = self.init_one_svd(self.density_n_comp, self.gridSize,
                                                                  0.1, device)
        self.app_plane, self.app_line = self.init_one_svd(self.app_n_comp, self.gridSize, 0.1, device)
        self.basis_mat = nn.Linear(sum(self.app_n_comp) * 3, self.app_dim, bias=False).to(device)

    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        g = [{"params": self.density_line, "lr": lr_init_spatialxyz},
             {"params": self.density_plane, "lr": lr_init_spatialxyz},
             {"params": self.blending_line, "lr": lr_init_spatialxyz},
             {"params": self.blending_plane, "lr": lr_init_spatialxyz},
             {"params": self.app_line, "lr": lr_init_spatialxyz},
             {"params": self.app_plane, "lr": lr_init_spatialxyz},
             {"params": self.basis_mat.parameters(), "lr": lr_init_network},
             {"params": self.scene_flow_mlp.parameters(), "lr": lr_init_network}]
        for m in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5, self.density_layer1,
                  self.density_layer2, self.blending_layer1, self.blending_layer2):
            g.append({"params": m.parameters(), "lr": lr_init_network})
        g.append({"params": self.renderModule.parameters(), "lr": lr_init_network})
        return g

    def _param_list(self):
        rm = self.renderModule
        t = (list(self.density_plane) + list(self.density_line) + list(self.blending_plane)
             + list(self.blending_line) + list(self.app_plane) + list(self.app_line))
        t += [self.basis_mat.weight, rm.mlp[0].weight, rm.mlp[0].bias, rm.mlp[2].weight,
              rm.mlp[2].bias, rm.mlp_view[0].weight, rm.mlp_view[0].bias]
        for m in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5, self.density_layer1,
                  self.density_layer2, self.blending_layer1, self.blending_layer2):
            t += [m.weight, m.bias]
        for i in (0, 2, 4, 6):
            t += [self.scene_flow_mlp[i].weight, self.scene_flow_mlp[i].bias]
        return t

    def forward(self, rays_chunk, ts_chunk, timeembeddings_chunk, xyz_sampled, z_vals, ray_valid,
                white_bg=True, is_train=False, ray_type="ndc", N_samples=-1, rgb=True):
        """rgb=False (extension): see TensorVMSplit.forward"""
        if timeembeddings_chunk is not None:
            raise NotImplementedError("timeembeddings_chunk is None at every reference call site")
        blending, weight, xyz_prime, rgb, sigma, dists = _DynamicFn.apply(
            self, ray_type if rgb else ray_type + "-norgb", rays_chunk, ts_chunk, xyz_sampled, z_vals, ray_valid,
            *self._param_list())
        return (None, None, blending, xyz_sampled, weight, xyz_prime, rgb, sigma, z_vals, dists)

    # models/tensoRF.py:378-416 (the reference's dynamic class has no vector_comp_diffs)
    def density_L1(self):
        return dense_l1(self, self.density_plane, self.density_line)

    def blending_L1(self):
        return dense_l1(self, self.blending_plane, self.blending_line)

    # models/tensoRF.py:418-444
    def TV_loss_blending(self, reg):
        return tv_family(self, reg, self.blending_plane, self.blending_line)

    def TV_loss_density(self, reg):
        return tv_family(self, reg, self.density_plane, self.density_line)

    def TV_loss_app(self, reg):
        return tv_family(self, reg, self.app_plane, self.app_line)

    def get_forward_backward_scene_flow(self, unnormalized_pts, t_sampled):
        return _SceneFlowFn.apply(self, unnormalized_pts, t_sampled, *self._param_list())

    def _features(self, want, x, t, normalized):
        x2 = x.reshape(-1, 3)
        t2 = t.reshape(-1)
        if t2.numel() != x2.shape[0]:
            raise L.RdrfError(f"compute_*: {x2.shape[0]} points but {t2.numel()} times")
        return _DynFeatFn.apply(self, want, normalized, x2, t2, *self._param_list())

    def warp_coordinate(self, unnormalized_xyz_sampled, t_sampled):
        """models/tensoRF.py:521-541: (...,3) un-normalised points, (...) times -> warped un-normalised
        points, same shape.  Differentiable wrt the warp MLP, the coordinates (not the times)."""
        out = self._features(("warp",), unnormalized_xyz_sampled, t_sampled, False)[3]
        return out.reshape(unnormalized_xyz_sampled.shape)

    def compute_densityfeature(self, xyz_sampled, t_sampled, time_embedding_sampled=None):
        """models/tensoRF.py:646-732: xyz_sampled [M,3] NORMALISED, t_sampled [M] -> density_layer2 output [M]
        (3-stride VM features at the warped point + [xn, PE10, t, PE8] -> 64 -> 1)."""
        return self._features(("density",), xyz_sampled, t_sampled, True)[0]

    def compute_blendingfeature(self, xyz_sampled, t_sampled, time_embedding_sampled=None):
        """models/tensoRF.py:543-629 (before the sigmoid)"""
        return self._features(("blending",), xyz_sampled, t_sampled, True)[1]

    def compute_appfeature(self, xyz_sampled, t_sampled, time_embedding_sampled=None):
        """models/tensoRF.py:734-811: basis_mat(216 3-stride VM features at the warped point) [M,27]"""
        return self._features(("app",), xyz_sampled, t_sampled, True)[2]

    @torch.no_grad()
    def upsample_volume_grid(self, res_target):
        self.app_plane, self.app_line = self.up_sampling_VM(self.app_plane, self.app_line, res_target)
        self.density_plane, self.density_line = self.up_sampling_VM(self.density_plane,
                                                                    self.density_line, res_target)
        self.blending_plane, self.blending_line = self.up_sampling_VM(self.blending_plane,
                                                                      self.blending_line, res_target)
        self.update_stepSize(res_target, 1)
