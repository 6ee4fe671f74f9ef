"""Host mirror of /root/reference/renderer.py:24-315 for the hot path: ``sampleXYZ``,
``raw2outputs`` and a working ``OctreeRender_trilinear_fast`` chunk loop.  All arithmetic runs in
the HIP kernels (rdrf_sample_*, rdrf_composite_*); torch only supplies memory, RNG and autograd."""
import ctypes as C

import torch

from . import _lib as L


class _SampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays, aabb_host, near, far, S, ray_type, jitter, jitter_outer):
        L.require_device(rays)
        rays = L.f32c(rays)
        N = rays.shape[0]
        dev = rays.device
        xyz = torch.empty(N, S, 3, device=dev)
        z = torch.empty(N, S, device=dev)
        valid = torch.empty(N, S, dtype=torch.uint8, device=dev)
        if ray_type == "ndc":
            ab = (C.c_float * 6)(*aabb_host)
            L.check(L.lib.rdrf_sample_ndc(L.ptr(rays), N, S, C.c_float(near), C.c_float(far),
                                          L.ptr(jitter), ab, L.ptr(xyz), L.ptr(z), L.ptr(valid),
                                          L.stream_of(rays)), "rdrf_sample_ndc")
        elif ray_type == "contract":
            L.check(L.lib.rdrf_sample_contract(L.ptr(rays), N, S, C.c_float(near), C.c_float(far),
                                               L.ptr(jitter), L.ptr(jitter_outer), L.ptr(xyz),
                                               L.ptr(z), L.ptr(valid), L.stream_of(rays)),
                    "rdrf_sample_contract")
        else:
            raise NotImplementedError("ray_type must be 'ndc' or 'contract' (the shipped configs)")
        ctx.ray_type = ray_type
        ctx.save_for_backward(rays, z)
        ctx.mark_non_differentiable(valid, z)
        return xyz, z, valid.view(torch.bool)

    @staticmethod
    def backward(ctx, g_xyz, g_z, g_valid):
        rays, z = ctx.saved_tensors
        N, S = z.shape
        g_rays = torch.zeros_like(rays)
        g_xyz = L.f32c(g_xyz)
        L.check(L.lib.rdrf_sample_bwd(L.ptr(rays), L.ptr(z), N, S, L.RAY_TYPES[ctx.ray_type],
                                      L.ptr(g_xyz), L.ptr(g_rays), L.stream_of(rays)),
                "rdrf_sample_bwd")
        return g_rays, None, None, None, None, None, None, None


def sample_rays(tensorf, rays, N_samples, ray_type="ndc", is_train=False, jitter=None,
                jitter_outer=None):
    """sample_ray_ndc / sample_ray_contracted with the z row tiled to [N,S]
    (models/tensorBase.py:487-559 + renderer.py:169). The train-time jitter is drawn with torch's
    device RNG here unless given."""
    near, far = tensorf.near_far
    S = int(N_samples)
    dev = rays.device
    if is_train and jitter is None:
        if ray_type == "ndc":
            jitter = torch.rand(S, device=dev)
        else:
            jitter = torch.rand(S - S // 2 + 1, device=dev)
            jitter_outer = torch.rand(S // 2 + 1, device=dev)
    if jitter is not None:
        jitter = L.f32c(jitter.reshape(-1))
    if jitter_outer is not None:
        jitter_outer = L.f32c(jitter_outer.reshape(-1))
    return _SampleFn.apply(rays, tensorf._aabb_host, float(near), float(far), S, ray_type, jitter,
                           jitter_outer)


def sampleXYZ(tensorf, rays_train, N_samples, ray_type="ndc", is_train=False, jitter=None,
              jitter_outer=None):
    """renderer.py:147-170 (extra keyword: explicit jitter vectors for reproducible tests)."""
    if N_samples is None or N_samples <= 0:
        N_samples = tensorf.nSamples
    return sample_rays(tensorf, rays_train, N_samples, ray_type, is_train, jitter, jitter_outer)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays, ray_type,
                add_white_bg):
        L.require_device(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays)
        ins = [L.f32c(t) for t in (rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays)]
        N, S = ins[1].shape
        dev = ins[1].device
        shapes = [(N, 3), (N,), (N,), (N, S)] * 3 + [(N,)]
        outs = [torch.empty(s, device=dev) for s in shapes]
        arr = (C.c_void_p * 13)(*[o.data_ptr() for o in outs])
        L.check(L.lib.rdrf_composite_fwd(*[L.ptr(t) for t in ins], N, S, L.RAY_TYPES.get(ray_type, 2),
                                         int(add_white_bg), arr, L.stream_of(ins[1])),
                "rdrf_composite_fwd")
        ctx.ray_type, ctx.white = ray_type, int(add_white_bg)
        ctx.save_for_backward(*ins)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_out):
        ins = ctx.saved_tensors
        N, S = ins[1].shape
        g_out = [None if g is None else L.f32c(g) for g in g_out]
        need = ctx.needs_input_grad[:8]
        g_in = [torch.zeros_like(t) if n else None for t, n in zip(ins, need)]
        ga = (C.c_void_p * 13)(*[0 if g is None else g.data_ptr() for g in g_out])
        gi = (C.c_void_p * 8)(*[0 if g is None else g.data_ptr() for g in g_in])
        L.check(L.lib.rdrf_composite_bwd(*[L.ptr(t) for t in ins], N, S,
                                         L.RAY_TYPES.get(ctx.ray_type, 2), ctx.white, ga, gi,
                                         L.stream_of(ins[1])), "rdrf_composite_bwd")
        return (*g_in, None, None)


def raw2outputs(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays_chunk, is_train=False,
                ray_type="ndc", add_white_bg=None):
    """renderer.py:173-315.  The train-time coin (`torch.rand((1,)) < 0.5`, renderer.py:269) is
    drawn here on the host unless `add_white_bg` is given."""
    if add_white_bg is None:
        add_white_bg = bool(is_train and (torch.rand((1,)) < 0.5).item())
    return _CompositeFn.apply(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays_chunk,
                              ray_type, bool(add_white_bg))


def OctreeRender_trilinear_fast(rays, ts, timeembeddings, tensorf, xyz_sampled, z_vals_input,
                                ray_valid, chunk=4096, N_samples=-1, ray_type="ndc", white_bg=True,
                                is_train=False, device="cuda"):
    """renderer.py:24-144 -- same signature and 11-tuple; unlike the reference (which raises on
    its own None entries) this chunk loop tolerates them."""
    keys = [[] for _ in range(10)]
    N = rays.shape[0]
    for c0 in range(0, N, chunk):
        sl = slice(c0, c0 + chunk)
        te = None if timeembeddings is None else timeembeddings[sl].to(device)
        out = tensorf(rays[sl].to(device), ts[sl].to(device), te, xyz_sampled[sl].to(device),
                      z_vals_input[sl].to(device), ray_valid[sl].to(device), is_train=is_train,
                      white_bg=white_bg, ray_type=ray_type, N_samples=N_samples)
        out = list(out)
        if out[5] is not None:
            out[5] = out[5] - xyz_sampled[sl].to(device)  # delta_xyz
        for k, v in zip(keys, out):
            k.append(v)
    cat = lambda l: None if any(v is None for v in l) else torch.cat(l)
    r = [cat(k) for k in keys]
    return (r[0], r[1], r[2], r[3], r[4], r[5], None, r[6], r[7], r[8], r[9])


@torch.no_grad()
def render_rays(tensorf_static, tensorf, rays, ts, N_samples=-1, ray_type="ndc"):
    """No-grad render of a ray chunk through ONE C-ABI call (rdrf_render_fwd): the loop body of
    renderer.py:740-812.  Returns (rgb_map_full[N,3], depth_map_full[N])."""
    from .fields import _cfg_struct, _dynamic_struct, _static_struct
    L.require_device(rays, ts)
    rays, ts = L.f32c(rays), L.f32c(ts)
    N = rays.shape[0]
    S = int(N_samples) if N_samples and N_samples > 0 else tensorf.nSamples
    dev = rays.device
    rgb = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    nbytes = int(L.lib.rdrf_render_workspace_bytes(N, S))
    ws = L.workspace(dev, nbytes)
    PS = _static_struct(tensorf_static._param_list())
    PD = _dynamic_struct(tensorf._param_list())
    cs, cd = _cfg_struct(tensorf_static, ray_type), _cfg_struct(tensorf, ray_type)
    near, far = tensorf.near_far
    L.check(L.lib.rdrf_render_fwd(C.byref(PS), C.byref(cs), C.byref(PD), C.byref(cd), L.ptr(rays),
                                  L.ptr(ts), N, S, C.c_float(near), C.c_float(far), L.ptr(rgb),
                                  L.ptr(depth), L.ptr(ws), C.c_size_t(ws.numel()), L.stream_of(rays)),
            "rdrf_render_fwd")
    return rgb, depth
