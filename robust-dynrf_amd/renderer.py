"""Host mirror of /root/reference/renderer.py:24-315 for the hot path: ``sampleXYZ``,
``raw2outputs`` and a working ``OctreeRender_trilinear_fast`` chunk loop.  All arithmetic runs in
the HIP kernels (rdrf_sample_*, rdrf_composite_*); torch only supplies memory, RNG and autograd."""
import ctypes as C

import torch

from . import _lib as L


class _SampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays, aabb_host, near, far, S, ray_type, jitter, jitter_outer):
        ctx.set_materialize_grads(False)
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
        if g_xyz is None:   # z_vals / valid carry no gradient to the rays
            return None, None, None, None, None, None, None, None
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
        ctx.set_materialize_grads(False)
        L.require_device(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays)
        ins = [L.f32c(t) for t in (rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays)]
        N, S = ins[1].shape
        dev = ins[1].device
        shapes = [(N, 3), (N,), (N,), (N, S)] * 3 + [(N,)]
        outs = [torch.empty(s, device=dev) for s in shapes]
        arr = (C.c_void_p * 13)(*[o.data_ptr() for o in outs])
        # the coin as a device float (one element, 0 / 1): read when the kernel runs, so a captured HIP graph of the
        # iteration follows each replay's coin (step.Trainer(graph=True)); a host bool otherwise
        white_dev = add_white_bg if torch.is_tensor(add_white_bg) else None
        if white_dev is not None and (white_dev.dtype != torch.float32 or white_dev.numel() != 1 or not white_dev.is_cuda):
            raise L.RdrfError("raw2outputs: a device coin is one fp32 element on the GPU")
        white = 0 if white_dev is not None else int(add_white_bg)
        L.check(L.lib.rdrf_composite_fwd(*[L.ptr(t) for t in ins], N, S, L.RAY_TYPES.get(ray_type, 2),
                                         white, L.ptr(white_dev), arr, L.stream_of(ins[1])),
                "rdrf_composite_fwd")
        ctx.ray_type, ctx.white, ctx.white_dev = ray_type, white, white_dev
        ctx.save_for_backward(*ins)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_out):
        ins = ctx.saved_tensors
        N, S = ins[1].shape
        g_out = [None if g is None else L.f32c(g) for g in g_out]
        need = list(ctx.needs_input_grad[:8])
        # like autograd on the reference's op graph, a branch no loss reaches gets NO gradient (not a
        # zero one): rgb_s only feeds rgb_map_full / rgb_map_s, rgb_d only rgb_map_full / rgb_map_d.
        # The field's backward then skips its whole appearance half (MLP, scatter, dW) for that pass.
        have = lambda idx: any(g_out[i] is not None for i in idx)
        need[0] = need[0] and have((0, 4))
        need[2] = need[2] and have((0, 8))
        # sigma_s feeds the full and the static maps, sigma_d / blending the full and the dynamic maps
        # (renderer.py:190-262): e.g. pass E of the trainer consumes only rgb_map_s / depth_map_s, so
        # the dynamic field gets no gradient at all from it
        need[1] = need[1] and have((0, 1, 2, 3, 4, 5, 6, 7, 12))
        need[3] = need[3] and have((0, 1, 2, 3, 8, 9, 10, 11, 12))
        # blending only enters T_full / weights_full (renderer.py:206-262): the dynamic-only maps (8..11) do not depend
        # on it, so a loss on depth_map_d / weights_d alone (passes B-D of the trainer) sends the blending head NO
        # gradient -- and the field's backward then leaves that head, its scatter set and its dW products out
        need[5] = need[5] and have((0, 1, 2, 3, 12))
        g_in = L.zeros_like_many(ins, need)   # one fill for the (up to eight) gradient tensors
        ga = (C.c_void_p * 13)(*[0 if g is None else g.data_ptr() for g in g_out])
        gi = (C.c_void_p * 8)(*[0 if g is None else g.data_ptr() for g in g_in])
        L.check(L.lib.rdrf_composite_bwd(*[L.ptr(t) for t in ins], N, S,
                                         L.RAY_TYPES.get(ctx.ray_type, 2), ctx.white, L.ptr(ctx.white_dev), ga, gi,
                                         L.stream_of(ins[1])), "rdrf_composite_bwd")
        return (*g_in, None, None)


def raw2outputs(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays_chunk, is_train=False,
                ray_type="ndc", add_white_bg=None):
    """renderer.py:173-315.  The train-time coin (`torch.rand((1,)) < 0.5`, renderer.py:269) is
    drawn here on the host unless `add_white_bg` is given: a bool, or a one-element fp32 DEVICE tensor (0 / 1) that the
    kernels read when they run."""
    if add_white_bg is None:
        add_white_bg = bool(is_train and (torch.rand((1,)) < 0.5).item())
    return _CompositeFn.apply(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, z_vals, rays_chunk,
                              ray_type, add_white_bg if torch.is_tensor(add_white_bg) else bool(add_white_bg))


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
def render_rays(tensorf_static, tensorf, rays, ts, N_samples=-1, ray_type="ndc", mode="auto"):
    """No-grad render of a ray chunk through ONE C-ABI call: the loop body of renderer.py:740-812.
    mode "auto" (rdrf_render_fwd: the per-phase launch sequence), "fused" (rdrf_render_fused_fwd:
    one cooperative launch) or "sequence" (rdrf_render_sequence_fwd); all three give the same bits.
    Returns (rgb_map_full[N,3], depth_map_full[N])."""
    from .fields import _attach_packed, _cfg_struct, _dynamic_struct, _static_struct
    L.require_device(rays, ts)
    rays, ts = L.f32c(rays), L.f32c(ts)
    N = rays.shape[0]
    S = int(N_samples) if N_samples and N_samples > 0 else tensorf.nSamples
    dev = rays.device
    rgb = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    nbytes = int(L.lib.rdrf_render_workspace_bytes(N, S))
    ws = L.workspace(dev, nbytes)
    ps_list, pd_list = tensorf_static._param_list(), tensorf._param_list()
    PS, PD = _static_struct(ps_list), _dynamic_struct(pd_list)
    _attach_packed(tensorf_static, PS, ps_list, False, False)   # packed once per (weights, stream), not per chunk
    _attach_packed(tensorf, PD, pd_list, False, True)
    cs, cd = _cfg_struct(tensorf_static, ray_type), _cfg_struct(tensorf, ray_type)
    near, far = tensorf.near_far
    fn = {"auto": L.lib.rdrf_render_fwd, "fused": L.lib.rdrf_render_fused_fwd, "sequence": L.lib.rdrf_render_sequence_fwd}[mode]
    L.check(fn(C.byref(PS), C.byref(cs), C.byref(PD), C.byref(cd), L.ptr(rays), L.ptr(ts), N, S, C.c_float(near),
               C.c_float(far), L.ptr(rgb), L.ptr(depth), L.ptr(ws), C.c_size_t(ws.numel()), L.stream_of(rays)),
            "rdrf_render_" + mode)
    return rgb, depth


_stream_pool = {}


@torch.no_grad()
def render_chunks(tensorf_static, tensorf, rays, ts, chunk, N_samples=-1, ray_type="ndc", streams=8):
    """The chunk loop of renderer.py:740-812 (`for chunk_idx in range(N_rays_all // chunk + ...)`, chunk = 512 at
    renderer.py:732) as ONE native call (rdrf_render_chunks_fwd): the chunks' launch sequences are issued from C,
    round-robin on `streams` HIP streams (0 / 1: all on the current stream).  Issued chunk by chunk from Python the loop
    is host-bound (~250 us of marshalling per chunk against ~190 us of GPU work); and one 512-ray chunk fills only
    64-110 of the 256 CUs, so independent chunks run side by side.  Same bits as render_rays on the whole batch.
    Returns (rgb_map [N,3], depth_map [N])."""
    from .fields import _attach_packed, _cfg_struct, _dynamic_struct, _static_struct
    L.require_device(rays, ts)
    rays, ts = L.f32c(rays), L.f32c(ts)
    N, dev = rays.shape[0], rays.device
    S = int(N_samples) if N_samples and N_samples > 0 else tensorf.nSamples
    chunk = int(chunk)
    rgb = torch.empty(N, 3, device=dev)
    depth = torch.empty(N, device=dev)
    if N == 0:
        return rgb, depth
    ns = int(streams) if streams and streams > 1 and N > chunk else 0
    ws = L.workspace(dev, int(L.lib.rdrf_render_chunks_workspace_bytes(min(chunk, N), S, max(ns, 1))))
    ps_list, pd_list = tensorf_static._param_list(), tensorf._param_list()
    PS, PD = _static_struct(ps_list), _dynamic_struct(pd_list)
    _attach_packed(tensorf_static, PS, ps_list, False, False)   # one image each, packed on the current stream, shared
    _attach_packed(tensorf, PD, pd_list, False, True)           # read-only by every side stream
    cs, cd = _cfg_struct(tensorf_static, ray_type), _cfg_struct(tensorf, ray_type)
    near, far = tensorf.near_far
    pool = _stream_pool.get((dev, ns))
    if pool is None:
        pool = _stream_pool[(dev, ns)] = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    arr = (C.c_void_p * max(ns, 1))(*[st.cuda_stream for st in pool]) if ns else None
    L.check(L.lib.rdrf_render_chunks_fwd(C.byref(PS), C.byref(cs), C.byref(PD), C.byref(cd), L.ptr(rays), L.ptr(ts), N, S, chunk,
                                         C.c_float(near), C.c_float(far), L.ptr(rgb), L.ptr(depth), L.ptr(ws),
                                         C.c_size_t(ws.numel()), L.stream_of(rays), arr, ns), "rdrf_render_chunks_fwd")
    for st in pool:   # the caching allocator must not hand these buffers to another stream's request while the side
        for t in (rays, ts, rgb, depth, ws):   # streams may still read / write them
            t.record_stream(st)
    return rgb, depth


@torch.no_grad()
def render_frame(tensorf_static, tensorf, poses9, focal, frame, H, W, N_samples=-1, ray_type="ndc",
                 chunk=None, t=None):
    """Whole-frame no-grad render (the per-image body of renderer.py:661-966 `evaluation`): rays of
    every pixel of `frame` are generated on the device and pushed through rdrf_render_fwd in chunks
    (default: the whole frame in one launch sequence).  `t` overrides the frame's own time in [-1,1].
    Returns (rgb [H,W,3] clamped to [0,1], depth [H,W])."""
    from .ray_utils import generate_rays
    dev = poses9.device
    T = poses9.shape[0]
    ids = torch.arange(H * W, device=dev) + int(frame) * H * W
    rays = generate_rays(ids, poses9, focal, H, W, ndc=ray_type == "ndc", near=1.0)
    tv = (2.0 * frame / max(T - 1, 1) - 1.0) if t is None else float(t)
    ts = torch.full((H * W,), tv, device=dev)
    S_ = int(N_samples) if N_samples and N_samples > 0 else tensorf.nSamples
    if not chunk:   # whole frame in one launch sequence, bounded by the kernels' 32-bit sample indices
        chunk = max(1, min(H * W, (2 ** 31 - 1) // (3 * S_) - 1))
    chunk = int(chunk)
    rgb = torch.empty(H * W, 3, device=dev)
    depth = torch.empty(H * W, device=dev)
    for c0 in range(0, H * W, chunk):
        r, d = render_rays(tensorf_static, tensorf, rays[c0:c0 + chunk], ts[c0:c0 + chunk], N_samples, ray_type)
        rgb[c0:c0 + chunk], depth[c0:c0 + chunk] = r, d
    return rgb.clamp_(0.0, 1.0).view(H, W, 3), depth.view(H, W)


def psnr(img, ref):
    """-10 log10(mse) on the device (renderer.py:905-906 computes it per image on the host)."""
    return -10.0 * torch.log10(((img - ref) ** 2).mean())


# --------------------------------------------------------------------------------------------
# induced optical flow / disparity (renderer.py:1266-1392) -- SURVEY.md 8f rank 1
# --------------------------------------------------------------------------------------------
def _focal_tensor(focal, dev):
    if torch.is_tensor(focal):
        L.require_device(focal)
        return focal.reshape(1).float()
    return torch.full((1,), float(focal), device=dev)


class _InduceFlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, W, focal, c2w, weights, pts, pts_2d, rays, ray_type):
        ctx.set_materialize_grads(False)
        L.require_device(c2w, weights, pts, pts_2d, rays)
        c2w, weights, pts, pts_2d, rays = (L.f32c(t) for t in (c2w, weights, pts, pts_2d, rays))
        focal = L.f32c(focal)
        N, S = weights.shape
        if c2w.shape != (N, 3, 4) or pts.shape != (N, S, 3) or rays.shape != (N, 6) or pts_2d.shape != (N, 2):
            raise L.RdrfError("induce_flow: expected pose [N,3,4], weights [N,S], pts [N,S,3], pts_2d [N,2], "
                              "rays [N,6]")
        if ray_type not in ("ndc", "contract"):
            raise L.RdrfError("induce_flow: ray_type must be 'ndc' or 'contract' (renderer.py:1343-1361)")
        flow = torch.empty(N, 2, device=weights.device)
        disp = torch.empty(N, 1, device=weights.device)
        L.check(L.lib.rdrf_induce_flow_fwd(int(H), int(W), L.ptr(focal), L.ptr(c2w), L.ptr(weights),
                                           L.ptr(pts), L.ptr(pts_2d), L.ptr(rays), N, S,
                                           L.RAY_TYPES[ray_type], L.ptr(flow), L.ptr(disp),
                                           L.stream_of(weights)), "rdrf_induce_flow_fwd")
        ctx.hw, ctx.ray_type = (int(H), int(W)), ray_type
        ctx.save_for_backward(focal, c2w, weights, pts, rays)
        return flow, disp

    @staticmethod
    def backward(ctx, g_flow, g_disp):
        focal, c2w, weights, pts, rays = ctx.saved_tensors
        N, S = weights.shape
        need = ctx.needs_input_grad     # H, W, focal, c2w, weights, pts, pts_2d, rays, ray_type
        g_focal, g_c2w, g_w, g_pts, g_rays = L.zeros_like_many([focal, c2w, weights, pts, rays],
                                                               [need[2], need[3], need[4], need[5], need[7]])
        g_flow = None if g_flow is None else L.f32c(g_flow)
        g_disp = None if g_disp is None else L.f32c(g_disp)
        L.check(L.lib.rdrf_induce_flow_bwd(ctx.hw[0], ctx.hw[1], L.ptr(focal), L.ptr(c2w), L.ptr(weights),
                                           L.ptr(pts), L.ptr(rays), N, S, L.RAY_TYPES[ctx.ray_type],
                                           L.ptr(g_flow), L.ptr(g_disp), L.ptr(g_w), L.ptr(g_pts),
                                           L.ptr(g_rays), L.ptr(g_c2w), L.ptr(g_focal),
                                           L.stream_of(weights)), "rdrf_induce_flow_bwd")
        g_p2d = None if (not need[6] or g_flow is None) else -g_flow
        return (None, None, g_focal, g_c2w, g_w, g_pts, g_p2d, g_rays, None)


def render_3d_point(H, W, f, c2w, weights, pts, rays, ray_type="ndc"):
    """renderer.py:1334-1378: weight-averaged 3-D point along each ray, projected into the camera
    c2w [N,3,4]; returns (pixel coordinates [N,2], NDC depth [N,1])."""
    dev = weights.device
    zero = torch.zeros(weights.shape[0], 2, device=dev)
    flow, disp = _InduceFlowFn.apply(H, W, _focal_tensor(f, dev), c2w, weights, pts, zero, rays, ray_type)
    return flow, disp


def induce_flow(H, W, focal, pose_neighbor, weights, pts_3d_neighbor, pts_2d, rays, ray_type="ndc"):
    """renderer.py:1381-1392: (induced_flow [N,2], induced_disp [N,1])."""
    return _InduceFlowFn.apply(H, W, _focal_tensor(focal, weights.device), pose_neighbor, weights,
                               pts_3d_neighbor, pts_2d, rays, ray_type)


def render_single_3d_point(H, W, f, c2w, pt_NDC):
    """renderer.py:1301-1331: the S = 1, weight 1 case; disparity is returned as (z_ndc + 1) / 2."""
    N = pt_NDC.shape[0]
    dev = pt_NDC.device
    one = torch.ones(N, 1, device=dev)
    plane, d = _InduceFlowFn.apply(H, W, _focal_tensor(f, dev), c2w, one, pt_NDC.reshape(N, 1, 3),
                                   torch.zeros(N, 2, device=dev), torch.zeros(N, 6, device=dev), "ndc")
    return plane, (d + 1.0) / 2.0


def induce_flow_single(H, W, focal, pose_neighbor, pts_3d_neighbor, pts_2d):
    """renderer.py:1381-1386 (induce_flow_single)."""
    return render_single_3d_point(H, W, focal, pose_neighbor, pts_3d_neighbor)[0] - pts_2d


# --------------------------------------------------------------------------------------------
# distortion loss (train.py:19-23 imports it from torch_efficient_distloss)
# --------------------------------------------------------------------------------------------
class _DistLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, m, interval):
        L.require_device(w, m)
        w, m = L.f32c(w), L.f32c(m)
        N, S = w.shape
        ipt = None
        if torch.is_tensor(interval):
            if interval.numel() == 1:
                interval = float(interval)
            else:
                L.require_device(interval)
                ipt, interval = L.f32c(interval).reshape(N, S), 0.0
        loss_ray = torch.empty(N, device=w.device)
        L.check(L.lib.rdrf_distloss_fwd(L.ptr(w), L.ptr(m), C.c_float(float(interval)), L.ptr(ipt), N, S,
                                        L.ptr(loss_ray), L.stream_of(w)), "rdrf_distloss_fwd")
        ctx.interval, ctx.ipt = float(interval), ipt
        ctx.save_for_backward(w, m)
        return loss_ray

    @staticmethod
    def backward(ctx, g_ray):
        w, m = ctx.saved_tensors
        N, S = w.shape
        g_w = torch.zeros_like(w)
        L.check(L.lib.rdrf_distloss_bwd(L.ptr(w), L.ptr(m), C.c_float(ctx.interval), L.ptr(ctx.ipt), N, S,
                                        L.ptr(L.f32c(g_ray)), L.ptr(g_w), L.stream_of(w)), "rdrf_distloss_bwd")
        return g_w, None, None


def distloss_rays(w, m, interval):
    """the per-ray values of eff_distloss [N] (their mean is the loss): lets the trainer fold the mean and the loss
    weight into its fused loss reduction instead of three scalar launches per call"""
    return _DistLossFn.apply(w, m, interval)


def eff_distloss(w, m, interval):
    """torch_efficient_distloss.eff_distloss: w, m [N,S] (m ascending along a ray), interval a scalar
    or [N,S]; mean over rays of  sum_ij w_i w_j |m_i - m_j| + (1/3) sum_i interval w_i^2."""
    return _DistLossFn.apply(w, m, interval).sum() / w.shape[0]


def flatten_eff_distloss(w, m, interval, ray_id, n_rays=None):
    """torch_efficient_distloss.flatten_eff_distloss as the reference calls it (train.py:1299-1312):
    flattened [N*S] points with ray_id = tile(arange(N), S).  Only that regular layout is built;
    `n_rays` (= ray_id.max() + 1) may be passed to avoid the device->host read."""
    if n_rays is None:
        n_rays = int(ray_id.max().item()) + 1
        S = w.numel() // n_rays
        if w.numel() != n_rays * S or not bool((ray_id.reshape(n_rays, S)
                                                 == torch.arange(n_rays, device=ray_id.device)[:, None]).all()):
            raise NotImplementedError("flatten_eff_distloss: only ray_id = tile(arange(N), S) is built")
    S = w.numel() // n_rays
    if torch.is_tensor(interval) and interval.numel() > 1:
        interval = interval.reshape(n_rays, S)
    return _DistLossFn.apply(w.reshape(n_rays, S), m.reshape(n_rays, S), interval).sum() / n_rays
