"""Generate golden input/output vectors by importing the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only). Nothing of the reference
is copied: this script imports it (with stub modules for never-executed imports and a CPU
``get_device`` shim), runs its ``TensorVMSplit`` /
``TensorVMSplit_TimeEmbedding`` / ``renderer.sampleXYZ`` / ``renderer.raw2outputs`` /
ray-generation functions on seeded inputs, and stores inputs, weights, outputs and autograd
gradients as ``tests/golden/*.npz``.  Those fixtures (data only) travel to the GPU box.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types
import io
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("RODYNRF_REFERENCE", "/root/reference")


def import_reference():
    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    stub("imageio")
    stub("plyfile")
    stub("cv2", COLORMAP_JET=2)
    stub("kornia").create_meshgrid = lambda H, W, normalized_coordinates=False: torch.stack(
        torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32),
                       indexing="xy"), -1)[None]
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    sk = stub("skimage")
    sk.measure = stub("skimage.measure")
    sk.morphology = stub("skimage.morphology")

    class _E(dict):
        __getattr__ = dict.__getitem__

    stub("easydict", EasyDict=_E)
    _gd = torch.Tensor.get_device
    torch.Tensor.get_device = lambda t: t.device if not t.is_cuda else _gd(t)
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from models.tensoRF import TensorVMSplit, TensorVMSplit_TimeEmbedding
        import renderer
        from dataLoader import ray_utils
        import camera
    return TensorVMSplit, TensorVMSplit_TimeEmbedding, renderer, ray_utils, camera


def build_fields(TS, TD, aabb, grid, act, static_head, density_shift, seed):
    common = dict(density_n_comp=[16, 4, 4], appearance_n_comp=[48, 12, 12], app_dim=27,
                  near_far=[0.0, 1.0], alphaMask_thres=1e-4, density_shift=density_shift,
                  distance_scale=25, pos_pe=6, view_pe=0, featureC=128, step_ratio=2.0,
                  fea2denseAct=act)
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        st = TS(aabb, grid, 12, "cpu", shadingMode=static_head, fea_pe=2, **common)
        dy = TD(aabb, grid, 12, "cpu", shadingMode="MLP_Fea_late_view", fea_pe=0, **common)
    return st, dy


def to_np(d, prefix, out):
    for k, v in d.items():
        out[prefix + k] = v.detach().cpu().numpy()


def gen_case(name, mods, ray_type, act, static_head, grid, N, S, seed, density_shift, jitter):
    TS, TD, renderer, _, _ = mods
    if ray_type == "ndc":
        aabb = torch.tensor([[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
    else:
        aabb = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]])
    st, dy = build_fields(TS, TD, aabb, grid, act, static_head, density_shift, seed)
    if ray_type == "contract":
        st.near_far = [0.0, 256.0]
        dy.near_far = [0.0, 256.0]
    # boost the dynamic density head a little so relu leaves a healthy mix of zero / non-zero sigma
    g = torch.Generator().manual_seed(seed + 1)
    if ray_type == "ndc":
        o = torch.stack([torch.empty(N).uniform_(-1.45, 1.45, generator=g),
                         torch.empty(N).uniform_(-1.6, 1.6, generator=g), -torch.ones(N)], -1)
        d = torch.stack([torch.randn(N, generator=g) * 0.15, torch.randn(N, generator=g) * 0.15,
                         2 * torch.ones(N)], -1)
    else:
        o = torch.randn(N, 3, generator=g) * 0.3
        d = torch.randn(N, 3, generator=g)
        d = d / d.norm(dim=-1, keepdim=True) * torch.empty(N, 1).uniform_(0.8, 1.2, generator=g)
    rays = torch.cat([o, d], -1).requires_grad_(True)
    ts = (torch.randint(0, 12, (N,), generator=g).float() * 2 / 11 - 1)

    out = {}
    out["meta.ray_type"] = np.array(ray_type)
    out["meta.act"] = np.array(act)
    out["meta.static_head"] = np.array(static_head)
    out["meta.grid"] = np.array(grid)
    out["meta.density_shift"] = np.array(density_shift, dtype=np.float32)
    out["meta.near_far"] = np.array(dy.near_far, dtype=np.float32)
    out["aabb"] = aabb.numpy()
    out["rays"] = rays.detach().numpy()
    out["ts"] = ts.numpy()

    # sampler (jitter reproduced by replaying the generator: the reference calls torch.rand_like
    # on a (1,S) / (1,inner+1) fp32 tensor)
    jseed = seed + 7
    if jitter:
        torch.manual_seed(jseed)
        if ray_type == "ndc":
            out["jitter"] = torch.rand(1, S).numpy()
        else:
            inner = S - S // 2
            out["jitter"] = torch.rand(1, inner + 1).numpy()
            out["jitter_outer"] = torch.rand(1, S // 2 + 1).numpy()
        torch.manual_seed(jseed)
    xyz, z, valid = renderer.sampleXYZ(dy, rays, N_samples=S, ray_type=ray_type, is_train=jitter)
    out["xyz"] = xyz.detach().numpy()
    out["z"] = z.detach().numpy()
    out["valid"] = valid.numpy()

    o_s = st(rays, ts, None, xyz, z, valid, is_train=True, white_bg=True, ray_type=ray_type,
             N_samples=S)
    o_d = dy(rays, ts, None, xyz, z, valid, is_train=True, white_bg=True, ray_type=ray_type,
             N_samples=S)
    names = ["_0", "_1", "blending", "pts_ref", "weight", "xyz_prime", "rgb", "sigma", "z", "dists"]
    for k, v in zip(names, o_s):
        if v is not None:
            out["fs." + k] = v.detach().numpy()
    for k, v in zip(names, o_d):
        if v is not None:
            out["fd." + k] = v.detach().numpy()

    onames = ["rgb_map_full", "depth_map_full", "acc_map_full", "weights_full", "rgb_map_s",
              "depth_map_s", "acc_map_s", "weights_s", "rgb_map_d", "depth_map_d", "acc_map_d",
              "weights_d", "dynamicness_map"]

    def comp(is_train, want_white):
        if is_train:
            sd_ = 0
            while True:
                torch.manual_seed(sd_)
                if (torch.rand((1,)) < 0.5).item() == want_white:
                    break
                sd_ += 1
            torch.manual_seed(sd_)
        return renderer.raw2outputs(o_s[6], o_s[7], o_d[6], o_d[7], o_d[9], o_d[2], o_d[8], rays,
                                    is_train=is_train, ray_type=ray_type)

    c_eval = comp(False, False)
    c0 = comp(True, False)
    c1 = comp(True, True)
    for k, a, b, c in zip(onames, c_eval, c0, c1):
        out["ce." + k] = a.detach().numpy()
        out["c0." + k] = b.detach().numpy()
        out["c1." + k] = c.detach().numpy()
    assert np.allclose(out["ce.rgb_map_full"], out["c0.rgb_map_full"])

    sf_f, sf_b = dy.get_forward_backward_scene_flow(o_d[3], ts)
    out["sf.f"] = sf_f.detach().numpy()
    out["sf.b"] = sf_b.detach().numpy()

    # fixed scalar loss over everything the trainer consumes, all paths live (pass-E-like)
    gl = torch.Generator().manual_seed(seed + 3)
    L = 0.0
    for k, v in zip(onames, c1):
        r = torch.randn(v.shape, generator=gl)
        out["lw.c1." + k] = r.numpy()
        L = L + (v * r).sum()
    for k, v in (("blending", o_d[2]), ("weight", o_d[4]), ("xyz_prime", o_d[5]),
                 ("weight_s", o_s[4])):
        r = torch.randn(v.shape, generator=gl)
        out["lw.f." + k] = r.numpy()
        L = L + (v * r).sum()
    for k, v in (("sf_f", sf_f), ("sf_b", sf_b)):
        r = torch.randn(v.shape, generator=gl)
        out["lw." + k] = r.numpy()
        L = L + (v * r).sum()
    out["loss"] = L.detach().numpy()
    ps = [p for p in st.parameters()]
    pd = [p for p in dy.parameters()]
    grads = torch.autograd.grad(L, ps + pd + [rays], allow_unused=True)
    for (k, _), gv in zip(st.named_parameters(), grads[: len(ps)]):
        out["gs." + k] = (gv if gv is not None else torch.zeros(())).numpy()
    for (k, _), gv in zip(dy.named_parameters(), grads[len(ps): len(ps) + len(pd)]):
        out["gd." + k] = (gv if gv is not None else torch.zeros(())).numpy()
    out["g.rays"] = grads[-1].numpy()

    to_np(st.state_dict(), "s.", out)
    to_np(dy.state_dict(), "d.", out)

    # function-level vectors incl. out-of-range coordinates (zero padding)
    M = 96
    xn = torch.empty(M, 3).uniform_(-1.3, 1.3, generator=g)
    xn[:8] = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [1.0, -1.0, 0.0], [0.0, 0.0, 0.0],
                           [1.0001, 0.5, -0.5], [-1.0001, 0.5, 0.5], [0.999999, -0.999999, 1.0],
                           [2.5, -3.0, 0.1]])
    tf = (torch.randint(0, 12, (M,), generator=g).float() * 2 / 11 - 1)
    out["fn.xn"] = xn.numpy()
    out["fn.t"] = tf.numpy()
    with torch.no_grad():
        out["fn.s_density"] = st.compute_densityfeature(xn, tf, None).numpy()
        out["fn.s_app"] = st.compute_appfeature(xn, tf, None).numpy()
        out["fn.d_density"] = dy.compute_densityfeature(xn, tf, None).numpy()
        out["fn.d_blending"] = dy.compute_blendingfeature(xn, tf, None).numpy()
        out["fn.d_app"] = dy.compute_appfeature(xn, tf, None).numpy()
        out["fn.d_warp"] = dy.warp_coordinate(dy.unnormalize_coord(xn), tf).numpy()
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    fa = float(((o_d[4] > 1e-4).float().mean()))
    print(f"{name}: loss {float(L):.5f} valid {float(valid.float().mean()):.3f} "
          f"app_mask_d {fa:.3f} app_mask_s {float((o_s[4] > 1e-4).float().mean()):.3f} "
          f"sigma_d>0 {float((o_d[7] > 0).float().mean()):.3f}")


def gen_raygen(mods):
    _, _, _, ru, camera = mods
    g = torch.Generator().manual_seed(99)
    T, H, W = 5, 27, 48
    poses = torch.zeros(T, 9)
    poses[:, 0] = 1
    poses[:, 4] = 1
    poses = poses + 0.05 * torch.randn(T, 9, generator=g)
    poses.requires_grad_(True)
    focal = torch.tensor(max(H, W) / 2.0 * 1.7320508, requires_grad=True)
    ids = torch.randint(0, T * H * W, (64,), generator=g)
    col, row, view = ids % W, (ids // W) % H, ids // (W * H)
    mtx = camera.pose_to_mtx(poses)
    dirs = ru.get_ray_directions_lean(col, row, [focal, focal], [W / 2, H / 2])
    ro, rd = ru.get_rays_lean(dirs, mtx[view])
    ro_n, rd_n = ru.ndc_rays_blender2(H, W, [focal, focal], 1.0, ro, rd)
    rays = torch.cat([ro_n, rd_n], -1)
    r = torch.randn(rays.shape, generator=g)
    gp, gf = torch.autograd.grad((rays * r).sum(), [poses, focal])
    np.savez(os.path.join(HERE, "raygen.npz"), poses=poses.detach().numpy(),
             focal=focal.detach().numpy(), ids=ids.numpy(), H=H, W=W,
             rays_world=torch.cat([ro, rd], -1).detach().numpy(), rays=rays.detach().numpy(),
             lw=r.numpy(), g_poses=gp.numpy(), g_focal=gf.numpy(), mtx=mtx.detach().numpy())
    print("raygen: ok")


def gen_induce_flow(mods):
    """renderer.py:1266-1392: induce_flow / render_3d_point / render_single_3d_point, both ray
    types, with gradients wrt every differentiable input (weights, pts, rays, c2w, focal)."""
    _, _, R, _, camera = mods
    out = {}
    for rt, seed in (("ndc", 501), ("contract", 502)):
        g = torch.Generator().manual_seed(seed)
        N, S, H, W = 24, 19, 27, 48
        focal = torch.tensor(max(H, W) / 2.0 * 1.7320508, requires_grad=True)
        p9 = torch.zeros(N, 9)
        p9[:, 0] = 1
        p9[:, 4] = 1
        p9 = p9 + 0.05 * torch.randn(N, 9, generator=g)
        c2w = camera.pose_to_mtx(p9).detach().clone().requires_grad_(True)     # [N,3,4]
        w = torch.rand(N, S, generator=g)
        w = w / w.sum(-1, keepdim=True) * torch.rand(N, 1, generator=g)         # acc in (0,1)
        w[:3] = 0.0                                                              # empty rays
        if rt == "ndc":
            pts = torch.empty(N, S, 3).uniform_(-0.9, 0.9, generator=g)
            pts[..., 2] = torch.empty(N, S).uniform_(-0.95, 0.97, generator=g)
            pts[5, :, 2] = 1.5       # beyond the clamp of NDC2world
            rays = torch.cat([torch.empty(N, 2).uniform_(-0.8, 0.8, generator=g), -torch.ones(N, 1),
                              torch.empty(N, 2).uniform_(-0.1, 0.1, generator=g), 2 * torch.ones(N, 1)], -1)
        else:
            pts = torch.empty(N, S, 3).uniform_(-1.9, 1.9, generator=g)
            pts[:6] *= 0.3           # inside the unit box: identity branch of contract2world
            rays = torch.cat([torch.empty(N, 3).uniform_(-0.2, 0.2, generator=g),
                              torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)], -1)
            rays[7, 3:] *= 1e-3      # farthest point stays inside the unit box
        w.requires_grad_(True)
        pts.requires_grad_(True)
        rays.requires_grad_(True)
        p2d = torch.stack([torch.rand(N, generator=g) * W, torch.rand(N, generator=g) * H], -1)
        flow, disp = R.induce_flow(H, W, focal, c2w, w, pts.clone() if rt == "contract" else pts, p2d, rays,
                                   ray_type=rt)
        lf, ld = torch.randn(flow.shape, generator=g), torch.randn(disp.shape, generator=g)
        gw, gp, gr, gc, gf = torch.autograd.grad((flow * lf).sum() + (disp * ld).sum(),
                                                 [w, pts, rays, c2w, focal])
        pre = f"{rt}."
        for k, v in dict(H=H, W=W, focal=focal, c2w=c2w, weights=w, pts=pts, pts_2d=p2d, rays=rays,
                         flow=flow, disp=disp, lw_flow=lf, lw_disp=ld, g_weights=gw, g_pts=gp,
                         g_rays=gr, g_c2w=gc, g_focal=gf).items():
            out[pre + k] = v.detach().numpy() if torch.is_tensor(v) else v
        if rt == "ndc":
            pt = pts.detach()[:, 0].clone().requires_grad_(True)
            pl, dd = R.render_single_3d_point(H, W, focal.detach(), c2w.detach(), pt)
            out["single.pt"] = pt.detach().numpy()
            out["single.plane"] = pl.detach().numpy()
            out["single.disp"] = dd.detach().numpy()
            out["single.flow"] = R.induce_flow_single(H, W, focal.detach(), c2w.detach(), pt, p2d).detach().numpy()
    np.savez(os.path.join(HERE, "induce_flow.npz"), **out)
    print("induce_flow: ok")


def gen_tv(mods):
    """utils.py:157-181 TVLoss on single tensors (incl. a line: value NaN, gradient finite) and the
    fields' TV_loss_* family sums (models/tensoRF.py:100-116, 418-444) on the weights of the
    ndc_relu case, with gradients."""
    TS, TD, _, _, _ = mods
    import utils as ref_utils
    tv = ref_utils.TVLoss()
    out = {}
    g = torch.Generator().manual_seed(77)
    for name, shape in (("plane", (1, 4, 9, 7)), ("line", (1, 16, 11, 1)), ("row", (1, 3, 1, 8))):
        x = torch.randn(shape, generator=g, requires_grad=True)
        out[f"t.{name}.x"] = x.detach().numpy()
        for ax in (None, "h", "w"):
            v = tv(x, ax) if ax else tv(x)
            gx, = torch.autograd.grad(v, x, allow_unused=True)
            out[f"t.{name}.v.{ax}"] = v.detach().numpy()
            out[f"t.{name}.g.{ax}"] = (torch.zeros_like(x) if gx is None else gx).numpy()
    case = np.load(os.path.join(HERE, "ndc_relu.npz"), allow_pickle=True)
    grid = [int(v) for v in case["meta.grid"]]
    st, dy = build_fields(TS, TD, torch.from_numpy(case["aabb"]), grid, "relu", "MLP_Fea", -10.0, 1)
    st.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("s.")})
    dy.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("d.")})
    for tag, mod, fams in (("s", st, ("density", "app")), ("d", dy, ("density", "blending", "app"))):
        for fam in fams:
            total = getattr(mod, f"TV_loss_{fam}")(tv)
            ps = list(getattr(mod, f"{fam}_plane")) + list(getattr(mod, f"{fam}_line"))
            gs = torch.autograd.grad(total, ps)
            out[f"f.{tag}.{fam}.total"] = total.detach().numpy()
            for i, gg in enumerate(gs):
                nm = f"{fam}_plane.{i}" if i < 3 else f"{fam}_line.{i - 3}"
                out[f"f.{tag}.{fam}.g.{nm}"] = gg.numpy()
    # factor-space regularisers with default weight 0 in Nvidia.txt, used by DAVIS.txt:
    # density_L1 / blending_L1 (models/tensoRF.py:80-98, 378-416), vector_comp_diffs (:63-78)
    for tag, mod, names in (("s", st, ("density_L1", "vector_comp_diffs")),
                            ("d", dy, ("density_L1", "blending_L1"))):   # the dynamic class has no vector_comp_diffs
        for nm in names:
            val = getattr(mod, nm)()
            fam = "blending" if nm == "blending_L1" else "density"
            ps = list(getattr(mod, f"{fam}_plane")) + list(getattr(mod, f"{fam}_line"))
            if nm == "vector_comp_diffs":
                ps = list(mod.density_line) + list(mod.app_line)
            gs = torch.autograd.grad(val, ps, allow_unused=True)
            out[f"r.{tag}.{nm}.value"] = val.detach().numpy()
            for i, gg in enumerate(gs):
                out[f"r.{tag}.{nm}.g{i}"] = (torch.zeros_like(ps[i]) if gg is None else gg).numpy()
    np.savez(os.path.join(HERE, "tv.npz"), **out)
    print("tv: ok", float(out["f.s.density.total"]), float(out["r.d.density_L1.value"]))


def gen_fn_grads(mods):
    """Gradients of the per-point building blocks (compute_densityfeature / compute_appfeature /
    compute_blendingfeature / warp_coordinate of both fields, models/tensoRF.py:118-196, 521-811) wrt
    every parameter and wrt the coordinates, on the weights and points of the ndc_relu case, from the
    reference's own autograd.  L = sum_k <fn_k, r_k> with fixed random r_k."""
    TS, TD, _, _, _ = mods
    case = np.load(os.path.join(HERE, "ndc_relu.npz"), allow_pickle=True)
    grid = [int(v) for v in case["meta.grid"]]
    st, dy = build_fields(TS, TD, torch.from_numpy(case["aabb"]), grid, "relu", "MLP_Fea", -10.0, 1)
    st.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("s.")})
    dy.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("d.")})
    xn = torch.from_numpy(case["fn.xn"]).clone().requires_grad_(True)
    xu = dy.unnormalize_coord(torch.from_numpy(case["fn.xn"])).clone().requires_grad_(True)
    tf = torch.from_numpy(case["fn.t"])
    g = torch.Generator().manual_seed(4242)
    outs = {"s_density": st.compute_densityfeature(xn, tf, None), "s_app": st.compute_appfeature(xn, tf, None),
            "d_density": dy.compute_densityfeature(xn, tf, None),
            "d_blending": dy.compute_blendingfeature(xn, tf, None),
            "d_app": dy.compute_appfeature(xn, tf, None), "d_warp": dy.warp_coordinate(xu, tf)}
    out = {}
    L = 0.0
    for k, v in outs.items():
        assert np.allclose(v.detach().numpy(), case["fn." + k], rtol=0, atol=0), k
        r = torch.randn(v.shape, generator=g)
        out["lw." + k] = r.numpy()
        L = L + (v * r).sum()
    ps, pd = list(st.parameters()), list(dy.parameters())
    grads = torch.autograd.grad(L, ps + pd + [xn, xu], allow_unused=True)
    for (k, _), gv in zip(st.named_parameters(), grads[: len(ps)]):
        out["gs." + k] = (gv if gv is not None else torch.zeros(())).numpy()
    for (k, _), gv in zip(dy.named_parameters(), grads[len(ps): len(ps) + len(pd)]):
        out["gd." + k] = (gv if gv is not None else torch.zeros(())).numpy()
    out["g.xn"] = grads[-2].numpy()
    out["g.xu"] = grads[-1].numpy()
    out["loss"] = L.detach().numpy()
    np.savez(os.path.join(HERE, "fn_grads.npz"), **out)
    print("fn_grads: ok", float(L))


def gen_pass_structure(mods):
    """the detach / liveness pattern of the trainer, re-enacted on the imported
    reference models.  train.py itself does not import here (configargparse, tensorboard, dataset files), so
    the two call sequences are replayed call by call with the argument lists of the trainer:

      pass A  train.py:1092-1162  sampleXYZ(rays.detach(), is_train=True) -> tensorf_static(rays.detach(), ...)
              -> tensorf(rays.detach(), ...) -> raw2outputs(rgb_s.detach(), sigma_s.detach(), ..., rays.detach(),
              is_train=True)
      pass E  train.py:1756-1823  the same three calls on rays WITH grad, static outputs live
      L = 3 mse(rgb_map_full_A) + mse(rgb_map_d_A)              train.py:1323-1332
          + sum((rgb_map_s_E - rgb)^2 (1 - fg)) / (sum(1 - fg) + 1e-8) / 3       train.py:1827-1835

    on the weights / rays / times of two existing cases (their state_dicts are NOT stored again).  The global
    RNG is seeded once; the draws (jitter A, coin A, jitter E, coin E -- models/tensorBase.py:492, 530-545,
    renderer.py:269) are recorded by replaying the generator."""
    TS, TD, renderer, _, _ = mods
    names = ["_0", "_1", "blending", "pts_ref", "weight", "xyz_prime", "rgb", "sigma", "z", "dists"]
    onames = ["rgb_map_full", "depth_map_full", "acc_map_full", "weights_full", "rgb_map_s",
              "depth_map_s", "acc_map_s", "weights_s", "rgb_map_d", "depth_map_d", "acc_map_d",
              "weights_d", "dynamicness_map"]
    for case_name, want in (("ndc_relu", (True, False)), ("contract_relu_te", (False, True))):
        case = np.load(os.path.join(HERE, case_name + ".npz"), allow_pickle=True)
        rt, act, head = str(case["meta.ray_type"]), str(case["meta.act"]), str(case["meta.static_head"])
        grid = [int(v) for v in case["meta.grid"]]
        st, dy = build_fields(TS, TD, torch.from_numpy(case["aabb"]), grid, act, head,
                              float(case["meta.density_shift"]), 1)
        st.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("s.")})
        dy.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("d.")})
        nf = [float(v) for v in case["meta.near_far"]]
        st.near_far = nf
        dy.near_far = nf
        rays = torch.from_numpy(case["rays"]).clone().requires_grad_(True)
        ts = torch.from_numpy(case["ts"])
        N, S = rays.shape[0], int(case["z"].shape[1])
        g = torch.Generator().manual_seed(606)
        rgb_t = torch.rand(N, 3, generator=g)
        fg = (torch.rand(N, 1, generator=g) < 0.3).float()       # allforegroundmasks_train[..., 0:1]

        def draws():
            if rt == "ndc":
                return (torch.rand(1, S),)
            return torch.rand(1, S - S // 2 + 1), torch.rand(1, S // 2 + 1)

        seed = 0
        while True:   # a seed whose two coins are `want` (white background in A / not in E, and vice versa)
            torch.manual_seed(seed)
            jA = draws()
            cA = bool(torch.rand((1,)) < 0.5)
            jE = draws()
            cE = bool(torch.rand((1,)) < 0.5)
            if (cA, cE) == want:
                break
            seed += 1
        out = {"meta.case": np.array(case_name), "meta.seed": np.array(seed), "rgb_train": rgb_t.numpy(),
               "fg": fg.numpy(), "A.white": np.array(cA), "E.white": np.array(cE),
               "A.jitter": jA[0].numpy(), "E.jitter": jE[0].numpy()}
        if rt != "ndc":
            out["A.jitter_outer"], out["E.jitter_outer"] = jA[1].numpy(), jE[1].numpy()
        torch.manual_seed(seed)
        kw = dict(is_train=True, white_bg=True, ray_type=rt, N_samples=S)
        # ---- pass A (train.py:1092-1162)
        xyz, z, valid = renderer.sampleXYZ(dy, rays.detach(), N_samples=S, ray_type=rt, is_train=True)
        oAs = st(rays.detach(), ts, None, xyz, z, valid, **kw)
        oAd = dy(rays.detach(), ts, None, xyz, z, valid, **kw)
        cA_out = renderer.raw2outputs(oAs[6].detach(), oAs[7].detach(), oAd[6], oAd[7], oAd[9], oAd[2], oAd[8],
                                      rays.detach(), is_train=True, ray_type=rt)
        # ---- pass E (train.py:1756-1823)
        xyzE, zE, validE = renderer.sampleXYZ(dy, rays, N_samples=S, ray_type=rt, is_train=True)
        oEs = st(rays, ts, None, xyzE, zE, validE, **kw)
        oEd = dy(rays, ts, None, xyzE, zE, validE, **kw)
        cE_out = renderer.raw2outputs(oEs[6], oEs[7], oEd[6], oEd[7], oEd[9], oEd[2], oEd[8], rays,
                                      is_train=True, ray_type=rt)
        for tag, smp, o_s, o_d, c in (("A", (xyz, z, valid), oAs, oAd, cA_out), ("E", (xyzE, zE, validE), oEs, oEd, cE_out)):
            out[tag + ".xyz"], out[tag + ".z"], out[tag + ".valid"] = (smp[0].detach().numpy(), smp[1].detach().numpy(),
                                                                       smp[2].numpy())
            for k, v in zip(names, o_s):
                if v is not None:
                    out[f"{tag}.fs.{k}"] = v.detach().numpy()
            for k, v in zip(names, o_d):
                if v is not None:
                    out[f"{tag}.fd.{k}"] = v.detach().numpy()
            for k, v in zip(onames, c):
                out[f"{tag}.c.{k}"] = v.detach().numpy()
        rgb_map_full, rgb_map_d, rgb_map_s = cA_out[0], cA_out[8], cE_out[4]
        l_full = torch.mean((rgb_map_full - rgb_t) ** 2)
        l_d = torch.mean((rgb_map_d - rgb_t) ** 2)
        l_s = torch.sum((rgb_map_s - rgb_t) ** 2 * (1.0 - fg)) / (torch.sum(1.0 - fg) + 1e-8) / rgb_map_s.shape[-1]
        L = 3.0 * l_full + 1.0 * l_d + 1.0 * l_s
        out["loss"], out["loss_full"], out["loss_d"], out["loss_s"] = (L.detach().numpy(), l_full.detach().numpy(),
                                                                     l_d.detach().numpy(), l_s.detach().numpy())
        ps, pd = list(st.parameters()), list(dy.parameters())
        grads = torch.autograd.grad(L, ps + pd + [rays], allow_unused=True)
        for (k, p), gv in zip(st.named_parameters(), grads[: len(ps)]):
            out["gs." + k] = (gv if gv is not None else torch.zeros_like(p)).numpy()
        for (k, p), gv in zip(dy.named_parameters(), grads[len(ps): len(ps) + len(pd)]):
            out["gd." + k] = (gv if gv is not None else torch.zeros_like(p)).numpy()
            out["gd_none." + k] = np.array(gv is None)      # which dynamic parameters the A + E graph never reaches
        out["g.rays"] = grads[-1].numpy()
        np.savez(os.path.join(HERE, f"pass_structure_{case_name}.npz"), **out)
        print(f"pass_structure_{case_name}: seed {seed} coins {cA, cE} loss {float(L):.6f} "
              f"|g.rays| {float(grads[-1].abs().max()):.3e}")


def gen_checkpoint(mods):
    """models/tensorBase.py:460-485: a `.th` checkpoint of each field WRITTEN BY THE REFERENCE's own save() (the file
    train.py:2413-2424 writes every progress_refresh_rate iterations: kwargs incl. se3_poses / focal_ratio_refine +
    state_dict), on the weights of the ndc_relu case, plus what the reference's own reload recipe (train.py:433-447:
    `Model(**kwargs)`, `.load(ckpt)`) computes from it at a few probe points -- so that loading it here can be checked
    for values, not only for keys."""
    TS, TD, _, _, _ = mods
    case = np.load(os.path.join(HERE, "ndc_relu.npz"), allow_pickle=True)
    grid = [int(v) for v in case["meta.grid"]]
    st, dy = build_fields(TS, TD, torch.from_numpy(case["aabb"]), grid, "relu", "MLP_Fea", -10.0, 1)
    st.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("s.")})
    dy.load_state_dict({k[2:]: torch.from_numpy(case[k]) for k in case.files if k.startswith("d.")})
    poses = torch.eye(3, 4)[None].repeat(12, 1, 1) + 0.01 * torch.randn(12, 3, 4, generator=torch.Generator().manual_seed(5))
    focal = torch.tensor(41.5)
    probe = {"xn": case["fn.xn"], "t": case["fn.t"]}
    for tag, m in (("static", st), ("dynamic", dy)):
        path = os.path.join(HERE, f"reference_ckpt_{tag}.th")
        m.save(poses, focal, path)
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        kwargs = dict(ckpt["kwargs"])
        kwargs.pop("se3_poses")
        kwargs.pop("focal_ratio_refine")
        kwargs.update({"device": "cpu"})
        with contextlib.redirect_stdout(io.StringIO()):
            m2 = type(m)(**kwargs)
        m2.load(ckpt)
        xn, tf = torch.from_numpy(probe["xn"]), torch.from_numpy(probe["t"])
        with torch.no_grad():
            probe[tag + ".density"] = m2.compute_densityfeature(xn, tf, None).numpy()
            probe[tag + ".app"] = m2.compute_appfeature(xn, tf, None).numpy()
        probe[tag + ".nSamples"] = np.array(m2.nSamples)
    np.savez(os.path.join(HERE, "reference_ckpt_probe.npz"), **probe)
    print("checkpoint: ok", os.path.getsize(os.path.join(HERE, "reference_ckpt_dynamic.th")), "bytes")


if __name__ == "__main__":
    mods = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "checkpoint":
        gen_checkpoint(mods)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fn_grads":
        gen_fn_grads(mods)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tv":
        gen_tv(mods)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pass_structure":
        gen_pass_structure(mods)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "induce_flow":
        gen_induce_flow(mods)
        sys.exit(0)
    gen_case("ndc_relu", mods, "ndc", "relu", "MLP_Fea", [18, 19, 11], 32, 13, 20211202, -10.0, False)
    gen_case("ndc_relu_long", mods, "ndc", "relu", "MLP_Fea", [10, 11, 7], 6, 130, 20211203, -10.0, True)
    gen_case("ndc_softplus", mods, "ndc", "softplus", "MLP_Fea", [10, 12, 7], 16, 13, 20211204, -1.0, True)
    gen_case("contract_relu_te", mods, "contract", "relu", "MLP_Fea_TimeEmbedding", [9, 9, 9], 16, 14,
             20211205, -10.0, True)
    gen_case("contract_softplus_te", mods, "contract", "softplus", "MLP_Fea_TimeEmbedding", [8, 8, 8],
             12, 13, 20211206, -1.0, False)
    gen_raygen(mods)
    gen_induce_flow(mods)
    gen_tv(mods)
    gen_fn_grads(mods)
    gen_pass_structure(mods)
    gen_checkpoint(mods)
