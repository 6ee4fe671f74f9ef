"""GPU parity (forward): HIP kernels through the C ABI vs (a) the golden vectors generated from the
reference and (b) the CPU oracle on a seeded Balloon1-stage-0-shaped batch.  Tolerance: 1e-4
relative (north_star), relative to each tensor's max magnitude; samples whose weight sits within
1e-6 of the app-mask threshold are excluded from the per-sample rgb check (a 1-ulp difference
legitimately flips `weight > 1e-4`)."""
import numpy as np
import pytest
import torch

from _util import CASES, FWD_ELEM, assert_close

pytestmark = pytest.mark.gpu

FNAMES = ["_0", "_1", "blending", "pts_ref", "weight", "xyz_prime", "rgb", "sigma", "z", "dists"]
ONAMES = ["rgb_map_full", "depth_map_full", "acc_map_full", "weights_full", "rgb_map_s",
          "depth_map_s", "acc_map_s", "weights_s", "rgb_map_d", "depth_map_d", "acc_map_d",
          "weights_d", "dynamicness_map"]


def _safe_mask(w_ref, thres=1e-4):
    w = torch.as_tensor(w_ref)
    return (w - thres).abs() > 1e-6


def _check_field(out, g, prefix, rtol=1e-4):
    safe = _safe_mask(g[prefix + "weight"])
    for k, v in zip(FNAMES, out):
        if v is None or (prefix + k) not in g:
            continue
        m = safe[..., None].expand(*safe.shape, 3) if k == "rgb" else None
        assert_close(v, g[prefix + k], prefix + k, rtol=rtol, mask=m, elem=FWD_ELEM)


@pytest.mark.parametrize("case", CASES)
def test_golden_forward(case):
    import rodynrf
    from _gpu_util import fields_from_case
    g, st, dy, _ = fields_from_case(case)
    rt = str(g["meta.ray_type"])
    dev = "cuda"
    rays = torch.from_numpy(g["rays"]).to(dev)
    ts = torch.from_numpy(g["ts"]).to(dev)
    S = g["z"].shape[1]
    jit = torch.from_numpy(g["jitter"]).to(dev) if "jitter" in g else None
    jo = torch.from_numpy(g["jitter_outer"]).to(dev) if "jitter_outer" in g else None
    with torch.no_grad():
        xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type=rt, is_train=jit is not None,
                                          jitter=jit, jitter_outer=jo)
        # byte / index work is bit-exact: the sampler reproduces ATen's linspace (fused multiply-add per
        # element, tests/test_oracle_golden.py::test_linspace_formula) and the un-fused o + d * z
        assert torch.equal(z.cpu(), torch.from_numpy(g["z"])), "z_vals differ from the reference bits"
        assert torch.equal(valid.cpu(), torch.from_numpy(g["valid"])), "valid mask differs from the reference"
        assert torch.equal(xyz.cpu(), torch.from_numpy(g["xyz"])), "sample points differ from the reference bits"
        o_s = st(rays, ts, None, xyz, z, valid, is_train=True, ray_type=rt, N_samples=S)
        o_d = dy(rays, ts, None, xyz, z, valid, is_train=True, ray_type=rt, N_samples=S)
        _check_field(o_s, g, "fs.")
        _check_field(o_d, g, "fd.")
        gt = lambda k: torch.from_numpy(g[k]).to(dev)
        for white, pre in ((False, "c0."), (True, "c1.")):
            outs = rodynrf.raw2outputs(gt("fs.rgb"), gt("fs.sigma"), gt("fd.rgb"), gt("fd.sigma"),
                                       gt("fd.dists"), gt("fd.blending"), gt("fd.z"), rays,
                                       is_train=True, ray_type=rt, add_white_bg=white)
            for k, v in zip(ONAMES, outs):
                # contract depth adds (1-acc)*256: fp32 rounding of acc (2^-23 relative to 1)
                # is amplified by 256, so that output carries an absolute term 256*2^-22
                at = 256.0 * 2.0 ** -22 if (rt == "contract" and "depth" in k) else 0.0
                assert_close(v, g[pre + k], pre + k, atol=at, elem=FWD_ELEM)
        sf_f, sf_b = dy.get_forward_backward_scene_flow(xyz, ts)
        assert_close(sf_f, g["sf.f"], "sf.f", elem=FWD_ELEM)
        assert_close(sf_b, g["sf.b"], "sf.b", elem=FWD_ELEM)


def test_raygen_golden():
    import os
    import rodynrf
    from _util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "raygen.npz"))
    dev = "cuda"
    rays = rodynrf.generate_rays(torch.from_numpy(z["ids"]).to(dev), torch.from_numpy(z["poses"]).to(dev),
                                 float(z["focal"]), int(z["H"]), int(z["W"]), ndc=True, near=1.0)
    assert_close(rays, z["rays"], "rays", rtol=1e-5)
    rw = rodynrf.generate_rays(torch.from_numpy(z["ids"]).to(dev), torch.from_numpy(z["poses"]).to(dev),
                               float(z["focal"]), int(z["H"]), int(z["W"]), ndc=False)
    assert_close(rw, z["rays_world"], "rays_world", rtol=1e-5)


@pytest.mark.parametrize("N,S,grid", [(256, 115, [141, 157, 94]), (96, 270, [331, 368, 220])])
def test_oracle_forward_balloon_shapes(N, S, grid):
    """Balloon1 stage-0 / final grids against the CPU oracle on seeded weights."""
    import rodynrf
    from _gpu_util import COMMON, make_rays, oracle_cfg, oracle_sd
    from oracle import rodynrf_oracle as O
    torch.manual_seed(20211202)
    aabb = torch.tensor([[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
    kw = dict(COMMON, near_far=[0.0, 1.0], density_shift=-10.0, fea2denseAct="relu")
    st = rodynrf.TensorVMSplit(aabb, grid, 12, "cuda", shadingMode="MLP_Fea", fea_pe=2, **kw)
    dy = rodynrf.TensorVMSplit_TimeEmbedding(aabb, grid, 12, "cuda", shadingMode="MLP_Fea_late_view",
                                             fea_pe=0, **kw)
    rays, ts = make_rays(N, 7)
    jit = torch.rand(S, generator=torch.Generator().manual_seed(3))
    sd_s, sd_d = oracle_sd(st), oracle_sd(dy)
    cfg_s, cfg_d = oracle_cfg(st), oracle_cfg(dy)
    with torch.no_grad():
        xyz, z, valid = O.sampleXYZ(rays, aabb, [0.0, 1.0], S, "ndc", jit)
        r_s = O.field_forward(sd_s, cfg_s, rays, ts, xyz, z, valid, "ndc", dynamic=False)
        r_d = O.field_forward(sd_d, cfg_d, rays, ts, xyz, z, valid, "ndc", dynamic=True)
        r_o = O.raw2outputs(r_s[6], r_s[7], r_d[6], r_d[7], r_d[9], r_d[2], r_d[8], rays, True, "ndc")
        dev = "cuda"
        cr, ct = rays.to(dev), ts.to(dev)
        gx, gz, gv = rodynrf.sampleXYZ(dy, cr, S, ray_type="ndc", is_train=True, jitter=jit.to(dev))
        assert_close(gx, xyz, "xyz", rtol=2e-6)
        o_s = st(cr, ct, None, xyz.to(dev), z.to(dev), valid.to(dev), ray_type="ndc")
        o_d = dy(cr, ct, None, xyz.to(dev), z.to(dev), valid.to(dev), ray_type="ndc")
        for name, o, r in (("s", o_s, r_s), ("d", o_d, r_d)):
            safe = _safe_mask(r[4])
            for k, a, b in zip(FNAMES, o, r):
                if a is None:
                    continue
                m = safe[..., None].expand(*safe.shape, 3) if k == "rgb" else None
                # weight = (1 - exp(-sigma dist)) T: the reference's own formula leaves alpha with the ABSOLUTE rounding
                # of exp() near 1 (1 ulp = 6e-8 on either side, whatever the exp implementation), which at these ray
                # lengths (max weight ~0.05, typical alpha ~1e-4) is above 1e-6 max|ref|: allow 2 ulp(1) absolute
                at = 2.0 * 2.0 ** -23 if k == "weight" else 0.0
                assert_close(a, b, f"{name}.{k}", mask=m, elem=FWD_ELEM, atol=at)
        outs = rodynrf.raw2outputs(o_s[6], o_s[7], o_d[6], o_d[7], o_d[9], o_d[2], o_d[8], cr,
                                   is_train=True, ray_type="ndc", add_white_bg=True)
        for k, a, b in zip(ONAMES, outs, r_o):
            assert_close(a, b, "c." + k, rtol=1e-4, elem=FWD_ELEM, atol=2.0 * 2.0 ** -23 if "weights" in k else 0.0)
        frac = float((r_d[4] > 1e-4).float().mean())
        print(f"app_mask fraction dynamic {frac:.3f} static {float((r_s[4] > 1e-4).float().mean()):.3f}")


def test_selftest_mfma_layer():
    """The MFMA layer primitive (pack -> LDS -> v_mfma_f32_32x32x2_f32 chain) against a matmul with
    an ASYMMETRIC weight (catches operand / output transposes)."""
    import ctypes as C
    import importlib
    L = importlib.import_module("robust-dynrf_amd._lib")
    g = torch.Generator().manual_seed(0)
    M = 77
    x = torch.randn(M, 64, generator=g).cuda()
    w = (torch.randn(64, 64, generator=g) + torch.arange(64)[:, None] * 0.01).cuda()
    b = torch.randn(64, generator=g).cuda()
    y = torch.empty(M, 64, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    L.check(L.lib.rdrf_selftest_mlp(L.ptr(x), L.ptr(w), L.ptr(b), M, 64, 64, L.ptr(y), L.ptr(ws),
                                    C.c_size_t(ws.numel()), L.stream_of(x)), "selftest")
    ref = torch.relu(x.double() @ w.double().T + b.double())
    assert_close(y, ref, "mfma layer", rtol=1e-5)


def test_fused_render_matches_unfused():
    import rodynrf
    from _gpu_util import fields_from_case
    g, st, dy, _ = fields_from_case("ndc_relu")
    dev = "cuda"
    rays = torch.from_numpy(g["rays"]).to(dev)
    ts = torch.from_numpy(g["ts"]).to(dev)
    S = g["z"].shape[1]
    rgb, depth = rodynrf.render_rays(st, dy, rays, ts, N_samples=S, ray_type="ndc")
    assert_close(rgb, g["ce.rgb_map_full"], "render rgb")
    assert_close(depth, g["ce.depth_map_full"], "render depth")


@pytest.mark.parametrize("case,N,S", [("ndc_relu", 32, 13), ("contract_relu_te", 16, 14), ("ndc_relu", 777, 115), ("contract_relu_te", 2100, 37)])
def test_fused_render_is_bit_identical_to_the_launch_sequence(case, N, S):
    """rdrf_render_fused_fwd (one cooperative launch: sampler, both density phases, both appearance phases and the
    compositor behind grid-wide barriers) runs the device bodies of the per-phase kernels: same bits as
    rdrf_render_sequence_fwd, for both ray types / static heads, a batch smaller than the grid and one that wraps it,
    ragged last tiles; and the same reference values as the unfused path on the golden rays."""
    import rodynrf
    from _gpu_util import fields_from_case, make_rays
    g, st, dy, _ = fields_from_case(case)
    rt = str(g["meta.ray_type"])
    if N == g["rays"].shape[0]:
        rays, ts = torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["ts"]).cuda()
    else:
        rays, ts = (t.cuda() for t in make_rays(N, 3, rt))
    a = rodynrf.render_rays(st, dy, rays, ts, N_samples=S, ray_type=rt, mode="sequence")
    for _ in range(3):   # repeated: the barrier word, counters and LDS images are re-initialised by every launch
        b = rodynrf.render_rays(st, dy, rays, ts, N_samples=S, ray_type=rt, mode="fused")
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    c = rodynrf.render_rays(st, dy, rays, ts, N_samples=S, ray_type=rt, mode="auto")
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    if N == g["rays"].shape[0] and S == g["z"].shape[1] and "jitter" not in g:
        assert_close(b[0], g["ce.rgb_map_full"], "fused rgb")
        assert_close(b[1], g["ce.depth_map_full"], "fused depth", atol=256.0 * 2.0 ** -22 if rt == "contract" else 0.0)


@pytest.mark.parametrize("case,N,S", [("ndc_relu", 7, 13), ("ndc_relu", 513, 115), ("contract_relu_te", 300, 37), ("ndc_relu", 2100, 33), ("ndc_relu", 1, 1), ("contract_relu_te", 3, 1000), ("ndc_relu", 17, 32)])
def test_inference_forward_is_bit_identical_to_the_training_forward(case, N, S):
    """Inference calls (no saved activations) run the dynamic field's density phase at tile granularity --
    k_dyn_density_flat over 32-sample tiles of the flattened [N * S] array + k_ray_scan -- and let the density kernels
    zero-fill the colours; training calls run the wave-per-ray kernel that also saves its activations.  Same arithmetic
    in the same order: every output of both fields (TensorBase.forward, models/tensorBase.py:704-850) has the same bits,
    for tiles that straddle rays (S not a multiple of 32), a last partial tile, and a batch smaller than one wave's share."""
    import rodynrf
    from _gpu_util import fields_from_case, make_rays
    g, st, dy, _ = fields_from_case(case)
    rt = str(g["meta.ray_type"])
    rays, ts = (t.cuda() for t in make_rays(N, 11, rt))
    with torch.no_grad():
        xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type=rt, is_train=False)
    for f in (st, dy):
        for _ in range(2):   # twice: the counters and the colour fill are re-initialised by every call
            with torch.no_grad():
                a = f(rays, ts, None, xyz, z, valid, is_train=False, ray_type=rt, N_samples=S)
            b = f(rays, ts, None, xyz, z, valid, is_train=False, ray_type=rt, N_samples=S)
            assert any(v is not None and v.requires_grad for v in b), "the second call must be the training path"
            for k, u, v in zip(FNAMES, a, b):
                if u is None or k.startswith("_"):
                    continue
                assert torch.equal(u, v.detach()), f"{type(f).__name__}.{k}: inference and training forward differ"


def test_render_frame_matches_oracle_pipeline():
    """whole-frame driver: device ray generation -> fused render, one launch sequence vs chunks of
    100 rays vs the oracle's generate_rays -> sampleXYZ -> fields -> raw2outputs on the CPU."""
    import rodynrf
    from _gpu_util import fields_from_case, oracle_cfg, oracle_sd
    from oracle import rodynrf_oracle as O
    g, st, dy, _ = fields_from_case("ndc_relu")
    T, H, W, frame = 5, 9, 16, 3
    gen = torch.Generator().manual_seed(2)
    poses = torch.zeros(T, 9)
    poses[:, 0] = 1
    poses[:, 4] = 1
    poses = poses + 0.02 * torch.randn(T, 9, generator=gen)
    focal = max(H, W) / 2.0 * 1.7320508
    S = 21
    a, da = rodynrf.render_frame(st, dy, poses.cuda(), focal, frame, H, W, N_samples=S)
    b, db = rodynrf.render_frame(st, dy, poses.cuda(), focal, frame, H, W, N_samples=S, chunk=100)
    assert a.shape == (H, W, 3) and da.shape == (H, W)
    assert torch.equal(a, b) and torch.equal(da, db)
    ids = torch.arange(H * W) + frame * H * W
    rays = O.generate_rays(ids, poses, focal, H, W, ndc=True, near=1.0)
    ts = torch.full((H * W,), 2.0 * frame / (T - 1) - 1.0)
    aabb = st.aabb.cpu()
    xyz, z, valid = O.sampleXYZ(rays, aabb, [float(v) for v in st.near_far], S, "ndc", None)
    r_s = O.field_forward(oracle_sd(st), oracle_cfg(st), rays, ts, xyz, z, valid, "ndc", dynamic=False)
    r_d = O.field_forward(oracle_sd(dy), oracle_cfg(dy), rays, ts, xyz, z, valid, "ndc", dynamic=True)
    outs = O.raw2outputs(r_s[6], r_s[7], r_d[6], r_d[7], r_d[9], r_d[2], r_d[8], rays, False, "ndc")
    assert_close(a.view(-1, 3), outs[0].clamp(0, 1), "frame rgb", rtol=2e-4)
    assert_close(da.view(-1), outs[1], "frame depth", rtol=2e-4)
    assert float(rodynrf.psnr(a, a + 0.1)) == pytest.approx(20.0, abs=1e-3)


@pytest.mark.parametrize("rt", ["ndc", "contract"])
def test_render_chunks_on_hip_streams_is_bit_identical(rt):
    """render_chunks: the 512-ray eval chunks of renderer.py:732-812 issued round-robin on 4 HIP streams (own scratch per
    stream, ONE packed weight image per field shared read-only by every stream) give the same bits as the sequential loop and as one call over all rays; repeated,
    with a weight update in between (the per-stream packed images must follow it)."""
    import rodynrf
    from _gpu_util import fields_from_case, make_rays
    g, st, dy, _ = fields_from_case("ndc_relu" if rt == "ndc" else "contract_relu_te")
    rays, ts = (t.cuda() for t in make_rays(3000, 5, rt))
    S = 40
    for rep in range(2):
        whole = rodynrf.render_rays(st, dy, rays, ts, N_samples=S, ray_type=rt)
        seq = rodynrf.render_chunks(st, dy, rays, ts, 512, N_samples=S, ray_type=rt, streams=1)
        par = rodynrf.render_chunks(st, dy, rays, ts, 512, N_samples=S, ray_type=rt, streams=4)
        torch.cuda.synchronize()
        for a, b in ((whole, seq), (whole, par)):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        with torch.no_grad():   # change the weights: the shared images are re-packed (on the main stream) before the next call's chunks
            dy.renderModule.mlp[0].weight.mul_(1.01)
            st.basis_mat.weight.mul_(0.99)


def test_upsample_volume_grid_then_forward():
    """upsample_volume_grid (models/tensoRF.py:223-232, 838-850; bilinear, align_corners): the new
    factors keep the channel-last storage, stepSize / nSamples follow, and both fields still match
    the oracle on the upsampled weights; the fused flat gradient buffer is rebuilt for the new shapes."""
    import rodynrf
    from _gpu_util import fields_from_case, make_rays, oracle_cfg, oracle_sd
    from oracle import rodynrf_oracle as O
    g, st, dy, _ = fields_from_case("ndc_relu")
    ref_planes = [p.detach().cpu().contiguous() for p in dy.density_plane]
    new = [25, 27, 16]
    n_before = dy.nSamples
    st.upsample_volume_grid(new)
    dy.upsample_volume_grid(new)
    assert list(dy.gridSize.tolist()) == new and dy.nSamples != n_before
    for i, p in enumerate(dy.density_plane):
        assert p.stride(1) == 1 and p.is_cuda
        want = torch.nn.functional.interpolate(ref_planes[i], size=p.shape[2:], mode="bilinear", align_corners=True)
        assert_close(p, want, f"upsampled density_plane.{i}", rtol=1e-6)
    N, S = 40, 33
    rays, ts = make_rays(N, 3)
    aabb = st.aabb.cpu()
    xyz, z, valid = O.sampleXYZ(rays, aabb, [float(v) for v in st.near_far], S, "ndc", None)
    r_s = O.field_forward(oracle_sd(st), oracle_cfg(st), rays, ts, xyz, z, valid, "ndc", dynamic=False)
    r_d = O.field_forward(oracle_sd(dy), oracle_cfg(dy), rays, ts, xyz, z, valid, "ndc", dynamic=True)
    dev = "cuda"
    args = (rays.to(dev), ts.to(dev), None, xyz.to(dev), z.to(dev), valid.to(dev))
    o_s = st(*args, ray_type="ndc")
    o_d = dy(*args, ray_type="ndc")
    for k in (4, 6, 7):
        assert_close(o_s[k], r_s[k], f"static out {k}")
        assert_close(o_d[k], r_d[k], f"dynamic out {k}")
    assert_close(o_d[2], r_d[2], "blending")
    dy.fused_grad = True
    dy.zero_grad_fused()
    (o_d[6].sum() + o_d[7].sum()).backward()
    assert all(p.grad is not None and p.grad.shape == p.shape for p in dy._param_list())


def test_c_abi_error_paths():
    """The C entry points never crash on bad arguments: they return a negative code and leave a
    message for rdrf_last_error() (which the ctypes binding turns into RdrfError)."""
    import ctypes as C
    import importlib
    import rodynrf
    from _gpu_util import fields_from_case
    L = importlib.import_module("robust-dynrf_amd._lib")
    F = importlib.import_module("robust-dynrf_amd.fields")
    g, st, dy, _ = fields_from_case("ndc_relu")
    dev = "cuda"
    rays = torch.from_numpy(g["rays"]).to(dev)
    ts = torch.from_numpy(g["ts"]).to(dev)
    xyz = torch.from_numpy(g["xyz"]).to(dev)
    z = torch.from_numpy(g["z"]).to(dev)
    valid = torch.from_numpy(g["valid"]).to(dev).view(torch.uint8)
    N, S = z.shape
    out = [torch.empty(N, S, 3, device=dev)] + [torch.empty(N, S, device=dev) for _ in range(3)]
    P = F._static_struct(st._param_list())
    cfg = F._cfg_struct(st, "ndc")
    tiny = torch.empty(64, dtype=torch.uint8, device=dev)           # workspace far too small
    rc = L.lib.rdrf_static_fwd(C.byref(P), C.byref(cfg), L.ptr(rays), L.ptr(ts), L.ptr(xyz), L.ptr(z),
                               L.ptr(valid), N, S, *[L.ptr(o) for o in out], None, C.c_size_t(0),
                               L.ptr(tiny), C.c_size_t(tiny.numel()), L.stream_of(z))
    assert rc < 0 and b"workspace" in L.lib.rdrf_last_error()
    rc = L.lib.rdrf_static_fwd(C.byref(P), C.byref(cfg), None, L.ptr(ts), L.ptr(xyz), L.ptr(z),
                               L.ptr(valid), N, S, *[L.ptr(o) for o in out], None, C.c_size_t(0),
                               L.ptr(tiny), C.c_size_t(tiny.numel()), L.stream_of(z))
    assert rc < 0 and len(L.lib.rdrf_last_error()) > 0                # null rays
    with pytest.raises(L.RdrfError):
        rodynrf.induce_flow(9, 16, 10.0, torch.zeros(N, 3, 4, device=dev), torch.zeros(N, S, device=dev),
                            torch.zeros(N, S, 3, device=dev), torch.zeros(N, 2, device=dev), rays, ray_type="other")
    with pytest.raises(L.RdrfError):                                  # CPU tensors: no fallback
        st(rays.cpu(), ts.cpu(), None, xyz.cpu(), z.cpu(), valid.cpu(), ray_type="ndc")
    with pytest.raises(NotImplementedError):                          # unsupported component counts
        rodynrf.TensorVMSplit(st.aabb, [8, 8, 8], 12, dev, density_n_comp=[8, 8, 8], appearance_n_comp=[48, 12, 12],
                              app_dim=27, featureC=128, view_pe=0, shadingMode="MLP_Fea", fea_pe=2)
    # the good path still works afterwards (no sticky error state)
    o = st(rays, ts, None, xyz, z, valid, ray_type="ndc")
    assert torch.isfinite(o[6]).all()


@pytest.mark.parametrize("case,N,S", [("contract_relu_te", 2100, 37), ("ndc_relu", 777, 115)])
def test_static_forward_is_bit_reproducible(case, N, S):
    """As below for the static field, whose hidden layers run on the bf16 matrix pipe with split storage (mfma_seg_b3s: hi + mid
    pieces in LDS, lo pieces streamed from the pack buffer): no atomics, so repeated calls must return the same bits."""
    import rodynrf
    from _gpu_util import fields_from_case, make_rays
    g, st, dy, _ = fields_from_case(case)
    rt = str(g["meta.ray_type"])
    rays, ts = (t.cuda() for t in make_rays(N, 3, rt))
    xyz, z, valid = rodynrf.sampleXYZ(st, rays, S, ray_type=rt, is_train=False)
    for grad in (False, True):
        ref = None
        for rep in range(25):
            with torch.set_grad_enabled(grad):
                o = st(rays, ts, None, xyz, z, valid, is_train=True, ray_type=rt)
            got = [t.detach().clone() for t in o if isinstance(t, torch.Tensor) and t.is_floating_point()]
            if ref is None:
                ref = got
            else:
                for a, b in zip(ref, got):
                    assert torch.equal(a, b), (grad, rep)


@pytest.mark.parametrize("case,N,S", [("contract_relu_te", 2100, 37), ("ndc_relu", 777, 115)])
def test_dynamic_forward_is_bit_reproducible(case, N, S):
    """The forward kernels do no atomics: repeated calls must return the same bits, in the inference (flat-tile) and the
    training (wave-per-ray, saved rows) instantiation.  Pins the operand hazards found when the heads' first layers moved to
    the bf16 matrix pipe (rdrf_common.hpp mfma_seg_b3: an `asm` statement inside the split, and lo pieces consumed by the
    third MFMA after the v_perm_b32 that packed them, each returned a stale piece for one sample in ~15 000, run to run)."""
    import rodynrf
    from _gpu_util import fields_from_case, make_rays
    g, st, dy, _ = fields_from_case(case)
    rt = str(g["meta.ray_type"])
    rays, ts = (t.cuda() for t in make_rays(N, 3, rt))
    xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type=rt, is_train=False)
    for grad in (False, True):
        ref = None
        for rep in range(25):
            with torch.set_grad_enabled(grad):
                o = dy(rays, ts, None, xyz, z, valid, is_train=True, ray_type=rt)
            got = [o[k].detach().clone() for k in (2, 4, 6, 7)]   # blending, weight, rgb, sigma
            if ref is None:
                ref = got
            else:
                for a, b in zip(ref, got):
                    assert torch.equal(a, b), (grad, rep)
