"""Edge cases of the call surface through the C ABI: empty and single-element batches, rays with no valid
sample, nothing / everything passing the appearance mask, the longest supported ray (S = 4096), S = 1.
The oracle (or the torch behaviour of the reference on the same input) is the checker."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu


def _fields(grid=(20, 22, 14), rt="ndc", thres=1e-4):
    import rodynrf
    from _gpu_util import COMMON
    torch.manual_seed(3)
    contract = rt == "contract"
    aabb = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]] if contract else [[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
    nf = [0.05, 256.0] if contract else [0.0, 1.0]
    kw = dict(COMMON, near_far=nf, density_shift=-10.0, fea2denseAct="relu")
    kw["alphaMask_thres"] = 1e-4
    st = rodynrf.TensorVMSplit(aabb, list(grid), 12, "cuda", shadingMode="MLP_Fea", fea_pe=2, **kw)
    dy = rodynrf.TensorVMSplit_TimeEmbedding(aabb, list(grid), 12, "cuda", shadingMode="MLP_Fea_late_view", fea_pe=0, **kw)
    st.rayMarch_weight_thres = thres
    dy.rayMarch_weight_thres = thres
    return st, dy, aabb, nf


def _oracle_pass(st, dy, aabb, nf, rays, ts, S, rt, valid_override=None):
    from _gpu_util import oracle_cfg, oracle_sd
    from oracle import rodynrf_oracle as O
    xyz, z, valid = O.sampleXYZ(rays, aabb, nf, S, rt, None, None)
    if valid_override is not None:
        valid = valid_override
    r_s = O.field_forward(oracle_sd(st), oracle_cfg(st), rays, ts, xyz, z, valid, rt, dynamic=False)
    r_d = O.field_forward(oracle_sd(dy), oracle_cfg(dy), rays, ts, xyz, z, valid, rt, dynamic=True)
    out = O.raw2outputs(r_s[6], r_s[7], r_d[6], r_d[7], r_d[9], r_d[2], r_d[8], rays, False, rt)
    return r_s, r_d, out


def _gpu_pass(st, dy, rays, ts, S, rt, valid_override=None):
    import rodynrf
    xyz, z, valid = rodynrf.sampleXYZ(dy, rays, S, ray_type=rt, is_train=False)
    if valid_override is not None:
        valid = valid_override.to(valid.device)
    o_s = st(rays, ts, None, xyz, z, valid, is_train=False, ray_type=rt)
    o_d = dy(rays, ts, None, xyz, z, valid, is_train=False, ray_type=rt)
    out = rodynrf.raw2outputs(o_s[6], o_s[7], o_d[6], o_d[7], o_d[9], o_d[2], o_d[8], rays, is_train=False, ray_type=rt)
    return o_s, o_d, out


def _compare(o_s, o_d, out, r_s, r_d, ref, rtol=1e-4):
    for i in (6, 7, 4):
        assert_close(o_s[i].cpu(), r_s[i], f"static[{i}]", rtol=rtol)
    for i in (2, 5, 6, 7, 4):
        assert_close(o_d[i].cpu(), r_d[i], f"dynamic[{i}]", rtol=rtol)
    for i, (a, b) in enumerate(zip(out, ref)):
        assert_close(a.cpu(), b, f"raw2outputs[{i}]", rtol=rtol)


@pytest.mark.parametrize("rt", ["ndc", "contract"])
def test_empty_batch_is_a_no_op(rt):
    """N = 0 rays: every entry point returns empty tensors of the reference's shapes (torch on empty tensors
    does the same), and a backward through them leaves zero gradients."""
    import rodynrf
    st, dy, aabb, nf = _fields(rt=rt)
    S = 12
    rays = torch.zeros(0, 6, device="cuda")
    ts = torch.zeros(0, device="cuda")
    o_s, o_d, out = _gpu_pass(st, dy, rays, ts, S, rt)
    assert o_s[6].shape == (0, S, 3) and o_s[7].shape == (0, S) and o_s[4].shape == (0, S)
    assert o_d[2].shape == (0, S) and o_d[5].shape == (0, S, 3) and o_d[3].shape == (0, S, 3)
    assert len(out) == 13 and out[0].shape == (0, 3) and out[5].shape == (0,) and out[11].shape == (0, S)
    sf_f, sf_b = dy.get_forward_backward_scene_flow(o_d[3], ts)
    assert sf_f.shape == (0, S, 3) and sf_b.shape == (0, S, 3)
    loss = out[0].sum() + out[8].sum() + sf_f.sum() + o_s[6].sum()
    loss.backward()
    for p in list(st.parameters()) + list(dy.parameters()):
        assert p.grad is None or float(p.grad.abs().max()) == 0.0
    poses = torch.zeros(3, 9, device="cuda"); poses[:, 0] = 1; poses[:, 4] = 1
    r = rodynrf.generate_rays(torch.zeros(0, dtype=torch.long, device="cuda"), poses, 100.0, 27, 48, ndc=rt == "ndc", near=1.0)
    assert r.shape == (0, 6)
    d = st.compute_densityfeature(torch.zeros(0, 3, device="cuda"), None, None)
    assert d.shape == (0,)
    rgb, depth = rodynrf.render_rays(st, dy, rays, ts, N_samples=S, ray_type=rt)
    assert rgb.shape == (0, 3) and depth.shape == (0,)


@pytest.mark.parametrize("rt,N,S", [("ndc", 1, 1), ("ndc", 1, 33), ("contract", 1, 2), ("ndc", 3, 1024), ("contract", 2, 1024), ("ndc", 2, 4096)])
def test_single_ray_and_extreme_sample_counts(rt, N, S):
    """one ray; one sample per ray; the longest ray the dynamic field supports (S = 1024: 32 tiles, the
    transmittance carries of all of them)"""
    from _gpu_util import make_rays
    st, dy, aabb, nf = _fields(rt=rt)
    rays, ts = make_rays(N, 5, rt)
    r_s, r_d, ref = _oracle_pass(st, dy, aabb, nf, rays, ts, S, rt)
    o_s, o_d, out = _gpu_pass(st, dy, rays.cuda(), ts.cuda(), S, rt)
    _compare(o_s, o_d, out, r_s, r_d, ref)


def test_rays_without_a_valid_sample_render_the_background():
    """`ray_valid` all False on some rays (every sample outside the box): sigma, rgb and weights are exactly 0
    there, the maps show the background, and the gradients of those rays' samples vanish."""
    import rodynrf
    from _gpu_util import make_rays
    st, dy, aabb, nf = _fields()
    N, S = 40, 24
    rays, ts = make_rays(N, 9)
    rays[::3, 0] = 5.0   # origins far outside the box in x: no valid sample on every third ray
    r_s, r_d, ref = _oracle_pass(st, dy, aabb, nf, rays, ts, S, "ndc")
    rays_g = rays.cuda()
    o_s, o_d, out = _gpu_pass(st, dy, rays_g, ts.cuda(), S, "ndc")
    xyz, z, valid = rodynrf.sampleXYZ(dy, rays_g, S, ray_type="ndc", is_train=False)
    assert not bool(valid[::3].any()) and bool(valid[1::3].any())
    _compare(o_s, o_d, out, r_s, r_d, ref)
    assert float(o_s[7][::3].abs().max()) == 0.0 and float(o_d[7][::3].abs().max()) == 0.0
    assert float(o_s[6][::3].abs().max()) == 0.0 and float(o_d[4][::3].abs().max()) == 0.0
    # all-invalid batch
    none = torch.zeros(N, S, dtype=torch.bool)
    r_s, r_d, ref = _oracle_pass(st, dy, aabb, nf, rays, ts, S, "ndc", valid_override=none)
    o_s, o_d, out = _gpu_pass(st, dy, rays_g, ts.cuda(), S, "ndc", valid_override=none)
    _compare(o_s, o_d, out, r_s, r_d, ref)


@pytest.mark.parametrize("thres", [1e9, -1.0])
def test_appearance_mask_nothing_and_everything(thres):
    """rayMarch_weight_thres so high that NO sample reaches the appearance networks (compacted list empty:
    rgb stays 0), and below zero so that EVERY sample does; forward and parameter gradients vs the oracle."""
    from _gpu_util import make_rays, oracle_cfg, oracle_sd
    from oracle import rodynrf_oracle as O
    st, dy, aabb, nf = _fields(thres=thres)
    N, S = 48, 40
    rays, ts = make_rays(N, 13)
    sd_s, sd_d = oracle_sd(st), oracle_sd(dy)
    for sd in (sd_s, sd_d):
        for v in sd.values():
            v.requires_grad_(True)
    xyz, z, valid = O.sampleXYZ(rays, aabb, nf, S, "ndc", None, None)
    r_s = O.field_forward(sd_s, oracle_cfg(st), rays, ts, xyz, z, valid, "ndc", dynamic=False)
    r_d = O.field_forward(sd_d, oracle_cfg(dy), rays, ts, xyz, z, valid, "ndc", dynamic=True)
    ref = O.raw2outputs(r_s[6], r_s[7], r_d[6], r_d[7], r_d[9], r_d[2], r_d[8], rays, False, "ndc")
    tgt = torch.rand(N, 3, generator=torch.Generator().manual_seed(1))
    lref = ((ref[0] - tgt) ** 2).mean() + ((ref[8] - tgt) ** 2).mean() + ((ref[4] - tgt) ** 2).mean() + ref[12].mean()
    lref.backward()
    o_s, o_d, out = _gpu_pass(st, dy, rays.cuda(), ts.cuda(), S, "ndc")
    _compare(o_s, o_d, out, [t.detach() if torch.is_tensor(t) else t for t in r_s],
             [t.detach() if torch.is_tensor(t) else t for t in r_d], [t.detach() for t in ref])
    if thres > 1:
        assert float(o_s[6].abs().max()) == 0.0 and float(o_d[6].abs().max()) == 0.0
    tg = tgt.cuda()
    l = ((out[0] - tg) ** 2).mean() + ((out[8] - tg) ** 2).mean() + ((out[4] - tg) ** 2).mean() + out[12].mean()
    l.backward()
    for mod, sd in ((st, sd_s), (dy, sd_d)):
        got = {k: v for k, v in mod.named_parameters()}
        for k, v in sd.items():
            g_ref = torch.zeros_like(v) if v.grad is None else v.grad
            g_got = got[k].grad
            g_got = torch.zeros_like(g_ref) if g_got is None else g_got.cpu()
            if float(g_ref.abs().max()) == 0.0:
                assert float(g_got.abs().max()) == 0.0, k
            else:
                assert_close(g_got, g_ref, "grad " + k, rtol=2e-4)
