"""Induced flow / disparity (renderer.py:1266-1392) through the C ABI against the
vectors generated from the reference and against the oracle at the benchmark shape."""
import importlib
import os

import numpy as np
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "induce_flow.npz"))


def _t(k, rg=False):
    return torch.from_numpy(np.asarray(G[k])).clone().cuda().requires_grad_(rg)


@pytest.mark.parametrize("rt", ["ndc", "contract"])
def test_induce_flow_golden(rt):
    import rodynrf
    f, c2w, w, pts, rays = (_t(f"{rt}.{k}", True) for k in ("focal", "c2w", "weights", "pts", "rays"))
    H, W = int(G[f"{rt}.H"]), int(G[f"{rt}.W"])
    flow, disp = rodynrf.induce_flow(H, W, f, c2w, w, pts, _t(f"{rt}.pts_2d"), rays, ray_type=rt)
    assert flow.shape == (w.shape[0], 2) and disp.shape == (w.shape[0], 1)
    assert_close(flow, G[f"{rt}.flow"], "flow", rtol=1e-4)
    assert_close(disp, G[f"{rt}.disp"], "disp", rtol=1e-4)
    grads = torch.autograd.grad((flow * _t(f"{rt}.lw_flow")).sum() + (disp * _t(f"{rt}.lw_disp")).sum(),
                                [w, pts, rays, c2w, f])
    for name, g in zip(("g_weights", "g_pts", "g_rays", "g_c2w", "g_focal"), grads):
        assert_close(g, G[f"{rt}.{name}"], name, rtol=2e-4)


def test_render_single_3d_point_golden():
    import rodynrf
    H, W = int(G["ndc.H"]), int(G["ndc.W"])
    pl, d = rodynrf.render_single_3d_point(H, W, _t("ndc.focal"), _t("ndc.c2w"), _t("single.pt"))
    assert_close(pl, G["single.plane"], "plane", rtol=1e-4)
    assert_close(d, G["single.disp"], "disp", rtol=1e-4)
    fl = rodynrf.induce_flow_single(H, W, _t("ndc.focal"), _t("ndc.c2w"), _t("single.pt"), _t("ndc.pts_2d"))
    assert_close(fl, G["single.flow"], "flow", rtol=1e-4)


@pytest.mark.parametrize("rt", ["ndc", "contract"])
def test_induce_flow_oracle_bench_shape(rt):
    """4096 rays x 115 samples (BASELINE configs[1] shape) against the oracle, forward + gradients."""
    import rodynrf
    from oracle import rodynrf_oracle as O
    g = torch.Generator().manual_seed(3)
    N, S, H, W = 4096, 115, 135, 240
    f = torch.tensor(max(H, W) / 2.0 * 1.7320508)
    c2w = torch.eye(3, 4).repeat(N, 1, 1) + 0.03 * torch.randn(N, 3, 4, generator=g)
    w = torch.rand(N, S, generator=g)
    w = w / w.sum(-1, keepdim=True) * torch.rand(N, 1, generator=g)
    lim = 0.9 if rt == "ndc" else 1.9
    pts = torch.empty(N, S, 3).uniform_(-lim, lim, generator=g)
    if rt == "ndc":
        rays = torch.cat([torch.empty(N, 2).uniform_(-0.8, 0.8, generator=g), -torch.ones(N, 1),
                          torch.empty(N, 2).uniform_(-0.1, 0.1, generator=g), 2 * torch.ones(N, 1)], -1)
    else:
        rays = torch.cat([torch.empty(N, 3).uniform_(-0.2, 0.2, generator=g),
                          torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)], -1)
    p2d = torch.rand(N, 2, generator=g) * 100
    lf, ld = torch.randn(N, 2, generator=g), torch.randn(N, 1, generator=g)

    def run(mod, dev):
        ins = [t.clone().to(dev).requires_grad_(True) for t in (f, c2w, w, pts, rays)]
        flow, disp = mod.induce_flow(H, W, ins[0], ins[1], ins[2], ins[3], p2d.to(dev), ins[4], rt)
        gr = torch.autograd.grad((flow * lf.to(dev)).sum() + (disp * ld.to(dev)).sum(), ins)
        return [flow, disp, *gr]

    got, ref = run(rodynrf, "cuda"), run(O, "cpu")
    # points that project close to the camera plane (|z_cam| tiny) amplify rounding: compare rays whose
    # reference outputs are moderate, the others only for finiteness
    ok = (ref[0].abs().max(-1)[0] < 1e4) & (ref[1][:, 0].abs() < 1e3)
    assert float(ok.float().mean()) > 0.9
    for name, a, b in zip(("flow", "disp", "g_focal", "g_c2w", "g_weights", "g_pts", "g_rays"), got, ref):
        a = a.cpu()
        assert torch.isfinite(a).all() or not torch.isfinite(b).all(), name
        if name == "g_focal":
            continue   # a sum over all rays, dominated by the ill-conditioned ones
        m = ok.reshape(-1, *([1] * (a.dim() - 1))).expand_as(a) if a.shape[0] == N else None
        assert_close(a, b.detach(), name, rtol=5e-4, mask=m)


@pytest.mark.parametrize("N,S", [(33, 13), (16, 64), (7, 270), (4096, 115)])
def test_distortion_loss_vs_bruteforce(N, S):
    """flatten_eff_distloss (prefix-sum kernel) against the O(S^2) double sum of the oracle, value and
    gradient, scalar and per-point interval, ragged tile counts (S not a multiple of 64)."""
    import rodynrf
    from oracle import rodynrf_oracle as O
    g = torch.Generator().manual_seed(N * 1000 + S)
    w = torch.rand(N, S, generator=g) / S
    w[0] = 0.0
    m = torch.sort(torch.rand(N, S, generator=g), dim=-1)[0]
    for interval in (1.0 / S, torch.rand(N, S, generator=g) / S):
        wr = w.clone().double().requires_grad_(True)
        iv_ref = interval.double() if torch.is_tensor(interval) else interval
        ref = O.eff_distloss(wr, m.double(), iv_ref)
        gref, = torch.autograd.grad(ref, wr)
        wg = w.clone().cuda().requires_grad_(True)
        iv = interval.cuda() if torch.is_tensor(interval) else interval
        ray_id = torch.arange(N, device="cuda")[:, None].expand(N, S).reshape(-1)
        got = rodynrf.flatten_eff_distloss(wg.reshape(-1), m.cuda().reshape(-1),
                                           iv.reshape(-1) if torch.is_tensor(iv) else iv, ray_id)
        ggot, = torch.autograd.grad(got, wg)
        assert_close(got, ref.float(), "distloss", rtol=2e-5)
        assert_close(ggot, gref.float(), "d distloss / dw", rtol=1e-4)
        # ... and against the library's PUBLISHED prefix-sum form and analytic gradient (second restatement)
        assert_close(got, O.eff_distloss_published(wr.detach(), m.double(), iv_ref).float(), "distloss (published form)", rtol=2e-5)
        assert_close(ggot, O.eff_distloss_published_grad(wr.detach(), m.double(), iv_ref).float(),
                     "d distloss / dw (published form)", rtol=1e-4)
    got2 = rodynrf.eff_distloss(w.cuda(), m.cuda(), 1.0 / S)
    assert_close(got2, O.eff_distloss(w.double(), m.double(), 1.0 / S).float(), "eff_distloss", rtol=2e-5)
    with pytest.raises(NotImplementedError):
        rodynrf.flatten_eff_distloss(w.cuda().reshape(-1), m.cuda().reshape(-1), 1.0 / S,
                                     torch.zeros(N * S, dtype=torch.long, device="cuda") + (N - 1))


@pytest.mark.parametrize("T,N", [(12, 4096), (50, 8192), (1000, 3000)])
def test_gather_rows_backward_matches_torch_index_backward(T, N):
    """ray_utils.gather_rows (the [T,3,4] camera matrices of the neighbour frames, train.py:1895-1948): forward = table[idx]
    bit for bit, backward (rdrf_rows_scatter_add: LDS accumulation per workgroup; the 1000-row table takes the global-atomic
    fallback) against torch's index backward."""
    RU = importlib.import_module("robust-dynrf_amd.ray_utils")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    table = torch.randn(T, 3, 4, generator=g).to(dev)
    idx = torch.randint(0, T, (N,), generator=g).to(dev)
    go = torch.randn(N, 3, 4, generator=g).to(dev)
    t1 = table.clone().requires_grad_(True)
    t2 = table.clone().requires_grad_(True)
    o1 = RU.gather_rows(t1, idx)
    o2 = t2[idx]
    assert torch.equal(o1, o2)
    o1.backward(go)
    o2.backward(go)
    assert_close(t1.grad, t2.grad, "gather_rows backward", rtol=1e-5)
    assert torch.equal(RU.gather_rows(table, idx), table[idx])   # no grad: plain indexing
