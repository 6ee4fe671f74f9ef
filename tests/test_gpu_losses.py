"""csrc/rdrf_loss.hip (LossTerms): every reducer / normaliser against the torch expression the reference
writes (train.py:1323-1421, 1522-1627, 1828-1832, 2293-2299), values and gradients, through the C ABI."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu


def masked_mean(x, m):
    return (x * m).sum() / (m.sum() + 1e-8)


def test_loss_terms_match_the_torch_expressions():
    import rodynrf
    from rodynrf import LossTerms
    g = torch.Generator().manual_seed(5)
    N, S = 777, 37   # ragged: not a multiple of the block size
    dev = "cuda"

    def rnd(*shape, grad=True):
        return (torch.rand(*shape, generator=g) * 2 - 1).to(dev).requires_grad_(grad)

    rgb, rgb_d, tgt = rnd(N, 3), rnd(N, 3), rnd(N, 3, grad=False)
    dyn, fg = rnd(N), (torch.rand(N, generator=g) < 0.3).float().to(dev)
    depth_a, depth_b = rnd(N), rnd(N)
    sf_f, sf_b = rnd(N, S, 3), rnd(N, S, 3)
    wts = torch.rand(N, S, generator=g).to(dev).requires_grad_(True)
    flow, flow_t = rnd(N, 2), rnd(N, 2, grad=False)
    mask = (torch.rand(N, 1, generator=g) < 0.8).float().to(dev)
    disp_a, disp_b = rnd(N, 1), rnd(N, 1)
    inv_d = rnd(N)
    leaves = [rgb, rgb_d, dyn, depth_a, depth_b, sf_f, sf_b, wts, flow, disp_a, disp_b, inv_d]

    def reference():
        w_d = wts.detach()[..., None]
        l = 3.0 * ((rgb - tgt) ** 2).mean() + ((rgb_d - tgt) ** 2).mean()
        l = l + 0.1 * (dyn - fg).abs().mean()
        l = l + 0.01 * dyn.mean() + 0.01 * (depth_a - depth_b.detach()).abs().mean()
        l = l + 0.01 * (sf_f.abs() * w_d).mean() + 0.01 * (sf_b.abs() * w_d).mean()
        l = l + 0.01 * ((sf_f + sf_b) ** 2 * w_d).mean()
        l = l + 0.02 * 0.5 * masked_mean((flow - flow_t).abs(), mask)
        l = l + 0.04 * masked_mean((disp_a - disp_b).abs(), mask)
        l = l + masked_mean((rgb - tgt) ** 2, (1.0 - fg)[:, None]) / 3.0
        l = l + 0.04 * ((depth_a - tgt[:, 0]).abs() * (1.0 - fg)).mean()
        l = l + 50.0 * ((inv_d - 1.0 / torch.clamp(depth_b, min=0.25)) ** 2).mean()
        l = l + 50.0 * ((inv_d - 1.0 / torch.clamp(depth_a, min=0.25)) ** 2).mean()   # inv_d in two terms
        return l

    def fused():
        T = LossTerms()
        w_d = wts.detach()
        T.add(3.0, "square", rgb, tgt).add(1.0, "square", rgb_d, tgt)
        T.add(0.1, "abs", dyn, fg)
        T.add(0.01, "identity", dyn).add(0.01, "abs", depth_a, depth_b.detach())
        T.add(0.01, "abs", sf_f, w=w_d).add(0.01, "abs", sf_b, w=w_d)
        T.add(0.01, "square", sf_f, sf_b, ysign=1.0, w=w_d)
        T.add(0.01, "abs", flow, flow_t, w=mask, norm="weight")
        T.add(0.04, "abs", disp_a, disp_b, w=mask, norm="weight")
        T.add(1.0 / 3.0, "square", rgb, tgt, w=(1.0 - fg)[:, None], norm="weight")
        T.add(0.04, "abs", depth_a, tgt[:, 0], w=1.0 - fg)
        T.add(50.0, "square", inv_d, 1.0 / torch.clamp(depth_b, min=0.25))
        T.add(50.0, "square", inv_d, 1.0 / torch.clamp(depth_a, min=0.25))
        l = T.total()
        assert T.values.shape == (len(T),)
        assert_close(T.values.sum(), l, "sum of the per-term values", rtol=1e-6)
        return l

    lr = reference()
    gr = torch.autograd.grad(lr, leaves, allow_unused=True)
    lf = fused()
    gf = torch.autograd.grad(lf, leaves, allow_unused=True)
    assert_close(lf, lr, "fused loss", rtol=2e-6)
    for name, a, b in zip("rgb rgb_d dyn depth_a depth_b sf_f sf_b wts flow disp_a disp_b inv_d".split(), gf, gr):
        if b is None:
            assert a is None, name   # the sample weights are constants of the scene-flow terms
            continue
        assert_close(a, b, "d / d " + name, rtol=2e-6)
    # deterministic: two evaluations are bit-identical (two-stage reduction, no float atomics)
    assert torch.equal(fused(), lf)


def test_loss_terms_errors_are_loud():
    import rodynrf
    from rodynrf import LossTerms
    x = torch.rand(8, 3, device="cuda")
    with pytest.raises(rodynrf.RdrfError):
        LossTerms().add(1.0, "abs", x, torch.rand(8, 2, device="cuda"))
    with pytest.raises(rodynrf.RdrfError):
        LossTerms().add(1.0, "abs", x, w=torch.rand(5, device="cuda"))
    with pytest.raises(rodynrf.RdrfError):
        LossTerms().add(1.0, "abs", x, norm="weight")
    with pytest.raises(rodynrf.RdrfError):
        LossTerms().add(1.0, "abs", x.cpu())
    T = LossTerms()
    for _ in range(32):   # RDRF_MAX_LOSS_TERMS
        T.add(1.0, "abs", x)
    assert torch.isfinite(T.total())
    with pytest.raises(rodynrf.RdrfError):
        T.add(1.0, "abs", x)


@pytest.mark.parametrize("N,T", [(257, 12), (4096, 12), (8192, 50), (32768, 12)])
def test_frame_depth_loss_matches_reference_loop(N, T):
    """rdrf_frame_depth_loss_fwd/bwd (one workgroup per frame: selection, LDS bitonic sort, loss + gradient in one pass)
    vs the reference's per-frame host loop (train.py:797-807, 1636-1664, 2097-2121, restated in the oracle with torch's
    own median / autograd): values and gradients, with empty frames, a single-ray frame (skipped), even / odd counts
    (lower median), a mask, exact ties at the median (the gradient is spread evenly over them, ATen's
    evenly_distribute_backward), a loss weight folded in, and one frame holding most of the batch."""
    import importlib
    from oracle import rodynrf_oracle as O
    LS = importlib.import_module("robust-dynrf_amd.losses")
    g = torch.Generator().manual_seed(3)
    frame = torch.randint(0, T - 2, (N,), generator=g)          # frames T-2, T-1 stay empty
    frame[:1] = T - 2                                           # one frame with a single ray: skipped
    frame[N // 3:] = 1                                          # one large frame (multi-chunk selection, big sort)
    pred0 = torch.randn(N, generator=g) * 3
    gt = torch.rand(N, generator=g)
    pred_ties = pred0.clone()
    idx0 = (frame == 0).nonzero()[:, 0]
    med0 = pred_ties[idx0].median()
    pred_ties[idx0[:3]] = med0                                  # frame 0: several elements equal to the median
    for pred_cpu, mask in ((pred0, None), (pred0, torch.rand(N, generator=g) < 0.6), (pred_ties, None)):
        p1 = pred_cpu.clone().cuda().requires_grad_(True)
        p2 = pred_cpu.clone().requires_grad_(True)
        a = LS.frame_depth_loss(p1, gt.cuda(), frame.cuda(), T, mask=None if mask is None else mask.cuda(), coef=0.04)
        b = 0.04 * O.frame_depth_loss(p2, gt, frame, T, mask=mask)
        assert_close(a, b, "loss", rtol=2e-5)
        (a * 1.7).backward()
        (b * 1.7).backward()
        assert_close(p1.grad, p2.grad, "d loss / d pred", rtol=1e-4, elem=(1e-3, 2e-6))
    # deterministic: same bits on repetition
    p3 = pred0.clone().cuda().requires_grad_(True)
    c1 = LS.frame_depth_loss(p3, gt.cuda(), frame.cuda(), T)
    c2 = LS.frame_depth_loss(p3, gt.cuda(), frame.cuda(), T)
    assert torch.equal(c1, c2)
    with pytest.raises(Exception):
        LS.frame_depth_loss(torch.zeros(40000, device="cuda"), torch.zeros(40000, device="cuda"),
                            torch.zeros(40000, dtype=torch.long, device="cuda"), T)    # N <= 32768
