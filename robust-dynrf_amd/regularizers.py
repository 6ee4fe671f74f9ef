"""Factor-space regularisers of the reference, fused.

TVLoss mirrors utils.py:157-181; the fields' TV_loss_density / TV_loss_blending / TV_loss_app
(models/tensoRF.py:100-116, 418-444) hand it all six tensors of a factor family at once, so a family
costs one forward and one backward launch instead of ~16 small torch kernels per tensor per
direction.  The reference's value arithmetic -- including its 0/0 = NaN for lines, whose count_w is
zero, with finite gradients -- is kept by doing the scalar part in torch on the kernel's sums."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L


def _tensor4(x, g=None):
    if x.dim() != 4 or x.shape[0] != 1:
        raise L.RdrfError("TVLoss: expected a (1, C, H, W) tensor (utils.py:163)")
    s = x.stride()
    return L.RdrfTensor4(x.data_ptr(), 0 if g is None else g.data_ptr(), x.shape[1], x.shape[2], x.shape[3],
                         s[1], s[2], s[3])


class _TVSumsFn(torch.autograd.Function):
    """sums[t] = (sum of squared H-differences, sum of squared W-differences) of each tensor"""

    @staticmethod
    def forward(ctx, field, *xs):
        L.require_device(*xs)
        if not 0 < len(xs) <= L.TV_MAX:
            raise L.RdrfError(f"TVLoss: 1..{L.TV_MAX} tensors per call")
        if any(x.dtype != torch.float32 for x in xs):
            raise L.RdrfError("TVLoss: fp32 tensors only")
        arr = (L.RdrfTensor4 * len(xs))(*[_tensor4(x) for x in xs])
        sums = torch.empty(len(xs), 2, device=xs[0].device)
        L.check(L.lib.rdrf_tv_fwd(arr, len(xs), L.ptr(sums), L.stream_of(xs[0])), "rdrf_tv_fwd")
        ctx.field = field
        ctx.save_for_backward(*xs)
        return sums

    @staticmethod
    def backward(ctx, g_sums):
        xs = ctx.saved_tensors
        field = ctx.field
        fused = field is not None and field.fused_grad
        if fused:   # accumulate straight into p.grad (views of the field's flat buffer)
            views = {p.data_ptr(): v for p, v in zip(field._param_list(), field.fused_grads())}
            grads = [views[x.data_ptr()] for x in xs]
        else:
            grads = [torch.zeros_like(x) for x in xs]   # preserve_format keeps the channel-last strides
        arr = (L.RdrfTensor4 * len(xs))(*[_tensor4(x, g) for x, g in zip(xs, grads)])
        L.check(L.lib.rdrf_tv_bwd(arr, len(xs), L.ptr(L.f32c(g_sums)), L.stream_of(xs[0])), "rdrf_tv_bwd")
        return (None, *([None] * len(xs) if fused else grads))


class TVLoss(nn.Module):
    """utils.py:157-181, same constructor / call signature."""

    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    _cache = {}

    def _counts(self, shapes, device):
        """(count_h, count_w, batch) of utils.py:166-167 as device vectors, cached per shape list"""
        key = (tuple(shapes), str(device))
        c = TVLoss._cache.get(key)
        if c is None:
            ch = torch.tensor([c_ * (h - 1) * w for _, c_, h, w in shapes], dtype=torch.float32, device=device)
            cw = torch.tensor([c_ * h * (w - 1) for _, c_, h, w in shapes], dtype=torch.float32, device=device)
            b = torch.tensor([b_ for b_, _, _, _ in shapes], dtype=torch.float32, device=device)
            c = TVLoss._cache[key] = (ch, cw, b)
        return c

    def _value(self, sums, shapes, ignore_axis=None):
        """per-tensor losses (a vector) from the kernel sums with the reference's scalar arithmetic,
        vectorised over the tensors: x / 0 is NaN / inf exactly as `tensor / 0` is in the reference"""
        ch, cw, b = self._counts(shapes, sums.device)
        h_tv, w_tv = sums[:, 0], sums[:, 1]
        if ignore_axis is None:
            return self.TVLoss_weight * 2 * (h_tv / ch + w_tv / cw) / b
        if ignore_axis == "h":
            return self.TVLoss_weight * 2 * (w_tv / cw) / b
        if ignore_axis == "w":
            return self.TVLoss_weight * 2 * (h_tv / ch) / b
        raise ValueError(ignore_axis)

    def many(self, xs, field=None):
        """vector of per-tensor losses for up to 16 tensors with ONE forward / backward launch"""
        sums = _TVSumsFn.apply(field, *xs)
        return self._value(sums, [tuple(x.shape) for x in xs])

    def forward(self, x, ignore_axis=None):
        sums = _TVSumsFn.apply(None, x)
        return self._value(sums, [tuple(x.shape)], ignore_axis)[0]


    @torch.no_grad()
    def accumulate_grad_(self, field, families, weights):
        """d(sum_f weights[f] * TV_loss_f)/d(factors) added straight to the field's gradients, with ONE launch
        for all families (rdrf_tv_grad) and no value: `families` = list of (planes, lines) as TV_loss_* takes
        them.  The value of these terms is NaN in the reference (line tensors: 0/0) -- only their gradient
        ever matters -- so the trainer uses this instead of building the sums and back-propagating them."""
        xs, coef = [], []
        for (planes, lines), wt in zip(families, weights):
            for x, fam in [(p, 1e-2) for p in planes] + [(l, 1e-3) for l in lines]:
                b, c, h, w = x.shape
                ch = self.TVLoss_weight * 2.0 / (c * (h - 1) * w) / b if h > 1 else 0.0
                cw = self.TVLoss_weight * 2.0 / (c * h * (w - 1)) / b if w > 1 else 0.0
                xs.append(x)
                coef += [float(wt) * fam * ch, float(wt) * fam * cw]
        L.require_device(*xs)
        if field.fused_grad:
            views = {p.data_ptr(): v for p, v in zip(field._param_list(), field.fused_grads())}
            grads = [views[x.data_ptr()] for x in xs]
        else:
            grads = []
            for x in xs:
                if x.grad is None:
                    x.grad = torch.zeros_like(x)
                grads.append(x.grad)
        arr = (L.RdrfTensor4 * len(xs))(*[_tensor4(x, g) for x, g in zip(xs, grads)])
        ca = (C.c_float * len(coef))(*coef)
        L.check(L.lib.rdrf_tv_grad(arr, len(xs), ca, L.stream_of(xs[0])), "rdrf_tv_grad")


_coef = {}


def tv_family(field, reg, planes, lines):
    """models/tensoRF.py:100-116 / 418-444: sum_i reg(plane_i) * 1e-2 + reg(line_i) * 1e-3, in the
    reference's accumulation order.  A rodynrf TVLoss takes the fused path; any other callable is
    applied tensor by tensor exactly as the reference does."""
    planes, lines = list(planes), list(lines)
    if isinstance(reg, TVLoss):
        vals = reg.many(planes + lines, field)
        key = (len(planes), len(lines), str(vals.device))
        coef = _coef.get(key)
        if coef is None:
            coef = _coef[key] = torch.tensor([1e-2] * len(planes) + [1e-3] * len(lines), device=vals.device)
        return (vals * coef).sum()
    total = 0
    for p, l in zip(planes, lines):
        total = total + reg(p) * 1e-2 + reg(l) * 1e-3
    return total


# --------------------------------------------------------------------------------------------
# regularisers that configs/Nvidia.txt leaves at weight 0 (DAVIS.txt uses density_L1).  vector_diffs is a
# handful of (C x L)(L x C) products on the line tensors: plain torch.  dense_l1 is a kernel.
# --------------------------------------------------------------------------------------------
def vector_diffs(lines):
    """models/tensoRF.py:63-75: mean |off-diagonal| of the component Gram matrix of every line"""
    total = 0
    for v in lines:
        n_comp, n_size = v.shape[1:-1]
        m = v.reshape(n_comp, n_size)
        dotp = m @ m.transpose(-1, -2)
        off = dotp.reshape(-1)[1:].view(n_comp - 1, n_comp + 1)[..., :-1]
        total = total + off.abs().mean()
    return total


class _DenseL1Fn(torch.autograd.Function):
    """mean |feature2density(sum_c plane x line)| over the grid through rdrf_dense_l1_fwd/bwd: no dense
    volume is ever materialised (the reference builds a [1,24,X,Y,Z] tensor, models/tensoRF.py:80-98)."""

    @staticmethod
    def forward(ctx, field, *xs):
        from .fields import _vm_struct
        L.require_device(*xs)
        planes, lines = xs[:3], xs[3:]
        vm = _vm_struct(planes, lines)
        out = torch.empty(1, device=xs[0].device)
        L.check(L.lib.rdrf_dense_l1_fwd(C.byref(vm), L.ACTS[field.fea2denseAct], C.c_float(float(field.density_shift)),
                                        L.ptr(out), L.stream_of(xs[0])), "rdrf_dense_l1_fwd")
        ctx.field = field
        ctx.save_for_backward(*xs)
        nv = planes[0].shape[2] * planes[0].shape[3] * lines[0].shape[2]
        return out[0] / nv

    @staticmethod
    def backward(ctx, g):
        from .fields import _vm_struct
        xs = ctx.saved_tensors
        field = ctx.field
        fused = field.fused_grad
        field._det_bind()
        if fused:
            views = {p.data_ptr(): v for p, v in zip(field._param_list(), field.fused_grads())}
            grads = [views[x.data_ptr()] for x in xs]
        else:
            grads = [torch.zeros_like(x) for x in xs]
        vm, gvm = _vm_struct(xs[:3], xs[3:]), _vm_struct(grads[:3], grads[3:])
        g = L.f32c(g.reshape(1))
        L.check(L.lib.rdrf_dense_l1_bwd(C.byref(vm), C.byref(gvm), L.ACTS[field.fea2denseAct],
                                        C.c_float(float(field.density_shift)), L.ptr(g), L.stream_of(xs[0])),
                "rdrf_dense_l1_bwd")
        return (None, *([None] * len(xs) if fused else grads))


def dense_l1(field, planes, lines):
    """models/tensoRF.py:80-98 / 378-416: mean |feature2density(sum_c plane x line)| over the grid"""
    return _DenseL1Fn.apply(field, *planes, *lines)
