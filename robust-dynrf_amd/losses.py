"""Fused loss terms (csrc/rdrf_loss.hip): the photometric / mask / flow / disparity / scene-flow terms of an
iteration -- each `coef * reduce(rho(x -+ y) * w) / Z` -- collected with LossTerms.add(...) and evaluated by
LossTerms.total() in one reduction launch forward and one launch backward, instead of the 5-15 elementwise
torch launches per term the expressions of train.py:1323-1421, 1522-1627, 1828-1832, 2293-2299 cost."""
import ctypes as C

import torch

from . import _lib as L

SQUARE, ABS, IDENTITY = 0, 1, 2
_KINDS = {"square": SQUARE, "abs": ABS, "identity": IDENTITY}


class _LossTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, reducer, *tensors):
        n = len(meta)
        dev = tensors[0].device
        arr = (L.RdrfLossTerm * n)()
        held = []
        for k, (kind, norm, ysign, coef, rows, cols, cdev) in enumerate(meta):
            x, y, w = tensors[3 * k: 3 * k + 3]
            x = L.f32c(x)
            y = None if y is None else L.f32c(y)
            w = None if w is None else L.f32c(w)
            held += [x, y, w]
            t = arr[k]
            t.x, t.y, t.w = x.data_ptr(), (0 if y is None else y.data_ptr()), (0 if w is None else w.data_ptr())
            t.gx = t.gy = 0
            t.rows, t.cols, t.kind, t.norm, t.ysign, t.coef = rows, cols, kind, norm, ysign, coef
            t.coef_dev = 0 if cdev is None else cdev.data_ptr()
        partial = torch.empty(int(L.lib.rdrf_loss_terms_workspace_floats(n)), device=dev)
        out = torch.empty(1 + 2 * n, device=dev)
        if reducer is None:
            L.check(L.lib.rdrf_loss_terms_fwd(arr, n, L.ptr(partial), L.ptr(out), L.stream_of(out)), "rdrf_loss_terms_fwd")
        else:   # data parallel, exact normalisers: all-reduce the per-term (sum, weight sum) pairs between the two stages
            stats = torch.empty(2 * n, device=dev)
            L.check(L.lib.rdrf_loss_terms_stats(arr, n, L.ptr(partial), L.ptr(stats), L.stream_of(out)), "rdrf_loss_terms_stats")
            glob, world = reducer(stats)
            L.check(L.lib.rdrf_loss_terms_finish(arr, n, L.ptr(stats), L.ptr(glob), int(world), L.ptr(out), L.stream_of(out)),
                    "rdrf_loss_terms_finish")
        ctx.meta, ctx.held, ctx.out = meta, held, out
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        meta, held, out = ctx.meta, ctx.held, ctx.out
        n = len(meta)
        arr = (L.RdrfLossTerm * n)()
        grads = []
        for k, (kind, norm, ysign, coef, rows, cols, cdev) in enumerate(meta):
            x, y, w = held[3 * k: 3 * k + 3]
            need_x, need_y = ctx.needs_input_grad[2 + 3 * k], ctx.needs_input_grad[3 + 3 * k]
            if ctx.needs_input_grad[4 + 3 * k]:
                raise L.RdrfError("LossTerms: the row weights are constants of the term (detach them)")
            gx = torch.empty_like(x) if need_x else None
            gy = torch.empty_like(y) if (need_y and y is not None) else None
            t = arr[k]
            t.x, t.y, t.w = x.data_ptr(), (0 if y is None else y.data_ptr()), (0 if w is None else w.data_ptr())
            t.gx, t.gy = (0 if gx is None else gx.data_ptr()), (0 if gy is None else gy.data_ptr())
            t.rows, t.cols, t.kind, t.norm, t.ysign, t.coef = rows, cols, kind, norm, ysign, coef
            t.coef_dev = 0 if cdev is None else cdev.data_ptr()   # (backward reads out[1 + k], which already holds it)
            grads += [gx, gy, None]
        g = L.f32c(g_loss.reshape(1))
        L.check(L.lib.rdrf_loss_terms_bwd(arr, n, L.ptr(out), L.ptr(g), L.stream_of(out)), "rdrf_loss_terms_bwd")
        return (None, None, *grads)


class _FrameDepthLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, frame, mask, T, coef, gscale):
        L.require_device(pred, gt, frame, mask)
        pred, gt = L.f32c(pred).reshape(-1), L.f32c(gt).reshape(-1)
        frame = frame.contiguous()
        if frame.dtype != torch.int64:
            frame = frame.long()
        if mask is not None:
            mask = mask.contiguous()
            mask = mask.view(torch.uint8) if mask.dtype == torch.bool else (mask != 0).view(torch.uint8)
        N = pred.numel()
        out = torch.empty(3, device=pred.device)
        g_raw = torch.empty(N, device=pred.device)
        nb = int(L.lib.rdrf_frame_depth_loss_workspace_bytes(N, int(T)))
        ws = L.workspace(pred.device, nb)
        L.check(L.lib.rdrf_frame_depth_loss_fwd(L.ptr(pred), L.ptr(gt), L.ptr(frame), L.ptr(mask), N, int(T), float(coef),
                                                L.ptr(out), L.ptr(g_raw), L.ptr(ws), C.c_size_t(nb), L.stream_of(pred)),
                "rdrf_frame_depth_loss_fwd")
        ctx.save_for_backward(g_raw, out)
        ctx.gscale = float(gscale)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g_loss):
        g_raw, out = ctx.saved_tensors
        g_pred = torch.empty_like(g_raw)
        L.check(L.lib.rdrf_frame_depth_loss_bwd(L.ptr(g_raw), L.ptr(out), L.ptr(L.f32c(g_loss.reshape(1))), g_raw.numel(),
                                                C.c_float(ctx.gscale), L.ptr(g_pred), L.stream_of(g_raw)),
                "rdrf_frame_depth_loss_bwd")
        return g_pred, None, None, None, None, None, None


class _GatherRaysFn(torch.autograd.Function):
    """all-gather of a per-ray tensor over the data-parallel group; backward keeps this rank's slice of the gradient
    (every rank computes the same global loss from the gathered batch, so the other slices' gradients belong to the
    other ranks)."""

    @staticmethod
    def forward(ctx, x, group, rank, world):
        import torch.distributed as dist
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        ctx.rank, ctx.n = rank, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.n: (ctx.rank + 1) * ctx.n].contiguous(), None, None, None


def frame_depth_loss(pred, gt, frame, T, mask=None, coef=1.0, dp=None):
    """dp = (group, rank, world): the statistics (per-frame medians, deviations, ray count) are those of the WHOLE
    data-parallel batch -- pred / gt / frame / mask are all-gathered (a few KB), every rank evaluates the loss of the
    gathered batch and keeps the gradient of its own rays, scaled by world for the exchange's mean-over-ranks."""
    if dp is not None:
        group, rank, world = dp
        pred = _GatherRaysFn.apply(pred, group, rank, world)
        gt = _GatherRaysFn.apply(gt, group, rank, world)
        frame = _GatherRaysFn.apply(frame, group, rank, world)
        mask = None if mask is None else _GatherRaysFn.apply(mask.to(torch.uint8), group, rank, world)
        return _frame_depth_loss(pred, gt, frame, T, mask, coef, gscale=float(world))
    return _frame_depth_loss(pred, gt, frame, T, mask, coef)


def _frame_depth_loss(pred, gt, frame, T, mask=None, coef=1.0, gscale=1.0):
    """coef * (train.py:797-807 compute_depth_loss summed over the frames of the batch with more than one ray) / (number
    of rays used): the per-frame median-normalised monocular depth loss of train.py:1636-1664 (dynamic) and
    2097-2121 (static, mask = background rays).  pred, gt [N]; frame [N] in [0, T).  One workgroup per frame
    (selection, LDS bitonic sort for the median, loss and gradient in the same pass): 2 launches instead of the ~40
    small sort / scatter / gather launches of a torch formulation, no host synchronisation."""
    if pred.shape != gt.shape or pred.numel() != frame.numel():
        raise L.RdrfError("frame_depth_loss: pred, gt, frame must have one entry per ray")
    return _FrameDepthLossFn.apply(pred, gt, frame, mask, T, coef, gscale)


class LossTerms:
    """terms = LossTerms(); terms.add(3.0, "square", rgb_map, rgb_gt); ...; loss = terms.total()"""

    def __init__(self, reducer=None):
        """reducer (data-parallel runs with exact loss statistics): callable(stats [2n] device tensor) -> (the same summed
        over the ranks, world size); None = this process' own statistics (the single-process arithmetic)."""
        self._reducer = reducer
        self._meta, self._tensors = [], []
        self.values = None   # after total(): per-term values (device tensor [n]), for logging

    def __len__(self):
        return len(self._meta)

    def add(self, coef, kind, x, y=None, ysign=-1.0, w=None, norm="mean"):
        """+= coef * sum(rho(x + ysign*y) * w[row]) / Z.   kind: "square" | "abs" | "identity";
        coef: a float, or (float, tensor): the float times a one-element fp32 DEVICE tensor read when the kernel runs (a
        weight that changes every iteration inside a captured HIP graph: Trainer(graph=True));
        w: one weight per row (x viewed as [w.numel(), -1]; no w: every element is a row);
        norm: "mean" (Z = x.numel()) or "weight" (Z = w.sum() + 1e-8: the reference's masked mean)."""
        if len(self._meta) >= L.MAX_LOSS_TERMS:
            raise L.RdrfError(f"LossTerms: more than {L.MAX_LOSS_TERMS} terms in one group")
        L.require_device(x, y, w)
        if x.dim() == 0:   # a scalar term (another fused loss' value): coef * x
            x = x.reshape(1)
        if y is not None and y.shape != x.shape:
            raise L.RdrfError(f"LossTerms: x {tuple(x.shape)} and y {tuple(y.shape)} differ")
        rows = x.numel() if w is None else w.numel()
        if rows == 0 or x.numel() % rows != 0:
            raise L.RdrfError(f"LossTerms: {x.numel()} elements do not split into {rows} weighted rows")
        if norm == "weight" and w is None:
            raise L.RdrfError("LossTerms: norm='weight' needs row weights")
        cdev = None
        if isinstance(coef, tuple):
            coef, cdev = coef
            if cdev is not None and (cdev.dtype != torch.float32 or cdev.numel() != 1 or not cdev.is_cuda or cdev.requires_grad):
                raise L.RdrfError("LossTerms: a device coefficient is one fp32 element on the GPU, not a trained tensor")
        self._meta.append((_KINDS[kind], 1 if norm == "weight" else 0, float(ysign), float(coef), rows, x.numel() // rows, cdev))
        self._tensors += [x, y, None if w is None else w.detach()]
        return self

    def total(self):
        loss, out = _LossTermsFn.apply(tuple(self._meta), self._reducer, *self._tensors)
        n = len(self._meta)
        self.values = out[1 + n:]
        return loss
