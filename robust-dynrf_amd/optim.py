"""Adam over the flat parameter buffers of the fields (rdrf_adam_step), replicated or sharded.

The reference builds ``torch.optim.Adam(grad_vars, betas=(0.9, 0.99))`` over 6 + 18 parameter groups with
two learning rates -- the VM factors at ``lr_init``, every network at ``lr_basis`` -- and multiplies every
group's rate by ``lr_factor`` each iteration (train.py:924-934, 2349-2351).  Here every parameter of a field
is a view of one flat buffer (TensorBase.flatten_params_) and so is every gradient (TensorBase.fused_grads),
so an optimiser step of a field is ONE launch over four flat ranges.

Data parallelism: rays are sharded, parameters replicated.  ``mode="allreduce"`` sums
the flat gradient buffers in place and every rank steps all parameters; ``mode="zero1"`` is the
reduce-scatter -> Adam on the owned 1/G slice -> all-gather form: each rank keeps Adam moments for its slice
only and updates 1/G of the parameters; both produce the same parameters.  The exchange of a field can be
started as soon as that field's backward is complete (``begin_exchange(i)``), so the static field's
reduce-scatter runs on RCCL's stream while the dynamic passes are still in their backward."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L
from .parallel import FlatExchange


class FlatAdam:
    def __init__(self, fields, lr_init=0.02, lr_basis=1e-3, betas=(0.9, 0.99), eps=1e-8, lr_factor=1.0,
                 mode="allreduce", group=None):
        self.fields = list(fields)
        self.lr0, self.lr1 = float(lr_init), float(lr_basis)
        self.betas, self.eps, self.lr_factor = betas, float(eps), float(lr_factor)
        self.t = 0
        self._mode, self._group = mode, group
        self.rebuild()

    # ---- state -----------------------------------------------------------------------------------
    def rebuild(self):
        """(re)allocate the flat views and zero moments: at construction and after upsample_volume_grid,
        where the reference also builds a NEW Adam (train.py:2582-2606: the moments are dropped)."""
        layouts = []
        for f in self.fields:
            pflat = f.flatten_params_()
            f.fused_grad = True
            gflat = f.zero_grad_fused()
            _, total, split = f._flat_layout()
            layouts.append((pflat, gflat, total, split))
        self.ex = FlatExchange([l[2] for l in layouts], self._mode, self._group)
        self.world, self.rank, self.mode = self.ex.world, self.ex.rank, self.ex.mode
        self.state = []
        for i, (pflat, gflat, total, split) in enumerate(layouts):
            lo, n = self.ex.slice(i)
            self.state.append(dict(p=pflat, g=gflat, lo=lo, n=n, split=min(max(split - lo, 0), n),
                                   m=torch.zeros(n, device=pflat.device), v=torch.zeros(n, device=pflat.device)))
        self.t = 0

    def grad_flats(self):
        return [st["g"] for st in self.state]

    def nbytes_exchanged(self):
        return sum(st["g"].numel() for st in self.state) * 4

    def zero_grad(self):
        for f, st in zip(self.fields, self.state):
            g = f.zero_grad_fused()
            if g.data_ptr() != st["g"].data_ptr():   # the field rebuilt its buffers (new shapes)
                raise L.RdrfError("the field's flat gradient buffer changed: call FlatAdam.rebuild() after "
                                  "upsample_volume_grid")

    def begin_exchange(self, i, async_op=True):
        """start the gradient exchange of field i (call when its backward is complete)."""
        self.ex.begin(i, self.state[i]["g"], async_op=async_op)

    # ---- step ------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self):
        """finish the exchange, Adam on the owned range, parameter all-gather (zero1), lr decay."""
        self.t += 1
        scale = 1.0 / self.world   # every rank's loss is a mean over ITS rays: mean over ranks = global mean
        gathers = []
        for i, st in enumerate(self.state):
            g, lo, n = self.ex.grads(i, st["g"])
            p = st["p"][lo: lo + n]
            L.check(L.lib.rdrf_adam_step(L.ptr(p), L.ptr(g), L.ptr(st["m"]), L.ptr(st["v"]), C.c_size_t(n),
                                         C.c_size_t(st["split"]), self.lr0, self.lr1, self.betas[0], self.betas[1],
                                         self.eps, self.t, scale, L.stream_of(p)), "rdrf_adam_step")
            gathers.append(self.ex.gather(i, st["p"]))
        for w in gathers:
            if w is not None:
                w.wait()
        for f in self.fields:   # the parameters changed behind torch's version counters: packed weight images are stale
            f._pack_epoch += 1
        self.lr0 *= self.lr_factor
        self.lr1 *= self.lr_factor
