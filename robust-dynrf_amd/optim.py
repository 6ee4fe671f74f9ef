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
                 mode="allreduce", group=None, lr_upsample_reset=True):
        self.fields = list(fields)
        self.lr_init, self.lr_basis = float(lr_init), float(lr_basis)
        self.lr0, self.lr1 = self.lr_init, self.lr_basis
        self.betas, self.eps, self.lr_factor = betas, float(eps), float(lr_factor)
        self.lr_upsample_reset = bool(lr_upsample_reset)   # opt.py:73-77, default 1 in every shipped config
        self.t = 0
        self._mode, self._group = mode, group
        self.rebuild()

    # ---- state -----------------------------------------------------------------------------------
    def rebuild(self, iteration=None):
        """(re)allocate the flat views and zero moments: at construction and after upsample_volume_grid,
        where the reference also builds a NEW Adam (train.py:2582-2606): the moments are dropped and the
        learning rates restart at lr_init / lr_basis (lr_upsample_reset, the default) or at
        lr * lr_decay_target_ratio ** (iteration / n_iters) = lr * lr_factor ** iteration."""
        if iteration is not None:
            scale = 1.0 if self.lr_upsample_reset else self.lr_factor ** int(iteration)
            self.lr0, self.lr1 = self.lr_init * scale, self.lr_basis * scale
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
        # every rank's loss is a mean over ITS rays, so the mean over ranks is the global mean for the plain means
        # (equal shard sizes).  The masked means (norm="weight": flow / disparity / static image terms) and the
        # per-frame median depth losses are normalised by per-SHARD statistics unless the trainer all-reduces them
        # (Trainer(dp_exact_stats=True): the numerators / denominators of train.py:1391-1394, 797-807): without
        # that an N-rank run optimises a slightly different objective from the 1-rank run.
        scale = 1.0 / self.world
        gathers = []
        for f in self.fields:
            f.det_fold_()
        for i, (f, st) in enumerate(zip(self.fields, self.state)):
            if f.flatten_params_().data_ptr() != st["p"].data_ptr():
                raise L.RdrfError("a field's parameters left the flat buffer FlatAdam updates (Module.to / .cuda / "
                                  "p.data = ... after construction): call FlatAdam.rebuild()")
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
