"""Ray-sharded data parallelism for the trainer step: one process per GPU, the batch
of rays is split across ranks (all passes of a ray stay on its rank, no data-path collective), the
parameters are replicated, and there is ONE exchange per iteration: a sum all-reduce of the
parameter gradients over RCCL/xGMI (backend "nccl" on ROCm), issued as a single flat bucket so the
ring moves few large messages.

The reference has no distributed code at all; this mirrors the replica-per-GPU / scatter-rays
vestige of its earlier nn.DataParallel wrapper (renderer.py:488, models/tensorBase.py:427)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns
    (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # RDRF_DIST_BACKEND=gloo: functional test of the N>1 path on one GPU
            backend = os.environ.get("RDRF_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def flat_view(t):
    """1-D view over the (dense, possibly channel-last) storage of t -- no copy."""
    return t.as_strided((t.numel(),), (1,))


class GradBucket:
    """One flat fp32 buffer holding every parameter gradient; all-reduced once per iteration."""

    def __init__(self, params, flats=None):
        """`flats`: callable returning the fields' fused flat gradient buffers (TensorBase.fused_grads):
        when given they are all-reduced in place, no gather/scatter copies."""
        self.flats = flats
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        total = sum(self.sizes)
        dev = self.params[0].device
        self.flat = None if flats is not None else torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        o = 0
        for n in self.sizes:
            self.offsets.append(o)
            o += n

    def nbytes(self):
        if self.flats is not None:
            return sum(f.numel() for f in self.flats()) * 4
        return self.flat.numel() * 4

    @torch.no_grad()
    def allreduce_(self, average=False):
        """sum (or mean) the gradients across ranks in place."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        if self.flats is not None:
            for f in self.flats():
                dist.all_reduce(f, op=dist.ReduceOp.SUM)
                if average:
                    f.div_(dist.get_world_size())
            return
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            if p.grad is None:
                self.flat[o:o + n].zero_()
            else:
                self.flat[o:o + n].copy_(flat_view(p.grad))
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if average:
            self.flat.div_(dist.get_world_size())
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            flat_view(p.grad).copy_(self.flat[o:o + n])


def shard_bounds(n, rank, world):
    return rank * n // world, (rank + 1) * n // world
