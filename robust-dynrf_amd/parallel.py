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


def force_collectives():
    """RDRF_FORCE_COLLECTIVES=1: create the process group and issue every collective even at world size 1,
    so that the RCCL call sequence (reduce_scatter_tensor / in-place all_gather_into_tensor / async handles)
    can be exercised on a single-GPU box (tests/test_gpu_trainer.py)."""
    return os.environ.get("RDRF_FORCE_COLLECTIVES", "0") == "1"


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns
    (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
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


def require_even_shards(batch_size, world):
    """The exact-statistics exchange of the trainer (step.Trainer(dp_exact_stats=True)) gathers the per-ray depths of the
    whole batch with all_gather_into_tensor, which needs equal shards on every rank: with a batch that does not divide by
    the world size the ranks would post collectives of different sizes and hang.  Every rank evaluates the same two
    integers, so every rank raises -- before any collective is issued."""
    if world > 0 and batch_size % world != 0:
        raise ValueError(f"dp_exact_stats needs batch_size ({batch_size}) divisible by the world size ({world}); "
                         "use --dp-per-shard-stats or a divisible batch")


class FlatExchange:
    """The data-parallel exchange of flat gradient buffers, separated from the optimiser
    arithmetic so that the choreography is testable on CPU tensors with gloo:

      mode "allreduce": sum-all-reduce of the whole buffer, every rank updates every parameter;
      mode "zero1":     reduce-scatter -> the rank updates the slice [rank*n, (rank+1)*n) it owns ->
                        all-gather of the updated parameter slices (ZeRO-1: moments only for the slice).

    begin(i, g) may be called as soon as buffer i is complete (async); grads(i) waits and returns
    (gradient tensor to use, lo, n); gather(i, p) publishes the updated slice of parameter buffer p."""

    def __init__(self, totals, mode="zero1", group=None):
        if mode not in ("allreduce", "zero1"):
            raise ValueError(mode)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.active = self.world > 1 or (force_collectives() and dist.is_available() and dist.is_initialized())
        self.mode = mode if self.active else "allreduce"
        self.totals = list(totals)
        if self.mode == "zero1":
            for t in self.totals:
                if t % (4 * self.world) != 0:
                    raise ValueError(f"flat buffer of {t} floats does not split into {self.world} 16-byte aligned shards")
        self._pending, self._g, self._shards = {}, {}, {}

    def plan(self, names=None):
        """the collectives one iteration issues on this exchange and what they move: a list of dicts (collective, buffer,
        bytes = size of the buffer the collective covers, ring_bytes_per_rank = what one rank sends (= receives) over its
        xGMI links on a ring: (G - 1) / G of the buffer for a reduce-scatter or an all-gather, twice that for an
        all-reduce).  bench.py prints it so that the first multi-GPU run can be checked against it."""
        out, G = [], max(self.world, 1)
        for i, t in enumerate(self.totals):
            nm = names[i] if names else f"buffer{i}"
            b = t * 4
            if self.mode == "zero1":
                out.append(dict(collective="reduce_scatter_tensor", buffer=nm + ".grad", bytes=b, ring_bytes_per_rank=b * (G - 1) // G))
                out.append(dict(collective="all_gather_into_tensor", buffer=nm + ".param", bytes=b, ring_bytes_per_rank=b * (G - 1) // G))
            else:
                out.append(dict(collective="all_reduce", buffer=nm + ".grad", bytes=b, ring_bytes_per_rank=2 * b * (G - 1) // G))
        return out

    def slice(self, i):
        if self.mode == "zero1":
            n = self.totals[i] // self.world
            return self.rank * n, n
        return 0, self.totals[i]

    def begin(self, i, g, async_op=True):
        if i in self._g:
            return
        self._g[i] = g
        if not self.active:
            return
        if self.mode == "zero1":
            lo, n = self.slice(i)
            sh = self._shards.get(i)
            if sh is None or sh.numel() != n or sh.device != g.device:
                sh = self._shards[i] = torch.empty(n, dtype=g.dtype, device=g.device)
            w = dist.reduce_scatter_tensor(sh, g, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        else:
            w = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        self._pending[i] = w if async_op else None

    def grads(self, i, g=None):
        """wait for buffer i; returns (tensor holding the summed gradient of this rank's slice, lo, n)"""
        if i not in self._g:
            self.begin(i, g, async_op=False)
        w = self._pending.pop(i, None)
        if w is not None:
            w.wait()
        g = self._g.pop(i)
        lo, n = self.slice(i)
        return (self._shards[i] if (self.mode == "zero1" and self.active) else g), lo, n

    def gather(self, i, p, async_op=True):
        """zero1: all-gather the updated parameter slices into the flat parameter buffer p (in place)"""
        if not self.active or self.mode != "zero1":
            return None
        lo, n = self.slice(i)
        src = p[lo: lo + n]
        if dist.get_backend(self.group) != "nccl":   # the in-place form (input = own slice of output) is an NCCL idiom
            src = src.clone()
        return dist.all_gather_into_tensor(p, src, group=self.group, async_op=async_op)
