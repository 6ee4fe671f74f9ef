"""world_size-2 gloo test of the data-parallel exchange (runs on CPU): sharded gradients summed
through GradBucket.allreduce_ equal the unsharded gradient, for dense and channel-last tensors."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    P = importlib.import_module("robust-dynrf_amd.parallel")
    F = importlib.import_module("robust-dynrf_amd.fields")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    P.init_distributed("gloo")
    torch.manual_seed(0)
    plane = torch.nn.Parameter(F.channel_last_(torch.randn(1, 4, 5, 6)))
    lin = torch.nn.Parameter(torch.randn(7, 3))
    x = torch.randn(10, 3)  # 10 "rays"
    lo, hi = P.shard_bounds(10, rank, world)
    loss = ((x[lo:hi] @ lin.T) ** 2).sum() + (plane.sum() * x[lo:hi].sum()) ** 2
    loss.backward()
    b = P.GradBucket([plane, lin])
    b.allreduce_()
    # reference: unsharded, same decomposition of the loss
    plane2 = torch.nn.Parameter(plane.detach().clone())
    lin2 = torch.nn.Parameter(lin.detach().clone())
    tot = 0
    for r in range(world):
        l2, h2 = P.shard_bounds(10, r, world)
        tot = tot + ((x[l2:h2] @ lin2.T) ** 2).sum() + (plane2.sum() * x[l2:h2].sum()) ** 2
    tot.backward()
    ok = torch.allclose(plane.grad, plane2.grad, rtol=1e-5, atol=1e-6) and \
        torch.allclose(lin.grad, lin2.grad, rtol=1e-5, atol=1e-6) and \
        plane.grad.stride() == plane.stride()
    # fused path: the field-style flat gradient buffer is all-reduced in place (no copies)
    flat = torch.zeros(64 + plane.numel())
    gl = flat[:lin.numel()].view_as(lin)
    gp = torch.as_strided(flat, plane.size(), plane.stride(), 64)
    lin3 = torch.nn.Parameter(lin.detach().clone())
    plane3 = torch.nn.Parameter(F.channel_last_(plane.detach().clone()))
    loss3 = ((x[lo:hi] @ lin3.T) ** 2).sum() + (plane3.sum() * x[lo:hi].sum()) ** 2
    loss3.backward()
    gl.copy_(lin3.grad)
    gp.copy_(plane3.grad)
    b2 = P.GradBucket([plane3, lin3], flats=lambda: [flat])
    b2.allreduce_()
    ok = ok and torch.allclose(gl, lin2.grad, rtol=1e-5, atol=1e-6) and \
        torch.allclose(gp, plane2.grad, rtol=1e-5, atol=1e-6) and b2.nbytes() == flat.numel() * 4
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29731
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def _ref_adam(p, g, m, v, t, lr_of, b1=0.9, b2=0.99, eps=1e-8):
    """torch.optim.Adam's update, element-wise (the arithmetic rdrf_adam_step implements)"""
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = v.sqrt() / (1 - b2 ** t) ** 0.5 + eps
    p.sub_(lr_of / (1 - b1 ** t) * m / denom)


def _worker_exchange(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import importlib
    P = importlib.import_module("robust-dynrf_amd.parallel")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    P.init_distributed("gloo")
    totals = [4096, 8192]
    split = [1024, 4096 + 640]         # lr boundary inside the second buffer's second shard
    gen = torch.Generator().manual_seed(0)
    p0 = [torch.randn(t, generator=gen) for t in totals]
    grads = [[torch.randn(t, generator=gen) for t in totals] for _ in range(world)]   # per rank
    results = {}
    for mode in ("allreduce", "zero1"):
        ex = P.FlatExchange(totals, mode)
        assert ex.mode == mode and ex.world == world
        params = [p.clone() for p in p0]
        moms = [(torch.zeros(ex.slice(i)[1]), torch.zeros(ex.slice(i)[1])) for i in range(2)]
        for step in range(1, 4):
            g_local = [g.clone() * step for g in grads[rank]]
            ex.begin(0, g_local[0])              # early start of buffer 0 (static field), async
            works = []
            for i in range(2):
                g, lo, n = ex.grads(i, g_local[i])
                lr = torch.where(torch.arange(lo, lo + n) < split[i], 0.02, 1e-3)
                _ref_adam(params[i][lo: lo + n], g / world, moms[i][0], moms[i][1], step, lr)
                works.append(ex.gather(i, params[i]))
            for w in works:
                if w is not None:
                    w.wait()
        results[mode] = params
    # single-process reference on the summed gradients
    ref = [p.clone() for p in p0]
    rm = [(torch.zeros(t), torch.zeros(t)) for t in totals]
    for step in range(1, 4):
        for i in range(2):
            g = sum(grads[r][i] for r in range(world)) * step / world
            lr = torch.where(torch.arange(totals[i]) < split[i], 0.02, 1e-3)
            _ref_adam(ref[i], g, rm[i][0], rm[i][1], step, lr)
    ok = all(torch.allclose(results["zero1"][i], ref[i], rtol=1e-5, atol=1e-7) and
             torch.allclose(results["allreduce"][i], ref[i], rtol=1e-5, atol=1e-7) for i in range(2))
    try:   # a buffer that does not split into aligned shards is refused
        P.FlatExchange([4098], "zero1")
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_flat_exchange_zero1_equals_allreduce_world2():
    """reduce-scatter -> Adam on the owned slice -> all-gather (ZeRO-1) gives the same parameters as
    all-reduce + replicated Adam and as one process on the summed gradients, over 3 steps, with the async
    early start of one buffer (the overlap the trainer uses for the static field)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, 29741, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def _worker_world4(rank, world, port, q):
    """world 4: the FlatExchange round trip of _worker_exchange, then a batch that does not divide by the world size: every
    rank must raise BEFORE a collective is posted (no hang), and a barrier afterwards must still complete"""
    sys.path.insert(0, ROOT)
    import importlib
    P = importlib.import_module("robust-dynrf_amd.parallel")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    P.init_distributed("gloo")
    ok = True
    try:
        P.require_even_shards(8190, world)   # 8190 = 2 * 4095: divides by 2, not by 4
        ok = False
    except ValueError as e:
        ok = ok and "divisible" in str(e)
    P.require_even_shards(8192, world)
    try:   # a flat buffer whose shards would not be 16-byte aligned is refused at construction, on every rank
        P.FlatExchange([4096 + 8], "zero1")
        ok = False
    except ValueError:
        pass
    dist.barrier()   # nothing is left pending on the group
    # uneven ray shards without the exact statistics are legal: shard_bounds covers the batch exactly once
    lo, hi = P.shard_bounds(8190, rank, world)
    cover = torch.zeros(8190)
    cover[lo:hi] = 1
    dist.all_reduce(cover)
    ok = ok and bool((cover == 1).all())
    totals = [4096, 8192]
    gen = torch.Generator().manual_seed(1)
    grads = [[torch.randn(t, generator=gen) for t in totals] for _ in range(world)]
    for mode in ("zero1", "allreduce"):
        ex = P.FlatExchange(totals, mode)
        plan = ex.plan(["static_field", "dynamic_field"])
        ok = ok and len(plan) == (4 if mode == "zero1" else 2)
        ok = ok and plan[0]["ring_bytes_per_rank"] == (totals[0] * 4 * 3 // 4) * (1 if mode == "zero1" else 2)
        for i in range(2):
            g, lo, n = ex.grads(i, grads[rank][i].clone())
            want = sum(grads[r][i] for r in range(world))[lo: lo + n]
            ok = ok and n == (totals[i] // world if mode == "zero1" else totals[i]) and torch.allclose(g, want, rtol=1e-5, atol=1e-6)
            p = torch.zeros(totals[i])
            p[lo: lo + n] = rank + 1.0
            w = ex.gather(i, p)
            if w is not None:
                w.wait()
                ok = ok and all(bool((p[r * n:(r + 1) * n] == r + 1.0).all()) for r in range(world))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_flat_exchange_world4_and_uneven_batch_raises():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_world4, args=(r, 4, 29751, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(60)
    assert len(res) == 4 and all(ok for _, ok in res), res
