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
