"""Host logic of robust-dynrf_amd/step.py that needs no GPU: the pass-batching plan and the draw order of the pooled
jitter generator (the batched and the per-pass paths must consume the same random numbers in the same order)."""
import importlib

import torch


def test_batch_groups_respect_the_sample_cap(monkeypatch):
    S_ = importlib.import_module("robust-dynrf_amd.step")
    monkeypatch.setattr(S_, "BATCH_MAX_SAMPLES", 6_000_000)
    assert S_.batch_groups(range(4), 4096 * 115) == [[0, 1, 2, 3]]                 # benchmark shape: one call
    assert S_.batch_groups([1, 2, 3], 4096 * 270) == [[1, 2, 3]]                   # final stage
    assert S_.batch_groups([1, 2, 3], 4096 * 578) == [[1, 2], [3]]                 # 640^3 grid: two passes at a time
    assert S_.batch_groups(range(4), 4096 * 578) == [[0, 1], [2, 3]]
    assert S_.batch_groups(range(4), 8192 * 221) == [[0, 1, 2], [3]]               # DAVIS final
    assert S_.batch_groups([2, 3], 10 ** 9) == [[2], [3]]                          # a pass larger than the cap: alone
    for idxs, n in (([0, 1, 2, 3], 1), ([1, 2, 3], 5_000_000), ([5], 7)):
        g = S_.batch_groups(idxs, n)
        assert [k for grp in g for k in grp] == list(idxs) and all(len(grp) * n <= max(n, 6_000_000) for grp in g)


def test_step_rng_pool_and_coin_streams():
    """StepRng: the coins come from its own seeded generator (the same sequence whatever happens in between); the
    jitter vectors are consecutive 64-float-aligned slices of one pooled uniform draw, so taking (jitter, coin) for four
    passes up front (ray_passes) or one pass at a time (ray_pass) consumes the same values in the same order."""
    S_ = importlib.import_module("robust-dynrf_amd.step")
    dev = torch.device("cpu")
    a, b = S_.StepRng(seed=11), S_.StepRng(seed=11)
    coins_a = [a.coin() for _ in range(16)]
    coins_b = []
    for _ in range(16):
        b.jitter(115, "ndc", dev)
        torch.rand(7)                                   # unrelated work on the global generator in between
        coins_b.append(b.coin())
    assert coins_a == coins_b and 0 < sum(coins_a) < 16
    c = S_.StepRng(seed=3)
    j0 = c.jitter(115, "ndc", dev)[0]
    pool = c._pool
    c.coin()
    j1 = c.jitter(115, "ndc", dev)[0]
    assert j0.shape == (115,) and torch.equal(j0, pool[0:115]) and torch.equal(j1, pool[128:243])
    jo = c.jitter(221, "contract", dev)
    assert jo[0].shape[0] == 221 - 221 // 2 + 1 and jo[1].shape[0] == 221 // 2 + 1
    assert torch.equal(jo[0], pool[256:256 + 111 + 1]) and torch.equal(jo[1], pool[384:384 + 111])
    assert float(pool.min()) >= 0.0 and float(pool.max()) < 1.0


def test_bench_reads_the_committed_pmc_summaries():
    """bench.py prices `roofline.traffic / hbm_real / l2_atomic_frac` from the committed rocprofv3 summaries under
    profiles/: the parsers must find the scatter family, the dW kernel and the step count in this round's files (a
    silent None would drop the fields from the driver's line)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    fam = ("void k_scatter<", "void k_scatter_sorted<", "void k_scatter_tiled<")
    fetch = bench.pmc_family_per_step("pmc_fetch", fam, "FETCH_SIZE")
    write = bench.pmc_family_per_step("pmc_write", fam, "WRITE_SIZE")
    atom = bench.pmc_family_per_step("sq_counters", fam, "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum")
    assert fetch and write and atom
    assert 1e5 < fetch < 1e7 and 1e5 < write < 1e7          # KB per training step
    assert 1e6 < atom < 1e8                                 # memory-side atomic requests per training step (r05: 6.8 M)
    dw = bench.pmc_traffic("k_dw2")
    assert dw and 1e8 < dw < 1e10                           # bytes per launch
    assert bench._profile_csv("pmc_fetch")[1] == bench.PROFILE_TAG
