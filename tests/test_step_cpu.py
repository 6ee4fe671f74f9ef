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
    dw = bench.pmc_traffic("k_dw3")
    assert dw and 1e8 < dw < 1e10                           # bytes per launch
    assert bench._profile_csv("pmc_fetch")[1] == bench.PROFILE_TAG
    # the configuration whose factors exceed the caches: a true HBM fraction of its dominant kernel
    r = bench.committed_hbm_roofline("640_", "640^3")
    assert r and r["bound"] == "hbm" and 0.2 < r["frac"] < 1.0 and 0.2 < r["all_kernels_frac"] < 1.0
    assert r["kernel"].startswith("k_") and r["ms_per_step"] > 1.0


def test_bench_selects_the_baseline_config_of_the_gpu_count():
    """bench.py --gpus N: by default configs[1] (Balloon1, the configuration BASELINE.json's metric is quoted on at 1/2/4/8
    GPUs) at 4096 rays per GPU; --baseline-config quoted selects the configuration quoted for N GPUs:
    N = 4 -> configs[3], DAVIS.txt final grid, 8192 rays global = 2048 per rank; N = 8 -> configs[4], the 640^3 grid, 32768
    rays global = 4096 per rank; explicit flags win."""
    import argparse
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    def sel(world, **kw):
        a = argparse.Namespace(config=None, stage=None, baseline_config="auto", rays_per_gpu=0)
        a.__dict__.update(kw)
        b, explicit = bench.select_baseline_config(a, world)
        return b, explicit, a.config, a.stage, a.rays_per_gpu

    for n in (1, 2, 4, 8):   # the metric is quoted on Balloon1 @1/2/4/8: the default at every GPU count (weak scaling)
        assert sel(n) == ("1", False, "nvidia", "stage0", 0)
    assert sel(1, baseline_config="quoted") == ("1", False, "nvidia", "stage0", 0)
    assert sel(4, baseline_config="quoted") == ("3", False, "davis", "final", 2048)
    assert sel(8, baseline_config="quoted") == ("4", False, "nvidia_no_poses", "final", 4096)
    assert sel(8, baseline_config="4") == ("4", False, "nvidia_no_poses", "final", 4096)
    assert sel(4, config="nvidia") == ("1", True, "nvidia", "stage0", 0)
    assert sel(4, rays_per_gpu=512)[4] == 512                       # an explicit per-GPU batch is kept
    assert sel(1, config="davis", stage="stage0")[0] is None        # not a BASELINE configuration


def test_resolution_schedule_matches_the_reference_formulas():
    """step.scene_config("nvidia", stage) for the five stages of configs/Nvidia.txt's schedule: the grids / sample counts are
    those of utils.py:58-65 N_to_reso / cal_n_samples (restated in the oracle) on the log-spaced voxel counts of
    train.py:937-947, and resolution_schedule() covers the 100 000 iterations once."""
    import importlib
    import numpy as np
    import torch
    from oracle import rodynrf_oracle as O
    S_ = importlib.import_module("robust-dynrf_amd.step")
    aabb = torch.tensor(S_.scene_config("nvidia", "stage0")["aabb"])
    nvox = torch.round(torch.exp(torch.linspace(np.log(2097156), np.log(27000000), 5))).long().tolist()
    sched = S_.resolution_schedule("nvidia")
    assert [s[0] for s in sched] == ["stage0", "up1", "up2", "up3", "final"]
    assert sched[0][1] == 0 and sched[-1][2] == 100000 and all(a[2] == b[1] for a, b in zip(sched, sched[1:]))
    assert [s[1] for s in sched[1:]] == [8000, 12000, 16000, 22000]   # configs/Nvidia.txt upsamp_list
    for (stage, first, _), n in zip(sched, nvox):
        cfg = S_.scene_config("nvidia", stage)
        reso = O.N_to_reso(n, aabb)
        assert cfg["grid"] == reso and cfg["n_samples"] == O.cal_n_samples(reso, 2.0), (stage, reso)
        assert cfg["start_iteration"] == (0 if stage == "stage0" else first)


def test_graph_rng_hands_out_the_same_static_slices_every_iteration():
    """GraphRng (Trainer(graph=True)): the draws of an iteration are fixed slices of static memory -- the same storage at
    capture and at every replay -- refilled by begin(); coins are one-element fp32 tensors holding 0 or 1; frozen keeps
    the contents (tests replay fixed draws)."""
    S_ = importlib.import_module("robust-dynrf_amd.step")
    dev = torch.device("cpu")
    r = S_.GraphRng(dev, pool_floats=4096)
    seen = []
    for it in range(3):
        r.begin()
        j = r.jitter(115, "ndc", dev)[0]
        c0, c1 = r.coin(), r.coin()
        jo = r.jitter(221, "contract", dev)
        seen.append((j.data_ptr(), c0.data_ptr(), c1.data_ptr(), jo[0].data_ptr(), jo[1].data_ptr(), j.clone(), float(c0), float(c1)))
        assert j.shape == (115,) and c0.shape == (1,) and c0.dtype == torch.float32
        assert float(c0) in (0.0, 1.0) and float(c1) in (0.0, 1.0)
        assert j.data_ptr() == r.pool[64:].data_ptr()           # the coins' source (pool head) is not handed out as jitter
        assert jo[0].shape[0] == 221 - 221 // 2 + 1 and jo[1].shape[0] == 221 // 2 + 1
        assert torch.equal(r.coins, torch.round(r.pool[:r.N_COINS]))
    assert all(s[:5] == seen[0][:5] for s in seen)             # same addresses every iteration
    assert not torch.equal(seen[0][5], seen[1][5])             # fresh values
    r.frozen = True
    before = r.pool.clone()
    r.begin()
    assert torch.equal(r.pool, before)
    import pytest
    with pytest.raises(RuntimeError):
        for _ in range(r.N_COINS + 1):
            r.coin()
    with pytest.raises(RuntimeError):
        r._take(1 << 20, dev)
