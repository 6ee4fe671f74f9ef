#!/usr/bin/env python
"""bench.py -- training rays/s (+ render Mpix/s) of the RoDynRF ray-batch hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config nvidia|nvidia_no_poses|davis] [--stage ...]

With --gpus N > 1 and no torch.distributed environment the script re-executes itself through
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` (one rank per GPU, RCCL = backend "nccl");
launched by the driver through torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.
--gpus must equal the world size (it fails loudly otherwise).

A "step" is one complete training iteration of the selected config (robust-dynrf_amd/step.py): for the default
``nvidia`` config (BASELINE.json configs[1]: Nvidia Balloon1, configs/Nvidia.txt, 4096 rays/iter, first
resolution stage 128^3 -> grid [141,157,94], 115 samples/ray) 5 dynamic + 5 static forward passes, scene-flow
MLP, induced flow / disparity, per-frame depth loss, distortion loss, the three-way compositor, TV
regularisers, one backward through all of it, the gradient exchange (N > 1) and Adam, on synthetic inputs that
are resident in HBM before the timed region.  For N > 1 every rank keeps 4096 rays (weak scaling): rays are
sharded, parameters replicated, the gradients go reduce-scatter -> Adam on the owned slice -> all-gather
(--dp zero1, default) or all-reduce (--dp allreduce).

Prints ONE JSON line (rank 0).  `value` = rays consumed by the whole job per second.  At N = 1 the same line
also carries: the kernel roofline (HIP-event timings of every launch + committed rocprofv3 PMC summaries), the
final-stage figure (grid [331,368,220], 270 samples/ray, where 78 % of the reference's iterations run), the
render legs (whole frame and the reference's 512-ray eval chunks), and the CPU baseline (the oracle's
re-enactment of the same step on the host cores)."""
import argparse
import ctypes as C
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOP per sample: f32 multiply-adds x 2
F_DYN_DENSITY = 26496 + 19584 + 19584      # warp + density head + blending head
F_DYN_APP = 11664 + 60946                  # basis + late-view head
F_STAT_APP = 72752                         # basis + MLP_Fea head
F_SCENE_FLOW = 21760
PEAK_F32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 16 * PEAK_F32_MFMA_TFLOPS   # same guide: the f32 MFMA rate is 1/16 of the bf16 rate (~2.5 PFLOP/s dense)
# Share of each kernel's algorithmic FLOP whose layers run as bf16 x 3 (csrc mfma_seg_b3 / _b3_pair / _b3s: three bf16 pieces per fp32
# value, SIX bf16 piece products per fp32 product), from the layer shapes: heads' first layers [72 features | X0 | X1[0..3]] of 152
# inputs; the appearance phases' two hidden layers (forward) and every backward-data layer but the 3-output one; the backward's
# first-layer blocks that carry a gradient.  A kernel's matrix-pipe floor is FLOP x ((1 - b) / f32 peak + 6 b / bf16 peak), i.e.
# (1 - 0.625 b) x its fp32 floor: `mfma_frac` (speed against the fp32 roof) x that factor = the physical pipe occupancy.
B3_FLOP_SHARE = {
    "dyn_density": 2 * (2 * 144 * 64) / F_DYN_DENSITY, "dyn_app": (2 * 107 * 128 + 2 * 128 * 128) / F_DYN_APP,
    "static_app": (2 * 138 * 128 + 2 * 128 * 128) / F_STAT_APP, "dyn_app_bwd": (F_DYN_APP - 2 * 131 * 3) / F_DYN_APP,
    "static_app_bwd": (F_STAT_APP - 2 * 128 * 3) / F_STAT_APP, "dyn_heads_bwd": (2 * 136 * 64) / 19584.0,
    "dyn_warp_bwd": (2 * 64 * 64 + 2 * 93 * 64) / 26496.0,
}


def pipe_factor(k):
    """matrix-pipe floor of kernel k / its fp32-only floor"""
    return 1.0 - (1.0 - 6.0 * PEAK_F32_MFMA_TFLOPS / PEAK_BF16_MFMA_TFLOPS) * B3_FLOP_SHARE.get(k, 0.0)


PEAK_HBM_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E peak (~6.3 TB/s achievable)
PEAK_L2_GBS = 34500.0                      # MI355X_MICROARCH.md: aggregate L2 bandwidth
L2_ATOMIC_REQ_PER_S = 20.8e9               # tools/ubench/atomics.hip: fp32 atomic requests the L2 retires
PROFILE_TAG = os.environ.get("RDRF_PROFILE_TAG", "r06")


def _profile_csv(name):
    for tag in (PROFILE_TAG, "r05", "r04", "r03", "r02", "r01"):
        fn = os.path.join(ROOT, "profiles", f"{tag}_{name}.csv")
        if os.path.exists(fn):
            return fn, tag
    return None, None


def pmc_value(csv_name, kernel, counter):
    """mean per dispatch of `counter` for `kernel` from a committed rocprofv3 summary (None if absent)"""
    fn, _ = _profile_csv(csv_name)
    if fn is None:
        return None
    got = None
    for line in open(fn).read().splitlines()[1:]:
        parts = line.split(",")
        if len(parts) >= 4 and parts[0] == kernel and parts[1] == counter:
            got = float(parts[3])
    return got


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/profile.sh -> profiles/rNN_pmc_{fetch,write}.csv; FETCH_SIZE / WRITE_SIZE are KB, and on gfx950
    FETCH_SIZE under-reports 16 B/lane streaming reads by 2x: MI355X_MICROARCH.md, HBM)."""
    f, w = pmc_value("pmc_fetch", kernel, "FETCH_SIZE"), pmc_value("pmc_write", kernel, "WRITE_SIZE")
    if f is None or w is None:
        return None
    return f * 1024.0 * 2.0 + w * 1024.0


def pmc_family_per_step(csv_name, prefixes, counter):
    """sum over the kernels whose name starts with one of `prefixes` of `counter` (mean x dispatches), per training
    step of the profiled command (steps = dispatches of k_adam / 2: one launch per field and step); None if absent"""
    fn, _ = _profile_csv(csv_name)
    if fn is None:
        return None
    tot, adam = 0.0, 0
    for line in open(fn).read().splitlines()[1:]:
        parts = line.split(",")
        if len(parts) < 4 or parts[1] != counter:
            continue
        if parts[0] == "k_adam":
            adam = int(float(parts[2]))
        if parts[0].startswith(prefixes):
            tot += float(parts[3]) * float(parts[2])
    return tot / (adam / 2.0) if adam else None


def committed_hbm_roofline(prefix, label):
    """The dominant kernel of a configuration whose factors exceed the caches, from the committed rocprofv3 passes of
    tools/profile_r6.sh (profiles/r06_<prefix>kernel_stats.csv + pmc_fetch / pmc_write): time per step from the kernel stats,
    HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE of the same command -- a TRUE HBM fraction.  Not re-measured
    by this run: `source` says so.  None if the files are absent."""
    import csv
    base = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{prefix}")
    try:
        stats, pmc = {}, {}
        with open(base + "kernel_stats.csv") as f:
            for r in csv.DictReader(f):
                k = r["Name"].split("(")[0].replace("void ", "").replace(",", ";").strip()
                c, t = stats.get(k, (0, 0.0))
                stats[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
        for fn, ctr in (("pmc_fetch.csv", "FETCH_SIZE"), ("pmc_write.csv", "WRITE_SIZE")):
            with open(base + fn) as f:
                for r in csv.DictReader(f):
                    if r["counter"] == ctr:
                        pmc.setdefault(r["kernel"].replace("void ", "").strip(), {})[ctr] = (float(r["dispatches"]), float(r["total"]))
    except OSError:
        return None
    steps = stats.get("k_adam", (0, 0))[0] / 2.0
    psteps = pmc.get("k_adam", {}).get("FETCH_SIZE", (0, 0))[0] / 2.0
    if not steps or not psteps:
        return None
    rows = []
    for k, (calls, ns) in stats.items():
        f, w = pmc.get(k, {}).get("FETCH_SIZE"), pmc.get(k, {}).get("WRITE_SIZE")
        if k.startswith("k_") and f and w:
            ms, b = ns / 1e6 / steps, (2.0 * f[1] + w[1]) * 1024.0 / psteps
            rows.append((ms, k, calls / steps, b))
    if not rows:
        return None
    rows.sort(reverse=True)
    ms, k, n, b = rows[0]
    tot_ms, tot_b = sum(r[0] for r in rows), sum(r[3] for r in rows)
    return {"workload": label, "kernel": k, "bound": "hbm", "achieved": b / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": b / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": b / n, "ms_per_step": ms, "launches_per_step": n,
            "all_kernels_gb_per_step": tot_b / 1e9, "all_kernels_ms_per_step": tot_ms,
            "all_kernels_frac": tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "source": f"profiles/{PROFILE_TAG}_{prefix}kernel_stats.csv / pmc_fetch.csv / pmc_write.csv + {PROFILE_TAG}_{prefix}hbm_table.txt "
                      "(committed rocprofv3 passes of tools/profile_r6.sh; not re-measured by this run)"}


def prof_get(L, name):
    ms, n = C.c_double(), C.c_int()
    L.lib.rdrf_prof_get(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


def cpu_baseline(trainer, n_rays, repeats, dead_work=True):
    """The oracle's re-enactment of the SAME step (oracle/rodynrf_oracle_step.py: torch-CPU restatement of
    the reference with its own grid_sample gather formulation) on `n_rays` rays on the host cores:
    kind "port".  Median of `repeats` timed steps after one warm-up on a quarter of the rays."""
    import torch
    from oracle import rodynrf_oracle as O
    from oracle import rodynrf_oracle_step as OS
    cfg = dict(trainer.cfg)
    sd_s = {k: v.detach().cpu().contiguous().clone() for k, v in trainer.st.state_dict().items()}
    sd_d = {k: v.detach().cpu().contiguous().clone() for k, v in trainer.dy.state_dict().items()}
    poses = trainer.pose_table().detach().cpu()
    foc = trainer.fov.detach().cpu() if trainer.optimize_poses else float(trainer.data.focal)

    def one(n):
        b = {k: v.cpu() for k, v in trainer.data.make_batch(0, n).items()}
        t0 = time.perf_counter()
        OS.step_gradients(cfg, sd_s, sd_d, b, poses, foc, 1000, OS.FixedRng(0), dead_work=dead_work)
        return time.perf_counter() - t0

    O.USE_GRID_SAMPLE = True   # the reference's own gather formulation (validated against the index-based one
    try:                       # in tests/test_oracle_golden.py): a fair CPU timing
        one(max(32, n_rays // (4 if repeats > 1 else 8)))
        times = sorted(one(n_rays) for _ in range(repeats))
    finally:
        O.USE_GRID_SAMPLE = False
    med = times[len(times) // 2]
    return dict(value=n_rays / med, unit="rays/s", cores=torch.get_num_threads(), kind="port", batch_rays=n_rays,
                sample=f"oracle/rodynrf_oracle_step.py (torch-CPU restatement of the reference, grid_sample gathers): the "
                       f"same {cfg['name']} step on {n_rays} rays x {cfg['n_samples']} samples, median of {repeats} after "
                       f"1 warm-up, {med:.1f} s per step (the CPU throughput grows with the batch)")


def quoted_baseline_config(world):
    """the BASELINE.json configs index quoted for this GPU count (configs[3]: 4 GPUs, configs[4]: 8 GPUs, else configs[1])"""
    return {4: "3", 8: "4"}.get(world, "1")


def select_baseline_config(args, world):
    """Which BASELINE.json configuration the line is for; fills args.config / args.stage /
    args.rays_per_gpu when they were not given and returns (configs index as a string or None, explicit?).  Default: configs[1]
    at every GPU count (the metric is quoted on Balloon1 @1/2/4/8); --baseline-config quoted: the one quoted for this count.
    configs[1]: Balloon1, Nvidia.txt, 4096 rays per GPU, stage 0 (N = 1, 2 and any N without an entry of its own);
    configs[2]: Nvidia_no_poses.txt; configs[3]: DAVIS.txt, final grid 256^3, 8192 rays GLOBAL (quoted on 4 GPUs: 2048 per
    rank); configs[4]: the 640^3 grid, 32768 rays global (quoted on 8 GPUs: 4096 per rank).  Explicit --config / --stage win."""
    BASE = {"1": ("nvidia", "stage0", 0), "2": ("nvidia_no_poses", "stage0", 0), "3": ("davis", "final", 8192),
            "4": ("nvidia_no_poses", "final", 32768)}
    bsel = args.baseline_config
    if bsel == "auto":
        bsel = "1"
    elif bsel == "quoted":
        bsel = quoted_baseline_config(world)
    b_cfg, b_stage, b_global = BASE[bsel]
    explicit = args.config is not None or args.stage is not None
    if explicit:
        args.config, args.stage = args.config or "nvidia", args.stage or "stage0"
        bsel = {("nvidia", "stage0"): "1", ("nvidia_no_poses", "stage0"): "2", ("davis", "final"): "3",
                ("nvidia_no_poses", "final"): "4"}.get((args.config, args.stage))
    else:
        args.config, args.stage = b_cfg, b_stage
        if not args.rays_per_gpu and b_global:
            if b_global % world:
                raise SystemExit(f"bench.py: configs[{bsel}] quotes a GLOBAL batch of {b_global} rays, which {world} ranks cannot "
                                 "split evenly; give --rays-per-gpu (the line then carries baseline_config_index null)")
            args.rays_per_gpu = b_global // world   # the config's GLOBAL batch split over the ranks it is quoted on
        elif args.rays_per_gpu and b_global and args.rays_per_gpu * world != b_global:
            bsel = None                             # not the quoted global batch: do not label the line with the config
    return bsel, explicit


def self_spawn(args):
    """python bench.py --gpus N without a torch.distributed environment: one rank per GPU through
    torch.distributed.run (the driver's own launch line)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def mask_fractions(S_, trainer, n_rays=None):
    """(valid, app_mask_static, app_mask_dynamic) of the CURRENT weights on a fixed probe batch: the work of the appearance
    kernels, their scatters and dW products follows the app-mask fractions, and those move as the weights train (random
    initialiser: ~0.5 / ~0.74; after ~40 iterations ~0.46 / ~0.35)"""
    import torch
    cfg = trainer.cfg
    with torch.no_grad():
        ids = trainer.data.batch(0, n_rays or min(cfg["batch_size"], 4096), 0)
        rays = trainer.rays_for(ids).detach()
        ts = trainer.data.ts_of(ids)
        o_s, o_d, _, smp = S_.ray_pass(trainer.st, trainer.dy, rays, ts, cfg["n_samples"], cfg["ray_type"], S_.StepRng(),
                                       is_train=False)
        _, _, vmask = S_.sampleXYZ(trainer.dy, rays, cfg["n_samples"], ray_type=cfg["ray_type"], is_train=False)
        return (float(vmask.float().mean()), float((o_s[4] > 1e-4).float().mean()), float((o_d[4] > 1e-4).float().mean()))


def timed_steps(trainer, shard, steps, warmup, world, dev, fractions=None):
    """W untimed + K timed iterations; `fractions` (dict) receives the app-mask fractions at the first and after the last
    timed iteration (probe forwards outside the timed region)"""
    import torch
    import torch.distributed as dist
    loss = None
    for _ in range(warmup):
        trainer.step(shard)
        trainer.finish_step()
    if fractions is not None:
        fractions["start"] = mask_fractions(fractions["S_"], trainer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = trainer.step(shard)
        trainer.finish_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if fractions is not None:
        fractions["end"] = mask_fractions(fractions.pop("S_"), trainer)
    return dt / steps, loss


KERNELS = ["pack", "generate_rays", "generate_rays_bwd", "sample_ndc", "sample_contract", "sample_bwd", "static_density",
           "static_app", "time_branch", "dyn_density", "dyn_app", "composite", "scene_flow", "induce_flow",
           "induce_flow_bwd", "distloss", "distloss_bwd", "tv_fwd", "tv_bwd", "tv_grad", "dense_l1", "dense_l1_bwd", "composite_bwd",
           "dyn_app_bwd", "scatter_dyn_app", "dyn_heads_bwd", "scatter_dyn_density", "dyn_warp_bwd", "time_branch_bwd",
           "dw_dyn", "static_app_bwd", "scatter_static_app", "static_density_bwd", "scatter_static_density", "sort",
           "dw_static", "scene_flow_bwd", "dw_sf", "adam"]


def roofline(L, S_, trainer, cfg, shard, rays_per_gpu, ms, window=None):
    """per-kernel HIP-event timings of the profiled steps + the algorithmic byte / FLOP counts of DESIGN.md 6.
    window = (make_trainer, warmup, steps): the profiled steps are iterations warmup .. warmup + steps of a FRESH trainer
    (same seeds, same batches, same draws as the timed run), i.e. the SAME training states the timed region covered --
    the appearance work follows the app-mask fractions, which fall from ~0.74 to ~0.35 (dynamic field) over the first
    ~40 iterations from the reference initialiser, so kernel times taken at a later state cannot be set against
    `ms_per_step`.  Without `window`: 3 steps at the trainer's current state."""
    import torch
    own = None
    if window is not None:
        make_trainer, w_steps, NP = window
        own = trainer = make_trainer()
        for _ in range(w_steps):
            trainer.step(shard)
            trainer.finish_step()
    else:
        NP = 3
    fr0 = mask_fractions(S_, trainer, rays_per_gpu)
    torch.cuda.synchronize()
    L.lib.rdrf_prof_enable(1)
    L.lib.rdrf_prof_reset()
    S_.PASSES.clear()
    # the fractions are not monotonic over the first iterations (they rise to ~0.74 around iteration 10 before they
    # fall), so they are probed about 20 times inside the window (HIP-event recording off around the probe forwards;
    # the profiled pass is not timed by the wall clock) and averaged by the trapezoid rule
    every = max(1, NP // 20)
    probes = [fr0]
    for i in range(NP):
        trainer.step(shard)
        trainer.finish_step()
        if (i + 1) % every == 0 and i + 1 < NP:
            torch.cuda.synchronize()
            L.lib.rdrf_prof_enable(0)
            counted = dict(S_.PASSES)   # the probe's own ray-pass is not part of the step
            probes.append(mask_fractions(S_, trainer, rays_per_gpu))
            S_.PASSES.clear()
            S_.PASSES.update(counted)
            torch.cuda.synchronize()
            L.lib.rdrf_prof_enable(1)
    torch.cuda.synchronize()
    L.lib.rdrf_prof_enable(0)
    counted = dict(S_.PASSES)
    fr1 = mask_fractions(S_, trainer, rays_per_gpu)
    S_.PASSES.clear()
    S_.PASSES.update(counted)
    probes.append(fr1)
    # the window's mean fractions price the appearance work of the profiled steps
    valid_frac, f_s, f_d = ((0.5 * (pr[0] + pr[-1]) + sum(pr[1:-1])) / (len(pr) - 1) for pr in zip(*probes))
    # ray-passes per step by kind: the algorithmic counts below are per PASS (4096 rays x S samples); a batched launch
    # (step.ray_passes) covers several
    np_stat, np_stat_g = S_.PASSES["static"] / NP, S_.PASSES["static_grad"] / NP
    np_dyn, np_dyn_bwd = S_.PASSES["dynamic"] / NP, (S_.PASSES["dynamic"] - S_.PASSES["dynamic_dead"]) / NP
    ns = rays_per_gpu * cfg["n_samples"]
    flops = {
        "dyn_density": ns * F_DYN_DENSITY, "dyn_heads_bwd": ns * (19584 + 19584), "dyn_warp_bwd": ns * 26496,
        "dyn_app": ns * f_d * F_DYN_APP, "dyn_app_bwd": ns * f_d * F_DYN_APP,
        "static_app": ns * f_s * F_STAT_APP, "static_app_bwd": ns * f_s * F_STAT_APP,
        "dw_dyn": ns * (F_DYN_DENSITY + f_d * F_DYN_APP), "dw_static": ns * f_s * F_STAT_APP,
        "scene_flow": ns * F_SCENE_FLOW, "scene_flow_bwd": ns * F_SCENE_FLOW, "dw_sf": ns * F_SCENE_FLOW,
    }
    table, tot_ms = {}, 0.0
    for k in KERNELS:
        msk, n = prof_get(L, k)
        if n:
            table[k] = {"ms_per_step": msk / NP, "launches_per_step": n / NP, "avg_us": msk / n * 1e3}
            if k != "sort":   # the radix sort runs INSIDE the scatter_dyn_* ranges (csrc: sorted_scatter_prepare): listed, not added twice
                tot_ms += msk / NP
    # passes per step each kernel family processes (its launches unless step.ray_passes batches them)
    mult = {k: v["launches_per_step"] for k, v in table.items()}
    for k in ("static_density", "static_app"):
        if k in mult: mult[k] = np_stat
    for k in ("dyn_density", "dyn_app", "time_branch"):
        if k in mult: mult[k] = np_dyn
    for k in ("dyn_heads_bwd", "dyn_warp_bwd", "scatter_dyn_density", "time_branch_bwd", "dw_dyn"):
        if k in mult: mult[k] = np_dyn_bwd
    for k, v in table.items():
        v["passes_per_step"] = mult[k]
    # per-pass averages of the two kernels whose work follows head liveness (the blending head's 19 584 FLOP per
    # sample is differentiated in `n_both_` of the dynamic backward launches, the appearance dW belongs to pass A)
    nb_ = mult.get("dw_dyn", 0.0)
    n_both_ = min(nb_, 2.0 if trainer.it >= cfg.get("upsamp_list", [0, 0, 0, 1 << 30])[3] else 1.0)
    if nb_ > 0:
        flops["dyn_heads_bwd"] = ns * (19584 + 19584 * n_both_ / nb_)
        flops["dw_dyn"] = ns * (26496 + 19584 + 19584 * n_both_ / nb_ + f_d * F_DYN_APP / nb_)
    step_flops = sum(flops[k] * mult[k] for k in table if k in flops)
    # k_dw (weight-gradient GEMMs; launches dw_dyn / dw_static / dw_sf): streams the saved activation rows and
    # the d(pre-activation) rows of every 32-sample tile; algorithmic bytes = UNIQUE rows x 128 B
    t1 = rays_per_gpu * ((cfg["n_samples"] + 31) // 32)
    t3d, t3s = (ns * f_d + 31) // 32, (ns * f_s + 31) // 32
    # head liveness (csrc/rdrf_bwd.hip): the blending head is differentiated only in the passes whose loss reaches it --
    # pass A, and pass B once the late mask terms are on (iteration >= upsamp_list[3]); the other dynamic backward
    # launches stage 640 instead of 864 row blocks per density tile and scatter one factor set instead of two.
    # The appearance rows (960 per compacted tile) belong to pass A alone.
    n_dyn_bwd = mult.get("dw_dyn", 0.0)
    n_both = min(n_dyn_bwd, 2.0 if trainer.it >= cfg.get("upsamp_list", [0, 0, 0, 1 << 30])[3] else 1.0)
    dw_step_bytes = {"dw_dyn": ((n_both * 864 + (n_dyn_bwd - n_both) * 640) * t1 + 960 * t3d) * 128.0,
                     "dw_static": 864 * t3s * 128.0 * table.get("dw_static", {}).get("launches_per_step", 0.0),
                     "dw_sf": 480 * t1 * 128.0 * table.get("dw_sf", {}).get("launches_per_step", 0.0)}
    dw_keys = [k for k in dw_step_bytes if k in table]
    out, dw_entry, dw_ms = {}, None, 0.0
    if dw_keys:
        dw_launch = sum(table[k]["launches_per_step"] for k in dw_keys)
        dw_ms = sum(table[k]["ms_per_step"] for k in dw_keys)
        dw_b = sum(dw_step_bytes[k] for k in dw_keys)
        dw_f = sum(flops[k] * mult[k] for k in dw_keys)
        tr = pmc_traffic("k_dw3")
        dw_entry = {
            "bound": "hbm", "achieved": dw_b / (dw_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": dw_b / (dw_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": tr,
            "hbm_real": None if tr is None else tr / (dw_ms / dw_launch * 1e-3) / 1e9,
            "source": {"traffic / hbm_real": f"profiles/{_profile_csv('pmc_fetch')[1]}_pmc_*.csv (committed, not re-measured here)"},
            "kernel": "k_dw3 (dw_dyn + dw_static + dw_sf launches)", "ms_per_step": dw_ms, "kernel_avg_us": dw_ms / dw_launch * 1e3,
            "launches_per_step": dw_launch, "algorithmic_bytes_per_launch": dw_b / dw_launch,
            "mfma": {"achieved_tflops": dw_f / (dw_ms * 1e-3) / 1e12, "peak_tflops": PEAK_F32_MFMA_TFLOPS,
                     "frac": dw_f / (dw_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}
    # k_scatter (VM gather backward): per sample re-gather the taps (B), read-modify-write the same texels of
    # the gradient factors (2B), read the d(feature) row entries (4 B per component); B = 1728 B for a
    # 72-component family, 5184 B for appearance.  The bytes are L2 / Infinity-Cache resident, so `achieved` may
    # exceed the HBM `traffic`; the kernel is bound by the L2 atomic request rate.
    n_sc = mult.get("scatter_dyn_density", 0.0)
    sc_step_bytes = {"scatter_dyn_density": (n_both * 2 + (n_sc - min(n_sc, n_both)) * 1) * ns * valid_frac * (3 * 1728 + 288.0),
                     "scatter_dyn_app": ns * f_d * (3 * 5184 + 864.0) * table.get("scatter_dyn_app", {}).get("launches_per_step", 0.0),
                     "scatter_static_app": ns * f_s * (3 * 1728 + 288.0) * table.get("scatter_static_app", {}).get("launches_per_step", 0.0)}
    sc_keys = [k for k in sc_step_bytes if k in table]
    sc_ms = sum(table[k]["ms_per_step"] for k in sc_keys)
    sc_launch = sum(table[k]["launches_per_step"] for k in sc_keys)
    sc_b = sum(sc_step_bytes[k] for k in sc_keys)
    # PMC figures of the whole k_scatter family (ray-tile and sorted kernels, all factor sets) per step of the committed
    # profile, set against this run's family time
    fam = ("void k_scatter<", "void k_scatter_sorted<", "void k_scatter_tiled<")
    f_, w_ = pmc_family_per_step("pmc_fetch", fam, "FETCH_SIZE"), pmc_family_per_step("pmc_write", fam, "WRITE_SIZE")
    tr_step = None if (f_ is None or w_ is None) else f_ * 1024.0 * 2.0 + w_ * 1024.0
    atom = pmc_family_per_step("sq_counters", fam, "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum")
    # `achieved` / `frac` of the scatter family are PHYSICAL: HBM-side bytes of the family's launches
    # (2 FETCH_SIZE + WRITE_SIZE of the committed PMC passes of this command) / this run's family time.  The algorithmic byte
    # model above prices the unsorted atomic scatter of round 2; the sorted / tiled kernels merge same-cell updates in
    # registers and LDS windows, so it stays in the detail record as `nominal_*` and is not a fraction of anything.
    hbm_real = None if tr_step is None else tr_step / (sc_ms * 1e-3) / 1e9
    sc_entry = {
        "bound": "hbm", "achieved": hbm_real, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": None if hbm_real is None else hbm_real / PEAK_HBM_GBS, "traffic": None if tr_step is None else tr_step / sc_launch,
        "l2_atomic_frac": None if atom is None else atom / (sc_ms * 1e-3) / L2_ATOMIC_REQ_PER_S,
        "kernel": "k_scatter family (k_scatter, k_scatter_sorted, k_scatter_tiled + key generation and radix sort)",
        "ms_per_step": sc_ms, "kernel_avg_us": sc_ms / sc_launch * 1e3, "launches_per_step": sc_launch,
        "nominal_bytes_per_launch_round2_model": sc_b / sc_launch,
        "nominal_gbs_round2_model": sc_b / (sc_ms * 1e-3) / 1e9,
        "nominal_frac_of_l2": sc_b / (sc_ms * 1e-3) / 1e9 / PEAK_L2_GBS,
        "bound_physical": "latency: gather round trips + the in-register run reduction; no roof nearby (HBM frac and memory-side "
                          "atomic fraction both < 0.25): plane sums form in LDS windows (ds_add_f64), line sums in LDS doubles",
        "source": {"ms_per_step / kernel_avg_us": "HIP events, this run",
                   "achieved / frac / traffic / l2_atomic_frac": f"profiles/{_profile_csv('pmc_fetch')[1]}_*.csv (committed rocprofv3 PMC "
                                                                 "summaries of the same command) over this run's family time"}}
    # forward / backward MLP families against the fp32-MFMA peak (useful = algorithmic FLOP of the samples they ran)
    def mlp_family(name, keys):
        ks = [k for k in keys if k in table and k in flops]
        if not ks:
            return None
        ms_f = sum(table[k]["ms_per_step"] for k in ks)
        fl = sum(flops[k] * mult[k] for k in ks)
        return {"bound": "mfma", "achieved": fl / (ms_f * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / (ms_f * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": None, "kernel": name, "ms_per_step": ms_f,
                "pipe_frac": sum(flops[k] * mult[k] * pipe_factor(k) for k in ks) / (ms_f * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "members_ms_per_step": {k: round(table[k]["ms_per_step"], 4) for k in ks}}
    fwd_entry = mlp_family("forward MLP kernels (k_dyn_density, k_dyn_app, k_static_app, k_scene_flow)",
                           ("dyn_density", "dyn_app", "static_app", "scene_flow"))
    bwd_entry = mlp_family("backward-data MLP kernels (k_dyn_heads_bwd, k_dyn_warp_bwd, k_dyn_app_bwd, k_static_app_bwd, k_scene_flow_bwd)",
                           ("dyn_heads_bwd", "dyn_warp_bwd", "dyn_app_bwd", "static_app_bwd", "scene_flow_bwd"))
    fams = [e for e in (fwd_entry, bwd_entry, sc_entry, dw_entry if dw_keys else None) if e]
    fams.sort(key=lambda e: -e["ms_per_step"])
    dom_e, oth_e = dict(fams[0]), (fams[1] if len(fams) > 1 else None)
    dom_e["note"] = "the kernel family with the most time per step; every `frac` here is physical (<= 1)"
    # The line's own roofline is the STEP against the fp32-MFMA peak: the path is MLP-bound
    # once the gathers are cache resident (SURVEY 8d "which roofline"), `achieved` = algorithmic FLOP of the work the step
    # executes (dead work included only when it runs) / the timed step; filled in by price_step() with the timed ms.
    out = {"bound": "mfma", "achieved": None, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": None,
           "traffic": None, "kernel": "the whole training step (every MLP kernel's algorithmic FLOP / step time)"}
    f_all, w_all = pmc_family_per_step("pmc_fetch", ("",), "FETCH_SIZE"), pmc_family_per_step("pmc_write", ("",), "WRITE_SIZE")
    if f_all is not None and w_all is not None:   # HBM-side bytes of one whole step of the committed profile (all kernels)
        out["traffic"] = f_all * 1024.0 * 2.0 + w_all * 1024.0
        out["traffic_note"] = (f"2 FETCH_SIZE + WRITE_SIZE summed over every kernel of one step, profiles/{_profile_csv('pmc_fetch')[1]}_pmc_*.csv "
                               "(committed rocprofv3 PMC passes of the driver's command; the guide's gfx950 correction), per STEP")
    # what the training forward leaves in HBM for the backward (rows of 32 samples, csrc/rdrf_kernels.hpp namespace sv)
    out["saved_bytes_per_sample"] = {
        "dynamic_density_phase": L.lib.rdrf_saved_row_bytes(0), "dynamic_appearance_phase_per_masked_sample": L.lib.rdrf_saved_row_bytes(1),
        "static_appearance_phase_per_masked_sample": L.lib.rdrf_saved_row_bytes(2)}
    out.update({
        "dominant_kernel": dom_e,
        "second_kernel": oth_e,
        "families": fams,
        "forward_mlp_ms_per_step": fwd_entry["ms_per_step"] if fwd_entry else None,
        "dw_ms_per_step": dw_ms if dw_keys else None,
        "launches_per_step": sum(v["launches_per_step"] for k, v in table.items() if k != "sort"),
        "step_algorithmic_tflop": step_flops / 1e12,
        "step_frac_of_peak": step_flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
        "kernel_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in table.items()},
        "kernel_launches_per_step": {k: round(v["launches_per_step"], 2) for k, v in table.items()},
        "ray_passes_per_step": {"static": np_stat, "static_with_grad": np_stat_g, "dynamic": np_dyn,
                                "dynamic_with_grad": np_dyn_bwd},
        "sum_kernel_ms_per_step": tot_ms,
        # the scatter family as one figure: every scatter_* range (the dynamic field's ranges contain their key generation
        # and radix sort, reported once more as `sort` for reference)
        "scatter_family_ms_per_step": sum(v["ms_per_step"] for k, v in table.items() if k.startswith("scatter_")),
        "sort_ms_per_step_nested_in_scatter": table.get("sort", {}).get("ms_per_step", 0.0),
        "profiled_steps": NP,
        "profiled_window": ("iterations warmup .. warmup + steps of a fresh trainer: the states of the timed region"
                            if window is not None else "3 iterations at the trainer's current state"),
        "fractions": {"valid": valid_frac, "app_mask_dynamic": f_d, "app_mask_static": f_s,
                      "probes_in_window": len(probes), "max_in_window": {"app_mask_static": max(pr[1] for pr in probes), "app_mask_dynamic": max(pr[2] for pr in probes)},
                      "at_window_start": {"app_mask_static": fr0[1], "app_mask_dynamic": fr0[2]},
                      "at_window_end": {"app_mask_static": fr1[1], "app_mask_dynamic": fr1[2]}},
        # fp32-MFMA fraction of every MLP kernel of the profiled steps (algorithmic FLOP of the passes it ran / its HIP-event
        # time / 157.3 TFLOP/s): the physical figure of the kernels that are not atomic-bound
        "mfma_frac": {k: round(flops[k] * mult[k] / (table[k]["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                      for k in table if k in flops and table[k]["ms_per_step"] > 0},
        # the same kernels' matrix-pipe occupancy: their instruction mix's floor (fp32 MFMAs + six bf16 MFMAs per bf16 x 3 product) /
        # their time -- physical, <= 1 by construction, where `mfma_frac` is the speed against the fp32 roof
        "mfma_pipe_frac": {k: round(flops[k] * mult[k] * pipe_factor(k) / (table[k]["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                           for k in table if k in flops and table[k]["ms_per_step"] > 0},
        "bf16x3_flop_share": {k: round(v, 4) for k, v in B3_FLOP_SHARE.items()},
        "step_pipe_tflop": sum(flops[k] * mult[k] * pipe_factor(k) for k in table if k in flops) / 1e12,   # -> step_pipe_frac (price_step)
        "peak_note": ("`frac` / `mfma_frac` = algorithmic fp32 FLOP / time / the fp32-MFMA peak (157.3 TFLOP/s): a speed against the fp32 roof. "
                      "Layers executed as bf16 x 3 (fp32-grade: 3 bf16 pieces, 6 piece products on v_mfma_f32_32x32x16_bf16, 16 x the fp32 rate) cost "
                      "6/16 of their fp32 MFMA cycles, so a kernel may pass the fp32 roof; `mfma_pipe_frac` / `pipe_frac` / `step_pipe_frac` price "
                      "the executed instruction mix instead (the physical matrix-pipe occupancy)"),
        "pmc_profile": _profile_csv("pmc_fetch")[1],
    })
    # SURVEY.md section 8(d) canonical constants: per sample B_fwd(f) = 4032 + 6912 f bytes, F_fwd(f) = 66004 +
    # 145362 f FLOP (f = app-mask fraction averaged over both fields); one training ray-pass = S * (4 B_fwd,
    # 3 F_fwd); training ray = 5 ray-passes (Nvidia.txt).
    f_avg = 0.5 * (f_d + f_s)
    Sn = cfg["n_samples"]
    npass = 7 if cfg.get("optimize_poses") else 5
    F_ray = npass * Sn * 3 * (66004 + 145362 * f_avg)
    B_ray = npass * Sn * 4 * (4032 + 6912 * f_avg)
    t_meas = ms * 1e-3 / rays_per_gpu
    t_star = max(F_ray / (PEAK_F32_MFMA_TFLOPS * 1e12), B_ray / 8e12)
    out["survey_canonical"] = {
        "f_app": f_avg, "flop_per_training_ray": F_ray, "gather_bytes_per_training_ray": B_ray,
        "achieved": t_star / t_meas, "frac_flop": F_ray / (PEAK_F32_MFMA_TFLOPS * 1e12 * t_meas),
        "frac_bytes": B_ray / (8e12 * t_meas),
        "note": f"SURVEY 8(d) constants with {npass} ray-passes per training ray; branches without a loss are not "
                "differentiated (as in the reference's autograd), so this counts more work than is executed; "
                "step_frac_of_peak counts only what runs; gather bytes are L2/MALL-resident algorithmic bytes"}
    if own is not None:
        del own
    return out


def price_step(rf, ms, rays_per_gpu):
    """fill the fields of a roofline() result that depend on the timed region's ms/step (the profiled replay runs BEFORE the
    timed region: it doubles as the clock warm-up of the process, see main())"""
    rf["step_frac_of_peak"] = rf["step_algorithmic_tflop"] / (ms * 1e-3) / PEAK_F32_MFMA_TFLOPS
    rf["achieved"] = rf["step_algorithmic_tflop"] / (ms * 1e-3)
    rf["frac"] = rf["step_frac_of_peak"]
    if "step_pipe_tflop" in rf:   # matrix-pipe occupancy of the whole step: the executed instruction mix's floor / the timed step
        rf["step_pipe_frac"] = rf["step_pipe_tflop"] / (ms * 1e-3) / PEAK_F32_MFMA_TFLOPS
    sc = rf["survey_canonical"]
    t_meas = ms * 1e-3 / rays_per_gpu
    sc["frac_flop"] = sc["flop_per_training_ray"] / (PEAK_F32_MFMA_TFLOPS * 1e12 * t_meas)
    sc["frac_bytes"] = sc["gather_bytes_per_training_ray"] / (8e12 * t_meas)
    sc["achieved"] = max(sc["frac_flop"], sc["frac_bytes"])
    rf["ms_per_step_minus_kernel_sum"] = ms - rf["sum_kernel_ms_per_step"]
    return rf


RENDER_KERNELS = ["pack", "sample_ndc", "sample_contract", "static_density", "static_app", "time_branch", "dyn_density", "dyn_app",
                  "composite", "render_fused"]


def render_leg(L, R, S_, trainer, cfg, dev, chunk, frames=5, streams=1):
    """BASELINE.json's second metric: Mpix/s of the no-grad chunk loop of renderer.py:740-812 (+ its roofline: SURVEY 8(d)
    F_fwd / B_fwd per sample x samples / time, with the two fields' measured app-mask fractions of THIS frame)."""
    import torch
    H, W = cfg["H"], cfg["W"]
    ids = torch.arange(H * W, device=dev)
    with torch.no_grad():
        rays_f = trainer.rays_for(ids + 3 * H * W).detach()
    ts_f = trainer.data.ts_of(ids + 3 * H * W)

    def frame():
        R.render_chunks(trainer.st, trainer.dy, rays_f, ts_f, chunk, N_samples=cfg["n_samples"], ray_type=cfg["ray_type"],
                        streams=streams)
    frame()
    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        frame()
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / frames
    out = {"value": H * W / dtf / 1e6, "unit": "Mpix/s", "frame": [H, W], "chunk": chunk, "hip_streams": streams,
           "samples_per_ray": cfg["n_samples"], "ms_per_frame": dtf * 1e3}
    # ---- roofline of the frame: per-kernel HIP events of one more frame, the frame's own app-mask fractions
    with torch.no_grad():
        o_s, o_d, _, _ = S_.ray_pass(trainer.st, trainer.dy, rays_f, ts_f, cfg["n_samples"], cfg["ray_type"], S_.StepRng(),
                                     is_train=False)
        f_s, f_d = float((o_s[4] > 1e-4).float().mean()), float((o_d[4] > 1e-4).float().mean())
    torch.cuda.synchronize()
    L.lib.rdrf_prof_enable(1)
    L.lib.rdrf_prof_reset()
    frame()
    torch.cuda.synchronize()
    L.lib.rdrf_prof_enable(0)
    table = {}
    for k in RENDER_KERNELS:
        msk, n = prof_get(L, k)
        if n:
            table[k] = {"ms_per_frame": round(msk, 4), "launches": n}
    ns = H * W * cfg["n_samples"]
    flop = ns * (340.0 + F_DYN_DENSITY + f_s * F_STAT_APP + f_d * F_DYN_APP)          # executed, per field fraction
    gbytes = ns * (576.0 + 3456.0 + f_s * 1728.0 + f_d * 5184.0)                       # VM gather bytes (L2 resident)
    f_avg = 0.5 * (f_s + f_d)
    flop_survey, bytes_survey = ns * (66004.0 + 145362.0 * f_avg), ns * (4032.0 + 6912.0 * f_avg)   # SURVEY 8(d) constants
    kern_ms = sum(v["ms_per_frame"] for k, v in table.items() if k != "render_fused" or len(table) == 1)
    mfma, mfma_pipe = {}, {}
    for k, f in (("dyn_density", ns * F_DYN_DENSITY), ("dyn_app", ns * f_d * F_DYN_APP), ("static_app", ns * f_s * F_STAT_APP)):
        if k in table and table[k]["ms_per_frame"] > 0:
            mfma[k] = round(f / (table[k]["ms_per_frame"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            mfma_pipe[k] = round(mfma[k] * pipe_factor(k), 4)
    out["roofline"] = {
        "bound": "mfma", "achieved": flop / dtf / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": flop / dtf / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": None,
        "gather_bytes_nominal": {"achieved": gbytes / dtf / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": gbytes / dtf / 1e9 / PEAK_HBM_GBS,
                                 "note": "algorithmic VM gather bytes / time against the HBM peak: the bytes are L2 / MALL resident"},
        "survey_canonical": {"f_app": f_avg, "frac_flop": flop_survey / dtf / 1e12 / PEAK_F32_MFMA_TFLOPS,
                             "frac_bytes": bytes_survey / dtf / 1e9 / PEAK_HBM_GBS},
        "fractions": {"app_mask_static": f_s, "app_mask_dynamic": f_d},
        "kernel_ms_per_frame": {k: v["ms_per_frame"] for k, v in table.items()},
        "kernel_launches_per_frame": {k: v["launches"] for k, v in table.items()},
        "sum_kernel_ms_per_frame": kern_ms, "mfma_frac": mfma, "mfma_pipe_frac": mfma_pipe,
        "source": "HIP events of one frame of this run (rdrf_prof_*); ms_per_frame is the wall clock of the timed frames"}
    return out


LINE_LIMIT = 8192   # bytes: the driver parses the LAST stdout line; r05's 21.5 KB line came back `parsed: null`
SCHEMA = 6          # r06: compact line + bench_detail.json; every `frac` of the line is a physical fraction (<= 1)


def _num(x, nd=5):
    """floats rounded to `nd` significant digits (the line is for reading; bench_detail.json keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def compact_line(out):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline` and the promoted
    scalars -- a few KB.  Everything else (per-kernel tables, exchange plan, stage list, sparse leg, notes) is written to
    bench_detail.json beside this file (write_detail).  tests/test_bench_line_cpu.py asserts len < LINE_LIMIT."""
    cfg = out.get("config", {})
    line = {k: _num(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                      "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["schema"] = SCHEMA
    line["config"] = {k: cfg[k] for k in ("workload", "baseline_config_index", "config", "stage", "grid", "samples_per_ray",
                                          "global_batch", "rays_per_gpu", "parallelism", "ranks", "backend", "exchange_bytes_per_step", "final_loss",
                                          "timed_region", "dead_work")
                      if k in cfg}
    rf = out.get("roofline")
    if rf:
        r = {k: _num(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")}
        for name in ("dominant_kernel", "second_kernel"):
            if rf.get(name):
                r[name] = {k: _num(rf[name].get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                               "ms_per_step")}
        r["mfma_frac"] = {k: _num(v) for k, v in rf.get("mfma_frac", {}).items()
                          if k in ("dyn_density", "dyn_app", "static_app", "dw_dyn", "dw_static")}
        if rf.get("mfma_pipe_frac"):   # matrix-pipe occupancy of the executed instruction mix (bf16 x 3 layers priced at 6/16)
            r["mfma_pipe_frac"] = {k: _num(v) for k, v in rf["mfma_pipe_frac"].items() if k in ("dyn_density", "dyn_app", "static_app")}
            r["step_pipe_frac"] = _num(rf.get("step_pipe_frac"))
            r["peak_note"] = "frac = fp32 FLOP / time / fp32-MFMA peak; bf16 x 3 layers (fp32-grade) cost 6/16 of their fp32 MFMA cycles: *_pipe_frac price the executed mix"
        for k in ("sum_kernel_ms_per_step", "scatter_family_ms_per_step", "forward_mlp_ms_per_step", "dw_ms_per_step",
                  "launches_per_step", "pmc_profile"):
            if k in rf:
                r[k] = _num(rf[k])
        line["roofline"] = r
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: _num(out["cpu_baseline"][k]) for k in ("value", "unit", "cores", "kind", "sample")}
    if "cpu_baseline_4096" in out:
        line["cpu_baseline_4096_value"] = _num(out["cpu_baseline_4096"]["value"])
    for k in ("final_stage_value", "final_stage_ms_per_step", "final_stage_frac_of_fp32_mfma_peak", "schedule_weighted_value",
              "liveness_exploited_value", "render_mpix_per_s", "render_final_stage_mpix_per_s", "render_chunk512_mpix_per_s",
              "step_frac_of_fp32_mfma_peak", "s13_stage_eager_ms_per_step", "s13_stage_graph_ms_per_step", "hbm_roofline_640"):
        if k in out:
            line[k] = _num(out[k]) if not isinstance(out[k], dict) else {a: _num(b) for a, b in out[k].items()}
    line["detail"] = "bench_detail.json"
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:   # never hand the driver a line it cannot parse: shed the optional objects
        for k in ("hbm_roofline_640", "cpu_baseline_4096_value"):
            line.pop(k, None)
        line.get("roofline", {}).pop("second_kernel", None)
        line["config"] = {"workload": cfg.get("workload", "")[:300]}
        text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_detail(out):
    """the full record (what rounds 1-5 printed on the line) next to bench.py and, on a gpurun box, under gpurun_out/"""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=["nvidia", "nvidia_no_poses", "davis"],
                    help="default: the BASELINE.json config quoted for this GPU count (see --baseline-config)")
    ap.add_argument("--stage", default=None, choices=["stage0", "up1", "up2", "up3", "final", "huge"])
    ap.add_argument("--baseline-config", default="auto", choices=["auto", "quoted", "1", "2", "3", "4"],
                    help="BASELINE.json configs[i] to run when --config / --stage are not given.  auto (default): configs[1] at "
                         "every GPU count -- BASELINE.json's metric is 'Nvidia Balloon1 @1/2/4/8 MI355X', 4096 rays per GPU "
                         "(weak scaling: the driver's efficiency figures compare like with like).  quoted: the configuration "
                         "BASELINE.json quotes for THIS GPU count -- N = 4 -> configs[3] (DAVIS.txt, contracted rays, 8192 rays "
                         "GLOBAL = 2048 per rank, final grid 256^3); N = 8 -> configs[4] (640^3 grid, 32768 rays global = 4096 "
                         "per rank); other N -> configs[1].  The line names both (config.baseline_config_*)")
    ap.add_argument("--weights", default="dense", choices=["dense", "sparse"])
    ap.add_argument("--rays-per-gpu", type=int, default=0, help="default: the config's batch size (4096; DAVIS 8192)")
    ap.add_argument("--dp", default="zero1", choices=["zero1", "allreduce"],
                    help="N > 1: reduce-scatter -> sharded Adam -> all-gather, or all-reduce + replicated Adam")
    ap.add_argument("--dp-per-shard-stats", action="store_true",
                    help="N > 1: normalise the masked-mean / per-frame depth losses by each rank's own batch statistics "
                         "instead of all-reducing them (default: exact single-process statistics, SURVEY 8e)")
    ap.add_argument("--scatter", default="auto", choices=["auto", "ray", "sorted", "sorted_plain"],
                    help="density / blending scatter of the dynamic field (rdrf_set_scatter_mode); auto = sorted from 300 k "
                         "samples per launch")
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--no-final-stage", action="store_true")
    ap.add_argument("--no-sparse", action="store_true", help="skip the W-sparse (app mask ~0.10) training / render leg")
    ap.add_argument("--no-cpu-4096", action="store_true", help="skip the single 4096-ray CPU step (about one minute)")
    ap.add_argument("--render-chunk", type=int, default=0, help="rays per render call (default: a whole frame)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record (bench_detail.json's content) as the last line instead of the compact one: for "
                         "the tools/ab_*.sh scripts, which read per-kernel tables from the line -- never for the driver")
    ap.add_argument("--no-liveness-leg", action="store_true",
                    help="skip the secondary liveness_exploited leg (profiling runs: every iteration of the process is then the "
                         "same workload)")
    ap.add_argument("--graph", action="store_true",
                    help="Trainer(graph=True): batch gather, forward passes and both backward phases of an iteration captured once "
                         "as a HIP graph and replayed (single process; for the launch-bound S = 13 stages of Nvidia_no_poses.txt / "
                         "DAVIS.txt).  Implies --no-roofline: the per-kernel HIP events cannot be recorded inside a capture")
    ap.add_argument("--no-graph-leg", action="store_true",
                    help="skip the launch-bound-stage leg (nvidia_no_poses stage 0, eager against Trainer(graph=True))")
    ap.add_argument("--exploit-liveness", action="store_true",
                    help="skip the work whose results nothing consumes (SURVEY 3.1 liveness table): the dynamic-field forward of "
                         "passes E / P3 / P4 and the colours of both fields in passes B-D / P1-P4; by default it is executed like "
                         "the reference does")
    args = ap.parse_args()
    if args.graph:
        if args.gpus > 1:
            raise SystemExit("bench.py --graph is the single-process path")
        args.no_roofline = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    import torch
    import torch.distributed as dist
    P = importlib.import_module("robust-dynrf_amd.parallel")
    rank, local, world = P.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the torch.distributed world size is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = importlib.import_module("robust-dynrf_amd._lib")
    S_ = importlib.import_module("robust-dynrf_amd.step")
    R = importlib.import_module("robust-dynrf_amd.renderer")

    L.set_scatter_mode(args.scatter)
    bsel, explicit = select_baseline_config(args, world)
    cfg = S_.scene_config(args.config, args.stage)
    rpg = args.rays_per_gpu or cfg["batch_size"]
    cfg["batch_size"] = rpg * world   # weak scaling: fixed rays per GPU
    trainer = S_.Trainer(cfg, dev, weights=args.weights, dead_work=not args.exploit_liveness, dp_mode=args.dp,
                         dp_exact_stats=not args.dp_per_shard_stats, graph=args.graph)
    shard = (rank, world)

    def make_trainer():
        return S_.Trainer(dict(cfg), dev, weights=args.weights, dead_work=not args.exploit_liveness, dp_mode=args.dp,
                          dp_exact_stats=not args.dp_per_shard_stats, graph=args.graph)

    # Order: the profiled replay FIRST (a fresh trainer with the same seeds runs iterations 0 .. warmup + steps, the last
    # `steps` of them under HIP events), then the timed region on the main trainer.  The replay gives the kernel table of
    # exactly the iterations that are timed afterwards, and it is the process's clock warm-up: the first ~second of GPU
    # work of a fresh process runs the MFMA-bound kernels up to ~10 % slower than steady state (measured: a 20-step
    # window timed right at start-up showed ms_per_step - kernel sum = 2.2 ms, of which 1.3 ms was clock ramp -- the same
    # kernels profiled 4 s later were that much faster), which a training run of 100 000 iterations never sees.
    rf = None
    if not args.no_roofline:   # collective: every rank runs the profiled steps (rank 0 reports)
        rf = roofline(L, S_, trainer, cfg, shard, rpg, 1.0, window=(make_trainer, args.warmup, min(args.steps, 400)))
    frs = {"S_": S_}
    dt, loss = timed_steps(trainer, shard, args.steps, args.warmup, world, dev, fractions=frs)
    ms = dt * 1e3
    value = cfg["batch_size"] / dt
    loss_val = float(loss.item())
    npass = "7 dynamic + 9 static" if cfg["optimize_poses"] else "5 dynamic + 5 static"
    workloads = {
        "nvidia": "BASELINE.json configs[1]: Nvidia Balloon1, configs/Nvidia.txt",
        "nvidia_no_poses": ("BASELINE.json configs[4]: Nvidia, N_voxel_final = 640^3 (configs/Nvidia_no_poses.txt schedule), 32k rays/iter "
                            "over 8 GPUs" if bsel == "4" else
                            "BASELINE.json configs[2]: Nvidia, configs/Nvidia_no_poses.txt, joint pose + focal optimisation"),
        "davis": "BASELINE.json configs[3]: DAVIS, configs/DAVIS.txt, contracted rays, flow + depth supervision, 8192 rays/iter over 4 GPUs"}
    out = {
        "metric": "training rays/sec (Nvidia Balloon1, configs/Nvidia.txt, static+dynamic TensorVMSplit)"
                  if args.config == "nvidia" else f"training rays/sec ({args.config})",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{workloads[args.config]}, {rpg} rays/iter/GPU, 1xMI355X per rank, static+dynamic "
                               f"TensorVMSplit; one step = {npass} forward passes, scene-flow MLP, induced flow/disparity, "
                               "per-frame depth loss, distortion loss, compositor, factor regularisers, full backward, Adam",
                   "baseline_config_index": None if bsel is None else int(bsel),
                   "baseline_config_selected": "explicit --config / --stage" if explicit else f"--baseline-config {args.baseline_config} at {world} GPU(s)",
                   "baseline_config_quoted_for_this_gpu_count": {
                       "index": int(quoted_baseline_config(world)),
                       "command": f"python bench.py --gpus {world} --baseline-config quoted",
                       "note": "BASELINE.json's metric is Balloon1 (configs[1]) at 1/2/4/8 GPUs, which is what the default runs at every "
                               "GPU count (weak scaling, 4096 rays per GPU); configs[3] (DAVIS, 8192 rays over 4 GPUs) and configs[4] "
                               "(640^3 grid, 32768 rays over 8 GPUs) are selected with --baseline-config quoted"},
                   "timed_region": ("preceded by a profiled REPLAY of the same window on a second trainer (iterations 0 .. warmup + steps, "
                                    "same seeds): it yields the per-kernel table of exactly the timed iterations and warms the clocks; "
                                    "then the driver's warmup + steps iterations run on the main trainer and `steps` of them are timed"
                                    if not args.no_roofline else "warmup + steps iterations on a fresh trainer"),
                   "config": args.config, "stage": args.stage, "grid": cfg["grid"], "samples_per_ray": cfg["n_samples"],
                   "global_batch": cfg["batch_size"], "rays_per_gpu": rpg, "weights": args.weights, "step_graph": bool(args.graph),
                   "parallelism": f"ray-sharded dp{world}" + (f" ({trainer.opt.mode})" if trainer.opt.ex.active else ""),
                   "ranks": world, "backend": dist.get_backend() if dist.is_initialized() else None,
                   "exchange_bytes_per_step": trainer.opt.nbytes_exchanged() if trainer.opt.ex.active else 0,
                   # what one iteration puts on RCCL (issued only for N > 1): the two flat gradient buffers (static first,
                   # asynchronously under the dynamic backward), the loss-statistics all-reduces and the per-ray depth gathers
                   "exchange_plan": {
                       "mode": trainer.opt.mode, "ranks": world,
                       "collectives": trainer.opt.ex.plan(["static_field", "dynamic_field"]) + (
                           [dict(collective="all_reduce", buffer="loss statistics (sum w rho, sum w) per loss group", bytes=2 * 32 * 2 * 4, count=2),
                            dict(collective="all_gather_into_tensor", buffer="per-ray depth / target / frame / mask of the per-frame depth loss",
                                 bytes=rpg * world * (4 + 4 + 8 + 1), count=2 if cfg["optimize_poses"] else 1)]
                           if not args.dp_per_shard_stats else []),
                       "issued": bool(trainer.opt.ex.active)},
                   "loss_statistics": ("whole batch (all-reduced mask sums, gathered per-frame depth statistics)"
                                       if trainer._dp() is not None else "single process"),
                   "final_loss": loss_val,
                   "app_mask_fractions_over_the_timed_steps": {
                       "first": {"static": frs["start"][1], "dynamic": frs["start"][2]},
                       "last": {"static": frs["end"][1], "dynamic": frs["end"][2]},
                       "note": "the appearance kernels' work follows these; they fall as the reference-initialised weights "
                               "train, so a 20-step window right after start-up times a heavier step than a 200-step one"},
                   "dead_work": "skipped (SURVEY 3.1: pass E's dynamic forward, the colours of passes B-D)" if args.exploit_liveness
                   else "executed (work the reference also computes although nothing consumes it)"},
    }

    if not args.exploit_liveness and not args.no_liveness_leg:
        # secondary figure: the same step without the dead dynamic forwards (identical results)
        trainer.dead_work = False
        dt2, _ = timed_steps(trainer, shard, max(10, args.steps // 4), 2, world, dev)
        out["liveness_exploited"] = {"value": cfg["batch_size"] / dt2, "unit": "rays/s", "ms_per_step": dt2 * 1e3,
                                     "note": "the same iterations without the work whose results nothing consumes (SURVEY 3.1): the "
                                             "dynamic forwards of pass E (P3 / P4) and the appearance phase (colours) of both fields in "
                                             "passes B-D (P1-P4) -- Trainer(dead_work=False), forward(rgb=False); losses and gradients "
                                             "identical (tests/test_gpu_trainer.py::test_dead_work_pruning_changes_nothing)"}
        trainer.dead_work = True
    if rf is not None and rank == 0:
        out["roofline"] = price_step(rf, ms, rpg)
    if rank == 0 and world == 1 and not args.no_render:
        # BASELINE.json's second metric: render Mpix/s through the no-grad chunk loop of renderer.py:740-812 --
        # whole frames per call, and the reference's own eval chunk of 512 rays (renderer.py:732)
        H, W = cfg["H"], cfg["W"]
        out["render"] = render_leg(L, R, S_, trainer, cfg, dev, args.render_chunk or H * W)
        # the reference's eval chunk (renderer.py:732): sequential on one stream, and the same chunks issued round-robin on
        # eight HIP streams (independent chunks; the MLP kernels of one chunk hold one workgroup per CU on a part of the
        # chip, the small kernels of the other chunks run beside them)
        out["render_chunk512"] = render_leg(L, R, S_, trainer, cfg, dev, 512, frames=4, streams=8)
        out["render_chunk512_one_stream"] = render_leg(L, R, S_, trainer, cfg, dev, 512, frames=2, streams=1)
    if rank == 0 and world == 1 and not args.no_final_stage and args.config == "nvidia" and args.stage == "stage0":
        # 78 % of the reference's iterations run after the last upsampling (configs/Nvidia.txt: upsamp_list[-1] =
        # 22000 of 100000): the same step at the final resolution
        cfg_f = S_.scene_config("nvidia", "final")
        cfg_f["batch_size"] = rpg
        def make_trainer_f():
            return S_.Trainer(dict(cfg_f), dev, weights=args.weights, dead_work=not args.exploit_liveness)
        n_f, w_f = max(10, args.steps // 5), 3
        rf_f = None
        if not args.no_roofline:   # profiled replay of iterations w_f .. w_f + n_f FIRST (as at stage 0): the kernel table
            rf_f = roofline(L, S_, None, cfg_f, shard, rpg, 1.0, window=(make_trainer_f, w_f, n_f))   # of the timed iterations
        tr_f = make_trainer_f()
        dtf, _ = timed_steps(tr_f, shard, n_f, w_f, 1, dev)
        fin = {"value": rpg / dtf, "unit": "rays/s", "ms_per_step": dtf * 1e3, "grid": cfg_f["grid"],
               "samples_per_ray": cfg_f["n_samples"], "steps": n_f, "warmup": w_f}
        if rf_f is not None:
            rf_f = price_step(rf_f, dtf * 1e3, rpg)
            fin["roofline"] = {k: rf_f[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "step_frac_of_peak",
                                                    "kernel_ms_per_step", "sum_kernel_ms_per_step", "ms_per_step_minus_kernel_sum",
                                                    "fractions", "mfma_frac", "mfma_pipe_frac", "step_pipe_frac", "families", "profiled_window")
                               if k in rf_f}
            fin["roofline"]["dominant_kernel"] = {k: rf_f["dominant_kernel"].get(k) for k in ("kernel", "bound", "achieved", "peak",
                                                                                             "unit", "frac", "ms_per_step")}
        if not args.no_render:
            fin["render"] = render_leg(L, R, S_, tr_f, cfg_f, dev, cfg_f["H"] * cfg_f["W"], frames=3)
        out["final_stage"] = fin
        del tr_f
        # time-weighted figure over the resolution schedule (configs/Nvidia.txt:14,19, train.py:937-947, 2582-2588): the
        # three intermediate grids once each (short windows, no per-kernel table), every stage weighted by the share of the
        # n_iters iterations it runs
        per_stage = {"stage0": ms, "final": dtf * 1e3}
        shapes = {"stage0": (cfg["grid"], cfg["n_samples"]), "final": (cfg_f["grid"], cfg_f["n_samples"])}
        for stg in ("up1", "up2", "up3"):
            cfg_i = S_.scene_config("nvidia", stg)
            cfg_i["batch_size"] = rpg
            tr_i = S_.Trainer(cfg_i, dev, weights=args.weights, dead_work=not args.exploit_liveness)
            dti, _ = timed_steps(tr_i, shard, 8, 3, 1, dev)
            per_stage[stg], shapes[stg] = dti * 1e3, (cfg_i["grid"], cfg_i["n_samples"])
            del tr_i
            torch.cuda.empty_cache()
        sched = S_.resolution_schedule("nvidia")
        n_it = float(sched[-1][2])
        mean_ms = sum(per_stage[st] * (b - a) / n_it for st, a, b in sched)
        out["schedule_weighted"] = {
            "value": rpg / (mean_ms * 1e-3), "unit": "rays/s", "mean_ms_per_step": mean_ms,
            "stages": [{"stage": st, "iterations": [a, b], "share": (b - a) / n_it, "grid": shapes[st][0],
                        "samples_per_ray": shapes[st][1], "ms_per_step": per_stage[st]} for st, a, b in sched],
            "note": "rays of a whole 100000-iteration Nvidia.txt run / its time: every stage of the resolution schedule measured "
                    "at its own grid and sample count from reference-initialised weights (stage 0 = the headline window; up1-up3 "
                    "8 timed steps after 3; final = the final_stage leg), weighted by its share of the iterations"}
    if (rank == 0 and world == 1 and not args.no_graph_leg and not args.graph and not trainer.opt.ex.active
            and args.config == "nvidia" and args.stage == "stage0"):
        # the launch-bound coarse stage of configs[2] (configs/Nvidia_no_poses.txt: grid [17,19,11], S = 13, the first 2000
        # iterations; 7 dynamic + 9 static passes with pose / focal optimisation): one iteration enqueued launch by launch from
        # Python against the same iteration captured once as a HIP graph and replayed (Trainer(graph=True))
        cfg_c = S_.scene_config("nvidia_no_poses", "stage0")
        leg = {"config": "nvidia_no_poses", "stage": "stage0", "grid": cfg_c["grid"], "samples_per_ray": cfg_c["n_samples"],
               "rays": cfg_c["batch_size"], "steps": 30, "warmup": 6}
        for mode in ("eager", "graph"):
            tr_c = S_.Trainer(dict(cfg_c), dev, weights=args.weights, dead_work=not args.exploit_liveness, graph=mode == "graph")
            dtc, _ = timed_steps(tr_c, None, 30, 6, 1, dev)
            leg[mode + "_ms_per_step"] = dtc * 1e3
            leg[mode + "_rays_per_s"] = cfg_c["batch_size"] / dtc
            if mode == "graph":
                leg["graphs_captured"] = len(tr_c._graphs)
            del tr_c
            torch.cuda.empty_cache()
        out["launch_bound_stage"] = leg
    if rank == 0 and world == 1 and not args.no_sparse and args.weights == "dense":
        # SURVEY 8d W-sparse: the same step / render with trained-scene-like occupancy (app mask ~0.10 in both fields)
        tr_s = S_.Trainer(dict(cfg), dev, weights="sparse", dead_work=not args.exploit_liveness)
        dts, _ = timed_steps(tr_s, shard, max(10, args.steps // 5), 3, 1, dev)
        sp = {"weights": "sparse", "value": rpg / dts, "unit": "rays/s", "ms_per_step": dts * 1e3, "steps": max(10, args.steps // 5)}
        if not args.no_roofline:
            rf = roofline(L, S_, tr_s, cfg, shard, rpg, dts * 1e3)
            sp["fractions"] = rf["fractions"]
            sp["kernel_ms_per_step"] = rf["kernel_ms_per_step"]
        if not args.no_render:
            sp["render"] = render_leg(L, R, S_, tr_s, cfg, dev, cfg["H"] * cfg["W"], frames=3)
        out["sparse_weights"] = sp
        del tr_s
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(trainer, args.cpu_rays, args.cpu_repeats, dead_work=not args.exploit_liveness)
        if args.cpu_rays < 4096 and not args.no_cpu_4096:   # BASELINE.json configs[0]: "4k rays/iter, PyTorch-CPU": one timed step
            out["cpu_baseline_4096"] = cpu_baseline(trainer, 4096, 1, dead_work=not args.exploit_liveness)
            out["cpu_baseline_4096"]["note"] = "the configs[0] batch size; one timed step after a 512-ray warm-up"
    if rank == 0:
        # what a reader of the driver's `parsed` summary should see without opening nested keys
        if "final_stage" in out:
            out["final_stage_value"] = out["final_stage"]["value"]
            out["final_stage_ms_per_step"] = out["final_stage"]["ms_per_step"]
        if "schedule_weighted" in out:
            out["schedule_weighted_value"] = out["schedule_weighted"]["value"]
        if "liveness_exploited" in out:
            out["liveness_exploited_value"] = out["liveness_exploited"]["value"]
        if "render" in out:
            out["render_mpix_per_s"] = out["render"]["value"]
            out["render_chunk512_mpix_per_s"] = out["render_chunk512"]["value"]
        if "launch_bound_stage" in out:
            out["s13_stage_eager_ms_per_step"] = out["launch_bound_stage"]["eager_ms_per_step"]
            out["s13_stage_graph_ms_per_step"] = out["launch_bound_stage"]["graph_ms_per_step"]
        if "roofline" in out:
            out["step_frac_of_fp32_mfma_peak"] = out["roofline"]["frac"]
        if "final_stage" in out:
            if "roofline" in out["final_stage"]:
                out["final_stage_frac_of_fp32_mfma_peak"] = out["final_stage"]["roofline"]["frac"]
            if "render" in out["final_stage"]:
                out["render_final_stage_mpix_per_s"] = out["final_stage"]["render"]["value"]
        if world == 1:
            r640 = committed_hbm_roofline("640_", "BASELINE.json configs[4] shape: nvidia_no_poses final, grid [706,786,471], S=578, 4096 rays, 1 GPU "
                                                  "(421 MB of factors > the 256 MB Infinity Cache: the HBM roof is physical here)")
            rdav = committed_hbm_roofline("davis_final_", "BASELINE.json configs[3] shape: davis final, grid 256^3, S=221, 8192 rays, 1 GPU")
            if r640:
                out["hbm_roofline_640_detail"] = r640
                out["hbm_roofline_640"] = {k: r640[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "ms_per_step")}
            if rdav:
                out["hbm_roofline_davis_final_detail"] = rdav
        out["schema"] = SCHEMA
        write_detail(out)
        sys.stdout.flush()
        print(json.dumps(out) if args.full_line else compact_line(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
