#!/usr/bin/env python
"""bench.py -- training rays/s of the RoDynRF ray-batch hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one Nvidia.txt-shaped training iteration (robust-dynrf_amd/step.py): 4 dynamic + 5
static forward passes, scene-flow MLP, the three-way compositor, one backward through all of it,
the gradient exchange (N > 1) and one Adam step, on synthetic Balloon1-shaped inputs that are
resident in HBM before the timed region.  Workload at N = 1: BASELINE.json configs[1] (Nvidia
Balloon1, configs/Nvidia.txt, 4096 rays/iter, static + dynamic TensorVMSplit, first resolution
stage 128^3 -> grid [141,157,94], 115 samples/ray).  For N > 1 every rank keeps 4096 rays
(weak scaling): rays are sharded, parameters replicated, one flat gradient all-reduce over RCCL.

Prints ONE JSON line (rank 0).  `value` = rays consumed by the whole job per second.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

# algorithmic FLOP per sample: f32 multiply-adds x 2
F_DYN_DENSITY = 26496 + 19584 + 19584      # warp + density head + blending head
F_DYN_APP = 11664 + 60946                  # basis + late-view head
F_STAT_APP = 72752                         # basis + MLP_Fea head
F_SCENE_FLOW = 21760
PEAK_F32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E peak (~6.3 TB/s achievable)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/profile.sh -> profiles/r01_pmc_{fetch,write}.csv; FETCH_SIZE / WRITE_SIZE are KB, and on
    gfx950 FETCH_SIZE under-reports 16 B/lane streaming reads by 2x: MI355X_MICROARCH.md, HBM).
    None when the profile is absent."""
    import csv
    tot = 0.0
    for tag, ctr, corr in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
        fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r01_pmc_{tag}.csv")
        if not os.path.exists(fn):
            return None
        got = None
        for line in open(fn).read().splitlines()[1:]:
            parts = line.rsplit(",", 4)
            if len(parts) == 5 and parts[0] == kernel and parts[1] == ctr:
                got = float(parts[3])
        if got is None:
            return None
        tot += got * 1024.0 * corr
    return tot


def prof_get(L, name):
    ms, n = C.c_double(), C.c_int()
    L.lib.rdrf_prof_get(name.encode(), C.byref(ms), C.byref(n))
    return ms.value, n.value


def cpu_baseline(trainer, n_rays, S, seed=0, dead_work=True):
    """The oracle (oracle/rodynrf_oracle.py, torch-CPU restatement of the reference) running the
    SAME step structure on `n_rays` rays on the host cores: kind "port"."""
    from oracle import rodynrf_oracle as O
    st, dy = trainer.st, trainer.dy
    cfg = trainer.cfg
    sd_s = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in st.state_dict().items()}
    sd_d = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in dy.state_dict().items()}
    aabb = st.aabb.detach().cpu()
    base = dict(aabb=aabb, act="relu", density_shift=-10.0, distance_scale=25.0, weight_thres=1e-4, view_pe=0)
    cfg_s = dict(base, head="MLP_Fea", fea_pe=2)
    cfg_d = dict(base, head="MLP_Fea_late_view", fea_pe=0)
    d = trainer.data
    g = torch.Generator().manual_seed(seed)
    ids = d.perm[:n_rays].cpu()
    ids2 = d.perm[n_rays:2 * n_rays].cpu()
    ids3 = d.perm[2 * n_rays:3 * n_rays].cpu()
    poses, focal = d.poses.cpu(), float(d.focal)
    H, W, T = cfg["H"], cfg["W"], cfg["T"]
    ts_of = lambda i: (i // (H * W)).float() * (2.0 / (T - 1)) - 1.0
    rgb_t, disp_t, fg = d.rgb[ids.to(d.device)].cpu(), d.disp[ids.to(d.device)].cpu(), d.fgmask[ids.to(d.device)].cpu()

    def rp(rays, ts, static_grad, dynamic, white):
        jit = torch.rand(S, generator=g)
        xyz, z, valid = O.sampleXYZ(rays, aabb, cfg["near_far"], S, "ndc", jit)
        if static_grad:
            o_s = O.field_forward(sd_s, cfg_s, rays, ts, xyz, z, valid, "ndc", dynamic=False)
        else:
            with torch.no_grad():
                o_s = O.field_forward(sd_s, cfg_s, rays, ts, xyz, z, valid, "ndc", dynamic=False)
        if dynamic:
            o_d = O.field_forward(sd_d, cfg_d, rays, ts, xyz, z, valid, "ndc", dynamic=True)
            a = (o_d[6], o_d[7], o_d[9], o_d[2], o_d[8])
        else:
            o_d = None
            a = (torch.zeros_like(o_s[6]), torch.zeros_like(o_s[7]), o_s[9], torch.zeros_like(o_s[7]), z)
        outs = O.raw2outputs(o_s[6], o_s[7], a[0], a[1], a[2], a[3], a[4], rays, white, "ndc")
        return o_s, o_d, outs, xyz

    def step():
        rays = O.generate_rays(ids, poses, focal, H, W, ndc=True, near=1.0)
        ts = ts_of(ids)
        dt = 2.0 / (T - 1)
        _, oA, outA, xyzA = rp(rays, ts, False, True, True)
        loss = 3.0 * ((outA[0] - rgb_t) ** 2).mean() + ((outA[8] - rgb_t) ** 2).mean()
        loss = loss + 0.1 * (outA[12] - fg).abs().mean() + 0.04 * (outA[9] - disp_t).abs().mean()
        w_dist = 0.01 * 1e-5
        loss = loss + w_dist * O.eff_distloss(outA[11], oA[8].detach(), 1.0 / S)
        _, oB, outB, _ = rp(rays, ts_of(ids2), False, True, False)
        loss = loss + 0.01 * outB[12].mean() + 0.01 * (outB[9] - outB[5].detach()).abs().mean()
        loss = loss + w_dist * O.eff_distloss(outB[11], oB[8].detach(), 1.0 / S)
        sf_f, sf_b = O.scene_flow(sd_d, aabb, oA[3], ts)
        w_d = outA[11].detach()[..., None]
        loss = loss + 0.01 * (sf_f.abs() * w_d).mean() + 0.01 * (sf_b.abs() * w_d).mean()
        loss = loss + 0.01 * ((sf_f + sf_b) ** 2 * w_d).mean()
        col, row, view = O.ids2pixel(W, H, ids)
        grid = torch.stack([col.float() + 0.5, row.float() + 0.5], -1)
        c2w_all = O.pose_to_mtx(poses)
        dev = d.device
        disp_A = {}
        for sgn, sf, fl, mk in ((1, sf_f, d.flow_f, d.flow_mask_f), (-1, sf_b, d.flow_b, d.flow_mask_b)):
            flow_t, mask_t = fl[ids.to(dev)].cpu(), mk[ids.to(dev)].cpu()
            pose_n = c2w_all[(view + sgn).clamp(0, T - 1)]
            ind_flow, ind_disp = O.induce_flow(H, W, focal, pose_n, outA[11], oA[3] + sf, grid, rays.detach(), "ndc")
            loss = loss + 0.02 * ((ind_flow - flow_t).abs() * mask_t).sum() / (mask_t.sum() + 1e-8) / 2.0
            disp_A[sgn] = (ind_disp, mask_t, pose_n)
        for ids_n, sgn in ((ids2, 1), (ids3, -1)):
            rays_n = O.generate_rays(ids_n, poses, focal, H, W, ndc=True, near=1.0)
            _, oN, outN, xyzN = rp(rays_n, (ts + sgn * dt).clamp(-1, 1), False, True, True)
            ind_disp, mask_t, pose_n = disp_A[sgn]
            _, ind_disp_n = O.induce_flow(H, W, focal, pose_n, outN[11], oN[3], grid, rays_n, "ndc")
            loss = loss + 0.04 * ((ind_disp - ind_disp_n).abs() * mask_t).sum() / (mask_t.sum() + 1e-8)
            loss = loss + w_dist * O.eff_distloss(outN[11], oN[8].detach(), 1.0 / S)
        _, _, outE, _ = rp(rays, ts, True, dead_work, False)   # pass E: dynamic forward is dead work
        m = (1.0 - fg)[:, None]
        loss = loss + (((outE[4] - rgb_t) ** 2) * m).sum() / (m.sum() + 1e-8) / 3.0
        loss = loss + 0.04 * ((outE[5] - disp_t).abs() * m[:, 0]).mean()
        ps = list(sd_s.values()) + list(sd_d.values())
        torch.autograd.grad(loss, ps, allow_unused=True)
        # TV regularisers of the five factor families (value NaN as in the reference: gradient only)
        tvl = 0
        for sd, fams in ((sd_d, ("density", "blending", "app")), (sd_s, ("density", "app"))):
            for fam in fams:
                tvl = tvl + O.tv_family([sd[f"{fam}_plane.{i}"] for i in range(3)],
                                        [sd[f"{fam}_line.{i}"] for i in range(3)])
        torch.autograd.grad(tvl, ps, allow_unused=True)

    O.USE_GRID_SAMPLE = True   # the reference's own gather formulation (validated against the
    try:                       # index-based one in tests/test_oracle_golden.py): a fair CPU timing
        step()  # warm-up
        t0 = time.perf_counter()
        step()
        dtm = time.perf_counter() - t0
    finally:
        O.USE_GRID_SAMPLE = False
    return dict(value=n_rays / dtm, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/rodynrf_oracle.py (torch-CPU restatement of the reference, grid_sample gathers), the same "
                       f"5-pass step on {n_rays} rays x {S} samples, 1 step after 1 warm-up, "
                       f"{dtm:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--stage", default="stage0", choices=["stage0", "final", "huge"])
    ap.add_argument("--weights", default="dense", choices=["dense", "sparse"])
    ap.add_argument("--rays-per-gpu", type=int, default=4096)
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--render-chunk", type=int, default=32768, help="rays per render call (default: a whole 240x135 frame)")
    ap.add_argument("--exploit-liveness", action="store_true",
                    help="skip pass E's dynamic-field forward, dead work the reference computes "
                         "(SURVEY 3.1 liveness table); by default it is executed like the reference does")
    args = ap.parse_args()

    P = importlib.import_module("robust-dynrf_amd.parallel")
    rank, local, world = P.init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = importlib.import_module("robust-dynrf_amd._lib")
    S_ = importlib.import_module("robust-dynrf_amd.step")

    cfg = S_.balloon1_config(args.stage)
    cfg["batch_size"] = args.rays_per_gpu * world   # weak scaling: fixed rays per GPU
    trainer = S_.Trainer(cfg, dev, weights=args.weights, dead_work=not args.exploit_liveness)
    params = [p for g in trainer.opt.param_groups for p in g["params"]]
    bucket = P.GradBucket(params, flats=lambda: trainer.grad_flats)
    shard = (rank, world)

    def one_step():
        loss = trainer.step(shard)
        # every rank's loss is a mean over ITS rays (and the full TV term): the mean over ranks is the
        # gradient of the global-batch loss
        bucket.allreduce_(average=True)
        trainer.finish_step()
        return loss

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    value = cfg["batch_size"] / (ms * 1e-3)
    loss_val = float(loss.item())

    out = {
        "metric": "training rays/sec (Nvidia Balloon1, configs/Nvidia.txt, static+dynamic TensorVMSplit)",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: Nvidia Balloon1, configs/Nvidia.txt, "
                               f"{args.rays_per_gpu} rays/iter/GPU, 1xMI355X per rank, static+dynamic "
                               "TensorVMSplit; one step = 5 dynamic + 5 static forward passes, scene-flow "
                               "MLP, induced flow/disparity x4, distortion loss x4, compositor, TV regularisers, full backward, Adam",
                   "stage": args.stage, "grid": cfg["grid"], "samples_per_ray": cfg["n_samples"],
                   "global_batch": cfg["batch_size"], "weights": args.weights,
                   "parallelism": f"ray-sharded dp{world}", "final_loss": loss_val,
                   "pass_E_dynamic_forward": "skipped (dead work, SURVEY 3.1)" if args.exploit_liveness
                   else "executed (dead work the reference also computes)"},
    }

    if not args.exploit_liveness:
        # secondary figure: the same step without pass E's dead dynamic forward (identical results)
        trainer.dead_work = False
        one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt2 = (time.perf_counter() - t0) / max(3, args.steps // 2)
        if world > 1:
            tt = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = float(tt.item())
        out["liveness_exploited"] = {"value": cfg["batch_size"] / dt2, "unit": "rays/s", "ms_per_step": dt2 * 1e3,
                                     "note": "pass E's dynamic forward skipped (SURVEY 3.1: nothing consumes it)"}
        trainer.dead_work = True
    if rank == 0 and not args.no_roofline:
        # measured sample fractions (enter the algorithmic FLOP counts)
        with torch.no_grad():
            ids = trainer.data.batch(0, args.rays_per_gpu, 0)
            rays = trainer.rays_for(ids)
            ts = trainer.data.ts_of(ids)
            o_s, o_d, _, _ = S_.ray_pass(trainer.st, trainer.dy, rays, ts, cfg["n_samples"], cfg["ray_type"])
            _, _, vmask = S_.sampleXYZ(trainer.dy, rays, cfg["n_samples"], ray_type=cfg["ray_type"], is_train=True)
            valid_frac = float(vmask.float().mean())
            f_d = float((o_d[4] > 1e-4).float().mean())
            f_s = float((o_s[4] > 1e-4).float().mean())
        L.lib.rdrf_prof_enable(1)
        L.lib.rdrf_prof_reset()
        NP = 2
        for _ in range(NP):
            trainer.step(shard)
            trainer.finish_step()
        torch.cuda.synchronize()
        ns = args.rays_per_gpu * cfg["n_samples"]
        flops = {
            "dyn_density": ns * F_DYN_DENSITY, "dyn_heads_bwd": ns * (19584 + 19584), "dyn_warp_bwd": ns * 26496,
            "dyn_app": ns * f_d * F_DYN_APP, "dyn_app_bwd": ns * f_d * F_DYN_APP,
            "static_app": ns * f_s * F_STAT_APP, "static_app_bwd": ns * f_s * F_STAT_APP,
            "dw_dyn": ns * (F_DYN_DENSITY + f_d * F_DYN_APP), "dw_static": ns * f_s * F_STAT_APP,
            "scene_flow": ns * F_SCENE_FLOW, "scene_flow_bwd": ns * F_SCENE_FLOW, "dw_sf": ns * F_SCENE_FLOW,
        }
        table, tot_ms = {}, 0.0
        for k in ["pack", "generate_rays", "sample_ndc", "static_density", "static_app", "time_branch",
                  "dyn_density", "dyn_app", "composite", "scene_flow", "composite_bwd", "dyn_app_bwd",
                  "scatter_dyn_app", "dyn_heads_bwd", "scatter_dyn_density", "dyn_warp_bwd",
                  "time_branch_bwd", "dw_dyn", "static_app_bwd", "scatter_static_app",
                  "static_density_bwd", "scatter_static_density", "dw_static", "scene_flow_bwd", "dw_sf"]:
            msk, n = prof_get(L, k)
            if n:
                table[k] = {"ms_per_step": msk / NP, "launches_per_step": n / NP, "avg_us": msk / n * 1e3}
                tot_ms += msk / NP
        L.lib.rdrf_prof_enable(0)
        step_flops = sum(flops[k] * table[k]["launches_per_step"] for k in table if k in flops)
        # Dominant kernel = k_dw (weight-gradient GEMMs; launches dw_dyn / dw_static / dw_sf).  It
        # streams the saved activation rows and the d(pre-activation) rows of every 32-sample tile
        # exactly once per (out-block, in-group) item and is HBM-bound (ablation in DESIGN.md s9:
        # loads only 4.15 ms, MFMA only 3.34 ms, both 5.0 ms per step): its roofline is bytes.
        # Algorithmic bytes = UNIQUE rows the jobs read x 128 B (rows are [32 samples] fp32).
        t1 = args.rays_per_gpu * ((cfg["n_samples"] + 31) // 32)
        t3d, t3s = (ns * f_d + 31) // 32, (ns * f_s + 31) // 32
        dw_bytes = {"dw_dyn": (864 * t1 + 960 * t3d) * 128.0, "dw_static": 864 * t3s * 128.0,
                    "dw_sf": 480 * t1 * 128.0}
        dw_keys = [k for k in dw_bytes if k in table]
        dw_launch = sum(table[k]["launches_per_step"] for k in dw_keys)
        dw_ms = sum(table[k]["ms_per_step"] for k in dw_keys)
        dw_b = sum(dw_bytes[k] * table[k]["launches_per_step"] for k in dw_keys)
        dw_f = sum(flops[k] * table[k]["launches_per_step"] for k in dw_keys)
        dw_entry = {
            "bound": "hbm", "achieved": dw_b / (dw_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": dw_b / (dw_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": pmc_traffic("k_dw"),
            "kernel": "k_dw", "ms_per_step": dw_ms, "kernel_avg_us": dw_ms / dw_launch * 1e3,
            "launches_per_step": dw_launch, "algorithmic_bytes_per_launch": dw_b / dw_launch,
            "mfma": {"achieved_tflops": dw_f / (dw_ms * 1e-3) / 1e12, "peak_tflops": PEAK_F32_MFMA_TFLOPS,
                     "frac": dw_f / (dw_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}
        # k_scatter (VM gather backward): per sample it re-gathers the taps (B), read-modify-writes the
        # same texels of the gradient factors (2B) and reads the d(feature) row entries (4 B per
        # component): B = 1728 B for a 72-component family, 5184 B for the 216-component appearance.
        # These bytes are L2 / Infinity-Cache resident (9-17 MB of factors), so `achieved` may exceed
        # what HBM sees (`traffic`); the kernel is bound by the L2 atomic request rate.
        sc_bytes = {"scatter_dyn_density": ns * valid_frac * 2 * (3 * 1728 + 288.0),
                    "scatter_dyn_app": ns * f_d * (3 * 5184 + 864.0),
                    "scatter_static_app": ns * f_s * (3 * 1728 + 288.0)}
        sc_keys = [k for k in sc_bytes if k in table]
        sc_ms = sum(table[k]["ms_per_step"] for k in sc_keys)
        sc_launch = sum(table[k]["launches_per_step"] for k in sc_keys)
        sc_b = sum(sc_bytes[k] * table[k]["launches_per_step"] for k in sc_keys)
        sc_entry = {
            "bound": "hbm", "achieved": sc_b / (sc_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": sc_b / (sc_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": pmc_traffic("void k_scatter<4; 1; 9>"),
            "kernel": "k_scatter", "ms_per_step": sc_ms, "kernel_avg_us": sc_ms / sc_launch * 1e3,
            "launches_per_step": sc_launch, "algorithmic_bytes_per_launch": sc_b / sc_launch,
            "limiter": "L2 fp32-atomic request rate (~20 G requests/s, tools/ubench/atomics.hip); bytes are "
                       "cache-resident, traffic is the PMC figure of the density/blending launch"}
        dom_e, oth_e = (sc_entry, dw_entry) if sc_ms >= dw_ms else (dw_entry, sc_entry)
        out["roofline"] = dict(dom_e)
        out["roofline"].update({
            "second_kernel": oth_e,
            "step_algorithmic_tflop": step_flops / 1e12,
            "step_frac_of_peak": step_flops / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "kernel_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in table.items()},
            "sum_kernel_ms_per_step": tot_ms,
            "fractions": {"valid": valid_frac, "app_mask_dynamic": f_d, "app_mask_static": f_s},
        })
        # SURVEY.md section 8(d) canonical constants: per sample B_fwd(f) = 4032 + 6912 f bytes,
        # F_fwd(f) = 66004 + 145362 f FLOP (f = app-mask fraction averaged over both fields); one
        # training ray-pass = S * (4 B_fwd, 3 F_fwd); training ray = 5 ray-passes (Nvidia.txt).
        f_avg = 0.5 * (f_d + f_s)
        Sn = cfg["n_samples"]
        F_ray = 5 * Sn * 3 * (66004 + 145362 * f_avg)
        B_ray = 5 * Sn * 4 * (4032 + 6912 * f_avg)
        t_meas = ms * 1e-3 / args.rays_per_gpu
        t_star = max(F_ray / (PEAK_F32_MFMA_TFLOPS * 1e12), B_ray / 8e12)
        out["roofline"]["survey_canonical"] = {
            "f_app": f_avg, "flop_per_training_ray": F_ray, "gather_bytes_per_training_ray": B_ray,
            "achieved": t_star / t_meas, "frac_flop": F_ray / (PEAK_F32_MFMA_TFLOPS * 1e12 * t_meas),
            "frac_bytes": B_ray / (8e12 * t_meas),
            "note": "SURVEY 8(d) constants with 5 ray-passes per training ray (the harness runs 5 static "
                    "and 4 dynamic forwards: pass E's dynamic evaluation is dead work, SURVEY 3.1 "
                    "liveness table, and branches without a loss are not differentiated, so this "
                    "figure counts more work than is executed; step_frac_of_peak counts only what "
                    "runs); gather bytes are L2/MALL-resident algorithmic bytes, not HBM traffic"}
    if rank == 0 and not args.no_render:
        # secondary metric of BASELINE.json: render Mpix/s -- whole 240x135 frames through the
        # no-grad chunk loop of renderer.py:740-812 (one C-ABI call per chunk; default: the whole frame)
        R = importlib.import_module("robust-dynrf_amd.renderer")
        H, W = cfg["H"], cfg["W"]
        ids = torch.arange(H * W, device=dev)
        rays_f = trainer.rays_for(ids + 3 * H * W)
        ts_f = trainer.data.ts_of(ids + 3 * H * W)
        chunk = args.render_chunk

        def frame():
            for c0 in range(0, H * W, chunk):
                R.render_rays(trainer.st, trainer.dy, rays_f[c0:c0 + chunk], ts_f[c0:c0 + chunk],
                              N_samples=cfg["n_samples"], ray_type=cfg["ray_type"])
        frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        NF = 5
        for _ in range(NF):
            frame()
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / NF
        out["render"] = {"value": H * W / dtf / 1e6, "unit": "Mpix/s", "frame": [H, W],
                         "chunk": chunk, "samples_per_ray": cfg["n_samples"], "ms_per_frame": dtf * 1e3}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(trainer, args.cpu_rays, cfg["n_samples"],
                                           dead_work=not args.exploit_liveness)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()   # ranks > 0 wait here while rank 0 finishes its roofline / render legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
