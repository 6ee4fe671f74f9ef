"""Training / rendering step harness reproducing the pass structure of the reference's trainer for the hot
path (/root/reference/train.py:1032-2351).

Per iteration, on the same ``batch_size`` rays (all configs):

    pass A  (rays, t)          static (value only) + dynamic (grad) + composite   train.py:1092-1162
    pass B  (rays, t_rand)     same                                              train.py:1166-1246
    scene-flow MLP on pass A's sample points                                     train.py:1319
    pass C  (flow-displaced pixel in frame t+1, t + 2/(T-1))                     train.py:1433-1528
    pass D  (frame t-1, t - 2/(T-1))                                             train.py:1530-1625
    pass E  (rays WITH grad -> pose / focal)  static (grad) + dynamic (dead) + composite   train.py:1756-1823

and, with ``optimize_poses`` (configs/Nvidia_no_poses.txt, configs/DAVIS.txt), the static-only block
train.py:1895-2311:

    static induced flow of pass E's weights into frames t+-1 (poses / focal live)           :1895-1948
    pass P1 / P2   static field along the flow-displaced rays of frames t+-1 (grad)         :1951-2095
    per-frame median-normalised monocular depth loss of the static depth                    :2097-2121
    pass P3 / P4   static (grad) + dynamic (dead) + composite on the x+1 / y+1 pixel rays   :2123-2311

i.e. 5 dynamic + 5 static forwards (Nvidia.txt) or 7 dynamic + 9 static (no-poses / DAVIS), one backward,
Adam on both fields (+ pose and focal Adam).  Every dynamic-field loss sees detached poses, focal and static
outputs (train.py:1380-1625), every static-field loss is independent of the dynamic field, so the backward
runs in two phases -- static first -- and the data-parallel exchange of the static gradients overlaps the
dynamic backward (optim.FlatAdam.begin_exchange).

The dataset (RGB, RAFT flow, DPT disparity, masks) is synthetic: random tensors of the right shape resident
in HBM.  Losses keep every term that decides which hot-path outputs carry gradient; there is no
per-iteration host sync (``.item()``): losses and the per-frame medians stay on the device.
"""
import collections
import math
import os
import types

import torch
import torch.nn as nn

from .fields import TensorVMSplit, TensorVMSplit_TimeEmbedding
from .optim import FlatAdam
from .parallel import require_even_shards
from .ray_utils import gather_rows, generate_rays, ids2pixel, pose_to_mtx
from .regularizers import TVLoss
from .losses import LossTerms, frame_depth_loss
from .renderer import distloss_rays, induce_flow, raw2outputs, sampleXYZ

NDC_AABB = [[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]]


def scene_config(name="nvidia", stage="stage0"):
    """Synthetic scene constants shaped like the three shipped configs:
    nvidia           configs/Nvidia.txt          Balloon1: NDC, GT poses, TV 1.0, grid 128^3 -> 300^3
    nvidia_no_poses  configs/Nvidia_no_poses.txt NDC, pose + focal optimisation, grid 16^3 -> 640^3
    davis            configs/DAVIS.txt           contracted rays, aabb +-2, T = 50, 16^3 -> 256^3, L1 + TV"""
    if name == "nvidia":
        cfg = dict(aabb=NDC_AABB, near_far=[0.0, 1.0], T=12, H=135, W=240, ray_type="ndc", batch_size=4096,
                   static_head="MLP_Fea", optimize_poses=False, tv_density=1.0, tv_app=1.0, dist_static=0.0,
                   dist_dynamic=0.01, l1_weight=0.0)
        # up1..up3: the intermediate grids of the resolution schedule (train.py:937-947: N_voxel_list linear in log
        # space between N_voxel_init = 128^3 and N_voxel_final = 300^3, utils.py:58-65 N_to_reso / cal_n_samples)
        stages = {"stage0": ([141, 157, 94], 115), "up1": ([174, 194, 116], 142), "up2": ([216, 240, 144], 176),
                  "up3": ([267, 298, 178], 218), "final": ([331, 368, 220], 270), "huge": ([706, 786, 471], 578)}
    elif name == "nvidia_no_poses":
        cfg = dict(aabb=NDC_AABB, near_far=[0.0, 1.0], T=12, H=135, W=240, ray_type="ndc", batch_size=4096,
                   static_head="MLP_Fea", optimize_poses=True, tv_density=0.0, tv_app=0.0, dist_static=0.01,
                   dist_dynamic=0.01, l1_weight=0.0)
        stages = {"stage0": ([17, 19, 11], 13), "final": ([706, 786, 471], 578), "huge": ([706, 786, 471], 578)}
    elif name == "davis":
        # DAVIS 480p frames (854 x 480) at downsample_train = 2; 50 frames; contracted rays to far = 256
        cfg = dict(aabb=[[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]], near_far=[0.0, 256.0], T=50, H=240, W=427,
                   ray_type="contract", batch_size=8192, static_head="MLP_Fea_TimeEmbedding", optimize_poses=True,
                   tv_density=0.1, tv_app=0.01, dist_static=0.02, dist_dynamic=0.005, l1_weight=8e-5)
        stages = {"stage0": ([16, 16, 16], 13), "final": ([256, 256, 256], 221)}
    else:
        raise ValueError(name)
    if stage not in stages:
        raise ValueError(f"{name}: stage must be one of {sorted(stages)}")
    cfg["grid"], cfg["n_samples"] = stages[stage]
    cfg.update(name=name, stage=stage, monodepth_static=0.04, monodepth_dynamic=0.04, n_iters=100000,
               lr_decay_target_ratio=0.1, small_scene_flow_weight=0.1, smooth_scene_flow_weight=0.1,   # opt.py:96-105
               upsamp_list=[8000, 12000, 16000, 22000] if name == "nvidia" else [2000, 4000, 6000, 8000, 12000, 16000, 22000])
    # the iteration a run of this stage starts at (train.py:2582-2588: grids change at upsamp_list)
    cfg["start_iteration"] = 0 if stage == "stage0" else cfg["upsamp_list"][-1]
    if stage in ("up1", "up2", "up3"):
        cfg["start_iteration"] = cfg["upsamp_list"][int(stage[2]) - 1]
    cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * math.sqrt(3.0)
    return cfg


def resolution_schedule(name="nvidia"):
    """[(stage, first iteration, last iteration)] of a config's resolution schedule (train.py:2582-2588: the grids change
    at the iterations of upsamp_list; configs/Nvidia.txt:14,19): the share of the n_iters iterations each stage runs."""
    if name != "nvidia":
        raise ValueError("resolution_schedule: only the Nvidia.txt schedule is tabulated")
    cfg = scene_config(name, "stage0")
    edges = [0] + cfg["upsamp_list"] + [cfg["n_iters"]]
    return [(st, edges[i], edges[i + 1]) for i, st in enumerate(["stage0", "up1", "up2", "up3", "final"])]


def balloon1_config(stage="stage0"):
    """BASELINE.json configs[1] (kept name: the Nvidia.txt / Balloon1 shape)."""
    return scene_config("nvidia", stage)


def build_fields(cfg, device, seed=20211202):
    """Both fields exactly as train.py:874-922 builds them (static fea_pe = 2, dynamic fea_pe = 0)."""
    torch.manual_seed(seed)
    common = dict(density_n_comp=[16, 4, 4], appearance_n_comp=[48, 12, 12], app_dim=27,
                  near_far=cfg["near_far"], alphaMask_thres=1e-4, density_shift=-10,
                  distance_scale=25, pos_pe=6, view_pe=0, featureC=128, step_ratio=2.0,
                  fea2denseAct="relu")
    aabb = torch.tensor(cfg["aabb"], dtype=torch.float32)
    st = TensorVMSplit(aabb, cfg["grid"], cfg["T"], device, shadingMode=cfg.get("static_head", "MLP_Fea"),
                       fea_pe=2, **common)
    dy = TensorVMSplit_TimeEmbedding(aabb, cfg["grid"], cfg["T"], device,
                                     shadingMode="MLP_Fea_late_view", fea_pe=0, **common)
    return st, dy


def sparsify_(st, dy, cfg, device, target=0.10, n_probe=1024):
    """W-sparse variant: a deterministic rescale of the static density factors and a
    shift of the dynamic density head's output bias, each found by bisection so that the measured
    app_mask fraction lands near `target` (trained-scene-like; the reference initialiser gives
    0.5-0.8).  The measured fractions are reported by bench.py and enter the roofline counts."""
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, cfg["T"] * cfg["H"] * cfg["W"], (n_probe,), generator=g).to(device)
    poses = torch.zeros(cfg["T"], 9, device=device)
    poses[:, 0] = 1.0
    poses[:, 4] = 1.0
    rays = generate_rays(ids, poses, torch.tensor(cfg["focal"], device=device), cfg["H"], cfg["W"],
                         ndc=cfg["ray_type"] == "ndc", near=1.0)
    ts = (ids // (cfg["H"] * cfg["W"])).float() * (2.0 / (cfg["T"] - 1)) - 1.0

    def frac(field):
        with torch.no_grad():
            xyz, z, valid = sampleXYZ(dy, rays, cfg["n_samples"], ray_type=cfg["ray_type"], is_train=False)
            o = field(rays, ts, None, xyz, z, valid, ray_type=cfg["ray_type"])
            return float((o[4] > 1e-4).float().mean())

    with torch.no_grad():
        base = [p.detach().clone() for p in st.density_plane]
        lo, hi = 0.0, 6.0   # log2 of the scale: denser -> earlier saturation -> fewer live samples
        for _ in range(12):
            mid = 0.5 * (lo + hi)
            for p, b in zip(st.density_plane, base):
                p.copy_(b * (2.0 ** mid))
            if frac(st) > target:
                lo = mid
            else:
                hi = mid
        b0 = dy.density_layer2.bias.detach().clone()
        lo, hi = 0.0, 8.0
        for _ in range(12):
            mid = 0.5 * (lo + hi)
            dy.density_layer2.bias.copy_(b0 + mid)
            if frac(dy) > target:
                lo = mid
            else:
                hi = mid
    return frac(st), frac(dy)


class SyntheticScene:
    """Synthetic dataset tensors, resident on the device (train.py:826-846, 953-990 keep the real ones
    as flat (T*H*W, ...) tensors the same way)."""

    def __init__(self, cfg, device, seed=20211202):
        g = torch.Generator().manual_seed(seed)
        T, H, W = cfg["T"], cfg["H"], cfg["W"]
        self.cfg = cfg
        self.total = T * H * W
        poses = torch.zeros(T, 9)
        poses[:, 0] = 1.0
        poses[:, 4] = 1.0
        poses[:, 6] = torch.linspace(-0.05, 0.05, T)
        self.poses = poses.to(device)
        self.focal = torch.tensor(cfg["focal"], device=device)
        self.rgb = torch.rand(self.total, 3, generator=g).to(device)
        self.disp = torch.rand(self.total, generator=g).to(device)
        self.fgmask = (torch.rand(self.total, generator=g) < 0.2).float().to(device)
        # optical-flow supervision (train.py:1380-1413): targets in pixels + validity masks
        self.flow_f = (2.0 * torch.randn(self.total, 2, generator=g)).to(device)
        self.flow_b = (2.0 * torch.randn(self.total, 2, generator=g)).to(device)
        self.flow_mask_f = (torch.rand(self.total, 1, generator=g) < 0.8).float().to(device)
        self.flow_mask_b = (torch.rand(self.total, 1, generator=g) < 0.8).float().to(device)
        self.perm = torch.randperm(self.total, generator=g).to(device)
        self.device = device
        # per-pixel tables the loader keeps next to the colours (train.py:826-846 stores rays / grids per pixel
        # the same way): pixel centre (col + 0.5, row + 0.5), frame index, normalised time -- one gather each
        # per batch instead of a dozen integer kernels
        pix = torch.arange(self.total)
        col, row, view = pix % W, (pix // W) % H, pix // (W * H)
        self.grid_table = torch.stack([col.float() + 0.5, row.float() + 0.5], -1).to(device)
        # `allgrids` of the reference (train.py:974-978): the INTEGER pixel coordinates -- what induce_flow subtracts
        self.px_table = torch.stack([col.float(), row.float()], -1).to(device)
        self.view_table = view.to(device)
        self.ts_table = (view.float() * (2.0 / (T - 1)) - 1.0).to(device)
        # packed copies for make_batch: one gather per SHAPE class instead of one per tensor (11 launches -> 5);
        # row k of a packed gather is a contiguous tensor
        self._scalars = torch.stack([self.ts_table, self.disp, self.fgmask, self.flow_mask_f[:, 0], self.flow_mask_b[:, 0]])
        self._vec2 = torch.stack([self.flow_f, self.flow_b, self.grid_table, self.px_table])

    def batch(self, it, bs, which=0):
        off = ((it * 3 + which) * bs) % (self.total - bs)
        return self.perm[off: off + bs]

    def ts_of(self, ids):
        return self.ts_table[ids]

    def make_batch(self, it, bs, shard=None, ids=None):
        """everything one iteration reads, as a dict of device tensors (train.py:1043-1060).  ids = (ids, ids2): the two
        index draws come from the caller's (static) buffers -- a captured iteration gathers from whatever they hold"""
        ids, ids2 = (self.batch(it, bs, 0), self.batch(it, bs, 1)) if ids is None else ids
        if shard is not None:  # (rank, world): ray-sharded data parallelism
            r, w = shard
            lo, hi = r * bs // w, (r + 1) * bs // w
            ids, ids2 = ids[lo:hi], ids2[lo:hi]
        sc, v2 = self._scalars[:, ids], self._vec2[:, ids]
        return dict(ids=ids, ts=sc[0], ts_rand=self.ts_of(ids2), grid=v2[2], px=v2[3], view=self.view_table[ids], rgb=self.rgb[ids],
                    disp=sc[1], fg=sc[2], flow_f=v2[0], flow_b=v2[1], mask_f=sc[3][:, None], mask_b=sc[4][:, None])


SyntheticBalloon = SyntheticScene   # round-1 name


class StepRng:
    """The three random draws of an iteration that live OUTSIDE the kernels: the per-call sampling jitter
    (models/tensorBase.py:492, 530-545), the white-background coin of raw2outputs (renderer.py:269).
    Tests replace this object to replay fixed draws on the GPU and in the oracle."""

    def __init__(self, seed=7):
        self.gen = torch.Generator().manual_seed(seed)
        self._pool, self._cur = None, 0

    def coin(self):
        return bool(torch.rand(1, generator=self.gen).item() < 0.5)

    def _take(self, n, device):
        """n uniform [0,1) floats from a device pool refilled by ONE rand launch per ~16 passes (a pass draws one
        or two small vectors: a launch each otherwise); 64-float aligned slices, contiguous"""
        n_al = (n + 63) // 64 * 64
        if self._pool is None or self._pool.device != torch.device(device) or self._cur + n_al > self._pool.numel():
            self._pool, self._cur = torch.rand(max(16 * n_al, 8192), device=device), 0
        v = self._pool[self._cur: self._cur + n]
        self._cur += n_al
        return v

    def jitter(self, S, ray_type, device):
        if ray_type == "ndc":
            return self._take(S, device), None
        return self._take(S - S // 2 + 1, device), self._take(S // 2 + 1, device)


class GraphRng:
    """The draws of StepRng held in STATIC device memory, so that a captured iteration (Trainer(graph=True)) reads fresh
    ones at every replay: begin() refills the jitter pool with one launch and derives the white-background coins from
    its head with a second (round(u) of a uniform u is 0 / 1 with probability 1/2 each); jitter() / coin() then hand out
    fixed slices in call order -- the same slices at capture and at every replay, since the iteration's structure is
    fixed.  coin() returns a one-element fp32 DEVICE tensor (raw2outputs' `add_white_bg`, rdrf_composite_*'s white_dev).
    frozen = True keeps the current contents (tests replay fixed draws through the eager and the captured path)."""
    N_COINS = 16

    def __init__(self, device, pool_floats=1 << 16):
        self.pool = torch.rand(pool_floats, device=device)
        self.coins = torch.round(self.pool[: self.N_COINS])
        self.frozen = False
        self._cur, self._ncoin = 64, 0

    def begin(self):
        if not self.frozen:
            self.pool.uniform_()
            torch.round(self.pool[: self.N_COINS], out=self.coins)
        self._cur, self._ncoin = 64, 0

    def coin(self):
        if self._ncoin >= self.N_COINS:
            raise RuntimeError("GraphRng: more than N_COINS coins in one iteration")
        k = self._ncoin
        self._ncoin += 1
        return self.coins[k: k + 1]

    def _take(self, n, device):
        n_al = (n + 63) // 64 * 64
        if self._cur + n_al > self.pool.numel():
            raise RuntimeError("GraphRng: the jitter pool is too small for one iteration of this shape")
        v = self.pool[self._cur: self._cur + n]
        self._cur += n_al
        return v

    jitter = StepRng.jitter


# ray-passes evaluated since the last reset, by kind (bench.py prices its algorithmic byte / FLOP counts per PASS: a
# batched launch covers several): static, static_grad, dynamic, dynamic_dead (dead work of passes E / P3 / P4)
PASSES = collections.Counter()


def _rgb_or_zeros(rgb, like):
    """the colours of a pass whose field ran with rgb=False: zeros (no map that depends on them is consumed)"""
    return rgb if rgb is not None else torch.zeros(*like.shape, 3, device=like.device)


def ray_pass(st, dy, rays, ts, n_samples, ray_type, rng, is_train=True, static_grad=False, dynamic=True, rgb_s=True,
             rgb_d=True):
    """sampleXYZ -> static -> dynamic -> raw2outputs (one ray-pass).  The static field runs value-only unless
    `static_grad`; with dynamic=False (dead work of passes E / P3 / P4 skipped) zeros stand in for the dynamic
    outputs, which the static maps do not depend on; rgb_s / rgb_d = False: that field's colours are not consumed by
    any loss of this pass (passes B-D, P1-P4) and its appearance phase is not run."""
    jit, jit_o = rng.jitter(n_samples, ray_type, rays.device) if is_train else (None, None)
    xyz, z, valid = sampleXYZ(dy, rays, n_samples, ray_type=ray_type, is_train=is_train, jitter=jit,
                              jitter_outer=jit_o)
    PASSES.update(static=1, static_grad=int(static_grad), dynamic=int(dynamic), dynamic_dead=int(dynamic and static_grad))
    if static_grad:
        o_s = st(rays, ts, None, xyz, z, valid, is_train=is_train, ray_type=ray_type, rgb=rgb_s)
    else:
        with torch.no_grad():
            o_s = st(rays, ts, None, xyz, z, valid, is_train=is_train, ray_type=ray_type, rgb=rgb_s)
    sigma_s = o_s[7]
    rgb_s = _rgb_or_zeros(o_s[6], sigma_s)
    if dynamic:   # in passes E / P3 / P4 this evaluation is dead work the reference performs (autograd graph and all)
        o_d = dy(rays, ts, None, xyz, z, valid, is_train=is_train, ray_type=ray_type, rgb=rgb_d)
        rgb_d, sigma_d, dists, blending, zv = _rgb_or_zeros(o_d[6], o_d[7]), o_d[7], o_d[9], o_d[2], o_d[8]
    else:
        o_d = None
        rgb_d = torch.zeros_like(rgb_s)
        sigma_d = torch.zeros_like(sigma_s)
        blending = torch.zeros_like(sigma_s)
        dists, zv = o_s[9], z
    white = rng.coin() if is_train else False
    outs = raw2outputs(rgb_s, sigma_s, rgb_d, sigma_d, dists, blending, zv, rays, is_train=is_train,
                       ray_type=ray_type, add_white_bg=white)
    return o_s, o_d, outs, xyz


# samples one batched field call may cover: beyond a few million the per-call buffers (saved rows: ~6 KB per sample)
# reach tens of GB and the caching allocator's handling of such blocks costs more than the launches saved (640^3 grid,
# 3 x 4096 x 578 samples in one call: kernel time unchanged, 84 -> 141 ms/step); the passes are then batched in smaller
# groups
BATCH_MAX_SAMPLES = int(float(os.environ.get("RDRF_BATCH_MAX_SAMPLES", 6e6)))


def batch_groups(idxs, n_samples_per_pass):
    """consecutive pass indices `idxs` cut into groups of at most BATCH_MAX_SAMPLES samples"""
    per = max(1, BATCH_MAX_SAMPLES // max(1, n_samples_per_pass))
    idxs = list(idxs)
    return [idxs[i:i + per] for i in range(0, len(idxs), per)]


def batched_field(field, rays_list, ts_list, samples, idxs, ray_type, grad=True, rgb=True):
    """`field` over the concatenated rays / samples of the passes `idxs` in ONE call; returns the 10-tuple of each pass
    (views of the batched outputs: unbind, whose backward stacks the per-pass gradients)"""
    if len(idxs) == 1:
        k = idxs[0]
        with torch.set_grad_enabled(grad and torch.is_grad_enabled()):
            return [field(rays_list[k], ts_list[k], None, *samples[k], is_train=True, ray_type=ray_type, rgb=rgb)]
    N = rays_list[idxs[0]].shape[0]
    cat = lambda ts: torch.cat([ts[k] for k in idxs])
    with torch.set_grad_enabled(grad and torch.is_grad_enabled()):
        o = field(cat(rays_list), cat(ts_list), None, *(torch.cat([samples[k][i] for k in idxs]) for i in range(3)),
                  is_train=True, ray_type=ray_type, rgb=rgb)
    parts = [None if t is None else t.view(len(idxs), N, *t.shape[1:]).unbind(0) for t in o]
    return [tuple(None if pt is None else pt[j] for pt in parts) for j in range(len(idxs))]


def ray_passes(st, dy, rays_list, ts_list, n_samples, ray_type, rng, groups, rgb=None):
    """Several ray-passes whose inputs do not depend on each other's outputs (passes A-D of an iteration: the rays of
    C / D come from the data's optical flow, not from pass A), evaluated as ONE value-only static forward over all their
    rays and one dynamic forward per GROUP of passes (lists of consecutive pass indices with the same gradient
    liveness, so that a batched call prunes what each of its passes would).  Same arithmetic and the same draw order
    (jitter, coin per pass, in pass order) as one ray_pass per entry; the persistent MLP kernels see 3-4x the tiles per
    launch (tail quantisation: 2.4 -> 3 tile rounds per wave becomes 7.3 -> 8) and fill their LDS images once.
    Calls are capped at BATCH_MAX_SAMPLES samples.  rgb (list of bool per pass, default all True): passes whose colours no
    loss consumes run both fields without their appearance phase (the static call is then cut at the flag changes).
    Returns one (o_s, o_d, outs, xyz) per pass; the per-pass tensors are views (unbind) of the batched outputs."""
    P, N, dev = len(rays_list), rays_list[0].shape[0], rays_list[0].device
    PASSES.update(static=P, dynamic=P)
    samples, coins = [], []
    for rays in rays_list:
        jit, jit_o = rng.jitter(n_samples, ray_type, dev)
        samples.append(sampleXYZ(dy, rays, n_samples, ray_type=ray_type, is_train=True, jitter=jit, jitter_outer=jit_o))
        coins.append(rng.coin())
    ns = N * samples[0][1].shape[1]
    rgb = [True] * P if rgb is None else list(rgb)
    o_ss, o_ds = [None] * P, [None] * P
    runs, lo = [], 0   # maximal runs of passes with the same colour flag: one static call each
    for p in range(1, P + 1):
        if p == P or rgb[p] != rgb[lo]:
            runs.append(list(range(lo, p)))
            lo = p
    for run in runs:
        for g in batch_groups(run, ns):
            for k, o in zip(g, batched_field(st, rays_list, ts_list, samples, g, ray_type, grad=False, rgb=rgb[run[0]])):
                o_ss[k] = o
    for group in groups:
        assert list(group) == list(range(group[0], group[0] + len(group)))
        assert all(rgb[k] == rgb[group[0]] for k in group), "a dynamic group must agree on its colour flag"
        for g in batch_groups(group, ns):
            for k, o in zip(g, batched_field(dy, rays_list, ts_list, samples, g, ray_type, rgb=rgb[group[0]])):
                o_ds[k] = o
    out = []
    for p in range(P):
        o_sp, o_d = o_ss[p], o_ds[p]
        outs = raw2outputs(_rgb_or_zeros(o_sp[6], o_sp[7]), o_sp[7], _rgb_or_zeros(o_d[6], o_d[7]), o_d[7], o_d[9], o_d[2],
                           o_d[8], rays_list[p], is_train=True,
                           ray_type=ray_type, add_white_bg=coins[p])
        out.append((o_sp, o_d, outs, samples[p][0]))
    return out


def masked_mean(x, m):
    return (x * m).sum() / (m.sum() + 1e-8)


class Trainer:
    def __init__(self, cfg, device, weights="dense", lr_init=0.02, lr_basis=1e-3, dead_work=False,
                 dp_mode="allreduce", lr_pose=3e-3, dp_exact_stats=False, batch_passes=None, graph=False):
        """dead_work: also run what the reference computes although nothing consumes it (SURVEY.md 3.1 liveness table):
        the dynamic-field forward of passes E / P3 / P4, and the appearance phase (colours) of both fields in passes B-D
        and P1-P4, whose rgb maps no loss term reads; off = skipped, losses and gradients identical.
        dp_exact_stats (data-parallel runs): the batch statistics of the losses -- the mask sums of the masked means
        (train.py:1391-1394, 1522-1524, 1828-1832, 1277-1291) and the per-frame medians / deviations / ray counts of
        the monocular depth losses (train.py:797-807) -- are those of the WHOLE batch (one all-reduce of ~30 pairs of
        floats per loss group, one all-gather of the per-ray depths), so that an N-rank run optimises exactly the
        single-process objective; off: per-shard statistics.
        batch_passes (default on; RDRF_BATCH_PASSES=0): passes A-D share one static forward and passes of equal gradient
        liveness one dynamic forward / backward (ray_passes); off = one launch sequence per pass, same arithmetic.
        graph (single-process runs): batch gather, every forward pass and both backward phases of an iteration are captured
        ONCE per shape / gate combination as a HIP graph (torch.cuda.CUDAGraph) and replayed -- the coarse stages of
        Nvidia_no_poses.txt / DAVIS.txt (S = 13) spend ~9.5 ms per iteration enqueueing ~300 launches of a few
        microseconds each from Python.  What changes from iteration to iteration reaches the kernels through static
        device memory: ray indices, sampling jitter and white-background coins (GraphRng), the loss weights that ramp
        (distortion ramp it / n_iters, Temp_static; LossTerms' device coefficients).  TV, Adam and the pose / focal Adam
        stay outside the graph (their weights and rates change every iteration; ~10 launches)."""
        self.batch_passes = (os.environ.get("RDRF_BATCH_PASSES", "1") != "0") if batch_passes is None else bool(batch_passes)
        self.dead_work = dead_work
        self.dp_exact_stats = bool(dp_exact_stats)
        self.cfg = cfg
        self.device = device
        self.st, self.dy = build_fields(cfg, device)
        if weights == "sparse":
            sparsify_(self.st, self.dy, cfg, device)
        self.data = SyntheticScene(cfg, device)
        lr_factor = cfg.get("lr_decay_target_ratio", 0.1) ** (1.0 / cfg.get("n_iters", 100000))   # train.py:926-930
        self.opt = FlatAdam([self.st, self.dy], lr_init, lr_basis, betas=(0.9, 0.99), lr_factor=lr_factor,
                            mode=dp_mode)
        self.lr_factor = lr_factor
        self.optimize_poses = bool(cfg.get("optimize_poses", False))
        self.it = int(cfg.get("start_iteration", 0))
        if self.optimize_poses:   # train.py:972-1009: 6-D pose table + field of view, their own Adam
            self.poses = nn.Parameter(self.data.poses.clone())
            self.fov = nn.Parameter(torch.full((1,), 30.0 / 180.0 * math.pi, device=device))
            self.lr_pose = lr_pose
            ups = cfg.get("upsamp_list", [0, 0, 0, 0])
            # ExponentialLR of both (train.py:993-1009); the focal Adam starts at lr 0 and is switched on by the first
            # upsample at or after upsamp_list[3] (train.py:2589-2595)
            self.pose_gamma = (1e-5 / lr_pose) ** (1.0 / max(1, cfg.get("n_iters", 100000) // 2 - ups[-1]))
            self.opt_pose = torch.optim.Adam([self.poses], lr=lr_pose)
            self.opt_focal = torch.optim.Adam([self.fov], lr=lr_pose if self.it >= ups[3] else 0.0)
        self.rng = StepRng()
        self.graph = bool(graph)
        self._graphs, self._graph_seen, self._gs = {}, None, None
        self._dyn = None   # name -> one-element device tensor: the iteration scalars of a captured iteration
        self.tv = TVLoss()
        self.grad_flats = self.opt.grad_flats()
        self.last = {}
        self._c2w_fixed = None

    # ---- data-parallel loss statistics ---------------------------------------------------------------
    def _dp(self):
        """(group, rank, world) when the exact-statistics exchange is on, else None"""
        ex = self.opt.ex
        return (self.opt._group, ex.rank, ex.world) if (self.dp_exact_stats and ex.active) else None

    def _loss_terms(self):
        dp = self._dp()
        if dp is None:
            return LossTerms()
        import torch.distributed as dist

        def reducer(stats):
            g = stats.clone()
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=dp[0])
            return g, dp[2]
        return LossTerms(reducer)

    # ---- geometry ----------------------------------------------------------------------------------
    def focal(self):
        if self.optimize_poses:   # train.py:1038-1041
            return max(self.cfg["H"], self.cfg["W"]) / 2.0 / torch.tan(self.fov[0])
        return self.data.focal

    def pose_table(self):
        return self.poses if self.optimize_poses else self.data.poses

    def _coef(self, const, name, value):
        """const * value as a LossTerms coefficient, for an iteration scalar `value` that changes every iteration:
        the host product, or -- while the iteration runs on static inputs (graph mode) -- (const, the device scalar
        `name`, which _graph_inputs() filled with `value`)"""
        if self._dyn is None:
            return const * value
        return (const, self._dyn[name])

    def rays_for(self, ids, poses=None, focal=None, uv=None, view_shift=0):
        c = self.cfg
        poses = self.pose_table() if poses is None else poses
        focal = self.focal() if focal is None else focal
        return generate_rays(ids, poses, focal, c["H"], c["W"], ndc=c["ray_type"] == "ndc", near=1.0, uv=uv,
                             view_shift=view_shift)

    # ---- one iteration -----------------------------------------------------------------------------
    def losses(self, b, terms="all", capture=None):
        """forward of one iteration on batch `b`; returns (loss_dynamic, loss_static): the two loss groups have
        disjoint parameter ancestries (dynamic field | static field + pose + focal).
        terms="image_AE" keeps pass A, pass E and the three image terms only (train.py:1323-1332, 1827-1835): the
        subset the reference-generated fixture tests/golden/pass_structure_*.npz pins; the
        two passes are the same calls either way.  b["rays"] (optional) replaces the generated rays;
        `capture` (dict) receives the per-pass tuples."""
        c = self.cfg
        S, rt, T, H, W = c["n_samples"], c["ray_type"], c["T"], c["H"], c["W"]
        it, rng = self.it, self.rng
        ids, ts, rgb_t, disp_t, fg = b["ids"], b["ts"], b["rgb"], b["disp"], b["fg"]
        N = ids.shape[0]
        poses, focal = self.pose_table(), self.focal()
        if terms not in ("all", "image_AE"):
            raise ValueError(terms)
        full = terms == "all"
        rays = b["rays"] if "rays" in b else self.rays_for(ids, poses, focal)   # with grad when the poses / focal are trained
        rays_d = rays.detach()
        poses_d, focal_d = poses.detach(), (focal.detach() if torch.is_tensor(focal) else focal)
        dt = 2.0 / (T - 1)
        if "grid" in b:
            grid, view = b["grid"], b["view"]
        else:
            col, row, view = ids2pixel(W, H, ids)
            grid = torch.stack([col.float() + 0.5, row.float() + 0.5], -1)
        # pts_2d of induce_flow is `grid_train = allgrids[ray_idx]` (train.py:1046, 974-978): the integer pixel coordinates,
        # while the flow-displaced rays start from the pixel centres (`v_ref + 0.5`, train.py:1433-1436) = `grid`
        px = b["px"] if "px" in b else grid - 0.5
        if self.optimize_poses:
            c2w_all = pose_to_mtx(poses)
        else:   # fixed poses: the [T,3,4] table is a constant of the run
            if self._c2w_fixed is None:
                self._c2w_fixed = pose_to_mtx(poses).detach()
            c2w_all = self._c2w_fixed
        temp = 1.0 / (10 ** (it // 100000))                # Temp (decay_iteration * 1000 = 100000), train.py:1034-1036
        temp_disp_tv = 1.0 / (10 ** (it // 50000))         # Temp_disp_TV
        temp_static = 1.0 / (10 ** (it / 100000.0))        # Temp_static
        ups = c.get("upsamp_list", [0, 0, 0, 0])
        early, late = it >= ups[0], it >= ups[3]           # gates of the mask terms (train.py:1338, 1349)
        gt_depth = -disp_t if rt == "ndc" else disp_t      # train.py:1645-1653
        to_depth = (lambda d: d) if rt == "ndc" else (lambda d: 1.0 / (d + 1e-6))
        # every term of a group goes through ONE fused reduction (losses.LossTerms): the per-element terms directly, the
        # other fused losses (per-frame depth loss, distortion loss, density L1) as their weighted values -- no scalar
        # torch arithmetic (each `loss = loss + w * x` on 0-dim tensors is 2 launches forward and 2 backward)
        Ld = self._loss_terms()

        def skewed(dyn):   # train.py:1349-1358: binary entropy of dynamicness^2 (late stages only)
            m2 = torch.clamp(dyn, min=1e-6, max=1.0 - 1e-6) ** 2
            return torch.mean(-(m2 * torch.log(m2) + (1 - m2) * torch.log(1 - m2)))

        def order_terms(outs):   # adaptive order loss, train.py:1277-1291 / 1666-1683
            return to_depth(outs[9]), to_depth(outs[5].detach()), (1.0 - outs[12].detach())

        # ---- pass A (and, batched with it where their inputs allow, passes B-D: see ray_passes)
        if full and self.batch_passes:
            rays_n_of = {sgn: self.rays_for(ids, poses_d, focal_d, uv=grid + b["flow_f" if sgn > 0 else "flow_b"], view_shift=sgn)
                         for sgn in (1, -1)}
            # groups of equal gradient liveness: A (everything live) | B, C, D (no appearance gradient; before
            # upsamp_list[3] no blending gradient either -- later only B has the dynamicness terms)
            groups = [[0], [1, 2, 3]] if not late else [[0], [1], [2, 3]]
            # passes B-D read depths, weights and the dynamicness only (train.py:1248-1312, 1515-1524): their colours are
            # dead work the reference computes (self.dead_work keeps it)
            pA, pB, pC, pD = ray_passes(self.st, self.dy, [rays_d, rays_d, rays_n_of[1], rays_n_of[-1]],
                                        [ts, b["ts_rand"], ts + dt, ts - dt], S, rt, rng, groups,
                                        rgb=[True] + [self.dead_work] * 3)
            osA, oA, outA, xyzA = pA
            batched = {"B": pB, 1: pC, -1: pD}
        else:
            osA, oA, outA, xyzA = ray_pass(self.st, self.dy, rays_d, ts, S, rt, rng)
            batched = None
        if capture is not None:
            capture["A"] = (osA, oA, outA, xyzA)
        Ld.add(3.0, "square", outA[0], rgb_t).add(1.0, "square", outA[8], rgb_t)      # train.py:1323, 1331
        if not full:
            oE, oEd, outE, xyzE = ray_pass(self.st, self.dy, rays, ts, S, rt, rng, static_grad=True, dynamic=self.dead_work)
            if capture is not None:
                capture["E"] = (oE, oEd, outE, xyzE)
            Ls = self._loss_terms().add(1.0 / 3.0, "square", outE[4], rgb_t, w=(1.0 - fg)[:, None], norm="weight")
            self.terms = (Ld, Ls)
            return Ld.total(), Ls.total()
        if early:
            Ld.add(0.1 * temp_disp_tv, "abs", outA[12], fg)                             # mask loss, :1338-1346
        if late:
            Ld.add(0.01, "identity", skewed(outA[12]))                                  # :1349-1364
            Ld.add(0.01, "abs", outA[12])                                               # mask_L1_reg_loss, :1366
        xo, yo, wo = order_terms(outA)
        Ld.add(10.0, "square", xo, yo, w=wo, norm="weight")                             # order_loss, :1666-1683
        Ld.add(1.0, "identity", frame_depth_loss(to_depth(outA[9]), gt_depth, view, T, coef=c["monodepth_dynamic"] * temp,
                                                 dp=self._dp()))
        # distortion loss of the dynamic weights (train.py:1299-1312, 1685-1716), ramped by iteration / n_iters
        w_dist = c["dist_dynamic"] * (it / c["n_iters"])
        k_dist = self._coef(c["dist_dynamic"], "ramp", it / c["n_iters"])   # == w_dist (a device scalar in graph mode)
        if w_dist > 0:   # mean over the rays of the per-ray loss (eff_distloss), weighted
            Ld.add(k_dist, "identity", distloss_rays(outA[11], oA[8].detach(), 1.0 / S))
        # ---- pass B (second random time)
        _, oB, outB, _ = batched["B"] if batched else ray_pass(self.st, self.dy, rays_d, b["ts_rand"], S, rt, rng,
                                                               rgb_s=self.dead_work, rgb_d=self.dead_work)
        if late:
            Ld.add(0.01, "identity", skewed(outB[12]))                                  # :1248-1266
            Ld.add(0.01, "abs", outB[12])                                               # novel_view_time_mask_loss, :1267
        xo, yo, wo = order_terms(outB)
        Ld.add(10.0, "square", xo, yo, w=wo, norm="weight")                             # novel_order_loss, :1277-1291
        if w_dist > 0:
            Ld.add(k_dist, "identity", distloss_rays(outB[11], oB[8].detach(), 1.0 / S))
        # ---- scene flow on pass A's sample points
        sf_f, sf_b = self.dy.get_forward_backward_scene_flow(oA[3], ts)
        Ld.add(c["small_scene_flow_weight"], "abs", sf_f).add(c["small_scene_flow_weight"], "abs", sf_b)   # :1421-1424
        Ld.add(c["smooth_scene_flow_weight"], "abs", sf_f, sf_b, ysign=1.0)             # :1627-1628
        # ---- induced flow of the dynamic field into the neighbour frames (train.py:1373-1413)
        weights_d, pts_ref = outA[11], oA[3]
        disp_A = {}
        for sgn, sf, flow_t, mask_t in ((1, sf_f, b["flow_f"], b["mask_f"]), (-1, sf_b, b["flow_b"], b["mask_b"])):
            pose_n = c2w_all[(view + sgn).clamp(0, T - 1)].detach()
            pts_n = pts_ref + sf if rt == "ndc" else torch.clamp(pts_ref + sf, min=-2.0 + 1e-6, max=2.0 - 1e-6)
            ind_flow, ind_disp = induce_flow(H, W, focal_d, pose_n, weights_d, pts_n, px, rays_d, ray_type=rt)
            Ld.add(0.01 * temp, "abs", ind_flow, flow_t, w=mask_t, norm="weight")       # :1392-1410 (x 0.02 / 2)
            disp_A[sgn] = (ind_disp, mask_t, pose_n, flow_t)
        # ---- pass C / D: the flow-displaced rays of the neighbour frames (train.py:1433-1528, 1530-1625)
        for sgn in (1, -1):
            ind_disp, mask_t, pose_n, flow_t = disp_A[sgn]
            if batched:
                rays_n = rays_n_of[sgn]
                _, oN, outN, _ = batched[sgn]
            else:
                rays_n = self.rays_for(ids, poses_d, focal_d, uv=grid + flow_t, view_shift=sgn)
                _, oN, outN, _ = ray_pass(self.st, self.dy, rays_n, ts + sgn * dt, S, rt, rng, rgb_s=self.dead_work,
                                          rgb_d=self.dead_work)
            _, ind_disp_n = induce_flow(H, W, focal_d, pose_n, outN[11], oN[3], px, rays_n, ray_type=rt)
            Ld.add(0.04 * temp, "abs", ind_disp, ind_disp_n, w=mask_t, norm="weight")   # :1522-1524, 1619-1621
            if w_dist > 0:
                Ld.add(k_dist, "identity", distloss_rays(outN[11], oN[8].detach(), 1.0 / S))
        # ---- pass E: static field with gradient, rays with gradient (pose / focal)
        oE, _, outE, _ = ray_pass(self.st, self.dy, rays, ts, S, rt, rng, static_grad=True, dynamic=self.dead_work)
        m = (1.0 - fg)[:, None]
        Ls = self._loss_terms()
        Ls.add(1.0 / 3.0, "square", outE[4], rgb_t, w=m, norm="weight")                  # :1828-1832
        if c["dist_static"] > 0 and it > 0:       # train.py:1841-1861
            Ls.add(self._coef(c["dist_static"], "ramp", it / c["n_iters"]), "identity",
                   distloss_rays(outE[7], oE[8].detach(), 1.0 / S))
        if self.optimize_poses:
            self._pose_block(b, rays, oE, outE, poses, focal, c2w_all, grid, px, view, m, temp_disp_tv, temp_static, gt_depth,
                             to_depth, Ls)
        # ---- factor-space regularisers (train.py:1718-1754, 1863-1885)
        if c["l1_weight"] > 0:
            Ld.add(c["l1_weight"], "identity", self.dy.density_L1())
            Ls.add(c["l1_weight"], "identity", self.st.density_L1())
        loss_s, loss_d = Ls.total(), Ld.total()
        self.terms = (Ld, Ls)
        # TV (train.py:1735-1754, 1872-1885): the VALUE is NaN in the reference (line tensors have count_w = 0)
        # while the gradients are finite; only the gradient is taken (step(): TVLoss.accumulate_grad_)
        return loss_d, loss_s

    def _pose_block(self, b, rays, oE, outE, poses, focal, c2w_all, grid, px, view, m, temp_disp_tv, temp_static, gt_depth,
                    to_depth, Ls):
        """train.py:1895-2311 (optimize_poses): every term reaches the static field, the poses and the focal."""
        c = self.cfg
        S, rt, T, H, W = c["n_samples"], c["ray_type"], c["T"], c["H"], c["W"]
        ids, ts, fg = b["ids"], b["ts"], b["fg"]
        rng = self.rng
        weights_s, pts_ref_s, depth_s = outE[7], oE[3], outE[5]
        col, row = grid[:, 0], grid[:, 1]
        uv_P34 = (torch.stack([torch.clamp(col + 1.0, max=W - 0.5), row], -1),
                  torch.stack([col, torch.clamp(row + 1.0, max=H - 0.5)], -1))
        batched = None
        if self.batch_passes:
            # P1-P4 are independent of each other and have the same gradient liveness (density branch of the static field
            # only): one static forward / backward over their 4 N rays, the draws in pass order (ray_passes)
            rays_P = [self.rays_for(ids, poses, focal, uv=grid + b["flow_f"], view_shift=1),
                      self.rays_for(ids, poses, focal, uv=grid + b["flow_b"], view_shift=-1),
                      self.rays_for(ids, poses, focal, uv=uv_P34[0]), self.rays_for(ids, poses, focal, uv=uv_P34[1])]
            smp, coins = [], []
            for k, r in enumerate(rays_P):
                jit, jit_o = rng.jitter(S, rt, rays.device)
                smp.append(sampleXYZ(self.st if k < 2 else self.dy, r, S, ray_type=rt, is_train=True, jitter=jit,
                                     jitter_outer=jit_o))
                coins.append(rng.coin() if k >= 2 else None)
            PASSES.update(static=4, static_grad=4)
            ts_P = [ts] * 4
            ns = rays.shape[0] * smp[0][1].shape[1]
            o_P = [None] * 4
            for g in batch_groups(range(4), ns):
                for k, o in zip(g, batched_field(self.st, rays_P, ts_P, smp, g, rt, rgb=self.dead_work)):
                    o_P[k] = o   # (P1-P4 read weights / depths only: train.py:1951-2311)
            d_P = [None, None]
            if self.dead_work:   # the dynamic forwards of P3 / P4: dead work the reference performs
                PASSES.update(dynamic=2, dynamic_dead=2)
                for g in batch_groups((2, 3), ns):
                    for k, o in zip(g, batched_field(self.dy, rays_P, ts_P, smp, g, rt)):
                        d_P[k - 2] = o
            batched = (rays_P, smp, coins, o_P, d_P)
        for k, (sgn, flow_t, mask_t) in enumerate(((1, b["flow_f"], b["mask_f"]), (-1, b["flow_b"], b["mask_b"]))):
            pose_n = gather_rows(c2w_all, (view + sgn).clamp(0, T - 1))         # live: allposes_refine_f / _b
            mm = mask_t * m
            ind_flow, ind_disp = induce_flow(H, W, focal, pose_n, weights_s, pts_ref_s, px, rays, ray_type=rt)
            Ls.add(self._coef(0.01, "temp_static", temp_static), "abs", ind_flow, flow_t, w=mm, norm="weight")    # :1909-1941 (x 0.02 / 2)
            # P1 / P2: the static field along the flow-displaced ray of the neighbour frame
            if batched:
                rays_n, o, xyz = batched[0][k], batched[3][k], batched[1][k][0]
            else:
                rays_n = self.rays_for(ids, poses, focal, uv=grid + flow_t, view_shift=sgn)
                jit, jit_o = rng.jitter(S, rt, rays.device)
                xyz, z, valid = sampleXYZ(self.st, rays_n, S, ray_type=rt, is_train=True, jitter=jit, jitter_outer=jit_o)
                PASSES.update(static=1, static_grad=1)
                o = self.st(rays_n, ts, None, xyz, z, valid, is_train=True, ray_type=rt, rgb=self.dead_work)
            _, ind_disp_n = induce_flow(H, W, focal, pose_n, o[4], xyz, px, rays_n, ray_type=rt)
            Ls.add(self._coef(0.04, "temp_static", temp_static), "abs", ind_disp, ind_disp_n, w=mm, norm="weight")  # :2012-2017, 2079-2084
        # per-frame median-normalised monocular depth of the static field on the background rays
        if self._dyn is None:
            Ls.add(1.0, "identity", frame_depth_loss(to_depth(depth_s), gt_depth, view, T, mask=fg < 0.5,
                                                     coef=c["monodepth_static"] * temp_static, dp=self._dp()))
        else:   # Temp_static multiplies the term's value instead of the kernel's coefficient (it changes every iteration)
            Ls.add(self._coef(1.0, "temp_static", temp_static), "identity",
                   frame_depth_loss(to_depth(depth_s), gt_depth, view, T, mask=fg < 0.5, coef=c["monodepth_static"], dp=self._dp()))
        # P3 / P4: disparity smoothness against the x+1 / y+1 pixel neighbours (train.py:2123-2311)
        inv_d = 1.0 / torch.clamp(depth_s, min=1e-6)
        for k, uv_n in enumerate(uv_P34):
            if batched:
                rays_n, o_s, o_d, z = batched[0][2 + k], batched[3][2 + k], batched[4][k], batched[1][2 + k][1]
                if o_d is None:
                    zero = torch.zeros_like(o_s[7])
                    dyn = (_rgb_or_zeros(None, zero), zero, o_s[9], zero, z)
                else:
                    dyn = (_rgb_or_zeros(o_d[6], o_d[7]), o_d[7], o_d[9], o_d[2], o_d[8])
                outN = raw2outputs(_rgb_or_zeros(o_s[6], o_s[7]), o_s[7], *dyn, rays_n, is_train=True, ray_type=rt,
                                   add_white_bg=batched[2][2 + k])
            else:
                rays_n = self.rays_for(ids, poses, focal, uv=uv_n)
                _, _, outN, _ = ray_pass(self.st, self.dy, rays_n, ts, S, rt, rng, static_grad=True, dynamic=self.dead_work,
                                         rgb_s=self.dead_work)
            Ls.add(50.0 * temp_disp_tv, "square", inv_d, 1.0 / torch.clamp(outN[5], min=1e-6))  # :2293-2305

    def step(self, shard=None):
        """One iteration on this rank's shard of the batch: forward of every pass, two-phase backward (static
        group first; its gradient exchange starts while the dynamic group is still differentiating).
        Returns the loss tensor (device, no sync)."""
        if shard is not None and self.dp_exact_stats:
            # the exact-statistics exchange all-gathers the per-ray depths with all_gather_into_tensor: equal shards only
            require_even_shards(self.cfg["batch_size"], shard[1])
        if self.graph:
            if (shard is not None and shard[1] > 1) or self.opt.ex.active:
                raise RuntimeError("Trainer(graph=True) is the single-process path: the gradient exchange of a data-parallel run "
                                   "overlaps the backward phases, which a captured iteration cannot expose")
            return self._step_graph()
        b = self.data.make_batch(self.it, self.cfg["batch_size"], shard)
        loss_d, loss_s = self._forward_backward(b, tv_between=True)
        return (loss_d + loss_s).detach()

    def _tv_weights(self):
        # TV_weight_density / TV_weight_app are multiplied by lr_factor every iteration BEFORE use (train.py:1734-1750;
        # the static terms of the same iteration use the decayed value, :1872-1885)
        c = self.cfg
        return c["tv_density"] * self.lr_factor ** (self.it + 1), c["tv_app"] * self.lr_factor ** (self.it + 1)

    def _tv(self, which):
        c = self.cfg
        if not (c["tv_density"] > 0 or c["tv_app"] > 0):
            return
        tv_d, tv_a = self._tv_weights()
        st, dy = self.st, self.dy
        if which == 0:
            self.tv.accumulate_grad_(st, [(st.density_plane, st.density_line), (st.app_plane, st.app_line)], [tv_d, tv_a])
        else:
            self.tv.accumulate_grad_(dy, [(dy.density_plane, dy.density_line), (dy.blending_plane, dy.blending_line),
                                          (dy.app_plane, dy.app_line)], [tv_d, tv_d, tv_a])

    def _forward_backward(self, b, tv_between):
        """losses + both backward phases on batch b.  tv_between: the TV gradients and the start of each field's gradient
        exchange sit between the phases (the eager order); a captured iteration leaves them to the caller."""
        loss_d, loss_s = self.losses(b)
        self.opt.zero_grad()
        if self.optimize_poses:
            self.poses.grad = None
            self.fov.grad = None
        st, dy = self.st, self.dy
        loss_s.backward()
        st.det_fold_()   # RDRF_DETERMINISTIC=1: fixed-point shadow -> fp32 gradients (no-op in the product build)
        if tv_between:
            self._tv(0)
            self.opt.begin_exchange(0)           # static field: complete
        loss_d.backward()
        dy.det_fold_()
        if tv_between:
            self._tv(1)
            self.opt.begin_exchange(1)
        self.last = dict(loss_dynamic=loss_d.detach(), loss_static=loss_s.detach())
        return loss_d, loss_s

    # ---- captured iteration (graph=True) -----------------------------------------------------------
    def _graph_key(self):
        """everything of an iteration that is baked into the captured launches: shapes, gates, step-function weights"""
        c, it = self.cfg, self.it
        ups = c.get("upsamp_list", [0, 0, 0, 0])
        return (tuple(c["grid"]), c["n_samples"], c["batch_size"], it // 100000, it // 50000, it >= ups[0], it >= ups[3],
                c["dist_dynamic"] * it > 0, c["dist_static"] > 0 and it > 0, self.dead_work, self.batch_passes,
                self.opt.state[0]["g"].data_ptr(), self.opt.state[1]["g"].data_ptr())

    def _graph_inputs(self):
        """refresh the static inputs of the iteration about to run: ray indices, draws, iteration scalars (6 launches)"""
        c, it, gs = self.cfg, self.it, self._gs
        bs = c["batch_size"]
        gs["ids"][0].copy_(self.data.batch(it, bs, 0))
        gs["ids"][1].copy_(self.data.batch(it, bs, 1))
        self.rng.begin()
        self._dyn["ramp"].fill_(it / c["n_iters"])
        self._dyn["temp_static"].fill_(1.0 / (10 ** (it / 100000.0)))

    def _step_graph(self):
        from . import _lib as L
        if L.DETERMINISTIC:
            raise RuntimeError("Trainer(graph=True): the deterministic build rebinds its shadow buffers from the host every launch")
        dev, bs = self.device, self.cfg["batch_size"]
        if self._gs is None:
            self._gs = dict(ids=(torch.zeros(bs, dtype=torch.long, device=dev), torch.zeros(bs, dtype=torch.long, device=dev)),
                            stream=torch.cuda.Stream(dev))
            scal = torch.zeros(2, device=dev)
            self._dyn = dict(ramp=scal[0:1], temp_static=scal[1:2])
            if not isinstance(self.rng, GraphRng):
                self.rng = GraphRng(dev)
        gs = self._gs
        key = self._graph_key()
        self._graph_inputs()
        body = lambda: self._forward_backward(self.data.make_batch(self.it, bs, ids=gs["ids"]), tv_between=False)
        entry = self._graphs.get(key)
        if entry is None and self._graph_seen != key:
            # first iteration of this shape / gate combination: eager on the static inputs (the warm-up of every launch path
            # a capture needs: lazily sized workspaces, packed weight images, function attributes)
            self._graph_seen = key
            loss_d, loss_s = body()
            total = (loss_d + loss_s).detach()
        else:
            if entry is None:
                before = collections.Counter(PASSES)
                # nothing may keep the previous iteration's autograd graph alive: the AccumulateGrad nodes of the pose table
                # and the field of view would be reused, and they belong to the stream they were created on -- the engine
                # would then tie that stream into the capture (torch.cuda.graph collects garbage on entry)
                self.terms, self.last = None, {}
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=gs["stream"]):
                    loss_d, loss_s = body()
                    total = (loss_d + loss_s).detach()
                del loss_d, loss_s
                delta = collections.Counter(PASSES)
                delta.subtract(before)
                PASSES.subtract(delta)   # a capture enqueues nothing: the replay below counts its passes
                # per-term values only (static memory): the LossTerms objects hold the captured autograd graph
                terms = tuple(types.SimpleNamespace(values=t.values) for t in self.terms)
                entry = dict(graph=g, total=total, last=self.last, terms=terms, passes=delta,
                             pose_grads=(self.poses.grad, self.fov.grad) if self.optimize_poses else None)
                self._graphs[key] = entry
            if self.optimize_poses:   # the gradient tensors the captured backward writes
                self.poses.grad, self.fov.grad = entry["pose_grads"]
            entry["graph"].replay()
            PASSES.update(entry["passes"])
            self.last, self.terms, total = entry["last"], entry["terms"], entry["total"]
        self._tv(0)
        self._tv(1)
        return total

    def finish_step(self):
        if self.optimize_poses:
            if self.opt.ex.active:   # the pose / focal gradients are a [T*9 + 1] vector: one tiny all-reduce
                import torch.distributed as dist
                buf = torch.cat([self.poses.grad.reshape(-1), self.fov.grad.reshape(-1)])
                dist.all_reduce(buf, group=self.opt._group)
                buf /= self.opt.world
                self.poses.grad.copy_(buf[:-1].view_as(self.poses))
                self.fov.grad.copy_(buf[-1:])
            self.opt_pose.step()
            self.opt_focal.step()
            for o in (self.opt_pose, self.opt_focal):   # scheduler.step() / scheduler_focal.step(), train.py:2326-2330
                o.param_groups[0]["lr"] *= self.pose_gamma
                if self.it > self.cfg.get("n_iters", 100000) // 2:   # train.py:2608-2610
                    o.param_groups[0]["lr"] = 0.0
        self.opt.step()
        self.it += 1

    def upsample(self, grid, n_samples=None):
        """train.py:2582-2606: both fields to the new grid, a NEW Adam (moments dropped) whose learning rates restart
        at lr_init / lr_basis (lr_upsample_reset = 1, opt.py:73-77); the pose rate restarts at lr_pose, the focal rate
        is switched on from upsamp_list[3]."""
        self.st.upsample_volume_grid(grid)
        self.dy.upsample_volume_grid(grid)
        self.cfg["grid"] = list(grid)
        if n_samples is not None:
            self.cfg["n_samples"] = int(n_samples)
        self.opt.rebuild(iteration=self.it)
        self._graphs, self._graph_seen = {}, None   # captured iterations hold the old factor tensors
        if self.optimize_poses and self.opt.lr_upsample_reset:
            self.opt_pose.param_groups[0]["lr"] = self.lr_pose
            if self.it >= self.cfg.get("upsamp_list", [0, 0, 0, 0])[3]:
                self.opt_focal.param_groups[0]["lr"] = self.lr_pose
        self.grad_flats = self.opt.grad_flats()


@torch.no_grad()
def render_chunk(st, dy, rays, ts, n_samples, ray_type="ndc"):
    """renderer.py:740-812 loop body (no-grad eval pass): returns rgb_map_full, depth_map_full."""
    _, _, outs, _ = ray_pass(st, dy, rays, ts, n_samples, ray_type, StepRng(), is_train=False)
    return outs[0], outs[1]
