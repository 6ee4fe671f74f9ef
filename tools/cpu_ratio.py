"""CPU baseline bridge: the reference's OWN PyTorch-CPU ray-pass and the oracle's
(oracle/rodynrf_oracle.py, what bench.py's `cpu_baseline` times on the GPU box) on the same rays, weights and cores, so that
`cpu_baseline` can be read as PyTorch-CPU-reference-equivalent through a stated ratio.  Runs ONLY in the build container
(it imports /root/reference; nothing of the reference travels to the GPU box).

    python tools/cpu_ratio.py [rays] [stage]        # default 1024 rays at the Balloon1 stage-0 shape

A training ray-pass (SURVEY 8d "unit of work"): sampleXYZ -> static forward -> dynamic forward -> raw2outputs -> the three
image terms of train.py:1323-1332 -> backward to every parameter of both fields; an eval ray-pass is the no-grad forward.
The oracle is timed both with its index-arithmetic gathers and with the reference's own F.grid_sample formulation
(USE_GRID_SAMPLE, what bench.py's cpu_baseline uses)."""
import contextlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG   # noqa: E402  (the committed import recipe: stub modules + get_device shim)
from oracle import rodynrf_oracle as O   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
stage = sys.argv[2] if len(sys.argv) > 2 else "stage0"
grid, S = {"stage0": ([141, 157, 94], 115), "final": ([331, 368, 220], 270)}[stage]
TS, TD, renderer, _, _ = MG.import_reference()
aabb = torch.tensor([[-1.5, -1.67, -1.0], [1.5, 1.67, 1.0]])
st, dy = MG.build_fields(TS, TD, aabb, grid, "relu", "MLP_Fea", -10, 20211202)
g = torch.Generator().manual_seed(1)
o = torch.stack([torch.empty(N).uniform_(-1.2, 1.2, generator=g), torch.empty(N).uniform_(-1.3, 1.3, generator=g), -torch.ones(N)], -1)
d = torch.stack([torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1, 2 * torch.ones(N)], -1)
rays = torch.cat([o, d], -1)
ts = torch.randint(0, 12, (N,), generator=g).float() * 2 / 11 - 1
target = torch.rand(N, 3, generator=g)
fg = (torch.rand(N, generator=g) < 0.3).float()


def loss_of(outs):
    full, rgb_s, rgb_d = outs[0], outs[4], outs[8]
    return 3.0 * ((full - target) ** 2).mean() + ((rgb_d - target) ** 2).mean() + (((rgb_s - target) ** 2) * (1 - fg)[:, None]).mean()


def ref_pass(train):
    with contextlib.redirect_stdout(io.StringIO()):
        xyz, z, valid = renderer.sampleXYZ(dy, rays, N_samples=S, ray_type="ndc", is_train=False)
        o_s = st(rays, ts, None, xyz, z, valid, ray_type="ndc")
        o_d = dy(rays, ts, None, xyz, z, valid, ray_type="ndc")
        outs = renderer.raw2outputs(o_s[6], o_s[7], o_d[6], o_d[7], o_d[9], o_d[2], o_d[8], rays, is_train=False, ray_type="ndc")
    if train:
        for m in (st, dy):
            m.zero_grad(set_to_none=True)
        loss_of(outs).backward()


sd_s = {k: v.detach().clone().requires_grad_(True) for k, v in st.state_dict().items()}
sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in dy.state_dict().items()}
base = dict(aabb=aabb, act="relu", density_shift=-10.0, distance_scale=25.0, weight_thres=1e-4, view_pe=0)
cfg_s, cfg_d = dict(base, head="MLP_Fea", fea_pe=2), dict(base, head="MLP_Fea_late_view", fea_pe=0)


def oracle_pass(train):
    _, _, outs = O.ray_pass(sd_s, cfg_s, sd_d, cfg_d, rays, ts, S, "ndc")
    if train:
        for v in list(sd_s.values()) + list(sd_d.values()):
            v.grad = None
        loss_of(outs).backward()


def med(fn, train, reps=3):
    ctx = contextlib.nullcontext() if train else torch.no_grad()
    with ctx:
        fn(train)
        ts_ = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn(train)
            ts_.append(time.perf_counter() - t0)
    return sorted(ts_)[len(ts_) // 2]


thr = torch.get_num_threads()
print(f"# {N} rays x {S} samples, grid {grid}, {thr} torch threads ({os.cpu_count()} cores), torch {torch.__version__}, median of 3 after 1 warm-up")
rows = {}
for train in (True, False):
    kind = "train (fwd+bwd)" if train else "eval (no-grad fwd)"
    t_ref = med(ref_pass, train)
    O.USE_GRID_SAMPLE = False
    t_idx = med(oracle_pass, train)
    O.USE_GRID_SAMPLE = True
    t_gs = med(oracle_pass, train)
    O.USE_GRID_SAMPLE = False
    rows[kind] = (t_ref, t_idx, t_gs)
    print(f"{kind:20s} reference PyTorch-CPU {N / t_ref:8.1f} ray-passes/s ({t_ref:6.2f} s)   oracle (index gathers) {N / t_idx:8.1f}   "
          f"oracle (grid_sample gathers, = bench.py cpu_baseline) {N / t_gs:8.1f}   reference / oracle(grid_sample) = {t_gs / t_ref:.2f}")
