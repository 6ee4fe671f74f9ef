"""End-to-end trainer harness (robust-dynrf_amd/step.py): a short run of the Nvidia.txt-shaped step on
a small synthetic scene stays finite and reduces the loss -- catches sign / accumulation mistakes in
the fused-gradient, pruning and optimiser wiring that per-kernel parity tests cannot see."""
import importlib

import numpy as np

import pytest
import torch

from _util import assert_close, record_margin

pytestmark = pytest.mark.gpu


def test_short_training_run_reduces_loss():
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.balloon1_config("stage0")
    cfg.update(grid=[36, 40, 24], n_samples=48, batch_size=512, H=27, W=48, T=6)
    cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * 3.0 ** 0.5
    tr = S_.Trainer(cfg, torch.device("cuda", 0))
    losses = []
    for _ in range(40):
        loss = tr.step()
        tr.finish_step()
        losses.append(float(loss.detach()))
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    for m in (tr.st, tr.dy):
        for n, p in m.named_parameters():
            assert torch.isfinite(p).all(), n
    first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    assert last < 0.9 * first, (first, last)


def dp_check_cfg(S_):
    cfg = S_.balloon1_config("stage0")
    cfg.update(grid=[24, 26, 16], n_samples=40, batch_size=256, H=27, W=48, T=6)
    cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * 3.0 ** 0.5
    return cfg


@pytest.mark.parametrize("mode", ["allreduce", "zero1"])
def test_two_rank_step_with_exact_statistics_matches_single_process(mode, tmp_path):
    """SURVEY 8e / train.py:1391-1394, 797-807: a 2-rank data-parallel step (ray-sharded, gloo, both ranks on this GPU)
    with dp_exact_stats -- mask sums of the masked means all-reduced, per-frame depth statistics from the gathered batch
    -- produces, after the exchange, the gradient of the SINGLE-process step on the whole batch (1e-3 relative L2 per
    flat buffer: atomics order and the split reductions are the only differences); with per-shard statistics (the
    default) the two differ visibly, which is what the option exists for."""
    import os
    import subprocess
    import sys
    S_ = importlib.import_module("robust-dynrf_amd.step")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dp_check_cfg(S_)
    torch.manual_seed(0)
    tr = S_.Trainer(cfg, torch.device("cuda", 0))
    tr.it = 9000
    tr.step()
    full = [f.detach().cpu().clone() for f in tr.grad_flats]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    rels = {}
    for exact in ("1", "0"):
        out = str(tmp_path / f"g{exact}.pt")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29711 + int(exact)), os.path.join(root, "tools", "dp_check.py"), out, exact, mode]
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        got = torch.load(out)
        rels[exact] = [float((a - b).norm() / b.norm()) for a, b in zip(got, full)]
    assert max(rels["1"]) < 1e-3, rels
    assert max(rels["0"]) > 3.0 * max(rels["1"]), rels    # per-shard statistics: a different objective


@pytest.mark.parametrize("dp", ["zero1", "allreduce"])
def test_bench_two_ranks_functional(dp):
    """`python bench.py --gpus 2` exactly as the driver may call it (no torch.distributed environment): the
    script re-executes itself under torch.distributed.run with 2 ranks.  The box has one GPU, so both ranks
    share cuda:0 and use gloo (RDRF_DIST_BACKEND) instead of RCCL: everything but the transport of the N>1
    path runs -- ray sharding, reduce-scatter -> sharded Adam -> all-gather (or the in-place all-reduce), the
    early asynchronous exchange of the static field, max-over-ranks timing, the rank-0 JSON."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RDRF_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rays-per-gpu", "512", "--no-render", "--no-final-stage", "--no-cpu-baseline", "--dp", dp]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 1024 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["ranks"] == 2 and dp in d["config"]["parallelism"]
    assert d["config"]["backend"] == "gloo" and d["config"]["exchange_bytes_per_step"] > 1e6
    assert "roofline" in d and len(line) < 8192
    detail = json.load(open(os.path.join(root, d["detail"])))   # the per-kernel tables live in the side file (bench.write_detail)
    assert detail["n_gpus"] == 2 and detail["roofline"]["kernel_ms_per_step"]["adam"] > 0
    assert detail["config"]["exchange_plan"]["issued"] is True


@pytest.mark.parametrize("dp", ["zero1", "allreduce"])
def test_bench_rccl_call_sequence_on_one_gpu(dp):
    """The RCCL transport itself: RDRF_FORCE_COLLECTIVES=1 makes a world-size-1 `nccl` process group and issues
    every collective of the N>1 step on it (reduce_scatter_tensor into the shard / in-place all_gather_into_tensor
    of the updated slice, or the in-place all_reduce, with their asynchronous handles and stream hand-over).  At
    world size 1 every collective is the identity, so the loss after 6 steps must equal the plain run's."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RDRF_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1",
           "--rays-per-gpu", "512", "--no-render", "--no-final-stage", "--no-cpu-baseline", "--dp", dp]
    res = []
    for force in ("0", "1"):
        with socket.socket() as sk:   # a free rendezvous port
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(base, RDRF_FORCE_COLLECTIVES=force, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        res.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
    plain, forced = res
    assert plain["config"]["backend"] is None and forced["config"]["backend"] == "nccl"
    assert forced["config"]["exchange_bytes_per_step"] > 1e6 and dp in forced["config"]["parallelism"]
    a, b = plain["config"]["final_loss"], forced["config"]["final_loss"]
    assert abs(a - b) <= 2e-3 * abs(a), (a, b)   # same arithmetic; atomics order differs run to run


def test_bench_refuses_world_size_mismatch():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "world size" in (out.stderr + out.stdout)


SMALL = {
    "nvidia": dict(grid=[24, 26, 16], n_samples=24, batch_size=64, H=27, W=48, T=6),
    "nvidia_no_poses": dict(grid=[17, 19, 11], n_samples=13, batch_size=64, H=27, W=48, T=6),
    "davis": dict(grid=[16, 16, 16], n_samples=14, batch_size=64, H=24, W=42, T=7),
}


@pytest.mark.parametrize("name,it", [("nvidia", 5000), ("nvidia", 30000), ("nvidia_no_poses", 5000), ("davis", 5000)])
def test_trainer_step_gradient_matches_oracle_step(name, it):
    """ONE complete iteration of every config -- 5 dynamic + 5 static forwards (Nvidia.txt)
    or 7 + 9 with the pose / focal block (Nvidia_no_poses.txt, DAVIS.txt: contracted rays, TimeEmbedding
    static head, density_L1, per-frame depth losses) -- on a small scene: the flat gradient of both fields
    (and of the pose table and the field of view) from Trainer.step vs the oracle's re-enactment of the same
    iteration (oracle/rodynrf_oracle_step.py) with identical batch, jitter vectors and white-background coins, at 3e-4
    of each tensor's max + the fp64 conditioning allowance, on weights conditioned like the reference-trainer fixtures
    (dynamic density bias + 0.3).  The error is NOT atomic-order noise: four runs on one box and runs on two boxes give the
    same margins to three digits (tools/noise_spread.py oracle, profiles/r05_noise_spread_*.txt: nvidia-5000 0.481 0.481
    0.481 0.481 of its 3e-4 bound; the other three cases are held to 1e-4, margins 0.12 / 0.28 / 0.20) -- it is the distance between two fp32 evaluation
    orders of a badly conditioned iteration, so the margin does not move with the box.  (1e-4 is NOT reachable for
    nvidia at iteration 5000: 1.44 x even with the conditioning, 2.67 x without it; the other three cases hold 1e-4 with
    margins 0.12-0.30.  The version of this comparison that is pinned to the REFERENCE's own trainer at 1e-4 is
    test_trainer_step_matches_reference_trainer_iteration.)"""
    from oracle import rodynrf_oracle_step as OS
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.scene_config(name, "stage0")
    cfg.update(SMALL[name])
    cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * 3.0 ** 0.5
    dev = torch.device("cuda", 0)
    tr = S_.Trainer(cfg, dev)
    # the conditioning of make_golden_trainer.py (DENSITY_BIAS_SHIFT): +0.3 on the dynamic density head's output bias gives
    # every ray dynamic density, so raw2outputs' weights_d / (sum + 1e-10) no longer amplifies fp32 rounding 1e17-fold
    with torch.no_grad():
        tr.dy.density_layer2.bias.add_(0.3)
    tr.dy.invalidate_packed()
    tr.it = it    # the ramped loss weights are non-zero; 5000 / 30000: before / after upsamp_list[0] and [3] (mask-term gates)
    tr.rng = OS.FixedRng(11)
    sd_s = {k: v.detach().cpu().contiguous().clone() for k, v in tr.st.state_dict().items()}
    sd_d = {k: v.detach().cpu().contiguous().clone() for k, v in tr.dy.state_dict().items()}
    batch = {k: v.cpu() for k, v in tr.data.make_batch(tr.it, cfg["batch_size"]).items()}
    poses = tr.pose_table().detach().cpu()
    foc = tr.fov.detach().cpu() if tr.optimize_poses else float(tr.data.focal)
    loss_ref, gref = OS.step_gradients(dict(cfg), sd_s, sd_d, batch, poses, foc, tr.it, OS.FixedRng(11))
    # conditioning: the same iteration in fp64 (identical draws: the jitter / coin stream is fp32 either way) measures how
    # far the fp32 REFERENCE arithmetic is from exact for each gradient entry (the order loss is 10 x a squared depth
    # difference: cancelling sums; a relu unit whose pre-activation rounds to the other side): the kernels are held to
    # tolerance + twice that distance, element by element, the allowance capped at the tolerance itself (no kink-free
    # ray selection here -- the trainer draws its own batch; the pass structure is pinned at 1e-4 by the reference fixture)
    torch.set_default_dtype(torch.float64)
    try:
        cv = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        _, g64 = OS.step_gradients(dict(cfg), {k: cv(v.detach()) for k, v in sd_s.items()},
                                   {k: cv(v.detach()) for k, v in sd_d.items()}, {k: cv(v) for k, v in batch.items()},
                                   cv(poses), cv(foc), tr.it, OS.FixedRng(11))
    finally:
        torch.set_default_dtype(torch.float32)
    loss = tr.step()
    assert abs(float(loss) - float(loss_ref)) <= 2e-4 * abs(float(loss_ref)), (float(loss), float(loss_ref))

    def check(name, got, ref, ref64, rtol):
        a, b = got.detach().cpu().double(), ref.double()
        cond = (2.0 * (b - ref64.double()).abs()).clamp(max=rtol * float(b.abs().max()))
        err = (a - b).abs()
        tol = rtol * float(b.abs().max()) + cond
        record_margin(name, float((err / tol.clamp_min(1e-30)).max()))
        if not bool((err <= tol).all()):
            i = int((err - tol).argmax())
            bad.append(f"{name}: |err| {float(err.flatten()[i]):.3e} > {rtol:.0e} * max|ref| ({float(b.abs().max()):.3e}) + "
                       f"conditioning {float(cond.flatten()[i]):.3e}")

    # north_star's 1e-4 for three of the four cases (measured margins 0.12 / 0.28 / 0.20 of it); nvidia at iteration 5000
    # needs 3e-4 (docstring)
    rtol_fields = 3e-4 if (name, it) == ("nvidia", 5000) else 1e-4
    bad, n = [], 0
    for mod, pre in ((tr.st, "s."), (tr.dy, "d.")):
        for k, p in mod.named_parameters():
            ref = gref[pre + k]
            if ref is None:
                assert float(p.grad.abs().max()) == 0.0, f"{pre}{k}: expected no gradient"
                continue
            n += 1
            check(pre + k, p.grad, ref, g64[pre + k], rtol_fields)
    if tr.optimize_poses:
        for nm, ten in (("poses", tr.poses), ("fov", tr.fov)):
            check(nm, ten.grad, gref[nm], g64[nm], 2e-4)
    assert n > 60 and not bad, "\n".join(bad)
    # and the optimiser step applies: parameters move, stay finite
    before = tr.st.flatten_params_().clone()
    tr.finish_step()
    assert torch.isfinite(tr.st.flatten_params_()).all() and not torch.equal(before, tr.st.flatten_params_())


@pytest.mark.parametrize("name,stage,rays", [("nvidia_no_poses", "stage0", 4096), ("davis", "final", 8192),
                                              ("nvidia_no_poses", "final", 4096)])
def test_config_workloads_run_at_full_shape(name, stage, rays):
    """BASELINE.json configs[2] / [3] / [4] at their own shapes -- grids [17,19,11] / S=13, [256,256,256] / S=221
    (contract, 8192 rays) and [706,786,471] / S=578 (the 640^3 grid) at the 4096 rays one of 8 ranks of the 32k-ray
    configuration takes -- two complete iterations: finite loss,
    finite parameters, gradients reach every factor family, poses and focal; and size-independent properties:
    repeat runs of the forward are bit-identical, a slice of the batch run alone reproduces its outputs."""
    import rodynrf
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.scene_config(name, stage)
    cfg["batch_size"] = rays
    dev = torch.device("cuda", 0)
    tr = S_.Trainer(cfg, dev)
    tr.it = 3000
    for _ in range(2):
        loss = tr.step()
        flats = [f.clone() for f in tr.grad_flats]
        tr.finish_step()
    assert torch.isfinite(loss) and all(torch.isfinite(f).all() and float(f.abs().max()) > 0 for f in flats)
    for m in (tr.st, tr.dy):
        for n_, p in m.named_parameters():
            assert torch.isfinite(p).all(), n_
            if "_plane" in n_ or "_line" in n_:
                assert float(p.grad.abs().max()) > 0, f"no gradient reached {n_}"
    assert float(tr.poses.grad.abs().max()) > 0 and float(tr.fov.grad.abs().max()) > 0
    with torch.no_grad():
        ids = tr.data.batch(0, min(rays, 512), 0)
        r = tr.rays_for(ids).detach()
        t = tr.data.ts_of(ids)
        S, rt = cfg["n_samples"], cfg["ray_type"]
        xyz, z, valid = rodynrf.sampleXYZ(tr.dy, r, S, ray_type=rt, is_train=False)
        a = tr.dy(r, t, None, xyz, z, valid, ray_type=rt)
        b = tr.dy(r, t, None, xyz, z, valid, ray_type=rt)
        assert all(torch.equal(x, y) for x, y in zip(a[2:], b[2:]) if x is not None)
        c = tr.dy(r[100:164], t[100:164], None, xyz[100:164], z[100:164], valid[100:164], ray_type=rt)
        for x, y in zip(a[2:], c[2:]):
            if x is not None:
                assert float((x[100:164] - y).abs().max()) <= 1e-6 * max(1.0, float(y.abs().max()))


def test_full_size_batch_independence_and_gradient_additivity():
    """BASELINE.json configs[1] at full size (4096 rays x 115 samples, grid [141,157,94]) is too large
    for the oracle; it is tied to the oracle-checked sizes through two size-independent properties:
    (1) rays are independent -- any slice of the full batch, run alone, gives the same outputs as
    inside the batch (so the 256-ray oracle comparison of test_oracle_forward_balloon_shapes speaks
    for every ray of the full batch); (2) gradients of a sum-type loss are additive over ray subsets
    (the scatter / dW accumulation of 471k samples equals the sum of its halves)."""
    import rodynrf
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.balloon1_config("stage0")
    dev = torch.device("cuda", 0)
    st, dy = S_.build_fields(cfg, dev)
    data = S_.SyntheticBalloon(cfg, dev)
    N, S = 4096, cfg["n_samples"]
    ids = data.batch(0, N, 0)
    rays = rodynrf.generate_rays(ids, data.poses, data.focal, cfg["H"], cfg["W"], ndc=True, near=1.0)
    ts = data.ts_of(ids)
    jit = torch.rand(S, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    g = torch.Generator(device=dev).manual_seed(2)
    w_rgb, w_dep = torch.rand(N, 3, device=dev, generator=g), torch.rand(N, device=dev, generator=g)

    def run(sl, grads):
        r, t = rays[sl], ts[sl]
        xyz, z, valid = rodynrf.sampleXYZ(dy, r, S, ray_type="ndc", is_train=True, jitter=jit)
        o_s = st(r, t, None, xyz, z, valid, ray_type="ndc")
        o_d = dy(r, t, None, xyz, z, valid, ray_type="ndc")
        outs = rodynrf.raw2outputs(o_s[6], o_s[7], o_d[6], o_d[7], o_d[9], o_d[2], o_d[8], r,
                                   is_train=True, ray_type="ndc", add_white_bg=False)
        if not grads:
            return [o.detach() for o in (outs[0], outs[1], outs[8], outs[11], o_d[2], o_d[5], o_s[4])]
        loss = (outs[0] * w_rgb[sl]).sum() + (outs[9] * w_dep[sl]).sum() + (outs[4] * w_rgb[sl]).sum()
        for m in (st, dy):
            for p in m.parameters():
                p.grad = None
        loss.backward()
        return {n: p.grad.detach().clone() for m, pre in ((st, "s."), (dy, "d.")) for n, p in
                ((pre + k, v) for k, v in m.named_parameters()) if p.grad is not None}

    with torch.no_grad():
        full = run(slice(0, N), False)
        again = run(slice(0, N), False)
        for a, b in zip(full, again):
            assert torch.equal(a, b), "forward is not deterministic"
        for lo, hi in ((0, 64), (1000, 1256), (4000, 4096)):
            part = run(slice(lo, hi), False)
            for a, b in zip(full, part):
                err = float((a[lo:hi] - b).abs().max())
                assert err <= 1e-6 * max(1.0, float(b.abs().max())), (lo, hi, err)
    g_full = run(slice(0, N), True)
    g_a, g_b = run(slice(0, N // 2), True), run(slice(N // 2, N), True)
    assert set(g_full) == set(g_a) == set(g_b) and len(g_full) > 60
    for k, v in g_full.items():
        s = g_a[k] + g_b[k]
        rel = float((v - s).abs().max() / v.abs().max().clamp_min(1e-30))
        record_margin(k + " additivity / 2e-4", rel / 2e-4)
        assert rel < 2e-4, (k, rel)


ONAMES = ["rgb_map_full", "depth_map_full", "acc_map_full", "weights_full", "rgb_map_s", "depth_map_s", "acc_map_s",
          "weights_s", "rgb_map_d", "depth_map_d", "acc_map_d", "weights_d", "dynamicness_map"]
FNAMES = ["_0", "_1", "blending", "pts_ref", "weight", "xyz_prime", "rgb", "sigma", "z", "dists"]


@pytest.mark.parametrize("dead_work", [True, False])
@pytest.mark.parametrize("case", ["ndc_relu", "contract_relu_te"])
def test_pass_structure_matches_reference_fixture(case, dead_work):
    """SURVEY 8a row 13, pinned to the REFERENCE: Trainer.losses' pass A (rays detached, static field value-only,
    its outputs detached before raw2outputs) and pass E (rays with grad, static field live, dynamic forward dead)
    with the three image terms, against tests/golden/pass_structure_*.npz -- the same two call sequences
    re-enacted on the imported reference models (make_golden.gen_pass_structure: train.py:1092-1162, 1756-1835,
    1323-1332): both passes' sampler outputs (bit-equal), forward 10-tuples, the 13 compositor outputs, the
    three loss terms, dL/dtheta of every parameter of both fields and dL/drays, with the reference's own
    jitter vectors and white-background coins replayed."""
    from _gpu_util import ELEM
    from _util import load_case, load_pass_structure, pass_structure_cfg, pass_structure_draws
    from oracle.rodynrf_oracle_step import ReplayRng
    S_ = importlib.import_module("robust-dynrf_amd.step")
    g, sd_s, _, sd_d, _ = load_case(case)
    p = load_pass_structure(case)
    cfg = pass_structure_cfg(g)
    dev = torch.device("cuda", 0)
    tr = S_.Trainer(cfg, dev, dead_work=dead_work)
    tr.st.load_state_dict(sd_s)
    tr.dy.load_state_dict(sd_d)
    tr.rng = ReplayRng(*pass_structure_draws(p, cfg["ray_type"]))
    rays = torch.from_numpy(g["rays"]).to(dev).requires_grad_(True)
    N = rays.shape[0]
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    batch = dict(ids=torch.zeros(N, dtype=torch.long, device=dev), ts=t(g["ts"]), rgb=t(p["rgb_train"]),
                 fg=t(p["fg"])[:, 0], disp=torch.zeros(N, device=dev), rays=rays)
    cap = {}
    tr.opt.zero_grad()
    loss_d, loss_s = tr.losses(batch, terms="image_AE", capture=cap)
    assert_close(loss_d, 3.0 * p["loss_full"] + p["loss_d"], "loss_d", rtol=5e-5)
    assert_close(loss_s, p["loss_s"], "loss_s", rtol=5e-5)
    el = (1e-4, 1e-6)
    for tag in ("A", "E"):
        o_s, o_d, outs, xyz = cap[tag]
        assert torch.equal(xyz.detach().cpu(), torch.from_numpy(p[tag + ".xyz"])), tag + ".xyz"
        for k, v in zip(FNAMES, o_s):
            if v is not None:
                assert_close(v, p[f"{tag}.fs.{k}"], f"{tag}.fs.{k}", elem=el)
        assert (o_d is None) == (tag == "E" and not dead_work)
        if o_d is not None:
            for k, v in zip(FNAMES, o_d):
                if v is not None:
                    assert_close(v, p[f"{tag}.fd.{k}"], f"{tag}.fd.{k}", elem=el)
        for i, (k, v) in enumerate(zip(ONAMES, outs)):
            if o_d is None and i not in (4, 5, 6, 7):
                continue
            assert_close(v, p[f"{tag}.c.{k}"], f"{tag}.c.{k}", elem=el)
    # pass A must not hold the static field in its graph; pass E must not reach the dynamic field
    assert not cap["A"][0][6].requires_grad and not cap["A"][0][7].requires_grad
    loss_s.backward()
    assert all(q.grad is None or float(q.grad.abs().max()) == 0.0 for q in tr.dy.parameters()), "pass E reached the dynamic field"
    g_rays_E = rays.grad.clone()
    loss_d.backward()
    assert torch.equal(rays.grad, g_rays_E), "pass A reached the rays (it sees rays.detach())"
    bad = []
    for mod, pre in ((tr.st, "gs."), (tr.dy, "gd.")):
        for k, q in mod.named_parameters():
            ref = p[pre + k]
            if pre == "gd." and bool(p["gd_none." + k]):
                assert q.grad is None or float(q.grad.abs().max()) == 0.0, f"{k}: the reference graph never reaches it"
                continue
            try:
                assert_close(q.grad, ref, pre + k, rtol=1e-4, elem=ELEM)
            except AssertionError as e:
                bad.append(str(e))
    try:
        assert_close(rays.grad, p["g.rays"], "g.rays", rtol=1e-4, elem=ELEM)
    except AssertionError as e:
        bad.append(str(e))
    assert not bad, "\n".join(bad)


def test_upsample_restarts_learning_rates_like_the_reference():
    """train.py:2582-2606 with lr_upsample_reset = 1 (opt.py:73-77, every shipped config): after an upsample the new
    Adam starts at lr_init / lr_basis again (not at the decayed rates), the moments and the step count are dropped,
    the pose rate restarts at lr_pose and the focal rate is switched on from upsamp_list[3]; with
    lr_upsample_reset = 0 the rates continue at lr * lr_decay_target_ratio ** (iteration / n_iters)."""
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.scene_config("nvidia_no_poses", "stage0")
    cfg.update(SMALL["nvidia_no_poses"])
    cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * 3.0 ** 0.5
    cfg["n_iters"] = 1000    # a visible decay per step
    tr = S_.Trainer(cfg, torch.device("cuda", 0))
    assert tr.opt_focal.param_groups[0]["lr"] == 0.0 and tr.opt_pose.param_groups[0]["lr"] == 3e-3
    for _ in range(3):
        tr.step()
        tr.finish_step()
    f = 0.1 ** (1.0 / 1000)
    assert tr.opt.lr0 == pytest.approx(0.02 * f ** 3) and tr.opt.lr1 == pytest.approx(1e-3 * f ** 3) and tr.opt.t == 3
    assert tr.opt_pose.param_groups[0]["lr"] < 3e-3
    tr.it = cfg["upsamp_list"][3]
    tr.upsample([20, 22, 13], 16)
    assert tr.opt.lr0 == 0.02 and tr.opt.lr1 == 1e-3 and tr.opt.t == 0
    assert all(float(st["m"].abs().max()) == 0.0 and float(st["v"].abs().max()) == 0.0 for st in tr.opt.state)
    assert tr.opt_pose.param_groups[0]["lr"] == 3e-3 and tr.opt_focal.param_groups[0]["lr"] == 3e-3
    loss = tr.step()
    tr.finish_step()
    assert torch.isfinite(loss)
    tr.opt.lr_upsample_reset = False
    tr.upsample([24, 26, 16], 18)
    assert tr.opt.lr0 == pytest.approx(0.02 * f ** tr.it) and tr.opt.lr1 == pytest.approx(1e-3 * f ** tr.it)


@pytest.mark.parametrize("name,stage,it", [("nvidia", "stage0", 5000), ("nvidia", "final", 30000),
                                           ("nvidia_no_poses", "final", 30000), ("davis", "stage0", 5000)])
def test_batched_passes_match_one_pass_per_launch(name, stage, it, monkeypatch):
    """step.ray_passes (passes A-D through one static forward, passes of equal gradient liveness through one dynamic
    forward / backward) against one launch sequence per pass: the same draws, the same loss values, gradients equal up to the order of the atomic accumulation -- at the benchmark shape, at the
    late stage (B alone, C + D batched), at [706,786,471] / S = 578 with 3 x 4096 rays in one dynamic call (7.1 M
    samples: the batched buffers pass 2^32 floats, the per-pass ones do not) and 4 x 4096 in the static call of the pose
    block's passes P1-P4, and on contracted rays (davis)."""
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.scene_config(name, stage)
    if name == "nvidia_no_poses":   # lift step.BATCH_MAX_SAMPLES (6 M): all of B-D / P1-P4 in one call each
        monkeypatch.setattr(S_, "BATCH_MAX_SAMPLES", 1 << 40)
    dev = torch.device("cuda", 0)
    got = {}
    for batched in (True, False):
        tr = S_.Trainer(cfg, dev, batch_passes=batched)
        tr.it = it
        b = tr.data.make_batch(tr.it, cfg["batch_size"], None)
        loss_d, loss_s = tr.losses(b)
        tr.opt.zero_grad()
        if tr.optimize_poses:
            tr.poses.grad = None
            tr.fov.grad = None
        loss_s.backward()
        loss_d.backward()
        torch.cuda.synchronize()
        got[batched] = (loss_d.detach().clone(), loss_s.detach().clone(), [g.detach().clone() for g in tr.grad_flats])
        del tr, b, loss_d, loss_s
        torch.cuda.empty_cache()
    assert_close(got[True][0], got[False][0], "dynamic loss group", rtol=1e-6)
    assert_close(got[True][1], got[False][1], "static loss group", rtol=1e-6)
    # Bounds from the measured spread, not from one run (tools/noise_spread.py batched 5, profiles/r05_noise_spread_box*.txt:
    # 5 fresh trainers per path, two boxes).  The quantity is the distance between two orders of fp32 accumulation (batched
    # calls take the sorted scatter, per-pass calls the ray-tile scatter); it varies run to run.  Measured worst case
    # (davis stage 0, static field: 8192 x 13 x 5 samples into a 16^3 grid, i.e. ~1e5 terms per texel): max-norm 1.98e-5 of
    # max|g| between the paths (6.5e-6 between two runs of ONE path), rel. L2 8.9e-6 (5.0e-6); every other config <= 2.2e-6 /
    # 1.5e-6.  Bounds: 1e-4 max-norm (north_star's tolerance, 5 x the worst spread) and 5e-5 rel. L2 (5.6 x).  A real
    # difference between the paths -- a pass dropped, a draw replayed for the wrong pass, a head pruned -- moves the
    # gradient by >= 1e-2.
    for a, b_ in zip(got[True][2], got[False][2]):
        assert float(b_.abs().max()) > 0
        rel = float((a - b_).norm() / b_.norm())
        record_margin("batched vs per-pass gradient (rel. L2 / 5e-5)", rel / 5e-5)
        assert rel < 5e-5, rel
        assert_close(a, b_, "batched vs per-pass gradient", rtol=1e-4)


@pytest.mark.parametrize("name", ["nvidia", "nvidia_late", "nvidia_no_poses", "davis"])
def test_trainer_step_matches_reference_trainer_iteration(name):
    """SURVEY 8a row 13, the whole iteration pinned to the REFERENCE: Trainer.step (batched passes, fused loss terms,
    two-phase backward, TV gradient) against the gradients the reference's own train.reconstruction() produced at
    iteration 0 (tests/golden/trainer_iter_*.npz, make_golden_trainer.py) -- passes A-E, the pose block P1-P4, every loss
    term with its gate and weight -- on the same weights, batch, jitter vectors and coins: every parameter of both
    fields, the pose table and the field of view at 1e-4 of each tensor's max, element by element.  Where two fp32
    evaluation orders of the same iteration legitimately differ (a relu unit on its kink: see
    test_oracle_step_matches_reference_trainer_iteration) the element is allowed the distance between the oracle's
    fp32 and fp64 evaluations on top."""
    from _util import load_trainer_iter
    from oracle import rodynrf_oracle_step as OS
    S_ = importlib.import_module("robust-dynrf_amd.step")
    p, cfg, batch, sd_s, sd_d, poses, focal, draws = load_trainer_iter(name)
    dev = torch.device("cuda", 0)
    tr = S_.Trainer(dict(cfg), dev, dead_work=True)
    tr.st.load_state_dict(sd_s)
    tr.dy.load_state_dict(sd_d)
    for f in (tr.st, tr.dy):
        f.invalidate_packed()
    if tr.optimize_poses:
        with torch.no_grad():
            tr.poses.copy_(poses.to(dev))
            tr.fov.copy_(focal.to(dev))
    else:
        tr.data.poses, tr.data.focal = poses.to(dev), torch.tensor(float(focal), device=dev)
    tr.it = 0
    tr.rng = OS.ReplayRng(*draws)
    H, W = cfg["H"], cfg["W"]
    b = {k: v.to(dev) for k, v in batch.items()}
    col, row, view = b["ids"] % W, (b["ids"] // W) % H, b["ids"] // (W * H)
    b.update(grid=torch.stack([col.float() + 0.5, row.float() + 0.5], -1), px=torch.stack([col.float(), row.float()], -1),
             view=view)
    tr.data.make_batch = lambda it, bs, shard=None: b
    tr.step()
    assert not tr.rng.jitters and not tr.rng.coins, "the trainer consumed a different number of draws than the reference"
    # the oracle's fp32 / fp64 evaluations of the same iteration: their distance is the kink / conditioning allowance
    _, g32 = OS.step_gradients(cfg, sd_s, sd_d, batch, poses, focal, 0, OS.ReplayRng(*draws), dead_work=True)
    torch.set_default_dtype(torch.float64)
    try:
        cv = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        _, g64 = OS.step_gradients(cfg, {k: cv(v.detach()) for k, v in sd_s.items()}, {k: cv(v.detach()) for k, v in sd_d.items()},
                                   {k: cv(v) for k, v in batch.items()}, cv(poses), cv(focal), 0, OS.ReplayRng(*draws),
                                   dead_work=True)
    finally:
        torch.set_default_dtype(torch.float32)
    bad, n = [], 0

    def check(label, got, ref, o32, o64, scale, rtol):
        a, r = got.detach().cpu().double().reshape(ref.shape), ref.double()
        # (4 x the fp32-vs-fp64 distance: tensors whose gradient is a 1e-6-scale residue of cancelling terms -- the blending
        # planes of nvidia_no_poses, max|ref| 3.7e-6 -- sit at 2.4 x that distance from the reference, i.e. in its rounding)
        tol = rtol * scale + 4.0 * (o32.detach().double().reshape(ref.shape) - o64.double().reshape(ref.shape)).abs()
        err = (a - r).abs()
        record_margin(label, float((err / tol.clamp_min(1e-30)).max()))
        if not bool((err <= tol + 1e-12).all()):
            i = int((err - tol).argmax())
            bad.append(f"{label}: |err| {float(err.flatten()[i]):.3e} > {rtol:.0e} * {scale:.3e} + allowance {float(tol.flatten()[i]) - rtol * scale:.3e}")

    for mod, pre, gp in ((tr.st, "s.", "gs."), (tr.dy, "d.", "gd.")):
        for k, q in mod.named_parameters():
            ref = torch.from_numpy(p[gp + k])
            if bool(p[gp.replace("g", "gnone_", 1) + k]):
                assert q.grad is None or float(q.grad.abs().max()) == 0.0, f"{pre}{k}: the reference graph never reaches it"
                continue
            n += 1
            check(pre + k, q.grad, ref, g32[pre + k], g64[pre + k], float(ref.abs().max()), 1e-4)
    if tr.optimize_poses:
        ps = float(np.abs(p["g.poses"]).max())
        check("poses", tr.poses.grad, torch.from_numpy(p["g.poses"]), g32["poses"], g64["poses"], ps, 1e-4)
        check("fov", tr.fov.grad, torch.from_numpy(p["g.fov"]), g32["fov"], g64["fov"],
              max(float(np.abs(p["g.fov"]).max()), 1e-2 * ps), 1e-4)
    assert n > 60 and not bad, "\n".join(bad)


@pytest.mark.parametrize("name,batched", [("nvidia", True), ("nvidia", False), ("nvidia_no_poses", True), ("davis", True)])
def test_dead_work_pruning_changes_nothing(name, batched):
    """Trainer(dead_work=False) (the default) skips what the reference computes and nothing consumes -- the dynamic forward
    of passes E / P3 / P4 and the appearance phase (colours) of both fields in passes B-D and P1-P4 (forward(rgb=False)) --
    against dead_work=True: the same draws, the same loss values, the same gradients (up to the order of the atomic
    accumulation) for both fields, the pose table and the field of view."""
    S_ = importlib.import_module("robust-dynrf_amd.step")
    cfg = S_.scene_config(name, "stage0")
    cfg.update(SMALL[name])
    cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * 3.0 ** 0.5
    dev = torch.device("cuda", 0)
    got = {}
    for dead in (True, False):
        tr = S_.Trainer(dict(cfg), dev, dead_work=dead, batch_passes=batched)
        tr.it = 30000   # every gate open, ramped weights non-zero
        tr.step()
        got[dead] = ([v.detach().clone() for v in tr.last.values()], [g.detach().clone() for g in tr.grad_flats],
                     [tr.poses.grad.clone(), tr.fov.grad.clone()] if tr.optimize_poses else [])
    for a, b in zip(got[True][0], got[False][0]):
        assert_close(a, b, "loss", rtol=1e-6)
    for a, b in zip(got[True][1] + got[True][2], got[False][1] + got[False][2]):
        assert float(a.abs().max()) > 0
        rel = float((a - b).norm() / a.norm())
        record_margin("pruned vs executed dead work (rel. L2 / 1e-5)", rel / 1e-5)
        assert rel < 1e-5, rel
