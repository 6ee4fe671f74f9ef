"""Run-to-run spread of the quantities two GPU tests bound: the tolerances of
test_batched_passes_match_one_pass_per_launch and test_trainer_step_gradient_matches_oracle_step are derived from what this
tool prints (profiles/r05_noise_spread_*.txt), not from one measurement.

    python tools/noise_spread.py batched [runs]     # gradient(batched passes) vs gradient(one launch sequence per pass)
    python tools/noise_spread.py oracle  [runs]     # Trainer.step vs oracle step, error / (rtol max|ref| + allowance)

`batched`: R fresh trainers per path (same seeds: same weights, batch and draws); for every flat gradient buffer
  same   = max over run pairs of ONE path   of max|a-b| / max|b|  and  ||a-b|| / ||b||   (the atomic order's own noise)
  cross  = max over runs across the two paths of the same two figures                     (what the test bounds)
`oracle`: the oracle step once per (config, iteration, bias shift), the trainer R times; prints the worst
  error / (rtol max|ref| + 2 |g32-g64| capped at rtol max|ref|) at rtol = 1e-4 and 3e-4 and which tensor it is, for the
  reference initialiser (shift 0) and with density_layer2.bias shifted by +0.3 (the conditioning of make_golden_trainer.py)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
S_ = importlib.import_module("robust-dynrf_amd.step")
dev = torch.device("cuda", 0)


def grads_of(cfg, it, batched):
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
    out = [g.detach().clone() for g in tr.grad_flats]
    del tr, b, loss_d, loss_s
    torch.cuda.empty_cache()
    return out


def dist(a, b):
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())


def batched_mode(runs):
    print(f"# batched vs per-pass, {runs} runs per path; columns: max-norm / max|g|, rel. L2   (per flat buffer: static, dynamic)")
    for name, stage, it, r in (("nvidia", "stage0", 5000, runs), ("nvidia", "final", 30000, max(2, runs // 2)),
                               ("davis", "stage0", 5000, runs), ("nvidia_no_poses", "stage0", 5000, runs)):
        cfg = S_.scene_config(name, stage)
        A = [grads_of(cfg, it, True) for _ in range(r)]
        B = [grads_of(cfg, it, False) for _ in range(r)]
        for k in range(len(A[0])):
            same = [dist(X[i][k], X[j][k]) for X in (A, B) for i in range(r) for j in range(i)]
            cross = [dist(a[k], b[k]) for a in A for b in B]
            print(f"{name:16s} {stage:7s} flat{k}: same-path max {max(s[0] for s in same):.3e} {max(s[1] for s in same):.3e}   "
                  f"cross-path max {max(c[0] for c in cross):.3e} {max(c[1] for c in cross):.3e}   "
                  f"cross-path min {min(c[0] for c in cross):.3e} {min(c[1] for c in cross):.3e}", flush=True)


SMALL = {
    "nvidia": dict(grid=[24, 26, 16], n_samples=24, batch_size=64, H=27, W=48, T=6),
    "nvidia_no_poses": dict(grid=[17, 19, 11], n_samples=13, batch_size=64, H=27, W=48, T=6),
    "davis": dict(grid=[16, 16, 16], n_samples=14, batch_size=64, H=24, W=42, T=7),
}


def oracle_mode(runs):
    from oracle import rodynrf_oracle_step as OS
    print(f"# Trainer.step vs oracle step, {runs} trainer runs; worst error / (rtol max|ref| + allowance) over the parameters")
    for name, it in (("nvidia", 5000), ("nvidia", 30000), ("nvidia_no_poses", 5000), ("davis", 5000)):
        for shift in (0.0, 0.3):
            cfg = S_.scene_config(name, "stage0")
            cfg.update(SMALL[name])
            cfg["focal"] = max(cfg["H"], cfg["W"]) / 2.0 * 3.0 ** 0.5
            worst = {1e-4: [], 3e-4: []}
            ref = None
            for r in range(runs):
                tr = S_.Trainer(cfg, dev)
                if shift:
                    with torch.no_grad():
                        tr.dy.density_layer2.bias.add_(shift)
                    tr.dy.invalidate_packed()
                tr.it = it
                tr.rng = OS.FixedRng(11)
                if ref is None:
                    sd_s = {k: v.detach().cpu().contiguous().clone() for k, v in tr.st.state_dict().items()}
                    sd_d = {k: v.detach().cpu().contiguous().clone() for k, v in tr.dy.state_dict().items()}
                    batch = {k: v.cpu() for k, v in tr.data.make_batch(tr.it, cfg["batch_size"]).items()}
                    poses = tr.pose_table().detach().cpu()
                    foc = tr.fov.detach().cpu() if tr.optimize_poses else float(tr.data.focal)
                    _, g32 = OS.step_gradients(dict(cfg), sd_s, sd_d, batch, poses, foc, tr.it, OS.FixedRng(11))
                    torch.set_default_dtype(torch.float64)
                    try:
                        cv = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
                        _, g64 = OS.step_gradients(dict(cfg), {k: cv(v) for k, v in sd_s.items()}, {k: cv(v) for k, v in sd_d.items()},
                                                   {k: cv(v) for k, v in batch.items()}, cv(poses), cv(foc), tr.it, OS.FixedRng(11))
                    finally:
                        torch.set_default_dtype(torch.float32)
                    ref = (g32, g64)
                tr.step()
                for rtol in worst:
                    w, wn = 0.0, ""
                    for mod, pre in ((tr.st, "s."), (tr.dy, "d.")):
                        for k, p in mod.named_parameters():
                            if ref[0][pre + k] is None:
                                continue
                            a, b = p.grad.detach().cpu().double(), ref[0][pre + k].double()
                            cond = (2.0 * (b - ref[1][pre + k].double()).abs()).clamp(max=rtol * float(b.abs().max()))
                            m = float(((a - b).abs() / (rtol * float(b.abs().max()) + cond).clamp_min(1e-30)).max())
                            if m > w:
                                w, wn = m, pre + k
                    worst[rtol].append((w, wn))
                del tr
            for rtol, ws in worst.items():
                print(f"{name:16s} it={it:5d} bias+{shift:.1f} rtol={rtol:.0e}: " +
                      " ".join(f"{w:.3f}" for w, _ in ws) + f"   worst tensor: {max(ws)[1]}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "batched"
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    (batched_mode if mode == "batched" else oracle_mode)(runs)
