"""Shared helpers for the parity tests: golden-case loading and tolerance checks."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["ndc_relu", "ndc_relu_long", "ndc_softplus", "contract_relu_te", "contract_softplus_te"]

# north_star tolerance: 1e-4 relative fp32. "Relative" is taken per tensor against its max
# magnitude (a per-element relative error is meaningless for values that are exactly 0).
RTOL = 1e-4


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    t = lambda k: torch.from_numpy(np.array(g[k]))
    sd_s = {k[2:]: t(k) for k in g if k.startswith("s.")}
    sd_d = {k[2:]: t(k) for k in g if k.startswith("d.")}
    aabb = t("aabb")
    base = dict(aabb=aabb, act=str(g["meta.act"]), density_shift=float(g["meta.density_shift"]),
                distance_scale=25.0, weight_thres=1e-4, view_pe=0)
    cfg_s = dict(base, head=str(g["meta.static_head"]), fea_pe=2)
    cfg_d = dict(base, head="MLP_Fea_late_view", fea_pe=0)
    return g, sd_s, cfg_s, sd_d, cfg_d


# element-wise forward tolerance (north_star: 1e-4 relative fp32): |err| <= 1e-4 |ref| + 1e-6 max|ref| for EVERY element
FWD_ELEM = (1e-4, 1e-6)


def elementwise_excess(a, b, elem_rtol, elem_atol_frac, atol=0.0):
    """max over the elements of |a-b| / (elem_rtol |b| + elem_atol_frac max|b|): <= 1 means every element
    satisfies |err| <= rtol |ref| + atol, with atol tied to the tensor's scale"""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    if a.numel() == 0:
        return 0.0
    scale = max(float(b.abs().max()), 1e-30)
    return float(((a - b).abs() / (elem_rtol * b.abs() + elem_atol_frac * scale + atol)).max())


def record_margin(name, ratio):
    """RDRF_MARGINS=<file>: append `test id <tab> check <tab> error / tolerance` for every tolerance check, so that one run of
    the suite shows how far each comparison sits from its bound (tools/margins.py prints the worst ones)"""
    path = os.environ.get("RDRF_MARGINS")
    if path:
        with open(path, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{name}\t{ratio:.4f}\n")


def assert_close(a, b, name="", rtol=RTOL, atol_scale=1.0, mask=None, atol=0.0, elem=None):
    """max-norm check: max|a-b| <= rtol * max|b| (* atol_scale) + atol.  elem = (elem_rtol, elem_atol_frac)
    adds the element-wise check |a-b| <= elem_rtol |b| + elem_atol_frac max|b| for EVERY element: small
    entries of a gradient must be right in relative terms too, up to the accumulation noise floor."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if mask is not None:
        a = a[mask]
        b = b[mask]
    if a.numel() == 0:
        return
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    record_margin(name, err / (rtol * scale * atol_scale + atol + 1e-30))
    assert err <= rtol * scale * atol_scale + atol + 1e-30, (
        f"{name}: max abs err {err:.3e} > {rtol * atol_scale:.1e} * max|ref| ({scale:.3e})")
    if elem is not None:
        ex = elementwise_excess(a, b, elem[0], elem[1], atol)
        record_margin(name + " (element-wise)", ex)
        assert ex <= 1.0, (f"{name}: element-wise |err| exceeds {elem[0]:.0e} |ref| + {elem[1]:.0e} max|ref| by a "
                           f"factor {ex:.2f}")


def load_pass_structure(case):
    """tests/golden/pass_structure_<case>.npz (make_golden.gen_pass_structure: pass A + pass E of the
    reference's trainer on the imported reference models) + the case it borrows weights / rays from"""
    p = np.load(os.path.join(GOLDEN, f"pass_structure_{case}.npz"))
    return {k: p[k] for k in p.files}


def pass_structure_cfg(g):
    """the trainer-config dict (robust-dynrf_amd/step.py / oracle/rodynrf_oracle_step.py) of a golden case"""
    return dict(aabb=g["aabb"].tolist(), near_far=[float(v) for v in g["meta.near_far"]], T=12, H=27, W=48,
                ray_type=str(g["meta.ray_type"]), batch_size=int(g["rays"].shape[0]), static_head=str(g["meta.static_head"]),
                optimize_poses=False, tv_density=0.0, tv_app=0.0, dist_static=0.0, dist_dynamic=0.0, l1_weight=0.0,
                grid=[int(v) for v in g["meta.grid"]], n_samples=int(g["z"].shape[1]), name="pass_structure",
                stage="fixture", monodepth_static=0.0, monodepth_dynamic=0.0, n_iters=100000,
                lr_decay_target_ratio=0.1, focal=48 / 2.0 * 3.0 ** 0.5)


def pass_structure_draws(p, ray_type):
    if ray_type == "ndc":
        jit = [p["A.jitter"], p["E.jitter"]]
    else:
        jit = [(p["A.jitter"], p["A.jitter_outer"]), (p["E.jitter"], p["E.jitter_outer"])]
    return jit, [bool(p["A.white"]), bool(p["E.white"])]


TRAINER_ITERS = ["nvidia", "nvidia_late", "nvidia_no_poses", "davis"]


def load_trainer_iter(name):
    """tests/golden/trainer_iter_<name>.npz (make_golden_trainer.py: iteration 0 of the reference's OWN
    train.reconstruction() on a synthetic dataset, stopped at the first optimizer.step()).  Returns
    (fixture dict, trainer-config dict, batch dict, sd_s, sd_d, poses9, focal_or_fov, (jitters, coins))."""
    z = np.load(os.path.join(GOLDEN, f"trainer_iter_{name}.npz"))
    p = {k: z[k] for k in z.files}
    t = lambda k: torch.from_numpy(np.array(p[k]))
    m = lambda k: p["meta." + k]
    rt, opt_poses = str(m("ray_type")), bool(m("optimize_poses"))
    H, W, T = int(m("H")), int(m("W")), int(m("T"))
    cfg = dict(aabb=m("aabb").tolist(), near_far=[float(v) for v in m("near_far")], T=T, H=H, W=W, ray_type=rt,
               batch_size=int(m("batch_size")), static_head=str(m("static_head")), optimize_poses=opt_poses,
               tv_density=float(m("tv_density")), tv_app=float(m("tv_app")), dist_static=float(m("dist_static")),
               dist_dynamic=float(m("dist_dynamic")), l1_weight=float(m("l1_weight")), grid=[int(v) for v in m("grid")],
               n_samples=int(m("n_samples")), name="trainer_iter_" + name, stage="fixture",
               monodepth_static=float(m("monodepth_static")), monodepth_dynamic=float(m("monodepth_dynamic")),
               n_iters=int(m("n_iters")), lr_decay_target_ratio=float(m("lr_decay_target_ratio")),
               small_scene_flow_weight=float(m("small_scene_flow_weight")),
               smooth_scene_flow_weight=float(m("smooth_scene_flow_weight")),
               upsamp_list=[int(v) for v in m("upsamp_list")], start_iteration=0, focal=float(p["focal_gt"]))
    batch = {k[2:]: t(k) for k in p if k.startswith("b.")}
    sd_s = {k[2:]: t(k) for k in p if k.startswith("s.")}
    sd_d = {k[2:]: t(k) for k in p if k.startswith("d.")}
    # draws in call order: per pass the jitter vector(s) (ndc: one [1,S]; contract: inner + outer) then -- when the pass
    # ends in raw2outputs -- the white-background coin (renderer.py:269: torch.rand((1,)) < 0.5)
    jit, coins, pend = [], [], []
    for i in range(int(p["n_draws"])):
        d = p[f"draw.{i:02d}"]
        if d.shape == (1,):
            coins.append(bool(d[0] < 0.5))
        else:
            pend.append(d.reshape(-1))
            if rt == "ndc" or len(pend) == 2:
                jit.append(pend[0] if rt == "ndc" else (pend[0], pend[1]))
                pend = []
    focal = t("fov") if opt_poses else float(p["focal_gt"])
    return p, cfg, batch, sd_s, sd_d, t("poses9"), focal, (jit, coins)
