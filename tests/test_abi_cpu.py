"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the header
declares, the host mirror keeps the reference's state_dict contract, and ops refuse CPU tensors
(no silent fallback)."""
import re
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import importlib
    L = importlib.import_module("robust-dynrf_amd._lib")
    hdr = open(os.path.join(ROOT, "include", "rodynrf.h")).read()
    declared = set(re.findall(r"\b(rdrf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rdrf_stream_t"}
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(L.lib, sym), f"librodynrf.so does not export {sym}"
    assert set(L.SYMBOLS) == declared
    assert L.lib.rdrf_abi_version() == L.ABI_VERSION == 6


def test_state_dict_contract_and_layout():
    import rodynrf
    from _util import load_case
    g, sd_s, _, sd_d, _ = load_case("ndc_relu")
    grid = [int(v) for v in g["meta.grid"]]
    kw = dict(density_n_comp=[16, 4, 4], appearance_n_comp=[48, 12, 12], app_dim=27,
              near_far=[0.0, 1.0], alphaMask_thres=1e-4, density_shift=-10, distance_scale=25,
              pos_pe=6, view_pe=0, featureC=128, step_ratio=2.0, fea2denseAct="relu")
    aabb = torch.from_numpy(g["aabb"])
    st = rodynrf.TensorVMSplit(aabb, grid, 12, "cpu", shadingMode="MLP_Fea", fea_pe=2, **kw)
    dy = rodynrf.TensorVMSplit_TimeEmbedding(aabb, grid, 12, "cpu", shadingMode="MLP_Fea_late_view",
                                             fea_pe=0, **kw)
    for mod, sd in ((st, sd_s), (dy, sd_d)):
        own = mod.state_dict()
        assert set(own.keys()) == set(sd.keys())
        for k in sd:
            assert tuple(own[k].shape) == tuple(sd[k].shape), k
        mod.load_state_dict(sd)
        for k, v in mod.state_dict().items():
            assert torch.equal(v.contiguous(), sd[k]), k
    p = dy.density_plane[0]
    _, c, h, w = p.shape
    assert p.stride() == (c * h * w, 1, w * c, c)  # XY plane: [y][x][C] storage
    p = dy.density_plane[1]
    _, c, h, w = p.shape
    F = __import__("importlib").import_module("robust-dynrf_amd.fields")
    want = (c * h * w, 1, c, h * c) if F.Z_FAST else (c * h * w, 1, w * c, c)
    assert p.stride() == want  # XZ plane: [z][x][C] (x fastest) unless fields.Z_FAST
    assert len(st.get_optparam_groups()) == 6 and len(dy.get_optparam_groups()) == 18
    assert st.nSamples == dy.nSamples


def test_no_cpu_fallback():
    import importlib
    import rodynrf
    L = importlib.import_module("robust-dynrf_amd._lib")
    rays = torch.zeros(4, 6)
    rays[:, 5] = 1
    with pytest.raises(L.RdrfError):
        rodynrf.raw2outputs(torch.zeros(4, 3, 3), torch.zeros(4, 3), torch.zeros(4, 3, 3),
                            torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 3),
                            rays)


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """TensorBase.save / the reload recipe of train.py:433-447 (`kwargs = ckpt["kwargs"]`, pop the two
    pose entries, `Model(**kwargs, device=...)`, `.load(ckpt)`): same file layout as
    models/tensorBase.py:438-485, parameters bit-identical after the round trip, channel-last kept."""
    import rodynrf
    from _util import load_case
    g, sd_s, _, sd_d, _ = load_case("ndc_relu")
    grid = [int(v) for v in g["meta.grid"]]
    kw = dict(density_n_comp=[16, 4, 4], appearance_n_comp=[48, 12, 12], app_dim=27,
              near_far=[0.0, 1.0], alphaMask_thres=1e-4, density_shift=-10, distance_scale=25,
              pos_pe=6, view_pe=0, featureC=128, step_ratio=2.0, fea2denseAct="relu")
    aabb = torch.from_numpy(g["aabb"])
    for cls, head, fea_pe, sd in ((rodynrf.TensorVMSplit, "MLP_Fea", 2, sd_s),
                                  (rodynrf.TensorVMSplit_TimeEmbedding, "MLP_Fea_late_view", 0, sd_d)):
        m = cls(aabb, grid, 12, "cpu", shadingMode=head, fea_pe=fea_pe, **kw)
        m.load_state_dict(sd)
        path = str(tmp_path / f"{cls.__name__}.th")
        m.save(torch.eye(3, 4)[None].repeat(12, 1, 1), torch.tensor(1.0), path)
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        assert set(ckpt.keys()) == {"kwargs", "state_dict"}
        kwargs = ckpt["kwargs"]
        assert kwargs.pop("se3_poses").shape == (12, 3, 4) and float(kwargs.pop("focal_ratio_refine")) == 1.0
        kwargs.update({"device": "cpu"})
        m2 = cls(**kwargs)
        m2.load(ckpt)
        for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert k == k2 and torch.equal(a, b), k
        assert m2.density_plane[1].stride(1) == 1 and m2.nSamples == m.nSamples


def test_reference_written_checkpoint_loads():
    """A `.th` file written by the REFERENCE's own TensorBase.save (tests/golden/reference_ckpt_*.th,
    make_golden.gen_checkpoint; models/tensorBase.py:460-470) goes through the reload recipe of train.py:433-447
    on this package's classes: every state_dict entry bit-identical to the weights the reference saved (the ndc_relu
    case), kwargs accepted as they are, VM factors re-strided channel-last, derived sizes (nSamples) equal to what the
    reference's own reload computed (reference_ckpt_probe.npz)."""
    import os
    import numpy as np
    import rodynrf
    from _util import GOLDEN, load_case
    g, sd_s, _, sd_d, _ = load_case("ndc_relu")
    probe = np.load(os.path.join(GOLDEN, "reference_ckpt_probe.npz"))
    for tag, cls, sd in (("static", rodynrf.TensorVMSplit, sd_s), ("dynamic", rodynrf.TensorVMSplit_TimeEmbedding, sd_d)):
        ckpt = torch.load(os.path.join(GOLDEN, f"reference_ckpt_{tag}.th"), map_location="cpu", weights_only=False)
        assert set(ckpt.keys()) == {"kwargs", "state_dict"}
        kwargs = dict(ckpt["kwargs"])
        assert kwargs.pop("se3_poses").shape == (12, 3, 4) and abs(float(kwargs.pop("focal_ratio_refine")) - 41.5) < 1e-6
        kwargs.update({"device": "cpu"})
        m = cls(**kwargs)
        m.load(ckpt)
        got = m.state_dict()
        assert set(got.keys()) == set(sd.keys()) == set(ckpt["state_dict"].keys())
        for k, v in sd.items():
            assert torch.equal(got[k], v), k
        assert all(pl.stride(1) == 1 for pl in m.density_plane) and all(pl.stride(1) == 1 for pl in m.app_plane)
        assert m.nSamples == int(probe[tag + ".nSamples"])
        assert m.get_kwargs().keys() == {k for k in ckpt["kwargs"] if k not in ("se3_poses", "focal_ratio_refine")}


def _sincos_pe_np(a):
    """numpy fp32 restatement of csrc/rdrf_common.hpp sincos_pe (fma emulated through fp64)"""
    import numpy as np
    f = np.float32

    def fma(a, b, c):
        return (a.astype(np.float64) * np.float64(b) + np.asarray(c, dtype=np.float64)).astype(f)

    a = a.astype(f)
    n = np.rint(a * f(0.63661977236758134)).astype(f)
    r = fma(n, f(-1.57079625129699707031e+00), a)
    r = fma(n, f(-7.54978941586159635335e-08), r)
    q = n.astype(np.int64)
    r2 = (r * r).astype(f)
    sp = fma(r2, f(-1.9515295891e-4), f(8.3321608736e-3))
    sp = (sp.astype(np.float64) * r2 + np.float64(f(-1.6666654611e-1))).astype(f)
    sr = ((sp * r2).astype(f).astype(np.float64) * r + r).astype(f)
    cp = fma(r2, f(2.443315711809948e-5), f(-1.388731625493765e-3))
    cp = (cp.astype(np.float64) * r2 + np.float64(f(4.166664568298827e-2))).astype(f)
    cr = ((cp * r2).astype(f).astype(np.float64) * r2 + fma(r2, f(-0.5), f(1.0))).astype(f)
    sw = (q & 1) == 1
    S, Cc = np.where(sw, cr, sr), np.where(sw, sr, cr)
    return np.where((q & 2) == 2, -S, S), np.where(((q + 1) & 2) == 2, -Cc, Cc)


def test_sincos_pe_formula():
    """csrc/rdrf_common.hpp sincos_pe (the positional encodings' sin / cos): two-constant Cody-Waite reduction +
    Cephes minimax polynomials, restated here in numpy fp32 (fma emulated through fp64) and bounded against fp64
    sin / cos over the encodings' argument range x * 2^k, |x| <= 1.5, k <= 9, and up to the 1e5 hand-over to OCML:
    max abs error < 1.2e-7 (2 ulp of 1; fp32 libm itself: 7e-8)."""
    import numpy as np
    f = np.float32
    sincos_pe = _sincos_pe_np
    rng = np.random.default_rng(0)
    x = rng.uniform(-1.5, 1.5, 100000).astype(f)
    worst = 0.0
    for k in range(10):
        a = (x * f(2 ** k)).astype(f)
        S, Cc = sincos_pe(a)
        worst = max(worst, np.abs(S - np.sin(a.astype(np.float64))).max(), np.abs(Cc - np.cos(a.astype(np.float64))).max())
    a = rng.uniform(-1e5, 1e5, 100000).astype(f)
    S, Cc = sincos_pe(a)
    worst = max(worst, np.abs(S - np.sin(a.astype(np.float64))).max(), np.abs(Cc - np.cos(a.astype(np.float64))).max())
    assert worst < 1.2e-7, worst


def test_product_library_does_not_read_the_environment():
    """a C-ABI call's behaviour must not depend on the caller's environment.  The product library (and its
    deterministic twin) import no getenv; the A/B switches live in the tools build only (make -C robust-dynrf_amd/csrc
    tools: -DRDRF_TOOLS)."""
    import subprocess
    for name in ("librodynrf.so", "librodynrf_det.so"):
        so = os.path.join(ROOT, "robust-dynrf_amd", name)
        if not os.path.exists(so):
            pytest.skip(name + " not built")
        syms = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
        assert "getenv" not in syms, f"{name} imports getenv"


def test_sincos_double_formula():
    """csrc/rdrf_common.hpp sincos_double: the odd octaves of a positional encoding from the even ones,
    sin 2a = (s + s) c, cos 2a = fma(-(s + s), s, 1) with (s, c) = sincos_pe(a): bounded against fp64 sin / cos of the
    DOUBLED fp32 argument (what the reference evaluates) over x * 2^k, |x| <= 1.5, even k <= 8: max abs error < 4e-7, and
    the static head's (F, 2F) pair over |F| <= 50."""
    import numpy as np
    f = np.float32
    rng = np.random.default_rng(1)

    def doubled(a):
        s, c = _sincos_pe_np(a)
        s, c = s.astype(f), c.astype(f)
        t = (s + s).astype(f)
        s2 = (t * c).astype(f)
        c2 = (np.float64(1.0) - t.astype(np.float64) * s.astype(np.float64)).astype(f)   # one rounding: the fma
        return s2, c2

    worst = 0.0
    x = rng.uniform(-1.5, 1.5, 100000).astype(f)
    for k in range(0, 10, 2):
        a = (x * f(2 ** k)).astype(f)
        s2, c2 = doubled(a)
        ref = (a * f(2)).astype(np.float64)   # exact in fp32
        worst = max(worst, np.abs(s2 - np.sin(ref)).max(), np.abs(c2 - np.cos(ref)).max())
    a = rng.uniform(-50, 50, 100000).astype(f)
    s2, c2 = doubled(a)
    worst = max(worst, np.abs(s2 - np.sin((a * f(2)).astype(np.float64))).max(), np.abs(c2 - np.cos((a * f(2)).astype(np.float64))).max())
    assert worst < 4e-7, worst
