"""RDRF_DETERMINISTIC=1 (librodynrf_det.so: 64-bit fixed-point gradient accumulation, sorted compaction lists, no LDS
line accumulators): the parameter gradients of a complete training step are bit-identical run after run, and equal
the product build's (hardware fp32 atomics, run-to-run noise) within fp32 accumulation noise -- the reproducible
reference a suspected race is diffed against (SURVEY.md 5 'race detection', 7 hard part 1)."""
import os
import subprocess
import sys

import pytest
import torch

from _util import record_margin

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, det, size, name="nvidia"):
    out = str(tmp_path / f"{tag}.pt")
    env = dict(os.environ, RDRF_DETERMINISTIC="1" if det else "0")
    env.pop("RDRF_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "det_check.py"), out, size, name], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return torch.load(out)


@pytest.mark.parametrize("size,name", [("small", "nvidia"), ("small", "davis"), ("bench", "nvidia")])
def test_deterministic_build_is_bit_reproducible_and_matches_the_atomic_build(tmp_path, size, name):
    a = _run(tmp_path, "det_a", True, size, name)
    b = _run(tmp_path, "det_b", True, size, name)
    c = _run(tmp_path, "atomic", False, size, name)
    for x, y in zip(a["flats"], b["flats"]):
        assert torch.equal(x, y), "the deterministic build is not bit-reproducible"
    assert float(a["loss"]) == float(b["loss"])
    for x, z in zip(a["flats"], c["flats"]):
        scale = float(x.abs().max())
        assert scale > 0
        err = float((x - z).abs().max())
        rel = float((x - z).norm() / x.norm())
        record_margin("deterministic vs atomic build (max-norm / 2e-5)", err / (2e-5 * scale))
        record_margin("deterministic vs atomic build (rel. L2 / 2e-6)", rel / 2e-6)
        # the atomic build's own order noise (fp32 sums of up to ~1e5 terms): run-to-run spread of one path at the benchmark
        # shape 7.6e-7 max-norm / 3.7e-7 rel. L2 (tools/noise_spread.py, profiles/r05_noise_spread_*.txt), up to 6.5e-6 /
        # 5.0e-6 where ~1e5 terms meet in one texel (16^3 grids)
        assert err <= 2e-5 * scale, (err, scale)
        assert rel < 2e-6, rel   # measured <= 5.2e-7 at these shapes (profiles/r05_parity_margins.txt): 4 x headroom


def test_product_build_refuses_the_deterministic_entry_points():
    import importlib
    L = importlib.import_module("robust-dynrf_amd._lib")
    if L.DETERMINISTIC:
        pytest.skip("running under the deterministic build")
    assert L.lib.rdrf_deterministic() == 0
    assert L.lib.rdrf_det_finish(0, None) < 0 and b"product build" in L.lib.rdrf_last_error()


def test_render_chunks_on_hip_streams_is_bit_identical_in_the_deterministic_build():
    """the deterministic build sorts every compaction list through one process-wide scratch
    (rdrf_sort_ints_inplace), so rdrf_render_chunks_fwd must not run chunks concurrently there: it serialises the loop on
    the caller's stream (csrc/rdrf_render.hip).  The stream test of the product build, re-run against librodynrf_det.so."""
    env = dict(os.environ, RDRF_DETERMINISTIC="1")
    env.pop("RDRF_LIB", None)
    env.pop("RDRF_MARGINS", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_forward.py"), "-q", "-m", "gpu",
                        "-k", "test_render_chunks_on_hip_streams_is_bit_identical", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize("case,N,S", [("ndc_relu", 777, 115), ("contract_relu_te", 2100, 37)])
def test_appearance_backward_is_bit_reproducible_in_the_deterministic_build(case, N, S):
    """d(<rgb, g>)/d(xyz_sampled) of both fields runs through every backward-data layer of k_static_app_bwd / k_dyn_app_bwd (bf16 x 3
    with split storage: transposed images, odd block counts, lo pieces streamed one step ahead) and the feature scatter into the
    per-sample coordinate gradients.  In the deterministic build nothing depends on arrival order, so repeated backward passes must
    return the same bits: pins the operand hazards of the bf16 x 3 steps (rdrf_common.hpp mfma_seg_b3s) for the backward images, as
    test_static_forward_is_bit_reproducible / test_dynamic_forward_is_bit_reproducible do for the forward ones.  (The product build's
    coordinate gradients vary run to run at this shape with fp32 AND bf16 x 3 layers alike: the order of its fp32 atomics,
    profiles/r06_det_bwd_app.txt.)"""
    env = dict(os.environ, RDRF_DETERMINISTIC="1", REPS="12")
    env.pop("RDRF_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "graph", "det_bwd_app.py"), case, str(N), str(S)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if "distinct:" in l]
    assert len(lines) == 2, r.stdout
    for l in lines:   # coordinate gradients and the parameter gradients (flat buffer): one signature each over the 12 passes
        assert l.count("distinct: [12]") == 2 and "nonzero 0 " not in l, l
