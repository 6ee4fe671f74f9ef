"""Register budget of the forward MLP kernels, checked at compile time (hipcc cross-compiles gfx950 without a GPU):
every instantiation fits its 256 registers WITHOUT spilling.  The training instantiation of k_dyn_density carried 29
spilled registers (104 B of scratch per lane) until the per-ray time-branch outputs were re-read per tile (round 6:
kernel -6 % / -7 %, profiles/r06_ab_t_reload.txt) -- a change that brings spills back shows up here, not in a profile."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc")
def test_forward_mlp_kernels_do_not_spill():
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    rows = kr.table(os.path.join(kr.CSRC, "rdrf_fwd.hip"))
    seen = 0
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        if not any(k in name for k in ("k_dyn_density", "k_dyn_app", "k_static_app")) or "static_app16" in name:
            continue
        seen += 1
        assert int(r["VGPRs"]) <= 256, (name, r["VGPRs"])
        assert int(r["VGPRs Spill"]) == 0, (name, r["VGPRs Spill"])
        assert int(r["ScratchSize [bytes/lane]"]) == 0, (name, r["ScratchSize [bytes/lane]"])
        assert int(r["Occupancy [waves/SIMD]"]) >= 2, (name, r["Occupancy [waves/SIMD]"])
    assert seen >= 10, seen   # every training / inference / feature-mode instantiation of the three kernels
