"""Gradient parity of the HIP path vs the oracle over seeds beyond the three the test suite pins
(tests/test_gpu_backward._midsize_once, both ray types): python tools/seed_sweep.py on an MI355X."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import test_gpu_backward as T
from _gpu_util import ELEM
fails = 0
for rt in ("ndc", "contract"):
    for seed in range(20, 32):
        try:
            bad, l2 = T._midsize_once(seed, "full", rt, elem=ELEM)
            status = "ok" if (not bad and l2 < 1e-3) else "FAIL"
        except AssertionError as e:
            bad, l2, status = [str(e)[:100]], float("nan"), "ASSERT"
        if status != "ok":
            fails += 1
        print(rt, seed, status, f"l2 {l2:.2e}", bad[:2] if bad else "")
print("failures:", fails)
