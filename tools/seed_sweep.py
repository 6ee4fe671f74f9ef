"""Gradient parity of the HIP path vs the oracle over seeds beyond the three the test suite pins
(tests/test_gpu_backward._midsize_once, both ray types): python tools/seed_sweep.py [--eps E] [--seeds a,b,...] [--rt ndc|contract]
on an MI355X.  --eps widens the kink guard band (default 2e-6 of each layer's scale): a seed that fails at 2e-6 and
passes at a wider band had a relu unit between the two (a branch the GPU and the CPU take differently), not an error."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import test_gpu_backward as T
from _gpu_util import ELEM
arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
eps = float(arg("--eps", 2e-6))
seeds = [int(v) for v in arg("--seeds", ",".join(str(v) for v in range(20, 32))).split(",")]
rts = [arg("--rt", None)] if "--rt" in sys.argv else ["ndc", "contract"]
fails = 0
for rt in rts:
    for seed in seeds:
        try:
            bad, l2 = T._midsize_once(seed, "full", rt, elem=ELEM, kink_eps=eps)
            status = "ok" if (not bad and l2 < 1e-3) else "FAIL"
        except AssertionError as e:
            bad, l2, status = [str(e)[:100]], float("nan"), "ASSERT"
        if status != "ok":
            fails += 1
        print(rt, seed, status, f"l2 {l2:.2e}", bad[:2] if bad else "")
print("failures:", fails)
