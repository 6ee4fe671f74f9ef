#!/usr/bin/env python
"""Per-kernel register / LDS / spill table from hipcc -Rpass-analysis=kernel-resource-usage
(cross-compiles without a GPU).  Usage: python tools/kernel_resources.py [file.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "robust-dynrf_amd", "csrc")


def table(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-disable-promote-alloca-to-lds=1", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-result",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    print(f"{'kernel':58s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
    for f in files:
        for r in table(f):
            name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name)[:58]
            print(f"{name:58s} {r.get('VGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>5s} "
                  f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('LDS Size [bytes/block]', '?'):>7s} "
                  f"{r.get('Occupancy [waves/SIMD]', '?'):>3s}")
