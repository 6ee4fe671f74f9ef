#!/usr/bin/env python
"""Section 9 of DESIGN.md (results) is regenerated from tools/docs/design_sec9.md: its @PLACEHOLDERS@ are filled from a bench_detail
record and the committed hbm tables; README.md likewise from readme_template.md:   python tools/docs/assemble_design.py profiles/r06_bench_detail_driver_cmd.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.load(open(sys.argv[1]))
fs, rf = d["final_stage"], d["roofline"]
rff = fs["roofline"]


sys.path.insert(0, ROOT)
import bench   # pipe_factor(): the matrix-pipe floor of a kernel's instruction mix / its fp32-only floor
PF = {n: bench.pipe_factor(n) for n in ("dyn_density", "dyn_app", "static_app")}


def k(v):
    return f"{v / 1e3:.0f}"


def fam(r, name):
    for f in r["families"]:
        if f["kernel"].startswith(name):
            return f["ms_per_step"]
    return float("nan")


def hbm_total(fn):
    for l in open(os.path.join(ROOT, "profiles", fn)):
        if l.startswith("TOTAL"):
            return float(l.split()[4])
    return float("nan")


sub = {
    "SCHED_STAGES": " / ".join(f"{s['ms_per_step']:.1f}" for s in d["schedule_weighted"]["stages"]),
    "SCHED": k(d["schedule_weighted"]["value"]), "SCHED_MS": f"{d['schedule_weighted']['mean_ms_per_step']:.1f}",
    "VALUE": k(d["value"]), "MS": f"{d['ms_per_step']:.2f}", "FINAL": k(fs["value"]), "FINAL_MS": f"{fs['ms_per_step']:.1f}",
    "LIVE": k(d["liveness_exploited"]["value"]), "SPARSE": k(d["sparse_weights"]["value"]),
    "RENDER": f"{d['render']['value']:.1f}", "RENDER512": f"{d['render_chunk512']['value']:.1f}", "RENDER_F": f"{fs['render']['value']:.1f}",
    "CPU": f"{d['cpu_baseline']['value']:.0f}", "CPU4096": f"{d['cpu_baseline_4096']['value']:.0f}",
    "FRAC": f"{rf['frac']:.2f}", "FRAC_F": f"{rff['frac']:.2f}",
    "FAM_FWD": f"{fam(rf, 'forward'):.2f} / {fam(rff, 'forward'):.2f}", "FAM_SC": f"{fam(rf, 'k_scatter'):.2f} / {fam(rff, 'k_scatter'):.2f}",
    "FAM_BWD": f"{fam(rf, 'backward'):.2f} / {fam(rff, 'backward'):.2f}", "FAM_DW": f"{fam(rf, 'k_dw'):.2f} / {fam(rff, 'k_dw'):.2f}",
    "MFMA_FWD": " / ".join(f"{rf['mfma_frac'][n]:.2f}" for n in ("dyn_density", "dyn_app", "static_app")) + " at stage 0, " +
                " / ".join(f"{rff['mfma_frac'][n]:.2f}" for n in ("dyn_density", "dyn_app", "static_app")) + " at the final stage (round 5: 0.54 / 0.46 / 0.59, 0.55 / 0.49 / 0.62)",
    "MFMA_PIPE": " / ".join(f"{rf['mfma_pipe_frac'][n]:.2f}" for n in ("dyn_density", "dyn_app", "static_app")) + " at stage 0, " +
                 " / ".join(f"{rff.get('mfma_pipe_frac', {}).get(n, rff['mfma_frac'][n] * PF[n]):.2f}" for n in ("dyn_density", "dyn_app", "static_app")) +
                 f" at the final stage; whole step (`step_pipe_frac`) {rf['step_pipe_frac']:.2f}",
    "S13_EAGER": f"{d['launch_bound_stage']['eager_ms_per_step']:.1f}" if "launch_bound_stage" in d else "-",
    "S13_GRAPH": f"{d['launch_bound_stage']['graph_ms_per_step']:.2f}" if "launch_bound_stage" in d else "-",
    "HBM_STEP": f"{hbm_total('r06_hbm_table.txt'):.1f}", "HBM_FINAL": f"{hbm_total('r06_final_stage_hbm_table.txt'):.1f}",
}
here = os.path.dirname(os.path.abspath(__file__))
sec9 = open(os.path.join(here, "design_sec9.md")).read()
sec9 = re.sub(r"@([A-Z0-9_]+)@", lambda m: sub[m.group(1)], sec9)
cur = open(os.path.join(ROOT, "DESIGN.md")).read()   # sections 0-8 and 10 are edited in DESIGN.md itself
i9, i10 = cur.index("## 9. Results"), cur.index("## 10. What comes next")
out = cur[:i9] + sec9 + cur[i10:]
open(os.path.join(ROOT, "DESIGN.md"), "w").write(out)
print("DESIGN.md:", len(out.splitlines()), "lines")

# README.md from tools/docs/readme_template.md: the same placeholders, plus the default run's (suffix 2) if its record is given
if os.path.exists(os.path.join(here, "readme_template.md")):
    sub2 = dict(sub)
    if len(sys.argv) > 2:
        e = json.load(open(sys.argv[2]))
        sub2.update({"SCHED2": k(e["schedule_weighted"]["value"]), "VALUE2": k(e["value"]), "MS2": f"{e['ms_per_step']:.2f}",
                     "FINAL2": k(e["final_stage"]["value"]), "LIVE2": k(e["liveness_exploited"]["value"]),
                     "RENDER2": f"{e['render']['value']:.1f}", "RENDER5122": f"{e['render_chunk512']['value']:.1f}"})
    else:
        sub2.update({kk: "n/a" for kk in ("SCHED2", "VALUE2", "MS2", "FINAL2", "LIVE2", "RENDER2", "RENDER5122")})
    t = open(os.path.join(here, "readme_template.md")).read()
    open(os.path.join(ROOT, "README.md"), "w").write(re.sub(r"@([A-Z0-9_]+)@", lambda m: sub2[m.group(1)], t))
    print("README.md written")
