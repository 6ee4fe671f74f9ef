"""The driver parses the LAST stdout line of bench.py; round 5's line had grown to 21.5 KB and came back `parsed: null`
(VERDICT r5 item 1).  bench.compact_line() must stay under 8 KB for the full single-GPU record and for the multi-GPU one,
carry the contract's keys, and report no fraction above 1."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _records():
    """full bench records kept under profiles/: this round's bench_detail files if any, else round 5's full line"""
    new = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_detail*.json")))
    return new or [os.path.join(ROOT, "profiles", "r05_bench_driver_cmd.json"), os.path.join(ROOT, "profiles", "r05_bench.json")]


def _walk(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _walk(v, f"{path}.{k}")
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _walk(v, f"{path}[{i}]")
    else:
        yield path, o


@pytest.mark.parametrize("fn", _records(), ids=os.path.basename)
def test_compact_line_fits_and_carries_the_contract(fn):
    import bench
    full = json.load(open(fn))
    text = bench.compact_line(full)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 8192, len(text)
    assert len(text) < 6144, f"{len(text)} B: keep a margin under the limit"
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    assert line["roofline"]["bound"] in ("hbm", "mfma")
    for k in ("achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    for k in ("final_stage_value", "schedule_weighted_value", "render_mpix_per_s", "render_chunk512_mpix_per_s",
              "step_frac_of_fp32_mfma_peak"):
        assert k in line, k
    if full.get("schema", 0) >= 6:   # every fraction of the line is physical
        for path, v in _walk(line):
            if path.split(".")[-1].startswith("frac") and v is not None:
                assert 0.0 <= v <= 1.0, (path, v)
        for path, v in _walk(line.get("roofline", {}).get("mfma_frac", {})):
            assert 0.0 <= v <= 1.0, (path, v)
        for path, v in _walk(line):   # matrix-pipe occupancies (bf16 x 3 layers priced at six bf16 MFMAs per product)
            if "pipe_frac" in path and isinstance(v, float):
                assert 0.0 <= v <= 1.0, (path, v)


def test_compact_line_multi_gpu_record():
    """the --gpus 8 record: no render / final-stage / CPU legs, an exchange plan in the config (detail only)"""
    import bench
    full = json.load(open(_records()[0]))
    for k in ("render", "render_chunk512", "render_chunk512_one_stream", "final_stage", "schedule_weighted", "sparse_weights",
              "cpu_baseline", "cpu_baseline_4096", "final_stage_value", "final_stage_ms_per_step", "schedule_weighted_value",
              "render_mpix_per_s", "render_chunk512_mpix_per_s"):
        full.pop(k, None)
    full["n_gpus"] = 8
    full["config"]["parallelism"] = "ray-sharded dp8 (zero1)"
    full["config"]["exchange_plan"] = {"collectives": [{"collective": "reduce_scatter_tensor", "bytes": 421 << 20}] * 8}
    text = bench.compact_line(full)
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    assert line["n_gpus"] == 8 and "exchange_plan" not in line["config"]


def test_oversized_record_is_shed_not_printed():
    import bench
    full = json.load(open(_records()[0]))
    full["config"]["timed_region"] = "x" * 9000
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT
    assert json.loads(text)["value"] == pytest.approx(full["value"], rel=1e-4)
