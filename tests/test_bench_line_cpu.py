"""bench.py's FINAL stdout line: the driver keeps the last 8 KB of stdout and parses the last line, so the compact record must
stay well below that whatever the legs add to the full record (round 3's 23-KB line left BENCH_r03.parsed null)."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

FULL = os.path.join(ROOT, "tests", "golden", "bench_full_record_r03.json")  # a full record of a real run (round 3, all legs)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def test_compact_line_fits_and_round_trips():
    full = json.load(open(FULL))
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in line
    assert len(line) < 6144, len(line)
    c = json.loads(line)
    for key in CONTRACT:
        assert key in c, key
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]
    r = c["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "units_per_launch"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["f32_operand_leg"]["frac"] == full["roofline"]["f32_operand_leg"]["frac"]
    assert r["hbm_from_pmc_GBps"] == round(full["roofline"]["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9, 1)
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c["cpu_baseline"], key
    assert set(c["step_ms"]) == {"min", "median", "max"}  # no prose
    assert "ivf" not in c["legs"]  # out of scope, full record only
    for nm in ("flat_f32_operands", "flat_B32", "flat_B1", "pq_flat", "hnsw", "hnsw_pq", "config1_gist_1000"):
        assert c["legs"][nm]["parity_ok"] is True, nm
    for nm in ("flat_f32_operands", "pq_flat", "hnsw", "hnsw_pq"):
        assert {"qps", "ms_per_step", "frac"} <= set(c["legs"][nm]), nm
    assert c["full_record"] == "gpurun_out/bench_full.json"


def test_compact_line_never_exceeds_the_limit():
    """prose creeping into the parts the line copies verbatim must cost the optional parts, not the parse"""
    full = json.load(open(FULL))
    fat = copy.deepcopy(full)
    for nm in list(fat["legs"]):
        for i in range(40):
            fat["legs"][f"{nm}_{i}"] = copy.deepcopy(fat["legs"][nm])
    line = bench.compact_line(fat, None)
    assert len(line) < 6144, len(line)
    c = json.loads(line)
    for key in CONTRACT:
        assert key in c, key


def test_compact_line_of_a_multi_rank_record():
    """N > 1 runs carry no legs, no cpu_baseline, no parity"""
    full = json.load(open(FULL))
    for key in ("legs", "cpu_baseline", "parity", "recall_at_10"):
        full.pop(key, None)
    full["n_gpus"] = 8
    c = json.loads(bench.compact_line(full, None))
    assert c["n_gpus"] == 8 and c["cpu_baseline"] is None and "legs" not in c and "full_record" not in c
