"""CPU: the ONE stdout line of bench.py is held to the driver's contract.

Round 5's line grew to 24.5 KB with free text in `scaling` and the driver could not parse it (BENCH_r05.json:
`parsed: null`).  The line is now `bench.driver_line(report)`; these tests build it from canned reports — round 5's
own full report (profiles/r05_bench_final.json: the shape `main()` assembles, 14 per-shard dicts and all) and a
synthetic worst case — and assert what the driver needs: strict JSON, one line, bounded length, the contract's keys
with the types rounds 1-4 used, `scaling` one of the two words, `roofline` and `cpu_baseline` complete.
"""
import io
import json
import math
import os
from contextlib import redirect_stdout

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT_TYPES = {
    "metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
    "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
    "config": dict, "roofline": dict, "cpu_baseline": dict,
}


def canned_report():
    with open(os.path.join(ROOT, "profiles", "r05_bench_final.json")) as f:
        rep = json.load(f)
    rep["scaling"] = "weak"
    return rep


def check_line(text, expect_cpu=True):
    assert "\n" not in text and len(text.encode()) <= bench.LINE_MAX_BYTES <= 12 * 1024

    def no_constants(name):  # json.loads accepts NaN / Infinity unless told otherwise
        raise AssertionError("non-finite constant %s in the line" % name)

    line = json.loads(text, parse_constant=no_constants)
    for k, t in CONTRACT_TYPES.items():
        if k == "cpu_baseline" and not expect_cpu:
            continue
        assert k in line, k
        assert isinstance(line[k], t), (k, type(line[k]))
        if t is int:
            assert not isinstance(line[k], bool)
    assert "vs_baseline" in line and line["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert line["scaling"] in ("weak", "strong")
    assert line["higher_is_better"] is True
    assert isinstance(line["config"].get("workload"), str) and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma")
    for k in ("achieved", "peak", "frac"):
        assert isinstance(roof[k], (int, float)) and math.isfinite(roof[k])
    assert roof["unit"] in ("GB/s", "TFLOP/s") and "traffic" in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    if expect_cpu:
        cb = line["cpu_baseline"]
        assert isinstance(cb["value"], (int, float)) and cb["kind"] in ("reference", "port")
        assert isinstance(cb["cores"], int) and isinstance(cb["sample"], str) and isinstance(cb["unit"], str)

    def walk(o, depth=0):
        assert depth <= 4, "the line nests deeper than a driver should have to follow"
        if isinstance(o, dict):
            for v in o.values():
                walk(v, depth + 1)
        elif isinstance(o, list):
            assert len(o) <= 8
            for v in o:
                walk(v, depth + 1)
        elif isinstance(o, str):
            assert len(o) <= 128
        elif isinstance(o, float):
            assert math.isfinite(o)

    walk(line)
    return line


def test_line_from_round5_report():
    rep = canned_report()
    rep["roofline"]["bound"] = "hbm"
    assert len(json.dumps(rep)) > 20_000  # the report itself is the 24.5 KB that broke round 5
    line = check_line(bench.line_text(rep))
    # the contract's two figures are carried exactly, not rounded
    assert line["value"] == rep["value"] and line["ms_per_step"] == rep["ms_per_step"]
    # per-config sub-records and the projection survive in compact form
    assert set(line["configs"]) == set(rep["configs"])
    for rec in line["configs"].values():
        assert set(rec) >= {"value", "ms_per_step", "roofline", "cpu_baseline"} and "frac" in rec["roofline"]
    assert set(line["scaling_projection"]["G"]) == {"2", "4", "8"}
    assert "per_shard" not in json.dumps(line)
    assert line["cpu_baseline"]["gpu_over_cpu"]["T2_pcie_inclusive"] > 1


def test_free_text_scaling_is_refused():
    rep = canned_report()
    rep["scaling"] = "n/a (1 GPU)"
    with pytest.raises(ValueError):
        bench.line_text(rep)


def test_non_finite_numbers_do_not_reach_the_line():
    rep = canned_report()
    rep["roofline"]["bound"] = "hbm"
    rep["roofline"]["traffic"] = float("nan")
    rep["ms_per_step_counts_only"] = float("inf")
    line = check_line(bench.line_text(rep))
    assert line["roofline"]["traffic"] is None and line["ms_per_step_counts_only"] is None


def test_worst_case_report_stays_bounded():
    """Long notes everywhere, 64 logical-shard counts' worth of per-shard records: the line does not grow with them."""
    rep = canned_report()
    rep["roofline"]["bound"] = "hbm"
    long = "x" * 5000
    rep["config"]["workload"] = long
    rep["config"]["sharding"] = long
    rep["roofline"]["note"] = long
    rep["cpu_baseline"]["sample"] = long
    rep["cpu_baseline"]["quota_note"] = long
    g8 = rep["scaling_projection"]["shards"]["8"]
    g8["per_shard"] = g8["per_shard"] * 8
    rep["kernels"]["groups"] = rep["kernels"]["groups"] * 10
    check_line(bench.line_text(rep))


def test_sharded_and_per_config_reports():
    """The --gpus N line (no cpu_baseline: rank 0 times it at N = 1 only) and a --workload line."""
    rep = canned_report()
    rep["roofline"]["bound"] = "hbm"
    multi = {k: rep[k] for k in bench.CONTRACT_KEYS if k != "cpu_baseline"}
    multi.update(n_gpus=8, scaling="strong")
    multi["weak_scaling_batch"] = {"value": 1.0e9, "unit": "proofs/s", "ms_per_step": 1.0, "scaling": "weak",
                                   "config": {"workload": "w"}, "kernels_ms_per_step": {}, "window": "T3"}
    line = check_line(bench.line_text(multi), expect_cpu=False)
    assert line["n_gpus"] == 8 and line["weak_scaling_batch"]["scaling"] == "weak"


def test_emit_prints_one_line_last_and_writes_the_report(tmp_path, monkeypatch):
    rep = canned_report()
    rep["roofline"]["bound"] = "hbm"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(rep)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1
    line = check_line(lines[-1])
    assert "bench_detail.json" in line["detail"]
    for where in (tmp_path, tmp_path / "gpurun_out"):
        with open(where / "bench_detail.json") as f:
            full = json.load(f)
        assert len(full["scaling_projection"]["shards"]["8"]["per_shard"]) == 8  # the whole report is kept, off stdout
