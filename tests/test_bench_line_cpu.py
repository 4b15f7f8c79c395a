"""bench.py's stdout contract: ONE compact JSON line the driver can parse — round 5's line had grown to 20.7 KB
and BENCH_r05.json recorded `parsed: null`. compact_line() is a pure function of the full line object; the fixtures are full line
objects of round 5's tree kept under profiles/."""
import io
import json
import os
import sys
import contextlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fixture(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", ["r5_final_bench7b_b64_benchline.json", "r5_final_bench13b_b8_benchline.json",
                                  "r5_final_train7b_b8_benchline.json"])
def test_compact_line_fits_the_budget_and_keeps_the_contract(name):
    full = _fixture(name)
    if full.get("parity"):
        full["parity"].setdefault("timed_mode", "fp32" if full["dtype"] == "f32" else "bf16")
    assert len(json.dumps(full)) > bench.LINE_BUDGET_BYTES      # the fixture IS one of the lines that were too long
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_BUDGET_BYTES, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-3) and line["config"]["workload"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(full["roofline"]["achieved"] / full["roofline"]["peak"], rel=1e-3)
    if full.get("cpu_baseline"):
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k
    if full.get("parity"):
        p = line["parity"]
        assert p["timed_mode"] in ("bf16", "fp32", "bf16_fp32_stream", "bf16_fp32_stream_f32neck")
        assert 0.9 < p["mask_iou_min"] <= 1.0 and isinstance(p["gate_failed"], list)
        want = full["parity"]["full_frame"][p["timed_mode"]]["mask_iou_min"]
        assert p["mask_iou_min"] == pytest.approx(want, abs=1e-5)        # the IoU printed belongs to the mode that was timed
        assert p["meets_iou_0.999"] == (want >= 0.999)


def test_compact_line_survives_absurdly_long_free_text():
    full = _fixture("r5_final_bench7b_b64_benchline.json")
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    assert len(json.dumps(bench.compact_line(full))) <= bench.LINE_BUDGET_BYTES
    # even a config object stuffed with junk cannot keep the line from being printed within the budget
    full["config"].update({"junk%d" % i: "x" * 200 for i in range(40)})
    line = bench.compact_line(full)
    assert len(json.dumps(line)) <= bench.LINE_BUDGET_BYTES and line["value"] == full["value"] and "frac" in line["roofline"]


def test_stub_run_prints_exactly_one_line_within_the_budget(tmp_path, monkeypatch):
    env_keys = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    for k in env_keys:
        monkeypatch.delenv(k, raising=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", "1", "--steps", "2", "--warmup", "1", "--stub-step-ms", "5", "--batch", "3"])
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_BUDGET_BYTES
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["config"]["stub"] is True and line["cpu_baseline"] is None
    detail = os.path.join(ROOT, line["detail"])
    assert os.path.exists(detail) and json.load(open(detail))["steps"] == 2
