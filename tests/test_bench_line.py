"""bench.py's ONE stdout line (benchlib/line.py) on a full result as a real run produced it (tests/golden/bench_full_result.json: the
side file of a round-6 run on one MI355X): the contract's keys are there, it is numbers rather than prose, and it stays far below the
8,000 characters of stdout the driver keeps -- round 5's 20.7 KB line never reached the driver's record.  No GPU."""
import contextlib
import io
import json
import os

from benchlib.line import LINE_LIMIT, compact_line, emit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def full_result():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_result.json")))


def strings(x):
    if isinstance(x, dict):
        for v in x.values():
            yield from strings(v)
    elif isinstance(x, str):
        yield x


def test_compact_line_carries_the_contract_and_stays_small():
    full = full_result()
    line = compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT <= 4000 and len(json.dumps(full)) > 4 * len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["value"] == full["value"] and line["unit"] == "queries/sec" and "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic_over_algorithmic"] > 0.9
    assert rf["row_operand"] == "fp16" and rf["bytes_per_element"] == 2 and 0 < rf["frac_f32_rows_kernel"] < 1.0   # both fractions side by side
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["gpu_matches_cpu_bitwise"] is True and isinstance(cb["sample"], str)
    assert max(len(s) for s in strings(line)) <= 100          # numbers, not prose


def test_emit_prints_exactly_one_line_and_writes_the_side_file(tmp_path):
    full = full_result()
    side = tmp_path / "extra.json"
    out, err = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        emit(full, str(side))
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < LINE_LIMIT
    assert json.loads(out.getvalue()[-8000:]) == json.loads(lines[0])   # what survives the driver's tail buffer parses on its own
    assert json.load(open(side))["extra"].keys() == full["extra"].keys()
    assert "[bench-extra] extra.single_query" in err.getvalue()


def test_a_leg_that_failed_or_was_skipped_does_not_break_the_line():
    full = full_result()
    full["extra"] = {"cfg4_rank": {"failed": "RuntimeError: out of memory"}}
    full["cpu_baseline"] = None
    full["cpu_baseline_all_cores"] = None
    full["cpu_baseline_kmeans"] = None
    line = compact_line(full)
    assert line["cpu_baseline"] is None and line["extra"] == {"cfg4_rank": {"failed": True}}
    assert len(json.dumps(line)) < LINE_LIMIT
