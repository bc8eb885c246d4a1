"""bench.py's contract line (bench_line.py): compact, strict JSON, at most 4 KB, a fixed key set -- built here from
detail records of earlier runs (profiles/r05_bench_*.json are what bench.py gathered in round 5, when the record itself was
printed as the line, grew to 21.6 KB and the driver could no longer parse it: BENCH_r05.json `parsed: null`)."""
import io
import json
import os
import contextlib

import pytest

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ["r05_bench_n1.json", "r05_bench_config4_n1.json", "r05_bench_config5_n1.json", "r05_bench_n8_pool_1dev.json",
           "r04_bench_n1.json", "r05_bench_n1_one_in_flight.json"]


def _strict(s):
    def no_constants(x):
        raise ValueError("NaN / Infinity are not JSON: " + x)
    return json.loads(s, parse_constant=no_constants)


@pytest.mark.parametrize("name", RECORDS)
def test_line_is_small_strict_and_complete(name):
    detail = json.load(open(os.path.join(ROOT, "profiles", name)))
    s = bench_line.line_of(detail)
    assert "\n" not in s and len(s.encode()) <= 4096
    d = _strict(s)
    assert set(d) == set(bench_line.COMPACT_KEYS)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "data"):
        assert d[k] == detail[k] or abs(d[k] - detail[k]) <= 1e-5 * abs(detail[k])
    assert d["higher_is_better"] is True and d["vs_baseline"] is None
    assert set(d["roofline"]) == set(bench_line.ROOFLINE_KEYS)
    assert d["roofline"]["bound"] == "int-alu" and d["roofline"]["peak"] == 39.32
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["kernel_ms"] > 0
    assert "workload" in d["config"] and ("batch_per_gpu" in d["config"] or "config" in name)   # (configs 4/5 name it since round 6)
    assert not any(isinstance(v, str) and len(v) > 300 for blk in (d, d["config"], d["roofline"]) for v in blk.values())


def test_headline_record_carries_cpu_baseline_and_the_five_api_numbers():
    detail = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    d = bench_line.compact(detail)
    assert set(d["cpu_baseline"]) == set(bench_line.CPU_KEYS)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16
    assert set(d["api_visible"]) == set(bench_line.API_KEYS)
    assert d["api_visible"]["sync_pageable"] == pytest.approx(detail["end_to_end"]["modexps_per_s"], rel=1e-5)
    assert d["api_visible"]["ipcl_1_thread"] == pytest.approx(detail["api_level"]["modexps_per_s"], rel=1e-5)
    assert d["roofline"]["chip_share"] == 0.25 and d["roofline"]["step_executed_frac"] == 0.77
    assert d["config"]["batches_in_flight_per_gpu"] == 4 and d["config"]["secret_table_access"] == "indexed"


def test_hostile_record_still_gives_a_parseable_line():
    detail = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    detail["roofline"]["kernel"] = "k" * 9000
    detail["config"]["workload"] = "w " * 9000
    detail["roofline"]["frac"] = float("nan")
    detail["cpu_baseline"] = {"error": "e" * 5000}
    detail["end_to_end"] = {"error": "boom"}
    s = bench_line.line_of(detail)
    d = _strict(s)
    assert len(s.encode()) <= 4096 and d["roofline"]["frac"] is None and d["api_visible"]["sync_pageable"] is None


def test_emit_prints_the_line_last_and_writes_the_detail(tmp_path, monkeypatch):
    detail = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    monkeypatch.setattr(bench_line, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        print("a banner some library wrote earlier")
        bench_line.emit(detail)
    last = out.getvalue().rstrip("\n").split("\n")[-1]
    assert _strict(last)["value"] == pytest.approx(detail["value"])
    assert json.load(open(tmp_path / "bench_detail.json")) == detail
    assert err.getvalue().startswith("bench_detail: ")
