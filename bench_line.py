"""The contract line of bench.py: ONE compact strict-JSON object (at most 4 KB) as the last line of stdout.

bench.py gathers a large detail record (sweeps, traces, notes, every side measurement).  Up to round 5 that record WAS
the line; it grew to 21.6 KB and the driver stopped parsing it (BENCH_r05.json: parsed null).  Now:

    stdout, last line   compact(detail)          the keys listed in COMPACT_KEYS and nothing else
    bench_detail.json   the whole detail record  next to bench.py (and under gpurun_out/ when that directory exists)
    stderr              the detail record again, one JSON line prefixed "bench_detail: "

This module is pure (json only): tests/test_bench_line.py builds the line from a recorded detail file on the CPU.
"""
import json
import os
import sys

MAX_LINE_BYTES = 4096
ROOT = os.path.dirname(os.path.abspath(__file__))

TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "batch_per_gpu", "batches_in_flight_per_gpu", "secret_table_access", "parallelism")
ROOFLINE_KEYS = ("bound", "kernel", "kernel_ms", "chip_share", "chip_ms_per_launch", "achieved", "peak", "unit", "frac",
                 "executed_mac32_per_launch", "canonical_frac", "step_executed_frac", "traffic",
                 "algorithmic_bytes_per_launch")
CPU_KEYS = ("value", "unit", "cores", "kind", "leg", "sample")
API_KEYS = ("sync_pageable", "sync_pinned", "pipelined_4_lanes", "ipcl_1_thread", "one_batch_in_flight")
COMPACT_KEYS = TOP_KEYS + ("config", "roofline", "cpu_baseline", "api_visible", "detail")


def _num(x, digits=6):
    """numbers to a bounded number of significant digits (the line is for a parser and a reader, not an archive)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None                     # strict JSON has no NaN / Infinity
        return float(f"{x:.{digits}g}")
    return x


def _clip(s, n):
    s = " ".join(str(s).split())
    return s if len(s) <= n else s[: n - 3] + "..."


def _rate(block, key="modexps_per_s"):
    """the one throughput number of a side-measurement block (None when the block is missing or failed)"""
    if isinstance(block, dict) and isinstance(block.get(key), (int, float)):
        return _num(float(block[key]))
    return None


def compact(detail):
    """detail record of a bench.py run -> the dict printed as the contract line"""
    out = {k: _num(detail.get(k)) for k in TOP_KEYS}
    cfg = detail.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in CONFIG_KEYS if k in cfg}
    if "workload" in out["config"]:
        out["config"]["workload"] = _clip(out["config"]["workload"], 260)
    if "parallelism" in out["config"]:
        out["config"]["parallelism"] = _clip(out["config"]["parallelism"], 100)
    r = detail.get("roofline") or {}
    roof = {k: _num(r.get(k)) for k in ROOFLINE_KEYS}
    if roof.get("kernel") is not None:
        roof["kernel"] = _clip(roof["kernel"], 200)
    if roof.get("chip_share") is None:
        roof["chip_share"] = 1.0            # a launch that has the chip to itself
    out["roofline"] = roof
    cb = detail.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {k: (_clip(cb[k], 200) if isinstance(cb[k], str) else _num(cb[k])) for k in CPU_KEYS if k in cb}
        if "error" in cb:
            out["cpu_baseline"]["error"] = _clip(cb["error"], 160)
    else:
        out["cpu_baseline"] = None
    api = detail.get("api_level") or {}
    av = {
        "sync_pageable": _rate(detail.get("end_to_end")),
        "sync_pinned": _rate(detail.get("end_to_end_pinned")),
        "pipelined_4_lanes": _rate(detail.get("end_to_end_pipelined_4_lanes")),
        "ipcl_1_thread": _rate(api),
        "one_batch_in_flight": _rate(detail.get("one_batch_in_flight")),
    }
    out["api_visible"] = av if any(v is not None for v in av.values()) else None
    out["detail"] = "bench_detail.json"
    line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    # a last guard: clip the long strings harder rather than ever print a line the driver cannot take
    for limit in (120, 60, 24):
        if len(line.encode()) <= MAX_LINE_BYTES:
            break
        for blk, key in ((out["config"], "workload"), (out["roofline"], "kernel"), (out["config"], "parallelism"),
                         (out.get("cpu_baseline") or {}, "sample")):
            if isinstance(blk.get(key), str):
                blk[key] = _clip(blk[key], limit)
        line = json.dumps(out, allow_nan=False, separators=(",", ":"))
    return out


def line_of(detail):
    s = json.dumps(compact(detail), allow_nan=False, separators=(",", ":"))
    if len(s.encode()) > MAX_LINE_BYTES:
        raise RuntimeError(f"bench line is {len(s.encode())} bytes, over {MAX_LINE_BYTES}")
    return s


def emit(detail, name="bench_detail.json"):
    """write the detail record beside the script (and to gpurun_out/ and stderr), print the compact line LAST on stdout"""
    blob = json.dumps(detail, default=str)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    f.write(blob + "\n")
            except OSError as e:                # a read-only checkout must not cost the line
                print(f"bench: cannot write {d}/{name}: {e}", file=sys.stderr)
    print("bench_detail: " + blob, file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(line_of(detail), flush=True)
