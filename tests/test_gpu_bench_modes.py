"""bench.py's multi-rank mode on the one GPU of a test box: the command line the driver uses for N > 1
(`python -m torch.distributed.run ... bench.py --gpus N`), with one rank over RCCL and with two ranks sharing the device
(gloo; RCCL refuses two ranks on one GPU).  Checks the contract line, not the speed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(nproc, port, extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "4", "--warmup", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 prints ONE JSON line
    assert r.stdout.rstrip("\n").endswith(lines[0]) and len(lines[0].encode()) <= 4096     # last on stdout, compact (bench_line.py)
    line = json.loads(lines[0])
    detail = json.load(open(os.path.join(ROOT, line["detail"])))      # the whole record, written beside the script
    assert detail["value"] == pytest.approx(line["value"], rel=1e-5)
    line["detail_record"] = detail
    return line


def _check(d, n):
    assert d["n_gpus"] == n and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["unit"] == "modexps/s" and d["higher_is_better"] is True and d["value"] > 0
    assert abs(d["value"] - 3 * 8192 * n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert d["roofline"]["bound"] == "int-alu" and 0 < d["roofline"]["frac"] < 1
    assert d["detail_record"]["config"]["resident_ciphertext_form"].startswith("pair rows")      # the same step as the pool path
    assert d["config"]["secret_table_access"] == "indexed"
    assert d["config"]["batches_in_flight_per_gpu"] == 4                          # (round 5: four lanes, a quarter of the chip each)


def test_one_rank_over_rccl():
    _check(_run(1, 29521, {}), 1)


def test_two_ranks_on_one_device_over_gloo():
    _check(_run(2, 29522, {"BENCH_SINGLE_DEVICE": "1", "BENCH_DIST_BACKEND": "gloo"}), 2)
