"""The in-process device pool (pgpu_init_all): sharded host-pointer entry points, sharded resident batches with
Montgomery-domain ciphertext chains, key replication.  A pool is a process-wide choice, so every scenario runs
tests/pool_worker.py in its own process; on the 1-GPU test box the pool entries wrap around the one physical
device (PGPU_POOL_OVERSUBSCRIBE=1): same code paths, per-entry streams / allocators / key copies."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(scenario, ndev, timeout=600, **env):
    e = dict(os.environ, PGPU_POOL_OVERSUBSCRIBE="1", PGPU_MIN_SHARD="8", **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pool_worker.py"), scenario, str(ndev)],
                       capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # (librccl prints a banner of its own)
    return json.loads(lines[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [1, 3])
def test_pool_host_api_sharded(engine, ndev):
    res = run_worker("host_api", ndev)
    assert res["pool"] == ndev
    assert res["transport"] == ("single" if ndev == 1 else "memcpy")   # entries share the physical GPU: no RCCL
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [1, 4])
def test_pool_resident_batch_chain(engine, ndev):
    res = run_worker("batch_chain", ndev)
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, bad


@pytest.mark.gpu
def test_rccl_broadcast_path_on_one_device(engine):
    """librccl is dlopen'ed and a one-rank communicator replicates the key images (self-test of the loader, the
    group call and the stream hand-over; the multi-GPU broadcast is the same call with more ranks)."""
    res = run_worker("host_api", 1, PGPU_RCCL_FORCE="1")
    assert res["transport"] == "rccl", res
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("geo410", ["1", "0"])
def test_key_with_unequal_square_widths(engine, geo410):
    """1008- and 1588-bit keys whose p^2 is one bit shorter than q^2: with PGPU_GEO_410=0 the two moduli straddle the
    unit-quotient threshold of (4,9) / (4,14) and must fall back to the same loop form."""
    res = run_worker("unequal_key", 1, PGPU_GEO_410=geo410)
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, bad


@pytest.mark.gpu
def test_process_exits_without_shutdown(engine):
    """A process that never calls pgpu_shutdown must still exit promptly (the worker lanes are stopped by an
    atexit hook); a hang here would burn the whole gpurun limit."""
    res = run_worker("no_terminate", 2, timeout=120)
    assert res["ok"]["enc"]


@pytest.mark.gpu
def test_corrupted_replica_is_detected_and_repaired(engine):
    """A wrong-but-successful replication (debug hook: one byte of a key image flipped on the last pool entry) is seen by
    the read-back check of rt::Replicated::upload, rewritten from the host and counted; results stay correct."""
    res = run_worker("corrupt_replica", 3)
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, (bad, res)


@pytest.mark.gpu
def test_objects_of_a_previous_pool_are_refused(engine):
    """pgpu_shutdown + pgpu_init_all with a different device set: stale keys fail with a clear error, the caches of the
    key-less seam (perfect-square moduli included) are rebuilt for the new pool."""
    res = run_worker("reinit", 2)
    assert res["pool"] == 3
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, (bad, res)


@pytest.mark.gpu
def test_fixed_base_tables_stay_within_budget(engine):
    """64 DJN keys take turns on one GPU under a 64 MiB table budget: LRU eviction keeps the live bytes under the cap and
    every ciphertext identical to the oracle's."""
    res = run_worker("fb_budget", 1, timeout=900)
    bad = [k for k, v in res["ok"].items() if not v]
    assert not bad, (bad, res)
