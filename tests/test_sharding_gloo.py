"""CPU suite, world_size 2 over gloo: the multi-GPU path's sharding + key broadcast logic
(pailliercryptolib_amd/sharding.py).  The per-shard compute is stood in for by the oracle here
(no GPU in this container); on the GPU box bench.py runs the same code path over RCCL."""
import os
import random
import socket

import numpy as np
import torch.multiprocessing as mp

from pailliercryptolib_amd.sharding import shard_bounds


def test_shard_bounds_cover_everything():
    for count in (0, 1, 7, 8, 9, 2100, 65536):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(count, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import paillier_oracle as orc
    from pailliercryptolib_amd.limbs import ints_to_limbs, limbs_to_ints
    from pailliercryptolib_amd.sharding import broadcast_key_words, gather_rows, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0 owns the key; the others start with garbage and must receive it
    rng = random.Random(5)
    key = ints_to_limbs([0x10001, 0xABCDEF], 2).reshape(-1) if rank == 0 else np.zeros(4, dtype=np.uint64)
    key = broadcast_key_words(key)
    assert limbs_to_ints(key.reshape(2, 2)) == [0x10001, 0xABCDEF]
    # the same seeded batch everywhere; each rank works only on its slice
    count = 37
    mod = (rng.getrandbits(256) | (1 << 255) | 1)
    base = [rng.randrange(mod) for _ in range(count)]
    exp = [rng.getrandbits(64) for _ in range(count)]
    lo, hi = shard_bounds(count, world, rank)
    local = ints_to_limbs(orc.mod_exp_batch(base[lo:hi], exp[lo:hi], [mod] * (hi - lo)), 4)
    full = gather_rows(local)
    if rank == 0:
        q.put(limbs_to_ints(full) == [pow(b, e, mod) for b, e in zip(base, exp)])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
