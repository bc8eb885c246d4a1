"""Batch sharding across the GPUs of one node.

Every element of every batch operation is independent (reference ipcl/mod_exp.cpp:612-633 already
treats chunks independently), so rank g of G processes the contiguous slice
[g*N/G, (g+1)*N/G) of the flat [element][limb] arrays; output order is preserved by construction.
The only collective on the data path is the one-off broadcast of the key material from rank 0
(torch.distributed: backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests);
results are gathered only when a caller on rank 0 wants the whole batch back.
"""
import numpy as np


def shard_bounds(count, world, rank):
    """Contiguous slice of rank; the remainder goes to the last rank."""
    per = count // world
    lo = rank * per
    hi = count if rank == world - 1 else lo + per
    return lo, hi


def broadcast_key_words(words, src=0, device=None):
    """Broadcast a uint64 numpy vector (key material) from rank `src` to all ranks."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(words, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t.cpu().numpy().view(np.uint64)


def gather_rows(local_rows, dst=0):
    """Collect every rank's output slice on `dst` in rank order (rows may differ per rank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_rows
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_rows)
    return np.concatenate(parts, axis=0) if dist.get_rank() == dst else None
