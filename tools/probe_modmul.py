"""One CT+CT launch of config 5's per-GPU shard (for rocprofv3 counter passes); diagnostics only."""
import json, sys
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pailliercryptolib_amd as pa
from pailliercryptolib_amd import torch_ops as T
pa.initialize()
k = json.load(open(os.path.join(ROOT, "tests/golden/iso_kat.json")))
N = int(k["p"], 16) * int(k["q"], 16)
rng = np.random.default_rng(7)
def rows(count, words, mask):
    a = np.frombuffer(rng.bytes(count * words * 8), dtype=np.uint64).reshape(count, words).copy()
    a[:, -1] &= np.uint64(mask)
    return a
da, db = T.to_device(rows(131072, 64, (1 << 62) - 1)), T.to_device(rows(131072, 64, (1 << 62) - 1))
for _ in range(3):
    T.mod_mul(da, db, N * N)
torch.cuda.synchronize()
