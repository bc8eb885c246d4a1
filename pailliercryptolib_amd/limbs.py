"""Conversions between Python integers and the C-ABI batch layout
([element][limb] little-endian uint64, numpy arrays)."""
import numpy as np


def ints_to_limbs(values, words):
    """list of non-negative ints -> np.uint64 array [len(values), words] (row-major)."""
    nbytes = words * 8
    buf = b"".join(int(v).to_bytes(nbytes, "little") for v in values)
    return np.frombuffer(buf, dtype="<u8").reshape(len(values), words).copy()


def limbs_to_ints(arr):
    arr = np.ascontiguousarray(arr, dtype="<u8")
    if arr.ndim == 1:
        arr = arr.reshape(1, -1)
    nbytes = arr.shape[1] * 8
    raw = arr.tobytes()
    return [int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "little") for i in range(arr.shape[0])]
