"""GPU check of the MFMA Montgomery-reduction experiment against model.py (run on an MI355X)."""
import ctypes, os, random, subprocess, sys, time
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from model import RedcModel, to_bytes

so = os.path.join(HERE, "libredc_mfma.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                    os.path.join(HERE, "redc_mfma.hip")], check=True)
torch.cuda.init()
lib = ctypes.CDLL(so)


class Args(ctypes.Structure):
    _fields_ = [("T", ctypes.c_void_p), ("U", ctypes.c_void_p), ("dump", ctypes.c_void_p), ("A1", ctypes.c_void_p),
                ("A2", ctypes.c_void_p), ("corr1", ctypes.c_void_p), ("corr2", ctypes.c_void_p),
                ("D", ctypes.c_int), ("tiles", ctypes.c_int), ("steps", ctypes.c_int), ("elems", ctypes.c_int),
                ("top", ctypes.c_int), ("reps", ctypes.c_int), ("clocks", ctypes.c_void_p)]


def lane_tiles(A, tiles, steps):
    """A: [rows, K] int64 -> [tiles][steps][64][16] int8 in MFMA A-operand lane order (row = lane % 16,
    k = 64 s + 16 (lane / 16) + byte)."""
    rows, K = A.shape
    P = np.zeros((tiles * 16, steps * 64), dtype=np.int8)
    P[:rows, :K] = A.astype(np.int8)
    out = np.zeros((tiles, steps, 64, 16), dtype=np.int8)
    for t in range(tiles):
        for s in range(steps):
            for lane in range(64):
                k0 = 64 * s + 16 * (lane // 16)
                out[t, s, lane] = P[16 * t + lane % 16, k0:k0 + 16]
    return out


def run(bits, D, elems, seed=1, reps=5):
    rng = random.Random(seed)
    N = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mdl = RedcModel(N, D)
    steps = (D + 63) // 64
    tiles = (max(D, mdl.A2.shape[0]) + 15) // 16
    A1 = lane_tiles(mdl.A1, tiles, steps)
    A2 = lane_tiles(mdl.A2, tiles, steps)
    corr1 = np.zeros(tiles * 16, dtype=np.int32); corr1[:D] = mdl.corr1
    corr2 = np.zeros(tiles * 16, dtype=np.int32); corr2[:mdl.A2.shape[0]] = mdl.corr2
    Ts = [rng.randrange(N * mdl.R) for _ in range(elems)]
    Ts[0] = 0; Ts[-1] = N * mdl.R - 1
    assert (2 * D) % 16 == 0
    flat = np.zeros(elems * 2 * D + 64, dtype=np.uint8)          # slack: the K padding reads past the last T_lo
    for i, T in enumerate(Ts):
        flat[i * 2 * D:(i + 1) * 2 * D] = to_bytes(T, 2 * D).astype(np.uint8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dT, dA1, dA2, dc1, dc2 = dev(flat), dev(A1), dev(A2), dev(corr1), dev(corr2)
    dU = torch.zeros(elems * (D + 8), dtype=torch.uint8, device="cuda")
    waves = (elems + 15) // 16
    ddump = torch.zeros(waves * 2 * tiles * 64 * 4, dtype=torch.int32, device="cuda")
    dclk = torch.zeros(waves * 5, dtype=torch.int64, device="cuda")
    a = Args(dT.data_ptr(), dU.data_ptr(), ddump.data_ptr(), dA1.data_ptr(), dA2.data_ptr(), dc1.data_ptr(),
             dc2.data_ptr(), D, tiles, steps, elems, mdl.TOP, 1, dclk.data_ptr())
    lds = 16 * tiles * 16 * 4 + 16 * steps * 64
    rc = lib.redc_launch(ctypes.byref(a), waves, lds)
    assert rc == 0, rc
    U = dU.cpu().numpy().reshape(elems, D + 8)
    check = range(elems) if elems <= 256 else list(range(64)) + list(range(elems - 64, elems))
    bad = sum(int.from_bytes(U[i].tobytes(), "little") != mdl.reduce(Ts[i]) for i in check)
    if bad:
        d = ddump.cpu().numpy().reshape(waves, 2, tiles, 64, 4)
        t = to_bytes(Ts[1], 2 * D)
        want = mdl.A1 @ (t[:D] - 128) + mdl.corr1
        got = np.zeros(tiles * 16, dtype=np.int64)
        for tl in range(tiles):
            for lane in range(64):
                if lane % 16 == 1:
                    got[16 * tl + 4 * (lane // 16): 16 * tl + 4 * (lane // 16) + 4] = d[0, 0, tl, lane]
        print("GEMM1 element 1: want", want[:8], "got", got[:8], "equal:", np.array_equal(want, got[:D]))
    a.dump = None
    a.reps = 4
    lib.redc_launch(ctypes.byref(a), waves, lds)
    clk = dclk.cpu().numpy().reshape(waves, 5).astype(np.float64).mean(axis=0)
    print("  cycles per 16-element reduction with warm caches (mean over waves): GEMM1 %.0f, carry1 %.0f, GEMM2 %.0f, "
          "final %.0f, total %.0f" % tuple(clk))
    a.reps = 1
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); lib.redc_launch(ctypes.byref(a), waves, lds); best = min(best, time.perf_counter() - t0)
    print(f"{bits}-bit modulus, D={D}: {elems} reductions, {bad} of {len(check)} checked wrong; {best * 1e3:.3f} ms "
          f"({elems / best / 1e6:.1f} M reductions/s, serial carry stages included)")
    return bad


if __name__ == "__main__":
    bad = run(2048, 264, 64) + run(2088, 272, 16384) + run(4096, 528, 4096)
    sys.exit(1 if bad else 0)
