"""GPU check + stage timing of the LDS-resident variant (redc_mfma_lds.hip) against model.py."""
import ctypes, os, random, subprocess, sys, time
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from model import RedcModel, to_bytes

STEPS, TILES = 5, 17
NDIAG = TILES + 4 * (STEPS - 1)
so = os.path.join(HERE, "libredc_mfma_lds.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                    os.path.join(HERE, "redc_mfma_lds.hip")], check=True)
torch.cuda.init()
lib = ctypes.CDLL(so)


class Args(ctypes.Structure):
    _fields_ = [("T", ctypes.c_void_p), ("U", ctypes.c_void_p), ("diag1", ctypes.c_void_p), ("diag2", ctypes.c_void_p),
                ("corr1", ctypes.c_void_p), ("corr2", ctypes.c_void_p), ("D", ctypes.c_int), ("elems", ctypes.c_int),
                ("c0", ctypes.c_int), ("reps", ctypes.c_int), ("clocks", ctypes.c_void_p)]


def diag_tiles(digit):
    """[NDIAG][64][16] int8: tile j holds A[i][k] = digit(16 (j - 4 (STEPS-1)) + i - k) in MFMA lane order."""
    out = np.zeros((NDIAG, 64, 16), dtype=np.int8)
    for j in range(NDIAG):
        delta = 16 * (j - 4 * (STEPS - 1))
        for lane in range(64):
            i = lane % 16
            for b in range(16):
                out[j, lane, b] = digit(delta + i - (16 * (lane // 16) + b))
    return out


def run(bits, D, elems, seed=1):
    rng = random.Random(seed)
    N = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    mdl = RedcModel(N, D)
    c0 = 16 * ((D - 5) // 16)
    assert D <= STEPS * 64 and 2 * D - c0 <= TILES * 16 and D <= TILES * 16
    d1 = lambda x: int(mdl.np_d[x]) if 0 <= x < D else 0
    d2 = lambda x: int(mdl.n_d[c0 + x]) if 0 <= c0 + x <= D else 0
    diag1, diag2 = diag_tiles(d1), diag_tiles(d2)
    # corrections 128 * sum over k < D of the matrix row
    corr1 = np.array([128 * sum(d1(i - k) for k in range(D)) for i in range(TILES * 16)], dtype=np.int32)
    corr2 = np.array([128 * sum(d2(r - k) for k in range(D)) for r in range(TILES * 16)], dtype=np.int32)
    Ts = [rng.randrange(N * mdl.R) for _ in range(elems)]
    Ts[0] = 0; Ts[-1] = N * mdl.R - 1
    flat = np.zeros(elems * 2 * D + 128, dtype=np.uint8)
    for i, T in enumerate(Ts):
        flat[i * 2 * D:(i + 1) * 2 * D] = to_bytes(T, 2 * D).astype(np.uint8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dT, dd1, dd2, dc1, dc2 = dev(flat), dev(diag1), dev(diag2), dev(corr1), dev(corr2)
    dU = torch.zeros(elems * (D + 8), dtype=torch.uint8, device="cuda")
    wgs = (elems + 63) // 64
    dclk = torch.zeros(wgs * 4 * 5, dtype=torch.int64, device="cuda")
    a = Args(dT.data_ptr(), dU.data_ptr(), dd1.data_ptr(), dd2.data_ptr(), dc1.data_ptr(), dc2.data_ptr(), D, elems, c0,
             1, dclk.data_ptr())
    rc = lib.redc_lds_launch(ctypes.byref(a), wgs)
    assert rc == 0, rc
    U = dU.cpu().numpy().reshape(elems, D + 8)
    check = range(elems) if elems <= 256 else list(range(64)) + list(range(elems - 64, elems))
    bad = sum(int.from_bytes(U[i].tobytes(), "little") != mdl.reduce(Ts[i]) for i in check)
    a.reps = 4
    lib.redc_lds_launch(ctypes.byref(a), wgs)
    clk = dclk.cpu().numpy().reshape(-1, 5).astype(np.float64)
    clk = clk[clk[:, 4] > 0].mean(axis=0)
    print(f"{bits}-bit modulus, D={D}, {elems} reductions: {bad} of {len(check)} checked wrong")
    print("  cycles per 16-element reduction, one wave: GEMM1 %.0f (90 MFMA), carry1 %.0f (serial), GEMM2 %.0f (85 MFMA), "
          "final %.0f (serial), total %.0f" % tuple(clk))
    return bad


if __name__ == "__main__":
    bad = run(2048, 264, 64) + run(2088, 264, 65536)
    sys.exit(1 if bad else 0)
