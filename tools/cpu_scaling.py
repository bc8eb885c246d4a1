"""CPU-baseline sanity: how the C oracle scales with threads on the host (not part of the bench)."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from oracle import c_oracle
    rng = np.random.default_rng(1)
    W, E, n = 64, 16, int(sys.argv[1])
    mod = np.frombuffer(rng.bytes(W * 8), dtype=np.uint64).copy(); mod[0] |= 1; mod[-1] |= 1 << 63
    base = np.frombuffer(rng.bytes(n * W * 8), dtype=np.uint64).reshape(n, W).copy(); base[:, -1] &= (1 << 62) - 1
    exp = np.frombuffer(rng.bytes(n * E * 8), dtype=np.uint64).reshape(n, E).copy()
    t0 = time.perf_counter(); c_oracle.modexp_batch(base, exp, mod); dt = time.perf_counter() - t0
    print(f"threads={os.environ.get('OMP_NUM_THREADS')} n={n}: {dt:.2f}s  {n/dt:.1f} modexp/s (4096-bit mod, 1024-bit exp)  per-thread {n/dt/int(os.environ.get('OMP_NUM_THREADS')):.1f}/s")
else:
    for t, n in ((1, 16), (8, 128), (32, 512), (64, 1024), (128, 2048), (256, 4096)):
        subprocess.run([sys.executable, __file__, str(n)], env=dict(os.environ, OMP_NUM_THREADS=str(t)))
