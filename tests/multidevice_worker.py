"""Scenario runner for tests/test_gpu_multidevice.py: a REAL pool over every visible GPU (no oversubscription unless asked),
in its own process.  BASELINE configs[3] (65536 x 3072-bit encrypt + CRT decrypt) and configs[4] (1 M x 2048-bit CT+CT and
CT x PT) sharded over the pool, every element checked by SHA-256 against the C oracle (the checker); how the key images
travelled; what the replication self-check saw; the decrypt-kernel time of every GPU.  Prints one JSON line.
usage: multidevice_worker.py <n_devices> <count_config4> <count_config5>
The fan-out this stands in for: /root/reference/module/heqat/heqat/ctrl.c:500-529 (one worker per accelerator instance)."""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
K_MODEXP = 1


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def rand_rows(rng, count, words, top_mask=None):
    a = np.frombuffer(rng.bytes(count * words * 8), dtype=np.uint64).reshape(count, words).copy()
    if top_mask is not None:
        a[:, -1] &= np.uint64(top_mask)
    return a


def main():
    ndev, count4, count5 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    from oracle import c_oracle, paillier_oracle as orc
    from pailliercryptolib_amd import _capi
    from pailliercryptolib_amd.limbs import ints_to_limbs
    import pailliercryptolib_amd as pa
    L = _capi.lib()
    _capi.check(L.pgpu_init_all(ndev))
    pa.engine._initialized = True
    ctypes.CDLL(None).fflush(None)
    res = {"visible": L.pgpu_device_count(), "pool": L.pgpu_pool_size(), "transport": L.pgpu_pool_transport().decode(),
           "rccl_note": L.pgpu_rccl_note().decode(), "ok": {}}
    ok = res["ok"]
    c_oracle.set_threads(min(c_oracle.lib().orc_max_threads(), c_oracle.usable_cpus()))
    be = c_oracle.ifma_modexp_batch if c_oracle.ifma_lib() is not None else (
        c_oracle.openssl_modexp_batch if c_oracle.openssl_lib() is not None else c_oracle.modexp_batch)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def up(a):
        h = ctypes.c_void_p()
        a = np.ascontiguousarray(a, dtype=np.uint64)
        _capi.check(L.pgpu_batch_upload(vp(a), a.shape[0], a.shape[1], a.shape[1], ctypes.byref(h)))
        return h

    def down(h):
        out = np.empty((L.pgpu_batch_count(h), L.pgpu_batch_words(h)), dtype=np.uint64)
        _capi.check(L.pgpu_batch_download(h, vp(out)))
        return out

    def op(fn, *a):
        h = ctypes.c_void_p()
        _capi.check(fn(*a, ctypes.byref(h)))
        return h

    # the cut every entry point follows: contiguous shards in order, sizes within one of each other, one per pool entry
    nsh, bounds = ctypes.c_int(), (ctypes.c_size_t * (ndev + 1))()
    _capi.check(L.pgpu_shard_plan(count4, ndev, ctypes.byref(nsh), bounds))
    sizes = [bounds[i + 1] - bounds[i] for i in range(nsh.value)]
    ok["shard_plan"] = nsh.value == ndev and bounds[0] == 0 and bounds[nsh.value] == count4 and max(sizes) - min(sizes) <= 1
    res["shards_config4"] = sizes

    # ---- configs[3]: 3072-bit DJN key, count4 elements over the pool ----
    if count4:
        bits = 3072
        case = [c for c in json.load(open(os.path.join(GOLD, "seeded_vectors.json")))["cases"] if c["bits"] == bits and c["djn"]][0]
        p, q, hs = int(case["p"], 16), int(case["q"], 16), int(case["hs"], 16)
        n = p * q
        nw, pw = bits // 64, bits // 128
        rng = np.random.default_rng(4004)
        m = rand_rows(rng, count4, nw, (1 << 62) - 1)
        r = rand_rows(rng, count4, pw)
        osk = orc.PrivateKey(n, p, q)
        n_l, hs_l = ints_to_limbs([n], nw)[0], ints_to_limbs([hs], 2 * nw)[0]
        sk_l = [ints_to_limbs([v], pw)[0] for v in (osk.p, osk.q, osk.hp, osk.hq, osk.pinv)]
        c_cpu = c_oracle.paillier_encrypt_with(be, n_l, hs_l, m, r)
        want_c, want_m = sha(c_cpu), sha(m)
        ok["c4_oracle_round_trip"] = sha(c_oracle.paillier_decrypt_crt_with(be, *sk_l, c_cpu)) == want_m
        ver0, rep0 = ctypes.c_uint64(), ctypes.c_uint64()
        L.pgpu_replication_stats(ctypes.byref(ver0), ctypes.byref(rep0))
        pk, sk = pa.PublicKey(n, bits, hs=hs), pa.PrivateKey(p, q)
        c_gpu = np.empty((count4, 2 * nw), dtype=np.uint64)
        m_gpu = np.empty((count4, nw), dtype=np.uint64)
        _capi.check(L.pgpu_paillier_encrypt(pk._h, vp(m), nw, nw, vp(r), pw, pw, 64 * pw, vp(c_gpu), count4))
        ok["c4_host_encrypt"] = sha(c_gpu) == want_c
        _capi.check(L.pgpu_paillier_decrypt_crt(sk._h, vp(c_gpu), vp(m_gpu), count4))
        ok["c4_host_decrypt"] = sha(m_gpu) == want_m
        bm, br = up(m), up(r)
        bc = op(L.pgpu_batch_encrypt, pk._h, bm, br, 64 * pw)
        _capi.check(L.pgpu_synchronize())
        # decrypt-kernel time per GPU: HIP events on each GPU's own batch stream, three launches each
        _capi.check(L.pgpu_set_timing(1))
        outs = [op(L.pgpu_batch_decrypt_crt, sk._h, bc) for _ in range(3)]
        _capi.check(L.pgpu_synchronize())
        per_gpu = {}
        kinds, ms = (ctypes.c_int * 256)(), (ctypes.c_double * 256)()
        for g in range(L.pgpu_pool_size()):
            _capi.check(L.pgpu_set_device(g))
            k = L.pgpu_timing_collect(kinds, ms, 256)
            t = [ms[i] for i in range(k) if kinds[i] == K_MODEXP]
            per_gpu[str(g)] = round(float(np.mean(t)), 3) if t else None
        _capi.check(L.pgpu_set_device(0))
        _capi.check(L.pgpu_set_timing(0))
        res["decrypt_kernel_ms"] = per_gpu
        ok["c4_resident_encrypt"] = sha(down(bc)) == want_c
        ok["c4_resident_decrypt"] = all(sha(down(o)) == want_m for o in outs)
        ok["every_gpu_ran_a_decrypt"] = all(v is not None for v in per_gpu.values())
        ver, rep = ctypes.c_uint64(), ctypes.c_uint64()
        L.pgpu_replication_stats(ctypes.byref(ver), ctypes.byref(rep))
        res["images_verified"], res["copies_repaired"] = ver.value - ver0.value, rep.value - rep0.value
        for h in [bm, br, bc] + outs:
            L.pgpu_batch_destroy(h)
        del pk, sk

    # ---- configs[4]: 2048-bit ISO key, count5 CT+CT and CT x PT (u32) over the pool ----
    if count5:
        k = json.load(open(os.path.join(GOLD, "iso_kat.json")))
        p, q, hs = int(k["p"], 16), int(k["q"], 16), int(k["bench_hs"], 16)
        n = p * q
        nsq, W = n * n, 64
        rng = np.random.default_rng(5005)
        a = rand_rows(rng, count5, W, (1 << 60) - 1)
        b = rand_rows(rng, count5, W, (1 << 60) - 1)
        e = rand_rows(rng, count5, 1, (1 << 32) - 1)
        mod = ints_to_limbs([nsq], W)[0]
        want_add, want_mul = sha(c_oracle.modmul_batch(a, b, mod)), sha(be(a, e, mod))
        out = np.empty_like(a)
        _capi.check(L.pgpu_modmul(vp(a), vp(b), W, vp(mod), W, vp(out), count5))
        ok["c5_host_modmul"] = sha(out) == want_add
        _capi.check(L.pgpu_modexp(vp(a), W, vp(e), 1, 1, 32, vp(mod), W, vp(out), count5))
        ok["c5_host_modexp"] = sha(out) == want_mul
        del out
        pk = pa.PublicKey(n, 2048, hs=hs)
        ba, bb, bex = up(a), up(b), up(e)
        one = up(np.array([[1] + [0] * (W - 1)], dtype=np.uint64))
        am, bmm = op(L.pgpu_batch_ct_add, pk._h, ba, one), op(L.pgpu_batch_ct_add, pk._h, bb, one)   # device-produced operands
        s2, t2 = op(L.pgpu_batch_ct_add, pk._h, am, bmm), op(L.pgpu_batch_ct_mul, pk._h, am, bex, 32)
        ok["c5_resident_ct_add"] = sha(down(s2)) == want_add
        ok["c5_resident_ct_mul"] = sha(down(t2)) == want_mul
        for h in (ba, bb, bex, one, am, bmm, s2, t2):
            L.pgpu_batch_destroy(h)
        del pk
    print(json.dumps(res), flush=True)
    pa.terminate()


if __name__ == "__main__":
    main()
