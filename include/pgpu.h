/* pgpu.h -- C-ABI of the MI355X (gfx950) batched modular-exponentiation engine.
 *
 * This is the drop-in boundary under the reference's `ipcl::` C++ API: every entry point takes
 * plain pointers and sizes (no C++ types, no torch types) and replaces one call the reference
 * makes into a third-party/accelerator primitive.  The precedent in the reference is its own
 * accelerator seam, the HE-QAT C library (module/heqat/heqat/include/heqat/bnops.h:63-148,
 * context.h:18-26) and the IPP-Crypto multi-buffer primitive mbx_exp_mb8 (call site
 * ipcl/mod_exp.cpp:508-516).
 *
 * DATA LAYOUT.  A big integer is `words` little-endian 64-bit limbs (same as the `int64u`
 * arrays handed to mbx_exp_mb8, mod_exp.cpp:486-506).  A batch is row-major
 * [element][limb] with a fixed element stride given in 64-bit words; a stride of 0 means
 * "one shared value for the whole batch" (every reference call site passes N copies of one
 * modulus / one exponent / one base: pub_key.cpp:53-54,67-69, pri_key.cpp:119-120,
 * ciphertext.cpp:151).  Values are zero padded to the stride.
 *
 * OWNERSHIP / THREADING.  The caller owns every buffer it passes.  Functions without the
 * `_dev` suffix take HOST pointers and are synchronous (like release_bnModExp_buffer,
 * bnops.h:134-148): on return the outputs are complete.  `_dev` functions take DEVICE
 * pointers plus a hipStream_t (as void*) and only enqueue work on that stream; calls on different
 * streams never share scratch memory (window tables and the CRT hand-over buffer are per stream), so
 * they may be issued concurrently.  All entry points may be called from several host threads (the
 * reference's APPLEVEL_OMP test calls encrypt/decrypt from 4 threads,
 * test/test_cryptography.cpp:45-57): host-pointer calls run as tasks on the worker lanes of the
 * device pool (two per GPU), so one caller's copies overlap another caller's kernels.
 *
 * DEVICE POOL.  pgpu_init binds the process to one GPU; pgpu_init_all builds a pool over several
 * (the counterpart of acquire_qat_devices taking every QAT instance, heqat/context.h:18-26, and
 * of the reference's own two-way batch split over a second std::thread, mod_exp.cpp:700-731).
 * With a pool, every host-pointer entry point cuts its batch [0, count) into contiguous shards,
 * one per GPU (output order is preserved by construction), and key objects hold one copy of their
 * constants per GPU: the image is uploaded to GPU 0 and reaches the others through one RCCL
 * broadcast over xGMI (per-device copies when RCCL is unavailable).  There is no other collective
 * on the data path.  The `_dev` functions and pgpu_dev_alloc/free/copy address ONE pool entry: the
 * calling thread's current one (pgpu_set_device, default 0).  pgpu_batch handles are the sharded,
 * device-resident form of a batch.
 *
 * SIDE CHANNELS.  Operation sequences do not depend on secret data unless the caller opts in:
 * exponents that are private-key constants (p-1, q-1) run a fixed-window schedule (always w
 * squarings and one multiplication, also for zero digits), the same instruction stream for every
 * key; pgpu_set_secret_exponent_policy(PGPU_EXP_SLIDING) trades that for ~5 % fewer
 * multiplications with a key-dependent schedule.  Table ADDRESSES are not hidden: the window table
 * of a wavefront is indexed by exponent digits (per-key constant pattern for p-1/q-1, per-element
 * for the randomness r of the DJN fixed-base product and for CT*PT exponents), i.e. the engine
 * is not hardened against a co-resident observer of the memory system -- unless
 * pgpu_set_table_gather_policy(1) is chosen, which makes the split-form kernels read every entry of
 * a window table and select; since round 4 that covers the DJN fixed-base product as well (a table of its own
 * with a 4-bit window: 256 products of 16 candidates each instead of 85 indexed ones).  Device copies of
 * private-key constants are zeroed before they are freed.
 *
 * ERRORS.  Every function returns PGPU_OK (0) or a negative pgpu_status; nothing calls
 * exit().  pgpu_last_error() returns a thread-local description.  The C++ layer turns a
 * non-zero status into the reference's ERROR_CHECK exception (utils/util.hpp:23-34).
 * There is NO CPU fallback anywhere behind this interface: without a usable gfx950 device
 * every compute entry point fails with PGPU_ERR_NO_DEVICE.
 */
#ifndef PGPU_H_
#define PGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pgpu_status {
  PGPU_OK = 0,
  PGPU_ERR_INVALID_PARAM = -1, /* null pointer, zero width, inconsistent sizes            */
  PGPU_ERR_EVEN_MODULUS = -2,  /* Montgomery arithmetic needs an odd modulus               */
  PGPU_ERR_UNSUPPORTED = -3,   /* operand width beyond the compiled kernel geometries      */
  PGPU_ERR_NO_DEVICE = -4,     /* no HIP device / not initialised                          */
  PGPU_ERR_HIP = -5,           /* a HIP runtime call failed (see pgpu_last_error)          */
  PGPU_ERR_NOT_INVERTIBLE = -6 /* key material is inconsistent (e.g. p == q)               */
} pgpu_status;

/* ---- context: behind ipcl::initializeContext / terminateContext (utils/context.cpp:40-71);
 *      replaces acquire_qat_devices / release_qat_devices (heqat/context.h:18-26) ---- */
int pgpu_init(int device /* HIP ordinal; -1 = keep the current device */);
/* Pool over the first n visible GPUs (0 = all).  With the environment variable
 * PGPU_POOL_OVERSUBSCRIBE=1, n may exceed the number of visible GPUs: pool entries then wrap around
 * the physical devices (validation of the multi-device paths on a single-GPU box). */
int pgpu_init_all(int n_devices_or_0);
void pgpu_shutdown(void);
int pgpu_device_count(void);       /* visible HIP devices */
int pgpu_is_initialized(void);
const char* pgpu_last_error(void);
const char* pgpu_device_name(void); /* of the calling thread's current pool entry */
/* What this library binary was built with (host-side query, needs no device): bit 0 = the split forms of the 4096-bit
 * key class (build switch PGPU_BUILD_4096=1; without them such keys run the full-width kernels).  Results never depend
 * on it.  (Bit 1 named the A/B-wavefront decrypt experiment of rounds 3-5; retired, always 0.) */
#define PGPU_FEATURE_4096_SPLIT 1
int pgpu_build_features(void);
int pgpu_pool_size(void);           /* entries of the pool (0 before init) */
int pgpu_set_device(int pool_index);/* pool entry addressed by this thread's `_dev` / dev_alloc / copy calls */
int pgpu_get_device(void);
/* "rccl" | "memcpy" | "single": how key images reached the pool devices (xGMI broadcast, per-device
 * copies, or a pool of one) */
const char* pgpu_pool_transport(void);
/* why RCCL is (not) in use ("ok", "pool entries share a physical device", the failing call, ...) */
const char* pgpu_rccl_note(void);
/* Replicated key images are read back from every GPU once per upload and compared with the host bytes; a copy
 * that differs is rewritten over PCIe, counted here, and RCCL is retired if it delivered it. */
int pgpu_replication_stats(uint64_t* images_verified, uint64_t* copies_repaired);
/* test hook: flips one byte of the NEXT replicated image on pool entry `pool_index` right after the broadcast */
int pgpu_debug_corrupt_next_replica(int pool_index);
/* batches smaller than min_shard * k elements are spread over at most k GPUs (default 256; env
 * PGPU_MIN_SHARD) */
int pgpu_set_min_shard(size_t min_elements_per_device);
/* waits for everything queued on the pool's own streams (resident-batch operations, worker lanes) on every GPU */
int pgpu_synchronize(void);
/* The cut a batch of `count` elements receives on a pool of `pool_size` GPUs: *n_shards shards, shard i =
 * elements [bounds[i], bounds[i+1]) on pool entry i (contiguous, in order, sizes differ by at most one;
 * bounds has room for pool_size + 1 entries).  Pure host-side query (needs no device): this is the rule every
 * host-pointer entry point and every pgpu_batch follows. */
int pgpu_shard_plan(size_t count, int pool_size, int* n_shards, size_t* bounds);

/* ---- generic batched modular exponentiation: out[i] = base[i]^exp[i] mod mod ----
 * Replaces mbx_exp_mb8 (mod_exp.cpp:508-516) / ippsMontExp (mod_exp.cpp:549-579) under
 * ipcl::modExp (mod_exp.hpp:72-83).  `exp_bits` is the maximum exponent bit length of the
 * batch (the reference computes the same maximum, mod_exp.cpp:484); exponent words above
 * exp_bits must be zero.  The modulus is shared and odd; bases may be any value that fits
 * mod_words words (they are reduced).  out has stride mod_words.
 * A modulus that is the square of an odd root of up to 3072 bits -- every modulus of the Paillier path: n^2, p^2,
 * q^2 -- is recognised (once per modulus) and runs the split form (PGPU_HENSEL=0: never). */
int pgpu_modexp(const uint64_t* base, size_t base_stride, const uint64_t* exp, size_t exp_stride,
                int exp_words, int exp_bits, const uint64_t* mod, int mod_words, uint64_t* out,
                size_t count);
int pgpu_modexp_dev(const uint64_t* d_base, size_t base_stride, const uint64_t* d_exp,
                    size_t exp_stride, int exp_words, int exp_bits, const uint64_t* h_mod,
                    int mod_words, uint64_t* d_out, size_t count, void* hip_stream);

/* ---- batched modular multiplication: out[i] = a[i]*b[i] mod mod ----
 * Replaces the per-element `a * b % sq` of CipherText::raw_add (ciphertext.cpp:135-141);
 * b_stride == 0 is the reference's vector (+) scalar case (ciphertext.cpp:51-58). */
int pgpu_modmul(const uint64_t* a, const uint64_t* b, size_t b_stride, const uint64_t* mod,
                int mod_words, uint64_t* out, size_t count);
int pgpu_modmul_dev(const uint64_t* d_a, const uint64_t* d_b, size_t b_stride,
                    const uint64_t* h_mod, int mod_words, uint64_t* d_out, size_t count,
                    void* hip_stream);


/* ---- Paillier public key: fused encrypt ----
 * Replaces PublicKey::raw_encrypt + applyObfuscator (pub_key.cpp:82-110) for make_secure=true:
 *   DJN key (hs != NULL):  c[i] = hs^r[i]      * (1 + n*m[i]) mod n^2   (pub_key.cpp:51-64)
 *   plain   (hs == NULL):  c[i] = r[i]^n       * (1 + n*m[i]) mod n^2   (pub_key.cpp:66-80)
 * n has n_words words (key bits / 64); hs and c have 2*n_words words.  r is the caller's
 * randomness (the reference draws it on the host too, pub_key.cpp:59-61,74-76, or injects it
 * with setRandom, pub_key.cpp:92-95); it is used as given: not reduced, not truncated. */
typedef struct pgpu_pubkey pgpu_pubkey;
int pgpu_pubkey_create(const uint64_t* n, int n_words, const uint64_t* hs_or_null,
                       pgpu_pubkey** out);
void pgpu_pubkey_destroy(pgpu_pubkey* key);
int pgpu_paillier_encrypt(const pgpu_pubkey* key, const uint64_t* m, size_t m_stride, int m_words,
                          const uint64_t* r, size_t r_stride, int r_words, int r_bits,
                          uint64_t* c, size_t count);
int pgpu_paillier_encrypt_dev(const pgpu_pubkey* key, const uint64_t* d_m, size_t m_stride,
                              int m_words, const uint64_t* d_r, size_t r_stride, int r_words,
                              int r_bits, uint64_t* d_c, size_t count, void* hip_stream);

/* DJN keys: hs is a key constant, so hs^r runs as a fixed-base product over a per-key table of
 * hs^(d*2^(w*i)) (built on the GPU at the first encrypt, no squarings afterwards).  w = 0 selects
 * the generic square-and-multiply kernel instead; default 13 (env PGPU_FB_WINDOW, 0..14): 79 table products
 * for a 1024-bit r, 373 MB of table per 2048-bit key and GPU, built in 73 ms (w = 12: 86 products, 203 MB, 40 ms -- the
 * default until round 4; w = 10: 103 products, 61 MB; encrypt launch of the bench batch 0.775 / 0.835 / 1.19 ms).  The
 * table budgets (pgpu_set_fixed_base_budget) narrow the window until the table fits.  Unless a window was set
 * explicitly, a key starts with w = 8 (13 MB, built in ~2 ms) and switches to the default after its first 4096
 * elements.  Results are identical. */
int pgpu_set_fixed_base_window(int w);
/* Table memory is bounded (round 3): per key and GPU by max_bytes_per_key (default 512 MiB, env
 * PGPU_FB_KEY_MAX_BYTES: the window narrows until the table fits), per GPU over ALL keys by max_bytes_per_device
 * (default 2 GiB, env PGPU_FB_MAX_BYTES: a new table first evicts the least recently used tables of other keys;
 * an evicted key rebuilds its table on its next encrypt).  0 leaves a limit unchanged.  Results never change. */
int pgpu_set_fixed_base_budget(size_t max_bytes_per_device, size_t max_bytes_per_key);
int pgpu_fixed_base_stats(int pool_index, size_t* live_bytes, uint64_t* evictions);
/* the table a key currently holds on a pool entry: window, bytes, build time in ms (0: none / still building) */
int pgpu_pubkey_fixed_base_info(const pgpu_pubkey* key, int pool_index, int* window, size_t* bytes, double* build_ms);

/* ---- Paillier private key: fused CRT decrypt ----
 * Replaces PrivateKey::decryptCRT + computeLfun + computeCRT (pri_key.cpp:114-157): per
 * ciphertext c (2*n_words words) two half-width exponentiations c^(p-1) mod p^2, c^(q-1) mod
 * q^2, the L function, *hp / *hq, and the CRT recombination; m has n_words words.
 * p and q have pq_words words each (either order: the smaller becomes p, pri_key.cpp:19-22). */
typedef struct pgpu_privkey pgpu_privkey;
int pgpu_privkey_create(const uint64_t* p, const uint64_t* q, int pq_words, pgpu_privkey** out);
void pgpu_privkey_destroy(pgpu_privkey* key);
int pgpu_paillier_decrypt_crt(const pgpu_privkey* key, const uint64_t* c, uint64_t* m,
                              size_t count);
int pgpu_paillier_decrypt_crt_dev(const pgpu_privkey* key, const uint64_t* d_c, uint64_t* d_m,
                                  size_t count, void* hip_stream);

/* ---- device buffers for callers that do not link the HIP runtime themselves ----
 * (the C++ ipcl:: layer keeps ciphertext batches resident in HBM between operations and feeds
 * them to the *_dev entry points on the default stream).  Copies are synchronous. */
int pgpu_dev_alloc(size_t bytes, void** out);
void pgpu_dev_free(void* d_ptr);
int pgpu_copy_h2d(void* d_dst, const void* h_src, size_t bytes);
int pgpu_copy_d2h(void* h_dst, const void* d_src, size_t bytes);

/* ---- pinned host memory ----
 * A caller buffer that lies inside a block from pgpu_host_alloc is the source / target of the DMA itself: every
 * host-pointer entry point, pgpu_batch_upload and pgpu_batch_download then skip their staging copies (a host thread moves
 * 8-10 GB/s between pageable and pinned memory, the link ~50 GB/s).  The reference's accelerator path does the same: HE-QAT
 * takes its operand buffers from the device driver's pinned allocator (module/heqat/heqat/bnops.c: qaeMemAllocNUMA).
 * pgpu_batch_upload from such a block only QUEUES the copy: the block must not be modified until pgpu_host_wait(ptr)
 * has returned (pgpu_host_free waits too; every other entry point is synchronous as before).  Results that land in a
 * block are complete on return.  Blocks are usable with every GPU of the pool and survive pgpu_shutdown. */
int pgpu_host_alloc(size_t bytes, void** out);
void pgpu_host_free(void* ptr);
int pgpu_host_wait(const void* ptr /* any address inside a block */);

/* ---- exponent schedules of private-key constants (see SIDE CHANNELS above) ---- */
typedef enum pgpu_exp_policy {
  PGPU_EXP_FIXED_WINDOW = 0, /* default: key-independent operation sequence                     */
  PGPU_EXP_SLIDING = 1       /* host-built sliding-window schedule of p-1 / q-1 (fewer products) */
} pgpu_exp_policy;
int pgpu_set_secret_exponent_policy(int policy);
int pgpu_get_secret_exponent_policy(void);
/* Window-table ACCESS of the split-form kernels (CRT decrypt, CT*PT and the key-less seam for 1024- to 3072-bit keys):
 * 0 (default) the entry is addressed by the exponent digit; 1 every entry of the table is read and the wanted one
 * selected, so that the address stream does not depend on the exponent -- what the reference's mbx_exp_mb8 does
 * (SURVEY Appendix B).  Costs 2^w x the table loads plus as many selects per multiplication (bench.py reports it).
 * DJN encrypt (round 4): with the policy on, hs^r runs as a fixed-base product over a SMALL-window table of its own
 * (w = 4, env PGPU_FB_MASKED_WINDOW 1..5: all 2^w entries of a window are read at every step and the wanted one is
 * selected) instead of the 2^12-entry windows addressed by the digits of r -- about 3x the indexed encrypt, 1/5 of a
 * masked square-and-multiply.  Both tables of a key may be live (2.4 MB + 203 MB for a 2048-bit key).  Not covered: the
 * full-width modexp_kernel (keys without a split form, PGPU_HENSEL=0).  Env PGPU_CT_GATHER=1. */
int pgpu_set_table_gather_policy(int masked);
int pgpu_get_table_gather_policy(void);

/* ---- sharded device-resident batches (SURVEY 8(f) N1 across the pool) ----
 * A pgpu_batch is [count][words] little-endian limbs living in GPU memory, cut into contiguous shards
 * over the pool (count == 1: one copy on every GPU, the reference's "vector (+) scalar" operand,
 * ciphertext.cpp:51-58).  Operations enqueue one launch per shard on the GPUs' batch streams and
 * return at once; pgpu_batch_download waits.  Batches are immutable once produced.
 * Ciphertext batches produced on the device stay in a device-side DOMAIN; the plain value only materialises in
 * pgpu_batch_download and callers never see the difference:
 *   - keys of 1024 / 2048 / 3072 bits (those with a split form): PAIR ROWS, c*R == a - (n*k)*b mod n^2 as 29-bit limbs
 *     (pgpu_batch_row_limbs) -- CT+CT is ONE pair product, CT+PT two half-width products, encrypt / CT*PT / CRT decrypt
 *     read and write the rows without conversion (ciphertext.cpp:135-141 does a multiply and a divide per CT+CT);
 *   - other keys, plaintext rows wider than n, PGPU_PAIR_ROWS=0: Montgomery-form words c*R mod n^2 -- CT+CT one
 *     full-width Montgomery product, encrypt emits the form for free, CRT decrypt absorbs it into its load constants.
 * Operations accept any mix (plain uploaded batches included) and convert on the way in. */
typedef struct pgpu_batch pgpu_batch;
int pgpu_batch_create(size_t count, int words, pgpu_batch** out);            /* uninitialised */
int pgpu_batch_upload(const uint64_t* host, size_t count, int words, size_t stride, pgpu_batch** out);
int pgpu_batch_download(const pgpu_batch* b, uint64_t* host /* [count][words] */);
/* Download that does not hold the caller (round 4): returns at once with a ticket; a worker of the pool runs the conversion
 * (if the batch is in a device-side domain) and queues the copy once the batch's kernels have finished -- never earlier, so
 * that it cannot block the copy engine for other lanes.  pgpu_ticket_wait blocks until the data has arrived, returns the
 * status of the download and frees the ticket.  `b` and `host` must stay alive until then.  With buffers from
 * pgpu_host_alloc this is what lets ONE thread pipeline whole host-to-host steps over two lanes (bench.py:
 * end_to_end_pipelined). */
/* Download with the rows `host_stride` words apart (the words between rows are overwritten with ZEROS -- write per-row
 * headers after the call): lets a host layer lay results out with room for its own per-row headers and use them in place
 * (the ipcl:: layer's BigNumbers point into the pinned block).  Pinned targets, one-GPU pools, plain or pair-row batches;
 * otherwise PGPU_ERR_UNSUPPORTED. */
int pgpu_batch_download_strided(const pgpu_batch* b, uint64_t* host, size_t host_stride_words);
/* 1: the batch belongs to the device pool that is up now; 0: its pool has been shut down (pgpu_shutdown, also followed by
 * a new pgpu_init*): every operation refuses it, only pgpu_batch_destroy takes it.  Lets a host layer that caches device
 * copies of caller data (the ipcl:: layer: injected randomness, texts uploaded at construction) fall back to its host
 * copy instead of failing. */
int pgpu_batch_is_current(const pgpu_batch* b);
typedef struct pgpu_ticket pgpu_ticket;
int pgpu_batch_download_async(const pgpu_batch* b, uint64_t* host /* [count][words] */, pgpu_ticket** out);
int pgpu_ticket_wait(pgpu_ticket* t);
void pgpu_batch_destroy(pgpu_batch* b);
size_t pgpu_batch_count(const pgpu_batch* b);
int pgpu_batch_words(const pgpu_batch* b);
int pgpu_batch_is_montgomery(const pgpu_batch* b);   /* 1: a device-side domain (Montgomery words or pair rows), 0: plain */
/* limbs per row when the batch is stored as PAIR ROWS (round 3; csrc/kargs.hpp), else 0.  Ciphertext batches produced
 * on the device for keys of 1024 / 2048 / 3072 bits are pairs (a, b) of 29-bit limbs with c*R == a - (n*k)*b mod n^2 --
 * the register image of the split-form kernels: CT+CT is ONE pair product, CT+PT two half-width products, encrypt /
 * CT*PT / CRT decrypt read and write the rows without any conversion.  PGPU_PAIR_ROWS=0 keeps Montgomery words. */
int pgpu_batch_row_limbs(const pgpu_batch* b);
/* Batch lanes: every GPU of the pool has pgpu_batch_lanes() (4) batch streams, so that many independent chains of
 * resident batches can be in flight at once (their kernels share the chip: a wavefront that is alone on a SIMD issues
 * ~8 % slower than two, and the sequential-halves kernels -- 11-17 % fewer instructions -- need 16384 ciphertexts in
 * flight to reach every SIMD).  Uploads and pgpu_batch_create take the calling thread's lane (the first thread of the
 * process that uploads gets lane 0, further threads the next lanes round-robin; pgpu_set_batch_lane overrides); every result
 * inherits the lane of the operation's first operand; operands of another lane are ordered in by events.
 * A thread that CHOOSES its lanes (pgpu_set_batch_lane) is taken to pipeline -- to keep several lanes fed without waiting in
 * between: what its launches look like depends on whether the neighbour lanes are busy when they are queued (the adaptive
 * policy, pgpu_decrypt_kernel_form_ex: part-chip launches side by side).  Threads on the round-robin lanes are synchronous
 * API callers (upload, operation, download): their lanes idle during the host part of every call, so their launches keep
 * the lone caller's full-chip forms and simply overlap one thread's copies and host work with another's kernels. */
int pgpu_set_batch_lane(int lane /* 0 .. pgpu_batch_lanes() - 1 */);
int pgpu_batch_lane(const pgpu_batch* b);
int pgpu_batch_lanes(void);
/* c = Enc(m; r): m, r batches of `count` elements (PublicKey::encrypt, pub_key.cpp:112-129) */
int pgpu_batch_encrypt(const pgpu_pubkey* key, const pgpu_batch* m, const pgpu_batch* r, int r_bits,
                       pgpu_batch** c);
/* m = Dec(c) (PrivateKey::decrypt CRT path, pri_key.cpp:65-90,114-157) */
int pgpu_batch_decrypt_crt(const pgpu_privkey* key, const pgpu_batch* c, pgpu_batch** m);
/* out = a*b mod n^2 under `key` (CipherText + CipherText, ciphertext.cpp:35-72); b may hold one element */
int pgpu_batch_ct_add(const pgpu_pubkey* key, const pgpu_batch* a, const pgpu_batch* b, pgpu_batch** out);
/* out = a * (1 + n*m) mod n^2 (CipherText + PlainText, ciphertext.cpp:75-80: g^m without obfuscator, then the
 * product -- here fused in one launch, no host loop); m may hold one element */
int pgpu_batch_ct_add_plain(const pgpu_pubkey* key, const pgpu_batch* a, const pgpu_batch* m, pgpu_batch** out);
/* out = a^e mod n^2 (CipherText * PlainText, ciphertext.cpp:83-106); e may hold one element */
int pgpu_batch_ct_mul(const pgpu_pubkey* key, const pgpu_batch* a, const pgpu_batch* e, int e_bits,
                      pgpu_batch** out);

/* ---- instrumentation used by bench.py (roofline) ----
 * With timing enabled every kernel launch is bracketed by two HIP events recorded on the stream
 * the kernel is launched on (no synchronisation at launch time).  pgpu_timing_collect waits for
 * the recorded launches, writes up to max_entries (kind, milliseconds) pairs in launch order,
 * clears the record and returns the number written. */
typedef enum pgpu_kernel_kind {
  PGPU_KERNEL_MODEXP = 1,     /* modexp_kernel (generic / encrypt / decrypt stage 1) */
  PGPU_KERNEL_MODMUL = 2,     /* modmul_kernel */
  PGPU_KERNEL_CRT = 3,        /* crt_kernel (decrypt stage 2) */
  PGPU_KERNEL_FB_ENCRYPT = 4  /* fb_encrypt_kernel (DJN encrypt, fixed-base) */
} pgpu_kernel_kind;
int pgpu_set_timing(int enabled);
int pgpu_timing_collect(int* kinds, double* ms, int max_entries);
/* the same with the kernel FORM the launcher picked for each launch (forms may be NULL): bit 0 the paired split form
 * (hensel.hpp: the two halves of a residue in neighbouring lanes), bit 1 the sequential-halves form (hensel_seq.hpp:
 * both halves in the same lanes), bit 4 the launch claimed whole CUs (a half-chip launch beside a busy neighbour lane);
 * 0: a full-width kernel.  With the adaptive policy the form depends on what the GPU's other batch lanes were doing
 * at launch time, so a measurement reports what actually ran. */
typedef enum pgpu_kernel_form {
  PGPU_FORM_FULL_WIDTH = 0,
  PGPU_FORM_PAIRED = 1,
  PGPU_FORM_SEQ = 2,
  PGPU_FORM_LANE = 4,      /* a whole exponentiation per lane; always together with PGPU_FORM_PS since round 6 */
  PGPU_FORM_PS = 8,        /* with PGPU_FORM_LANE: by product scanning (hensel_ps.hpp: 1024- to 3072-bit keys; round 5) */
  PGPU_FORM_CU_CLAIM = 16,
  PGPU_FORM_WAVE = 32      /* one exponentiation per WAVEFRONT (hensel_wave.hpp, hensel_wave_n2.hpp: small launches of CRT decrypt, DJN
                              encrypt onto pair rows, CT x PT on pair rows; round 6) */
} pgpu_kernel_form;
int pgpu_timing_collect_ex(int* kinds, int* forms, double* ms, int max_entries);
/* ... and as a small kernel trace: batch lane of each launch (-1: another stream) and its start relative to the first
 * recorded launch, from the same HIP events (any of forms / lanes / start_ms may be NULL) */
int pgpu_timing_collect_trace(int* kinds, int* forms, int* lanes, double* start_ms, double* ms, int max_entries);
/* Kernel geometry a batch of `count` exponentiations / products under an odd modulus of mod_bits bits
 * (operand rows of in_words 64-bit words) is launched with: *lanes lanes per element, *limbs 29-bit limbs
 * per lane (modexp_kernel; small batches take a 16-lane latency split, large ones the wide split).  Pure
 * host-side query (needs no device); lets a profile be labelled with the kernel instantiation that ran.  PGPU_ERR_UNSUPPORTED when no compiled geometry is wide enough. */
int pgpu_kernel_geometry(int in_words, int mod_bits, size_t count, int* lanes, int* limbs);
/* Exponentiation kernel a CRT decrypt of `count` ciphertexts (per device) runs under this key: *split = 1:
 * hensel_decrypt_kernel<*lanes / 2, *limbs> -- residues modulo p^2 / q^2 as pairs of half-width numbers (DESIGN.md
 * section 3; compiled for 1024- to 4096-bit keys; PGPU_HENSEL=0 turns it off); *split = 2: for ciphertexts of a
 * resident batch (pair rows), hensel_decrypt_seq_kernel<*lanes, *limbs> -- both halves of a pair in the same *lanes
 * lanes, launches that still put a wavefront on every SIMD that way (PGPU_SEQ_DECRYPT=0 turns it off); (*split = 3 named
 * round 4's operand-scanning one-lane kernel, retired in round 6;) *split = 4 (round 5):
 * hensel_decrypt_ps_kernel<*limbs, 28 | 29> -- a whole exponentiation in one lane by product scanning, *limbs limbs per half
 * (38 / 56 limbs of 28 bits: 2048- / 3072-bit keys, 19 limbs of 29 bits: 1024-bit keys; launches of
 * more than 16384 ciphertexts (3072-bit keys: 24576) in rounds of 32768 -- the form reported is that of the full rounds; a
 * mostly empty last round runs as a launch of its own in the form of its size -- or smaller ones that cover the SIMDs
 * together with busy neighbour lanes: 8192 beside three;
 * PGPU_PS_DECRYPT=0 turns it off); *split = 5 (round 6): hensel_decrypt_wave_kernel<*limbs, 28 | 29> -- one exponentiation per
 * wavefront (*lanes = 64), a limb per lane: lone decrypts of up to 512 ciphertexts of 1024- to 3072-bit keys;
 * *split = 0: the full-width modexp_kernel<Geo<*lanes, *limbs>>.
 * Host-side query. */
int pgpu_decrypt_kernel_form(const pgpu_privkey* key, size_t count, int* split, int* lanes, int* limbs);
/* ... when `busy_lanes` OTHER batch lanes of the GPU have work queued at launch time (round 4, the adaptive policy: a
 * launch that will share the chip anyway takes the sequential-halves form as soon as waves * (1 + busy_lanes) covers the
 * SIMDs; pgpu_decrypt_kernel_form is the busy_lanes = 0 case, a lone caller) */
int pgpu_decrypt_kernel_form_ex(const pgpu_privkey* key, size_t count, int busy_lanes, int* split, int* lanes, int* limbs);
/* The same for an encrypt of `count` plaintext rows of m_words words: *split = 1: hensel_fb_encrypt_kernel<*lanes / 2,
 * *limbs> (DJN keys with a fixed-base window, 1024- to 3072-bit keys, plaintext rows no wider than n, batches that
 * fill the chip); *split = 2: results that stay resident as pair rows, launches that still put a wavefront on every SIMD
 * with half the lanes per element: hensel_fb_encrypt_seq_kernel<*lanes, *limbs> (PGPU_SEQ_DECRYPT=0 turns it off);
 * *split = 5 (round 6; asked for with busy_lanes = -1 through the _ex form: "the results stay resident as pair rows"): at most
 * 1024 elements: hensel_fb_encrypt_wave_kernel<*limbs, ...> -- one wavefront per element (*lanes = 64), *limbs limbs per half of the row; *split = 0: fb_encrypt_kernel / modexp_kernel
 * <Geo<*lanes, *limbs>>. */
int pgpu_encrypt_kernel_form(const pgpu_pubkey* key, int m_words, size_t count, int* split, int* lanes, int* limbs);
int pgpu_encrypt_kernel_form_ex(const pgpu_pubkey* key, int m_words, size_t count, int busy_lanes, int* split, int* lanes,
                                int* limbs);
/* The same for the exponentiations modulo n^2 with per-element bases (CT x PT of a resident batch, the non-DJN
 * obfuscator r^n): *split = 1: hensel_modexp_kernel<*lanes / 2, *limbs>; *split = 2: the same for r^n, and for CT x PT
 * of a resident batch (pair rows in and out, per-element exponents) hensel_modexp_seq_kernel<*lanes, *limbs> (both halves
 * of a pair in the same *lanes lanes; PGPU_SEQ_DECRYPT=0 turns it off); *split = 5 (round 6): CT x PT of at most 1024 resident
 * elements, hensel_modexp_wave_kernel<*limbs, ...> -- one wavefront per element; *split = 0: modexp_kernel<Geo<*lanes, *limbs>>. */
int pgpu_modexp_n2_kernel_form(const pgpu_pubkey* key, size_t count, int* split, int* lanes, int* limbs);
/* The same for CT + CT of two resident batches of `count` elements (per device): *split = 1: pair_ops_kernel<*lanes / 2,
 * *limbs> (one pair product on pair rows); *split = 2: pair_mul_seq_kernel<*lanes, *limbs> (both halves of a pair in the
 * same *lanes lanes: launches that still put a wavefront on every SIMD that way); *split = 0: modmul_kernel<Geo<*lanes,
 * *limbs>> on Montgomery-form words (keys without a pair form, PGPU_PAIR_ROWS=0). */
int pgpu_ct_add_kernel_form(const pgpu_pubkey* key, size_t count, int* split, int* lanes, int* limbs);

#ifdef __cplusplus
}
#endif

#endif /* PGPU_H_ */
