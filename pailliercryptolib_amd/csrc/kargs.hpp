// pailliercryptolib_amd -- kernel argument blocks shared by the device code (kernels.hpp) and the host
// runtime (capi.cpp, runtime.cpp).  Plain data only: this header is included by files that are
// compiled without any device code.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_KARGS_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_KARGS_HPP_

#include <stddef.h>
#include <stdint.h>

namespace pgpu {

constexpr int kWave = 64;
constexpr int kLimbBits = 29;
constexpr uint32_t kLimbMask = (1u << kLimbBits) - 1;
// Workgroups are kWavesPerWG independent wavefronts (one per SIMD of a CU, for even SIMD load)
constexpr int kWavesPerWG = 4;
constexpr int kWGThreads = kWave * kWavesPerWG;

// Montgomery context of one odd modulus N, resident in device memory (built by the host,
// capi.hip: build_modctx).  R = 2^(29*L) for the geometry the context was built for.
struct ModCtxDev {
  const uint32_t* n;    // [L]   N, 29-bit limbs
  const uint32_t* r2;   // [L]   R^2 mod N
  const uint32_t* one;  // [L]   R mod N
  const uint32_t* r2s;  // [L]   R^2 * 2^(64*mod_words) mod N   (wide-base reduction)   | may be null
  const uint32_t* fc;   // [L]   constant final multiplier, plain domain (hp, hq)       | may be null
  const uint32_t* nr;   // [L]   n*R mod N for N = n^2 (Paillier g^m = 1 + n*m)          | may be null
  const uint64_t* n64;  // [W64+1] N as little-endian 64-bit words, zero padded
  // Quotient-digit shortcut.  If nhat != null the exponentiation loop runs modulo Nhat = N*k with
  // k = -N^-1 mod 2^29, i.e. Nhat == -1 mod 2^29 and the Montgomery constant n0' is 1: the
  // quotient digit is just the low limb (no multiply).  n / r2 / one / r2s / nr above are then
  // all taken modulo Nhat (lazy values stay correct modulo N); the multiplication that leaves the
  // Montgomery domain switches back to the true modulus held here.
  const uint32_t* nhat; // [L] Nhat limbs, or null (then n is the true modulus everywhere)
  // Montgomery-form output of the Paillier encrypt forms: with gadd != null the kernel adds these L limbs
  // (R mod N, true modulus) instead of 1 to m*nr, and the host points nr at n*R^2 mod N, so that
  // g^m arrives as (1 + n*m)*R and the result c*R stays in the Montgomery domain of N.
  const uint32_t* gadd;
  uint32_t n0inv;       // -N^-1 mod 2^29 (true modulus)
  int mod_words;        // 64-bit words per element of this modulus in the C-ABI layout
};

enum FinalMul : int {
  FM_UNIT = 0,       // multiply by 1: plain modexp
  FM_CTX_CONST = 1,  // multiply by ctx.fc
  FM_PAILLIER_G = 2  // multiply by (1 + n*m) mod n^2, m read from fm_words
};

struct ModexpArgs {
  ModCtxDev ctx[2];
  int nctx;              // 1, or 2: instance i uses ctx[i % 2] and base element i / 2.  With 2 contexts a
                         // wavefront takes instances of ONE parity (wave W: parity W & 1, elements
                         // (W >> 1)*IPW ...), so context and exponent are wave-uniform
  const uint64_t* base;  // [.][base_stride] (0: one shared base)
  size_t base_stride;
  int base_words;        // valid words per base; may be 2*mod_words (reduced on load) if ctx.r2s
  const uint64_t* exp;   // [.][exp_stride]; exp_per_ctx: row (i % nctx), else row i (0: shared)
  size_t exp_stride;
  int exp_per_ctx;
  int exp_words;
  int exp_bits;          // max exponent bit length over the batch (mod_exp.cpp:484)
  int window;            // fixed window width w, 1..5 (table of 2^w entries); with a schedule: 2^w ODD powers
  // Shared exponents known to the host (key constants p-1, q-1, n) come with a sliding-window
  // schedule per context instead of being scanned digit by digit: step k = (nsq << 6) | (idx + 1):
  // nsq squarings, then a multiplication by base^(2*idx+1) (idx + 1 == 0: squarings only; step 0 has
  // nsq == 0 and loads the entry).  null: fixed-window scan of exp.
  const uint16_t* sched[2];
  int sched_len[2];
  int parity_waves;      // nctx == 2 only: 1 = a wavefront takes one parity (required by a schedule)
  int final_mul;         // FinalMul
  const uint64_t* fm_words;  // FM_PAILLIER_G: plaintexts [count][fm_stride]
  size_t fm_stride;
  int fm_nwords;
  uint64_t* out;         // [count][out_stride]
  size_t out_stride;
  uint32_t* table;       // [count rounded up to IPW][2^w][L] workspace
  size_t count;          // number of instances (= 2 * ciphertexts when nctx == 2)
  uint64_t* wave_clocks; // optional diagnostics (tools/wave_spread.py): [waves][3] = start, end
                         // (s_memtime ticks), XCC id | HW_ID << 8
};

struct ModmulArgs {
  ModCtxDev ctx;
  const uint64_t* a;     // [count][a_stride]
  size_t a_stride;
  const uint64_t* b;     // [count][b_stride]  (b_stride == 0: scalar broadcast)
  size_t b_stride;
  int in_words;          // valid words per operand
  uint64_t* out;         // [count][ctx.mod_words]
  size_t count;
  int mode;              // ModmulMode
  int b_words;           // MM_GM: valid words per plaintext
};

enum ModmulMode : int {
  MM_PLAIN = 0,   // out = a*b mod N, plain operands and result (two Montgomery products)
  MM_SINGLE = 1,  // out = montmul(a, b): Montgomery-form operands and result (one product)
  MM_BY_R2 = 2,   // out = montmul(a, R^2): plain -> Montgomery form
  MM_BY_ONE = 3,  // out = montmul(a, 1): Montgomery form -> plain
  MM_GM = 4       // out = a * (1 + n*b) mod n^2 (CT + PT): b = plaintexts of b_words words; keeps a's form
};

// Second half of CRT decryption.  All constants are host-precomputed for the geometry of this
// launch (R = 2^(29*L)); M is the auxiliary modulus 2^(29*(L-1)) - 1 (odd, coprime to p and q)
// under which "multiply by p^-1" is an exact division and "multiply by p" an exact product.
struct CrtArgs {
  ModCtxDev ctxM;        // modulus M
  ModCtxDev ctxQ;        // modulus q
  const uint32_t* cp;    // [L] p^-1 * R mod M
  const uint32_t* cq;    // [L] q^-1 * R mod M
  const uint32_t* pinvR; // [L] (p^-1 mod q) * R mod q
  const uint32_t* pRM;   // [L] p * R mod M
  const uint64_t* hp64;  // [vw] hp
  const uint64_t* hq64;  // [vw] hq
  const uint64_t* p2_64; // [vw] p^2
  const uint64_t* q2_64; // [vw] q^2
  const uint64_t* q64;   // [vw] q (zero padded)
  const uint64_t* v;     // [2*count][vw]: row 2i = xp*hp mod p^2, row 2i+1 = xq*hq mod q^2
  int vw;                // words per row of v (= words of p^2)
  int have_m;            // 1: the rows of v already hold mp (row 2i) and mq (row 2i+1), canonical (hensel.hpp)
  uint64_t* out;         // [count][out_words]  plaintexts (< n)
  int out_words;
  size_t count;
};

// CRT-decrypt exponentiation in split form (hensel.hpp): one side (p or q) of the key.  L2 = H*K limbs per half;
// "pair" arrays hold 2*L2 limbs in group-lane order: the L2 limbs of a, then the L2 limbs of b, for
// x == a - P*b (mod P^2).
// Resident ciphertext batches of keys that have a split form live in HBM as PAIRS of limbs ("pair rows", round 3):
// row i = 2*L2n 29-bit limbs (one uint32_t each) in group-lane order -- the L2n limbs of a, then the L2n limbs of b --
// with  c*R == a - P*b  (mod n^2),  P = n*k == -1 (mod 2^29),  R = 2^(29*L2n),  0 <= a, b < 4P (lazy, relaxed limbs as
// the products leave them).  That IS the register image of the split-form kernels: no word <-> limb conversion, no
// canonical reduction and no full-width product on the way in or out; CT+CT is one pair product (5 instead of 8 s^2
// limb products), CT+PT two half-width products, encrypt / CT x PT / decrypt enter and leave without their
// conversion products.  Only pgpu_batch_download materialises the plain value.
struct HenselCtxDev {
  const uint32_t* nhat;  // [L2]  P = p * k == -1 mod 2^29: the loop modulus
  const uint32_t* n;     // [L2]  p
  const uint32_t* one;   // pair  R mod P^2                      (R = 2^(29*L2))
  const uint32_t* conv;  // [nchunks] pairs  2^(64*chunk_words*i) * R^2 mod P^2  (divided by the R of the n^2
                         //       context when the ciphertexts arrive in its Montgomery form)
  const uint32_t* h;     // [L2]  hp (hq): the constant multiplier of pri_key.cpp:153-154
  const uint32_t* kr;    // [L2]  (P / p) * R mod p
  uint32_t n0inv;        // -p^-1 mod 2^29
  // entry from a pair row of the n^2 domain (HenselArgs::ct_pair): the a part enters as pchunks chunks of pchunk_limbs
  // limbs, (z_i, 0) (x) pconv[i]; the b part only matters modulo p and enters as b_i (x) pcb[i], half-width
  const uint32_t* pconv; // [pchunks] pairs  2^(29*pchunk_limbs*i) * R^2 * Rn^-1 mod P^2       (Rn: the radix of the rows)
  const uint32_t* pcb;   // [pchunks][L2]    (n/p) * kn * k^-1 * 2^(29*pchunk_limbs*i) * R^2 * Rn^-1 mod P
};

struct HenselArgs {
  HenselCtxDev ctx[2];   // p side (even wavefronts), q side (odd wavefronts)
  const uint64_t* ct;    // [count][ct_stride] ciphertexts
  size_t ct_stride;
  int ct_words;          // valid words per ciphertext
  int chunk_words;       // a ciphertext enters as nchunks chunks of chunk_words words (each < 4P)
  int nchunks;
  const uint64_t* exp;   // [2][exp_stride]: p-1, q-1
  size_t exp_stride;
  int exp_words;
  int exp_bits;
  int window;            // as ModexpArgs::window
  const uint16_t* sched[2];   // as ModexpArgs::sched (null: fixed-window scan)
  int sched_len[2];
  uint64_t* out;         // [2*count][out_stride]: row 2i = mp, row 2i+1 = mq
  size_t out_stride;
  int out_words;
  uint32_t* table;       // [wavefronts * 64/(2H)][entries][2*L2] workspace
  size_t count;          // ciphertexts
  int ct_gather;             // 1: window-table entries are fetched by reading ALL entries and selecting (addresses do
                             //    not depend on exponent digits; pgpu_set_table_gather_policy)
  const uint32_t* ct_pair;   // non-null: ciphertexts as pair rows [count][ct_pair_stride] (ct unused)
  size_t ct_pair_stride;     // limbs per row = 2 * pair_l2
  int pair_l2;               // L2n of the rows
  int pchunk_limbs, pchunks;
};

// Split form (hensel.hpp) of the public modulus n^2 = (n)^2: DJN encrypt with fixed-base tables of pairs.
struct HenselPubDev {
  const uint32_t* nhat;  // [L2]  P = n * k == -1 mod 2^29
  const uint32_t* n;     // [L2]  n
  const uint32_t* kr;    // [L2]  k * R mod n          (R = 2^(29*L2))
  const uint32_t* one;   // pair  R mod P^2
  const uint32_t* conv;  // [nchunks] pairs  2^(64*chunk_words*i) * R^2 mod P^2
  const uint32_t* gm;    // [L2]  (-k^-1 mod n) * R^2 mod n: multiplying a resident pair by g^m = 1 + n*m is
                         //       b += montmul(montmul(m, gm), a) under the true modulus n (hensel.hpp: pair_times_gm)
  uint32_t n0inv;        // -n^-1 mod 2^29
};

struct HenselFbBuildArgs {
  HenselPubDev ctx;
  const uint64_t* base;  // hs, base_words words
  int base_words;
  int chunk_words;
  int nchunks;
  uint32_t* table;       // [nwin][2^w] pairs
  int nwin;
  int w;
};

// The way back from a canonical pair (a, b), a, b < n, to the full-width residue c = a + n*b modulo n^2, in the
// geometry Geo<2H,K> of the n^2 context (R' = 2^(29*2*L2)):  c = montmul(b, n*R') + a;  its Montgomery form
// c*R' = montmul(b, n*R'^2) + montmul(a, R'^2).
struct HenselFullDev {
  const uint32_t* n;     // [2*L2] n^2
  const uint32_t* nr;    // [2*L2] n*R' mod n^2   (Montgomery-form result: n*R'^2 mod n^2)
  const uint32_t* r2;    // [2*L2] R'^2 mod (a multiple of n^2); null: plain result
  uint32_t n0inv;        // -(n^2)^-1 mod 2^29
  int mod_words;         // 64-bit words per residue
};

struct HenselFbArgs {
  HenselPubDev ctx;
  HenselFullDev full;
  const uint32_t* table;     // [nwin][2^w] pairs
  int nwin;
  int w;
  const uint64_t* exp;       // [count][exp_stride] the randomness r
  size_t exp_stride;
  int exp_words;
  const uint64_t* fm_words;  // plaintexts [count][fm_stride], fm_nwords <= words of n
  size_t fm_stride;
  int fm_nwords;
  uint64_t* out;             // [count][out_stride]
  size_t out_stride;
  size_t count;
  uint32_t* out_pair;        // non-null: ciphertexts leave as pair rows [count][2*L2] (out unused)
  int ct_gather;             // 1: every entry of a window is read and the wanted one selected (masked fixed-base product)
};

// base[i]^exp[i] modulo n^2 in split form (hensel.hpp: hensel_modexp_kernel): CT x PT and the non-DJN obfuscator r^n.
struct HenselModexpArgs {
  HenselPubDev ctx;          // ctx.conv: the chunk constants for the form the bases arrive in (plain, or c*R' mod n^2)
  HenselFullDev full;
  const uint64_t* base;      // [count][base_stride]
  size_t base_stride;
  int base_words;
  int chunk_words;
  int nchunks;
  const uint64_t* exp;       // [count][exp_stride]; exp_stride == 0: one shared exponent
  size_t exp_stride;
  int exp_words;
  int exp_bits;
  int window;                // as ModexpArgs::window
  const uint16_t* sched;     // as ModexpArgs::sched (shared exponents the host knows), or null
  int sched_len;
  int final_mul;             // FM_UNIT, or FM_PAILLIER_G: times 1 + n*m, m from fm_words (rows no wider than n)
  const uint64_t* fm_words;
  size_t fm_stride;
  int fm_nwords;
  uint64_t* out;             // [count][out_stride]
  size_t out_stride;
  uint32_t* table;           // [wavefronts * 64/(2H)][entries][2*L2] workspace
  size_t count;
  int ct_gather;             // as HenselArgs::ct_gather
  const uint32_t* base_pair; // non-null: bases as pair rows [count][2*L2] (base unused; base_pair_stride 0: one shared row)
  size_t base_pair_stride;
  uint32_t* out_pair;        // non-null: results leave as pair rows [count][2*L2] (out unused)
};

// Element-wise operations on pair rows (hensel.hpp: pair_ops_kernel).
enum PairOp : int {
  PO_MUL = 0,         // out = a (x) b                 CT + CT (ciphertext.cpp:135-141): ONE pair product
  PO_TIMES_GM = 1,    // out = a * (1 + n*m)           CT + PT (ciphertext.cpp:75-80): two half-width products
  PO_FROM_WORDS = 2,  // words (plain, or c*Rs mod n^2 with the chunk constants of that form) -> pair row
  PO_TO_WORDS = 3     // pair row -> canonical words (plain)
};
struct PairOpsArgs {
  HenselPubDev ctx;
  HenselFullDev full;        // PO_TO_WORDS
  int op;
  const uint32_t* a;         // [count][2*L2] pair rows (PO_MUL, PO_TIMES_GM, PO_TO_WORDS)
  const uint32_t* b;         // PO_MUL: [count][2*L2] pair rows (b_stride 0: one shared row)
  size_t b_stride;           // in limbs
  const uint64_t* words;     // PO_FROM_WORDS: [count][words_stride]; PO_TIMES_GM: plaintexts (words_stride 0: one shared)
  size_t words_stride;
  int nwords;                // valid words per row of `words`
  int chunk_words, nchunks;  // PO_FROM_WORDS
  uint32_t* out;             // pair rows (PO_TO_WORDS: unused)
  uint64_t* out_words;       // PO_TO_WORDS: [count][out_stride]
  size_t out_stride;
  size_t count;
};

struct FixedBaseArgs {
  ModCtxDev ctx;         // modulus n^2 (nr set)
  const uint32_t* table; // [nwin][2^w][L]
  int nwin;
  int w;
  const uint64_t* exp;   // [count][exp_stride] the randomness r
  size_t exp_stride;
  int exp_words;
  const uint64_t* fm_words;  // plaintexts [count][fm_stride]
  size_t fm_stride;
  int fm_nwords;
  uint64_t* out;         // [count][out_stride]
  size_t out_stride;
  size_t count;
  int ct_gather;         // 1: every entry of a window is read and the wanted one selected (include/pgpu.h, SIDE CHANNELS)
};

struct FixedBaseBuildArgs {
  ModCtxDev ctx;
  const uint64_t* base;  // hs, ctx.mod_words words
  uint32_t* table;       // [nwin][2^w][L]
  int nwin;
  int w;
};

}  // namespace pgpu

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_KARGS_HPP_
