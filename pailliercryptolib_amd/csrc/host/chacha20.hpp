// ChaCha20 block function (RFC 8439 section 2.3) -- the expander behind detail::fill_random for bulk requests:
// the kernel CSPRNG delivers ~0.35 GB/s through getrandom(), so the 1 MiB of obfuscator exponents of an 8192-element
// DJN encrypt cost ~3 ms, three times the GPU kernel; a fresh 256-bit key and 96-bit nonce from getrandom() per request
// expanded in user space cost ~0.4 ms (the construction of arc4random / randombytes).  Header-only so that the
// known-answer test (tests/test_host_random.py) needs nothing else.
#ifndef PAILLIERCRYPTOLIB_AMD_CSRC_HOST_CHACHA20_HPP_
#define PAILLIERCRYPTOLIB_AMD_CSRC_HOST_CHACHA20_HPP_

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace ipcl {
namespace detail {

inline std::uint32_t chacha_rotl(std::uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

// state: 4 constants | 8 key words | block counter | 3 nonce words (little-endian words); out: 64 key-stream bytes
inline void chacha20_block(const std::uint32_t state[16], unsigned char out[64]) {
  std::uint32_t x[16];
  for (int i = 0; i < 16; ++i) x[i] = state[i];
#define PGPU_CHACHA_QR(a, b, c, d)                 \
  x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 16); \
  x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 12); \
  x[a] += x[b]; x[d] = chacha_rotl(x[d] ^ x[a], 8);  \
  x[c] += x[d]; x[b] = chacha_rotl(x[b] ^ x[c], 7);
  for (int round = 0; round < 10; ++round) {
    PGPU_CHACHA_QR(0, 4, 8, 12) PGPU_CHACHA_QR(1, 5, 9, 13) PGPU_CHACHA_QR(2, 6, 10, 14) PGPU_CHACHA_QR(3, 7, 11, 15)
    PGPU_CHACHA_QR(0, 5, 10, 15) PGPU_CHACHA_QR(1, 6, 11, 12) PGPU_CHACHA_QR(2, 7, 8, 13) PGPU_CHACHA_QR(3, 4, 9, 14)
  }
#undef PGPU_CHACHA_QR
  for (int i = 0; i < 16; ++i) {
    const std::uint32_t v = x[i] + state[i];
    out[4 * i] = (unsigned char)v;
    out[4 * i + 1] = (unsigned char)(v >> 8);
    out[4 * i + 2] = (unsigned char)(v >> 16);
    out[4 * i + 3] = (unsigned char)(v >> 24);
  }
}

// LANES consecutive blocks at once (counters state[12] .. state[12]+LANES-1) on LANES-wide vectors (GCC/Clang vector
// extensions): 4 lanes = SSE2 on a plain x86-64 build, 8 lanes in a function compiled for AVX2 and chosen at run time;
// out: 64*LANES key-stream bytes
#define PGPU_CHACHA_BLOCKS(NAME, LANES, ATTR)                                                       \
  ATTR inline void NAME(const std::uint32_t state[16], unsigned char* out) {                        \
    typedef std::uint32_t vec __attribute__((vector_size(4 * LANES)));                              \
    vec x[16], in[16];                                                                              \
    for (int i = 0; i < 16; ++i)                                                                    \
      for (int b = 0; b < LANES; ++b) in[i][b] = state[i] + (i == 12 ? (std::uint32_t)b : 0u);      \
    for (int i = 0; i < 16; ++i) x[i] = in[i];                                                      \
    for (int round = 0; round < 10; ++round) {                                                      \
      PGPU_CHACHA_QRV(0, 4, 8, 12) PGPU_CHACHA_QRV(1, 5, 9, 13) PGPU_CHACHA_QRV(2, 6, 10, 14)       \
      PGPU_CHACHA_QRV(3, 7, 11, 15) PGPU_CHACHA_QRV(0, 5, 10, 15) PGPU_CHACHA_QRV(1, 6, 11, 12)     \
      PGPU_CHACHA_QRV(2, 7, 8, 13) PGPU_CHACHA_QRV(3, 4, 9, 14)                                     \
    }                                                                                               \
    std::uint32_t t[16][LANES];                                                                     \
    for (int i = 0; i < 16; ++i) {                                                                  \
      const vec v = x[i] + in[i];                                                                   \
      std::memcpy(t[i], &v, sizeof(v));                                                             \
    }                                                                                               \
    for (int b = 0; b < LANES; ++b)                                                                 \
      for (int i = 0; i < 16; ++i) {                                                                \
        const std::uint32_t w = t[i][b];                                                            \
        unsigned char* o = out + 64 * b + 4 * i;                                                    \
        o[0] = (unsigned char)w;                                                                    \
        o[1] = (unsigned char)(w >> 8);                                                             \
        o[2] = (unsigned char)(w >> 16);                                                            \
        o[3] = (unsigned char)(w >> 24);                                                            \
      }                                                                                             \
  }
#define PGPU_CHACHA_ROT(v, c) (((v) << (c)) | ((v) >> (32 - (c))))
#define PGPU_CHACHA_QRV(a, b, c, d)                                \
  x[a] += x[b]; x[d] ^= x[a]; x[d] = PGPU_CHACHA_ROT(x[d], 16);    \
  x[c] += x[d]; x[b] ^= x[c]; x[b] = PGPU_CHACHA_ROT(x[b], 12);    \
  x[a] += x[b]; x[d] ^= x[a]; x[d] = PGPU_CHACHA_ROT(x[d], 8);     \
  x[c] += x[d]; x[b] ^= x[c]; x[b] = PGPU_CHACHA_ROT(x[b], 7);
PGPU_CHACHA_BLOCKS(chacha20_blocks4, 4, )
#if defined(__x86_64__)
PGPU_CHACHA_BLOCKS(chacha20_blocks8, 8, __attribute__((target("avx2"))))
#endif
#undef PGPU_CHACHA_QRV
#undef PGPU_CHACHA_ROT
#undef PGPU_CHACHA_BLOCKS

// n key-stream bytes for (key, nonce), block counter starting at `counter`
inline void chacha20_stream(const unsigned char key[32], const unsigned char nonce[12], std::uint32_t counter,
                            unsigned char* dst, std::size_t n) {
  std::uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; ++i)
    st[4 + i] = (std::uint32_t)key[4 * i] | ((std::uint32_t)key[4 * i + 1] << 8) | ((std::uint32_t)key[4 * i + 2] << 16) |
                ((std::uint32_t)key[4 * i + 3] << 24);
  st[12] = counter;
  for (int i = 0; i < 3; ++i)
    st[13 + i] = (std::uint32_t)nonce[4 * i] | ((std::uint32_t)nonce[4 * i + 1] << 8) |
                 ((std::uint32_t)nonce[4 * i + 2] << 16) | ((std::uint32_t)nonce[4 * i + 3] << 24);
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  while (avx2 && n >= 512) {
    chacha20_blocks8(st, dst);
    st[12] += 8;
    dst += 512;
    n -= 512;
  }
#endif
  while (n >= 256) {
    chacha20_blocks4(st, dst);
    st[12] += 4;
    dst += 256;
    n -= 256;
  }
  unsigned char block[64];
  while (n > 0) {
    chacha20_block(st, block);
    ++st[12];
    const std::size_t take = n < 64 ? n : 64;
    std::memcpy(dst, block, take);
    dst += take;
    n -= take;
  }
  volatile unsigned char* wipe = block;
  for (int i = 0; i < 64; ++i) wipe[i] = 0;
  volatile std::uint32_t* ws = st;
  for (int i = 0; i < 16; ++i) ws[i] = 0;
}

}  // namespace detail
}  // namespace ipcl

#endif  // PAILLIERCRYPTOLIB_AMD_CSRC_HOST_CHACHA20_HPP_
