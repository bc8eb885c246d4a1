// Known-answer driver for the ChaCha20 expander of detail::fill_random (tests/test_host_random.py):
//   block <key hex 64> <nonce hex 24> <counter> <bytes>   ->  key stream, hex
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "chacha20.hpp"

static std::vector<unsigned char> unhex(const std::string& s) {
  std::vector<unsigned char> v;
  for (size_t i = 0; i + 1 < s.size(); i += 2) v.push_back((unsigned char)std::stoul(s.substr(i, 2), nullptr, 16));
  return v;
}

int main() {
  std::string op, key, nonce;
  unsigned long counter, n;
  while (std::cin >> op >> key >> nonce >> counter >> n) {
    std::vector<unsigned char> k = unhex(key), no = unhex(nonce), out(n);
    if (k.size() != 32 || no.size() != 12) { std::puts("bad"); continue; }
    ipcl::detail::chacha20_stream(k.data(), no.data(), (std::uint32_t)counter, out.data(), out.size());
    for (unsigned char c : out) std::printf("%02x", c);
    std::puts("");
  }
  return 0;
}
