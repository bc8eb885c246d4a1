// Test driver for tests/test_serialization_layout.py: reads "op a [b c]" lines (hex operands, optional '-') and prints
// the bytes ipcl::serializer writes, as hex.  Host-only (no GPU context is created): BigNumber and PlainText.
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "ipcl/bignum.h"
#include "ipcl/plaintext.hpp"

static std::string hexbytes(const std::string& s) {
  static const char* d = "0123456789abcdef";
  std::string o;
  for (unsigned char c : s) { o += d[c >> 4]; o += d[c & 15]; }
  return o;
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string op, sa, sb, sc;
    is >> op >> sa >> sb >> sc;
    try {
      BigNumber a(sa.c_str()), b(sb.empty() ? "0x0" : sb.c_str()), c(sc.empty() ? "0x0" : sc.c_str());
      std::string out = "?";
      if (op == "ser") {
        std::ostringstream os;
        ipcl::serializer::serialize(os, a);
        BigNumber back;
        std::istringstream in(os.str());
        ipcl::serializer::deserialize(in, back);
        out = hexbytes(os.str()) + (back == a ? "" : " BAD");
      } else if (op == "serpt") {   // PlainText of {a, b, c}
        std::ostringstream os;
        ipcl::serializer::serialize(os, ipcl::PlainText(std::vector<BigNumber>{a, b, c}));
        ipcl::PlainText back;
        std::istringstream in(os.str());
        ipcl::serializer::deserialize(in, back);
        out = hexbytes(os.str()) + (back.getSize() == 3 && back.getElement(0) == a && back.getElement(2) == c ? " ok" : " BAD");
      }
      std::cout << out << "\n";
    } catch (const std::exception& e) {
      std::cout << "EXC\n";
    }
  }
  return 0;
}
