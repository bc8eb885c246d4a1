// Test driver for the host BigNumber: reads "op a b [c]" lines (hex operands, optional '-'),
// prints the result as num2hex.  Driven by tests/test_host_bignum.py against Python integers.
#include <iostream>
#include <sstream>
#include <string>
#include "ipcl/bignum.h"

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string op, sa, sb, sc;
    is >> op >> sa >> sb >> sc;
    try {
      BigNumber a(sa.c_str()), b(sb.empty() ? "0x0" : sb.c_str()), c(sc.empty() ? "0x0" : sc.c_str());
      std::string out;
      if (op == "add") (a + b).num2hex(out);
      else if (op == "sub") (a - b).num2hex(out);
      else if (op == "mul") (a * b).num2hex(out);
      else if (op == "div") (a / b).num2hex(out);
      else if (op == "mod") (a % b).num2hex(out);
      else if (op == "gcd") a.gcd(b).num2hex(out);
      else if (op == "inv") b.InverseMul(a).num2hex(out);          // a^-1 mod b
      else if (op == "modmul") c.ModMul(a, b).num2hex(out);        // a*b mod c
      else if (op == "modsub") c.ModSub(a, b).num2hex(out);
      else if (op == "cmp") out = std::to_string(a.compare(b));
      else if (op == "bits") out = std::to_string(a.BitSize()) + " " + std::to_string(a.LSB()) + " " + std::to_string(a.DwordSize());
      else if (op == "dec") { BigNumber d(sa.c_str()); d.num2hex(out); }
      else if (op == "vec") { std::vector<Ipp32u> v; a.num2vec(v); std::ostringstream os; os << v.size(); for (auto w : v) os << " " << w; out = os.str(); }
      else if (op == "bin") { unsigned char buf[64] = {0}; BigNumber::toBin(buf, 64, a); BigNumber r; BigNumber::fromBin(r, buf, 64); r.num2hex(out); }
      else out = "?";
      std::cout << out << "\n";
    } catch (const std::exception& e) {
      std::cout << "EXC\n";
    }
  }
  return 0;
}
