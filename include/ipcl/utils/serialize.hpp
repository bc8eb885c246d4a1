// pailliercryptolib_amd -- serialization (reference ipcl/include/ipcl/utils/serialize.hpp:25-62 and the
// save/load members in bignum.h:131-153, pub_key.hpp:133-164, pri_key.hpp:93-133,
// base_text.hpp:108-114, plaintext.hpp:92-98, ciphertext.hpp:69-74).
//
// The reference serialises through cereal's PortableBinary archives.  cereal is not vendored in
// the reference tree and not available here, so this header re-creates the small subset of that
// wire format the reference uses, from cereal's documented behaviour:
//   * one leading byte: 1 on a little-endian writer;
//   * arithmetic values little-endian, bool as one byte, enums as their 4-byte underlying int;
//   * std::vector: element count as uint64, then the elements (raw for arithmetic types);
//   * a class with a versioned save/load/serialize writes its class version (uint32, 0 here:
//     the reference never calls CEREAL_CLASS_VERSION) ONCE per archive, before the first
//     instance of that type.
// PARITY: pinned against hand-built byte vectors that follow cereal v1.3.2's PortableBinary framing rule by
// rule (tests/test_serialization_layout.py for BigNumber / PlainText, tests/cpp/ipcl_api_tests.cpp for the
// keys) -- NOT against a stream written by cereal itself: cereal is fetched from GitHub by the reference's
// build (cmake/cereal.cmake:7-8) and exists neither in its tree nor in this image, and the tree holds no
// serialized fixture.  Interoperability with archives of a real IPCL build therefore remains unverified.
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_SERIALIZE_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_SERIALIZE_HPP_

#include <algorithm>   // (the reference's headers provide it transitively through cereal: test_serialization.cpp uses std::for_each)
#include <cstdint>
#include <cstring>
#include <fstream>
#include <istream>
#include <ostream>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace ipcl {
namespace serializer {

class OutputArchive {
 public:
  explicit OutputArchive(std::ostream& os) : m_os(os) { u8(1); }  // little-endian writer
  void u8(uint8_t v) { m_os.write(reinterpret_cast<const char*>(&v), 1); }
  void u32(uint32_t v) { le(v); }
  void i32(int32_t v) { le(static_cast<uint32_t>(v)); }
  void u64(uint64_t v) { le(v); }
  void boolean(bool v) { u8(v ? 1 : 0); }
  void vec_u32(const std::vector<uint32_t>& v) {
    u64(v.size());
    for (uint32_t x : v) u32(x);
  }
  // class version record: written the first time a type name is seen in this archive
  void class_version(const char* type_name, uint32_t version = 0) {
    if (m_seen.insert(type_name).second) u32(version);
  }

 private:
  template <typename T>
  void le(T v) {
    unsigned char b[sizeof(T)];
    for (size_t i = 0; i < sizeof(T); ++i) b[i] = static_cast<unsigned char>(v >> (8 * i));
    m_os.write(reinterpret_cast<const char*>(b), sizeof(T));
  }
  std::ostream& m_os;
  std::set<std::string> m_seen;
};

class InputArchive {
 public:
  explicit InputArchive(std::istream& is) : m_is(is) {
    uint8_t flag = u8();
    if (flag != 1) throw std::runtime_error("ipcl::serializer: big-endian archives are not supported");
  }
  uint8_t u8() {
    char c;
    if (!m_is.read(&c, 1)) throw std::runtime_error("ipcl::serializer: unexpected end of archive");
    return static_cast<uint8_t>(c);
  }
  uint32_t u32() { return le<uint32_t>(); }
  int32_t i32() { return static_cast<int32_t>(le<uint32_t>()); }
  uint64_t u64() { return le<uint64_t>(); }
  bool boolean() { return u8() != 0; }
  std::vector<uint32_t> vec_u32() {
    uint64_t n = u64();
    if (n > (1u << 24)) throw std::runtime_error("ipcl::serializer: implausible vector size");
    std::vector<uint32_t> v(n);
    for (auto& x : v) x = u32();
    return v;
  }
  uint32_t class_version(const char* type_name) {
    if (m_seen.insert(type_name).second) return u32();
    return 0;
  }

 private:
  template <typename T>
  T le() {
    unsigned char b[sizeof(T)];
    if (!m_is.read(reinterpret_cast<char*>(b), sizeof(T)))
      throw std::runtime_error("ipcl::serializer: unexpected end of archive");
    T v = 0;
    for (size_t i = 0; i < sizeof(T); ++i) v |= static_cast<T>(b[i]) << (8 * i);
    return v;
  }
  std::istream& m_is;
  std::set<std::string> m_seen;
};

// every serialisable type T provides:  void save(OutputArchive&) const;  void load(InputArchive&);
template <typename T>
void serialize(std::ostream& ss, const T& obj) {
  OutputArchive archive(ss);
  obj.save(archive);
}

template <typename T>
void deserialize(std::istream& ss, T& obj) {
  InputArchive archive(ss);
  obj.load(archive);
}

template <typename T>
bool serializeToFile(const std::string& fn, const T& obj) {
  std::ofstream ofs(fn, std::ios::out | std::ios::binary);
  if (!ofs.is_open()) return false;
  serializer::serialize(ofs, obj);
  return true;
}

template <typename T>
bool deserializeFromFile(const std::string& fn, T& obj) {
  std::ifstream ifs(fn, std::ios::in | std::ios::binary);
  if (!ifs.is_open()) return false;
  serializer::deserialize(ifs, obj);
  return true;
}

}  // namespace serializer
}  // namespace ipcl

#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_UTILS_SERIALIZE_HPP_
