// pailliercryptolib_amd -- BaseText container (reference ipcl/include/ipcl/base_text.hpp:14-115).
#ifndef PAILLIERCRYPTOLIB_AMD_IPCL_BASE_TEXT_HPP_
#define PAILLIERCRYPTOLIB_AMD_IPCL_BASE_TEXT_HPP_

#include <cstdint>
#include <string>
#include <vector>

#include "ipcl/bignum.h"

namespace ipcl {

class BaseText {
 public:
  BaseText() = default;
  ~BaseText() = default;
  explicit BaseText(const uint32_t& n);
  explicit BaseText(const std::vector<uint32_t>& n_v);
  explicit BaseText(const BigNumber& bn);
  explicit BaseText(const std::vector<BigNumber>& bn_v);
  BaseText(const BaseText& bt) = default;
  BaseText& operator=(const BaseText& other) = default;

  BigNumber& operator[](const std::size_t idx);
  void insert(const std::size_t pos, BigNumber& bn);
  void clear();
  void remove(const std::size_t pos, const std::size_t length = 1);
  BigNumber getElement(const std::size_t& idx) const;
  std::vector<uint32_t> getElementVec(const std::size_t& idx) const;
  std::string getElementHex(const std::size_t& idx) const;
  std::vector<BigNumber> getChunk(const std::size_t& start, const std::size_t& size) const;
  std::vector<BigNumber> getTexts() const;
  std::size_t getSize() const;

 protected:
  std::vector<BigNumber> m_texts;
  std::size_t m_size = 0;
};

}  // namespace ipcl
#endif  // PAILLIERCRYPTOLIB_AMD_IPCL_BASE_TEXT_HPP_
