// ipcl::modExp -- the seam (reference ipcl/mod_exp.cpp:655-749).  The reference slices the batch
// into chunks of 8 for mbx_exp_mb8 under OpenMP; here the whole batch is marshalled into flat
// limb arrays once and handed to the GPU engine in one call per distinct modulus.
#include "ipcl/mod_exp.hpp"

#include <map>

#include "detail.hpp"
#include "ipcl/utils/util.hpp"

namespace ipcl {

// ---- hybrid knobs: thread-local state, no effect (reference mod_exp.cpp:22-64) ----
namespace {
struct HybridParams {
  float ratio = 0.0f;
  HybridMode mode = HybridMode::OPTIMAL;
};
thread_local HybridParams g_hybrid;
}  // namespace

void setHybridMode(HybridMode) {}
void setHybridRatio(float, bool) {}
void setHybridOff() {}
float getHybridRatio() { return g_hybrid.ratio; }
HybridMode getHybridMode() { return g_hybrid.mode; }
bool isHybridOptimal() { return g_hybrid.mode == HybridMode::OPTIMAL; }

namespace {

// one launch: all elements share `mod`
void modexp_shared_mod(const std::vector<BigNumber>& base, const std::vector<BigNumber>& exp,
                       const BigNumber& mod, const std::vector<size_t>& idx,
                       std::vector<BigNumber>& out) {
  ERROR_CHECK(!mod.isNegative() && !mod.isZero(), "modExp: modulus must be positive");
  const int mw = detail::words_for_bits(mod.BitSize());
  int ebits = 0;
  bool neg_exp = false;
  for (size_t i : idx) {
    neg_exp = neg_exp || exp[i].isNegative();
    ebits = std::max(ebits, exp[i].isZero() ? 0 : exp[i].BitSize());
  }
  ERROR_CHECK(!neg_exp, "modExp: negative exponent");
  const int ew = detail::words_for_bits(ebits);
  std::vector<uint64_t> fb(idx.size() * (size_t)mw), fe(idx.size() * (size_t)ew), fm((size_t)mw),
      fo(idx.size() * (size_t)mw);
  mod.toLimbs64(fm.data(), (size_t)mw);
  // per-element marshalling on the host team (the reference's chunk loop runs under OpenMP: mod_exp.cpp:607-612)
  detail::parallel_for(idx.size(), 512, [&](size_t k) {
    const BigNumber& b = base[idx[k]];
    // the engine reduces any base that fits the modulus width; wider / negative ones first go
    // through the host (callers of the reference guarantee base < mod)
    if (b.isNegative() || b.limbs64().size() > (size_t)mw) (b % mod).toLimbs64(fb.data() + k * (size_t)mw, (size_t)mw);
    else b.toLimbs64(fb.data() + k * (size_t)mw, (size_t)mw);
    exp[idx[k]].toLimbs64(fe.data() + k * (size_t)ew, (size_t)ew);
  });
  IPCL_GPU_CHECK(pgpu_modexp(fb.data(), (size_t)mw, fe.data(), (size_t)ew, ew, ebits, fm.data(), mw,
                             fo.data(), idx.size()),
                 "modExp");
  detail::parallel_for(idx.size(), 512, [&](size_t k) {
    out[idx[k]] = BigNumber::fromLimbs64(fo.data() + k * (size_t)mw, (size_t)mw);
  });
}

}  // namespace

std::vector<BigNumber> modExp(const std::vector<BigNumber>& base, const std::vector<BigNumber>& exp,
                              const std::vector<BigNumber>& mod) {
  ERROR_CHECK(base.size() == exp.size() && exp.size() == mod.size(),
              "modExp: input vector size error");  // reference mod_exp.cpp:452-454
  detail::ensure_context();
  std::vector<BigNumber> out(base.size());
  if (base.empty()) return out;
  // group by modulus: every reference call site passes N copies of one modulus
  bool all_same = true;
  for (size_t i = 1; i < mod.size() && all_same; ++i) all_same = (mod[i] == mod[0]);
  if (all_same) {
    std::vector<size_t> idx(base.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
    modexp_shared_mod(base, exp, mod[0], idx, out);
  } else {
    std::map<BigNumber::Limbs, std::vector<size_t>> groups;
    for (size_t i = 0; i < mod.size(); ++i) groups[mod[i].limbs64()].push_back(i);
    for (auto& g : groups) modexp_shared_mod(base, exp, mod[g.second[0]], g.second, out);
  }
  return out;
}

BigNumber modExp(const BigNumber& base, const BigNumber& exp, const BigNumber& mod) {
  return modExp(std::vector<BigNumber>{base}, std::vector<BigNumber>{exp}, std::vector<BigNumber>{mod})[0];
}

std::vector<BigNumber> ippModExp(const std::vector<BigNumber>& base, const std::vector<BigNumber>& exp,
                                 const std::vector<BigNumber>& mod) {
  return modExp(base, exp, mod);
}
BigNumber ippModExp(const BigNumber& base, const BigNumber& exp, const BigNumber& mod) {
  return modExp(base, exp, mod);
}
std::vector<BigNumber> qatModExp(const std::vector<BigNumber>&, const std::vector<BigNumber>&,
                                 const std::vector<BigNumber>&) {
  ERROR_CHECK(false, "qatModExp: Need to turn on IPCL_ENABLE_QAT");  // reference mod_exp.cpp:593
  return {};
}

std::vector<BigNumber> modMul(const std::vector<BigNumber>& a, const std::vector<BigNumber>& b,
                              const BigNumber& mod) {
  ERROR_CHECK(b.size() == a.size() || b.size() == 1, "modMul: size mismatch");
  detail::ensure_context();
  if (a.empty()) return {};
  const int mw = detail::words_for_bits(mod.BitSize());
  auto reduce_fit = [&](const std::vector<BigNumber>& v) {
    std::vector<uint64_t> flat(v.size() * (size_t)mw);
    detail::parallel_for(v.size(), 512, [&](size_t i) {   // (reference ciphertext.cpp:53-68: OpenMP per element)
      if (v[i].isNegative() || v[i].limbs64().size() > (size_t)mw) (v[i] % mod).toLimbs64(flat.data() + i * (size_t)mw, (size_t)mw);
      else v[i].toLimbs64(flat.data() + i * (size_t)mw, (size_t)mw);
    });
    return flat;
  };
  std::vector<uint64_t> fa = reduce_fit(a), fb = reduce_fit(b), fm((size_t)mw), fo(a.size() * (size_t)mw);
  mod.toLimbs64(fm.data(), (size_t)mw);
  // b.size()==1 with a.size()==1 is an ordinary element-wise product
  size_t bstride = (b.size() == a.size()) ? (size_t)mw : 0;
  IPCL_GPU_CHECK(pgpu_modmul(fa.data(), fb.data(), bstride, fm.data(), mw, fo.data(), a.size()), "modMul");
  return detail::unpack(fo, a.size(), mw);
}

}  // namespace ipcl
