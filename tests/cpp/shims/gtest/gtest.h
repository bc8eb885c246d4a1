// Syntax-check stand-in for <gtest/gtest.h> (tests/test_dropin_compile.py): just enough of the googletest surface that
// the reference's OWN client sources -- test/*.cpp, compiled where they lie under /root/reference, never copied -- parse
// and type-check against include/ipcl.  Not a test framework: nothing here runs.
#ifndef PAILLIERCRYPTOLIB_AMD_TESTS_SHIMS_GTEST_H_
#define PAILLIERCRYPTOLIB_AMD_TESTS_SHIMS_GTEST_H_
#include <iostream>

namespace testing {
inline void InitGoogleTest(int*, char**) {}
struct Message {
  template <class T>
  Message& operator<<(const T&) { return *this; }
};
template <class A, class B>
Message expect_eq(const A& a, const B& b) {
  (void)(a == b);      // the comparison the macro would make must exist
  return Message();
}
}  // namespace testing
inline int RUN_ALL_TESTS() { return 0; }
#define TEST(suite, name) void suite##_##name##_body()
#define EXPECT_EQ(a, b) ::testing::expect_eq((a), (b))
#define ASSERT_EQ(a, b) ::testing::expect_eq((a), (b))
#define EXPECT_TRUE(a) ((void)(a), ::testing::Message())
#define EXPECT_FALSE(a) ((void)(a), ::testing::Message())
#endif
