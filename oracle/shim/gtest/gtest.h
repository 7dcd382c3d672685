// ORACLE — TEST INFRASTRUCTURE ONLY.
// Stand-in for <gtest/gtest.h> (googletest is not installed here): just enough to compile the reference's own test sources
// unmodified.  TEST(suite, name) registers a function; oracle/ref/reference_test_main.cpp runs the registered functions and
// turns an escaping exception (OKVIS_ASSERT_TRUE throws) or a failed EXPECT/ASSERT into a non-zero exit code.
#pragma once
#include <cstdio>
#include <utility>
#include <vector>

namespace gtest_shim {
typedef void (*TestFn)();
inline std::vector<std::pair<const char*, TestFn> >& registry() {
  static std::vector<std::pair<const char*, TestFn> > r;
  return r;
}
inline int& failures() {
  static int n = 0;
  return n;
}
struct Registrar {
  Registrar(const char* name, TestFn f) { registry().push_back(std::make_pair(name, f)); }
};
inline void check(bool ok, const char* expr, const char* file, int line) {
  if (!ok) {
    std::printf("%s:%d: failed: %s\n", file, line, expr);
    ++failures();
  }
}
}  // namespace gtest_shim

#define TEST(suite, name)                                                                              \
  static void gtest_shim_##suite##_##name();                                                           \
  static gtest_shim::Registrar gtest_shim_reg_##suite##_##name(#suite "." #name, &gtest_shim_##suite##_##name); \
  static void gtest_shim_##suite##_##name()
#define EXPECT_TRUE(x) gtest_shim::check((x), #x, __FILE__, __LINE__)
#define ASSERT_TRUE(x) gtest_shim::check((x), #x, __FILE__, __LINE__)
#define EXPECT_FALSE(x) gtest_shim::check(!(x), "!(" #x ")", __FILE__, __LINE__)
#define EXPECT_LT(a, b) gtest_shim::check((a) < (b), #a " < " #b, __FILE__, __LINE__)
#define EXPECT_NEAR(a, b, tol) gtest_shim::check(((a) - (b)) < (tol) && ((b) - (a)) < (tol), #a " ~ " #b, __FILE__, __LINE__)
