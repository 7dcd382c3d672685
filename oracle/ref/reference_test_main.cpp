// ORACLE — TEST INFRASTRUCTURE ONLY.
// main() for the reference's own test sources compiled with oracle/shim/gtest/gtest.h: runs every TEST() and reports.
#include <gtest/gtest.h>

#ifndef OKVIS_REF_SIDE   // (OKVIS_REF_SIDE: the same main() around the reference's own Estimator, no backend to warm up)
#include "okvis_amd_ba.h"
#endif

#include <cstddef>
#include <cstdlib>
#include <cstdio>
#include <exception>

#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

namespace google {
int eshim_log_warnings = 0;
}

// TestEstimator.cpp:85 takes t0 = okvis::Time::now(): the sub-second phase of the wall clock enters every time stamp of the
// scenario.  For a repeatable run the executable answers CLOCK_REALTIME itself: a fixed epoch plus the whole seconds this
// process has been running (everything else, and every other clock, is the kernel's).
extern "C" int clock_gettime(clockid_t id, struct timespec* ts) {
  if (id != CLOCK_REALTIME) return (int)syscall(SYS_clock_gettime, id, ts);
  static time_t start = 0;
  struct timespec mono;
  const int rc = (int)syscall(SYS_clock_gettime, CLOCK_MONOTONIC, &mono);
  if (start == 0) start = mono.tv_sec;
  ts->tv_sec = 1400000000 + (mono.tv_sec - start);
  ts->tv_nsec = 0;
  return rc;
}

// TestEstimator.cpp:64-75 fills an okvis::ImuParameters member by member and leaves sigma_bg / sigma_ba (Parameters.hpp:111-113,
// no default initialisers) unset; Estimator.cpp:272-276 turns them into the weights of the first speed/bias prior.  What the
// reference's test reads there is whatever the stack held.  To make that read deterministic here (zeros would mean infinite
// weights and NaN residuals on any backend) the stack below main() is painted with 0.03 = the sigma_bg of the EuRoC
// configuration before each test runs.
__attribute__((noinline)) static void paint_stack(double value) {
  volatile double region[1 << 17];
  for (size_t i = 0; i < sizeof(region) / sizeof(region[0]); ++i) region[i] = value;
}

int main() {
  int failed = 0;
  for (const auto& t : gtest_shim::registry()) {
    std::printf("[ RUN      ] %s\n", t.first);
    std::fflush(stdout);
    const int before = gtest_shim::failures();
    bool threw = false;
    try {
      // the scenario draws its noise from std::rand(): the HIP runtime's start-up (inside the first Estimator) was seen to
      // change the sequence from run to run, and with the noise the final errors the test asserts on (translation error
      // 0.03 ... 0.11 m against the 0.1 m bound over six runs).  Bring the runtime up first, then start from a fixed seed
#ifndef OKVIS_REF_SIDE
      okvis_ba_solver* warm = nullptr;
      if (okvis_ba_create(&warm, 0) == OKVIS_BA_OK) okvis_ba_destroy(warm);
#endif
      // OKVIS_TEST_SEED: the seed sweep of scripts/test_estimator_seeds.py (both sides over the same seeds)
      const char* seed_env = std::getenv("OKVIS_TEST_SEED");
      std::srand(seed_env ? (unsigned)std::atoi(seed_env) : 1u);
      paint_stack(0.03);
      t.second();
    } catch (const std::exception& e) {
      std::printf("exception: %s\n", e.what());
      threw = true;
    }
    const bool ok = !threw && gtest_shim::failures() == before;
    std::printf("[ %s ] %s\n", ok ? "      OK" : "  FAILED", t.first);
    failed += !ok;
  }
  std::printf("%zu tests, %d failed\n", gtest_shim::registry().size(), failed);
  return failed ? 1 : 0;
}
