// ORACLE — TEST INFRASTRUCTURE ONLY: stand-in for <glog/logging.h> (messages are dropped, FATAL aborts).
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace google {
enum { GLOG_INFO = 0, GLOG_WARNING = 1, GLOG_ERROR = 2, GLOG_FATAL = 3 };
extern int eshim_log_warnings;  // counts LOG(WARNING) (e.g. "gyr saturation") for the tests
struct LogSink {
  int sev;
  std::ostringstream os;
  explicit LogSink(int s) : sev(s) {}
  ~LogSink() {
    if (sev == GLOG_FATAL) {
      std::cerr << "FATAL: " << os.str() << std::endl;
      std::abort();
    }
  }
  template <class T>
  LogSink& operator<<(const T& v) {
    if (sev >= GLOG_ERROR) os << v;
    return *this;
  }
  LogSink& operator<<(std::ostream& (*f)(std::ostream&)) {
    if (sev >= GLOG_ERROR) os << f;
    return *this;
  }
};
inline void InitGoogleLogging(const char*) {}
}  // namespace google
#define LOG(sev) ::google::LogSink(::google::GLOG_##sev)
#define VLOG(n) ::google::LogSink(::google::GLOG_INFO)
#define DLOG(sev) LOG(sev)
#define LOG_IF(sev, c) if (c) LOG(sev)
#define CHECK(c) if (!(c)) LOG(FATAL) << "Check failed: " #c " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_NOTNULL(p) (p)
#define DCHECK(c) CHECK(c)
