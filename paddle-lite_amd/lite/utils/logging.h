// logging.h — LOG / CHECK in the convention of lite/utils/logging.h:186-206.  LOG(FATAL) and a failed CHECK
// throw paddle::lite::PaddleLiteException (the reference's behaviour under LITE_WITH_EXCEPTION; otherwise it
// aborts) so that hosts embedding the kernels can report the message.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <string>

namespace paddle {
namespace lite {

struct PaddleLiteException : public std::runtime_error {
  explicit PaddleLiteException(const std::string& m) : std::runtime_error(m) {}
};

class LogMessage {
 public:
  LogMessage(const char* file, int line, const char* level, bool fatal) : fatal_(fatal) {
    os_ << "[" << level << " " << file << ":" << line << "] ";
  }
  std::ostream& stream() { return os_; }
  ~LogMessage() noexcept(false) {
    if (fatal_) throw PaddleLiteException(os_.str());
    const char* v = std::getenv("GLOG_v");
    if (v && std::atoi(v) > 0) std::fprintf(stderr, "%s\n", os_.str().c_str());
  }

 private:
  std::ostringstream os_;
  bool fatal_;
};

struct LogVoidify {
  void operator&(std::ostream&) {}
};

}  // namespace lite
}  // namespace paddle

#define LITE_LOG_INFO paddle::lite::LogMessage(__FILE__, __LINE__, "I", false).stream()
#define LITE_LOG_WARNING paddle::lite::LogMessage(__FILE__, __LINE__, "W", false).stream()
#define LITE_LOG_ERROR paddle::lite::LogMessage(__FILE__, __LINE__, "E", false).stream()
#define LITE_LOG_FATAL paddle::lite::LogMessage(__FILE__, __LINE__, "F", true).stream()
#define LOG(level) LITE_LOG_##level
#define VLOG(n) LITE_LOG_INFO
#define CHECK(cond) \
  (cond) ? (void)0 : paddle::lite::LogVoidify() & LITE_LOG_FATAL << "Check failed: " #cond " "
#define CHECK_OP_(a, b, op) CHECK((a)op(b)) << "(" << (a) << " vs " << (b) << ") "
#define CHECK_EQ(a, b) CHECK_OP_(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP_(a, b, !=)
#define CHECK_LT(a, b) CHECK_OP_(a, b, <)
#define CHECK_LE(a, b) CHECK_OP_(a, b, <=)
#define CHECK_GT(a, b) CHECK_OP_(a, b, >)
#define CHECK_GE(a, b) CHECK_OP_(a, b, >=)
#define UNUSED __attribute__((unused))
