// TEST INFRASTRUCTURE (oracle/ref_shim/README.md). Stands in for the reference's glog front end
// (src/colmap/util/logging.h): glog is not installed in this image. Same macro names and
// streaming syntax; THROW_CHECK* throw std::invalid_argument, LOG(x) writes to stderr.
#pragma once

#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>

namespace ref_shim {

class LogLine {
 public:
  explicit LogLine(const char* severity) { stream_ << "[ref " << severity << "] "; }
  ~LogLine() { std::cerr << stream_.str() << std::endl; }
  std::ostream& stream() { return stream_; }

 private:
  std::ostringstream stream_;
};

class ThrowLine {
 public:
  ThrowLine(const char* file, int line, const char* what) {
    stream_ << "[" << file << ":" << line << "] Check failed: " << what << " ";
  }
  ~ThrowLine() noexcept(false) { throw std::invalid_argument(stream_.str()); }
  std::ostream& stream() { return stream_; }

 private:
  std::ostringstream stream_;
};

struct Voidify {
  void operator&(std::ostream&) {}
};

}  // namespace ref_shim

#define LOG(severity) ref_shim::LogLine(#severity).stream()
#define VLOG(level) \
  if (true) {       \
  } else            \
    ref_shim::LogLine("V").stream()

#define THROW_CHECK(condition) \
  (condition) ? (void)0 : ref_shim::Voidify() & ref_shim::ThrowLine(__FILE__, __LINE__, #condition).stream()
#define REF_SHIM_CHECK_OP(op, a, b) \
  ((a)op(b)) ? (void)0 : ref_shim::Voidify() & ref_shim::ThrowLine(__FILE__, __LINE__, #a " " #op " " #b).stream()
#define THROW_CHECK_EQ(a, b) REF_SHIM_CHECK_OP(==, a, b)
#define THROW_CHECK_NE(a, b) REF_SHIM_CHECK_OP(!=, a, b)
#define THROW_CHECK_LE(a, b) REF_SHIM_CHECK_OP(<=, a, b)
#define THROW_CHECK_LT(a, b) REF_SHIM_CHECK_OP(<, a, b)
#define THROW_CHECK_GE(a, b) REF_SHIM_CHECK_OP(>=, a, b)
#define THROW_CHECK_GT(a, b) REF_SHIM_CHECK_OP(>, a, b)
#define CHECK(condition) THROW_CHECK(condition)
#define CHECK_EQ(a, b) THROW_CHECK_EQ(a, b)
#define CHECK_NE(a, b) THROW_CHECK_NE(a, b)
#define CHECK_LE(a, b) THROW_CHECK_LE(a, b)
#define CHECK_LT(a, b) THROW_CHECK_LT(a, b)
#define CHECK_GE(a, b) THROW_CHECK_GE(a, b)
#define CHECK_GT(a, b) THROW_CHECK_GT(a, b)
