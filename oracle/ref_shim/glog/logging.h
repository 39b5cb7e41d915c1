// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for <glog/logging.h> (the functor headers log nothing).
#pragma once
#include <ostream>
// LOG(severity) << ... : swallowed (src/backend.cpp:38 logs its tick time)
namespace lvf_ref_shim { struct NullLog { template <typename T> NullLog& operator<<(const T&) { return *this; } NullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; } }; }
#define LOG(severity) ::lvf_ref_shim::NullLog()
