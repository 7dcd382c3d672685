// stand-in for okvis_util/include/okvis/assert_macros.hpp:52-58
#pragma once
#include <stdexcept>
#include <string>
#define OKVIS_DEFINE_EXCEPTION(exceptionName, exceptionParent)               \
  class exceptionName : public exceptionParent {                             \
   public:                                                                   \
    exceptionName(const char* message) : exceptionParent(message) {}         \
    exceptionName(std::string const& message) : exceptionParent(message) {}  \
    virtual ~exceptionName() throw() {}                                      \
  };
