// stand-in (declarations only)
#pragma once
#include <cstddef>
namespace brisk {
struct Hamming {
  typedef unsigned char ValueType;
  typedef int ResultType;
  static unsigned int PopcntofSize(const unsigned char* a, const unsigned char* b, size_t size);
  static unsigned int PopcntofXORed(const unsigned char* signature1, const unsigned char* signature2, const int numberOf128BitWords);
  ResultType operator()(const unsigned char* a, const unsigned char* b, const int size) const;
};
}
