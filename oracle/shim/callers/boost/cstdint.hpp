#pragma once
#include <cstdint>
namespace boost { using std::uint8_t; using std::uint16_t; using std::uint32_t; using std::uint64_t; using std::int32_t; using std::int64_t; }
