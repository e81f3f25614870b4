// lzss_legacy.h -- lz.Compress (lzss.go:224), host only: see lzss_legacy.cpp
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace rsn {
void lzss_compress_legacy_host(const uint8_t *in, size_t n, int64_t window, std::string &out);
}
