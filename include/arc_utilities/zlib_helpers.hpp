// zlib wrappers with the interface of arc_utilities' ZlibHelpers (used at src/sdf_tools/sdf.cpp:392-483).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <stdexcept>
#include <vector>

namespace ZlibHelpers {

inline std::vector<uint8_t> CompressBytes(const std::vector<uint8_t>& uncompressed) {
    uLongf bound = compressBound((uLong)uncompressed.size());
    std::vector<uint8_t> out(bound);
    const int rc = compress2(out.data(), &bound, uncompressed.data(), (uLong)uncompressed.size(), Z_BEST_COMPRESSION);
    if (rc != Z_OK) throw std::runtime_error("ZLIB compression failed");
    out.resize(bound);
    return out;
}

inline std::vector<uint8_t> DecompressBytes(const std::vector<uint8_t>& compressed) {
    z_stream strm{};
    if (inflateInit(&strm) != Z_OK) throw std::runtime_error("ZLIB inflateInit failed");
    strm.next_in = const_cast<Bytef*>(compressed.data());
    strm.avail_in = (uInt)compressed.size();
    std::vector<uint8_t> out;
    std::vector<uint8_t> chunk(1 << 16);
    int rc = Z_OK;
    do {
        strm.next_out = chunk.data();
        strm.avail_out = (uInt)chunk.size();
        rc = inflate(&strm, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&strm); throw std::runtime_error("ZLIB decompression failed"); }
        out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - strm.avail_out));
    } while (rc != Z_STREAM_END);
    inflateEnd(&strm);
    return out;
}

}  // namespace ZlibHelpers
