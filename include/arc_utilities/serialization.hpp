// In-tree stand-in for the slice of UM-ARM-Lab/arc_utilities' serialization.hpp that
// SignedDistanceField / CollisionMapGrid use (reference call sites: src/sdf_tools/sdf.cpp:213-258,
// src/sdf_tools/collision_map.cpp:21-67).  arc_utilities is an un-vendored, un-pinned dependency of
// the reference (package.xml:21,35), so the byte format below follows its published behaviour
// (little-endian memcpy of PODs; uint64 element count before vectors and strings; 16 column-major
// doubles for an Isometry3d) and is marked "wire-format parity unpinned" in DESIGN.md.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sdf_tools/eigen_lite.hpp"

namespace arc_utilities {

template <typename T>
inline uint64_t SerializeFixedSizePOD(const T& item, std::vector<uint8_t>& buffer) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&item);
    buffer.insert(buffer.end(), p, p + sizeof(T));
    return sizeof(T);
}

template <typename T>
inline std::pair<T, uint64_t> DeserializeFixedSizePOD(const std::vector<uint8_t>& buffer, const uint64_t current) {
    if (current > buffer.size() || sizeof(T) > buffer.size() - current) throw std::invalid_argument("Not enough room in the provided buffer");
    T item;
    std::memcpy(&item, buffer.data() + current, sizeof(T));
    return std::make_pair(item, (uint64_t)sizeof(T));
}

template <typename T>
inline uint64_t SerializeVector(const std::vector<T>& v, std::vector<uint8_t>& buffer,
                                const std::function<uint64_t(const T&, std::vector<uint8_t>&)>& item_serializer) {
    const uint64_t start = buffer.size();
    SerializeFixedSizePOD<uint64_t>((uint64_t)v.size(), buffer);
    for (const T& item : v) item_serializer(item, buffer);
    return buffer.size() - start;
}

template <typename T>
inline std::pair<std::vector<T>, uint64_t> DeserializeVector(
    const std::vector<uint8_t>& buffer, const uint64_t current,
    const std::function<std::pair<T, uint64_t>(const std::vector<uint8_t>&, const uint64_t)>& item_deserializer) {
    uint64_t pos = current;
    const auto n = DeserializeFixedSizePOD<uint64_t>(buffer, pos);
    pos += n.second;
    if (pos > buffer.size() || n.first > (uint64_t)(buffer.size() - pos)) throw std::invalid_argument("Not enough room in the provided buffer");
    std::vector<T> out;
    out.reserve((size_t)n.first);
    for (uint64_t i = 0; i < n.first; i++) {
        const auto item = item_deserializer(buffer, pos);
        out.push_back(item.first);
        pos += item.second;
    }
    return std::make_pair(std::move(out), pos - current);
}

inline uint64_t SerializeString(const std::string& s, std::vector<uint8_t>& buffer) {
    SerializeFixedSizePOD<uint64_t>((uint64_t)s.size(), buffer);
    buffer.insert(buffer.end(), s.begin(), s.end());
    return sizeof(uint64_t) + s.size();
}

inline std::pair<std::string, uint64_t> DeserializeString(const std::vector<uint8_t>& buffer, const uint64_t current) {
    const auto n = DeserializeFixedSizePOD<uint64_t>(buffer, current);
    if (current + n.second > buffer.size() || n.first > (uint64_t)(buffer.size() - current - n.second)) throw std::invalid_argument("Not enough room in the provided buffer");
    std::string s(reinterpret_cast<const char*>(buffer.data() + current + n.second), (size_t)n.first);
    return std::make_pair(s, n.second + n.first);
}

inline uint64_t SerializeIsometry3d(const Eigen::Isometry3d& t, std::vector<uint8_t>& buffer) {
    double m[16];
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) m[c * 4 + r] = t.matrix()(r, c);
    const uint8_t* p = reinterpret_cast<const uint8_t*>(m);
    buffer.insert(buffer.end(), p, p + sizeof m);
    return sizeof m;
}

inline std::pair<Eigen::Isometry3d, uint64_t> DeserializeIsometry3d(const std::vector<uint8_t>& buffer, const uint64_t current) {
    double m[16];
    if (current + sizeof m > buffer.size()) throw std::invalid_argument("Not enough room in the provided buffer");
    std::memcpy(m, buffer.data() + current, sizeof m);
    Eigen::Isometry3d t = Eigen::Isometry3d::Identity();
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) t.matrix()(r, c) = m[c * 4 + r];
    return std::make_pair(t, (uint64_t)sizeof m);
}

// vector<double> as a fixed-size POD-ish item (bindings.cpp:85 DeserializeFixedSizePOD<vector<double>>
// is used for 3-element gradient cells): uint64 count + doubles.
inline uint64_t SerializeVectorOfDoubles(const std::vector<double>& v, std::vector<uint8_t>& buffer) {
    return SerializeVector<double>(v, buffer, SerializeFixedSizePOD<double>);
}
inline std::pair<std::vector<double>, uint64_t> DeserializeVectorOfDoubles(const std::vector<uint8_t>& buffer, const uint64_t current) {
    return DeserializeVector<double>(buffer, current, DeserializeFixedSizePOD<double>);
}

}  // namespace arc_utilities
