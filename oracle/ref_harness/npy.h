// Minimal .npy (format 1.0) writer so dump tools can hand arrays to numpy.load.
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace npy
{

// descr: "<f8", "<f4", "<i4", "<i8", "<u4", "|u1" ...
inline void write(const std::string& path, const std::string& descr,
                  const std::vector<size_t>& shape, const void* data, size_t itemSize)
{
    std::string shapeText = "(";
    size_t count = 1;
    for (size_t i = 0; i < shape.size(); ++i)
    {
        shapeText += std::to_string(shape[i]);
        shapeText += (shape.size() == 1 || i + 1 < shape.size()) ? "," : "";
        count *= shape[i];
    }
    shapeText += ")";
    std::string header = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': " + shapeText + ", }";
    const size_t preamble = 10;
    size_t total = preamble + header.size() + 1;
    const size_t padded = (total + 63) / 64 * 64;
    header.append(padded - total, ' ');
    header.push_back('\n');

    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f)
    {
        throw std::runtime_error("cannot open " + path);
    }
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const uint16_t headerLength = static_cast<uint16_t>(header.size());
    std::fwrite(magic, 1, 8, f);
    std::fwrite(&headerLength, 2, 1, f);
    std::fwrite(header.data(), 1, header.size(), f);
    if (count > 0)
    {
        std::fwrite(data, itemSize, count, f);
    }
    std::fclose(f);
}

inline void writeF64(const std::string& path, const std::vector<double>& v, const std::vector<size_t>& shape)
{
    write(path, "<f8", shape, v.data(), 8);
}
inline void writeF32(const std::string& path, const std::vector<float>& v, const std::vector<size_t>& shape)
{
    write(path, "<f4", shape, v.data(), 4);
}
inline void writeI32(const std::string& path, const std::vector<int32_t>& v, const std::vector<size_t>& shape)
{
    write(path, "<i4", shape, v.data(), 4);
}
inline void writeI64(const std::string& path, const std::vector<int64_t>& v, const std::vector<size_t>& shape)
{
    write(path, "<i8", shape, v.data(), 8);
}

}  // namespace npy
