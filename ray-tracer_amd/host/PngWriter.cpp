#include "PngWriter.h"

#include <cstdint>
#include <fstream>
#include <iostream>
#include <vector>

namespace
{

std::uint32_t crcTable[256];
bool crcReady = false;

std::uint32_t crc32(const std::uint8_t* data, size_t n, std::uint32_t crc = 0)
{
    if (!crcReady)
    {
        for (std::uint32_t i = 0; i < 256; ++i)
        {
            std::uint32_t c = i;
            for (int k = 0; k < 8; ++k)
            {
                c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            }
            crcTable[i] = c;
        }
        crcReady = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i)
    {
        crc = crcTable[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    }
    return ~crc;
}

void put32(std::vector<std::uint8_t>& out, std::uint32_t v)
{
    out.insert(out.end(), {static_cast<std::uint8_t>(v >> 24), static_cast<std::uint8_t>(v >> 16),
                           static_cast<std::uint8_t>(v >> 8), static_cast<std::uint8_t>(v)});
}

void chunk(std::ofstream& file, const char type[4], const std::vector<std::uint8_t>& payload)
{
    std::vector<std::uint8_t> body(type, type + 4);
    body.insert(body.end(), payload.begin(), payload.end());
    std::vector<std::uint8_t> head;
    put32(head, static_cast<std::uint32_t>(payload.size()));
    std::vector<std::uint8_t> tail;
    put32(tail, crc32(body.data(), body.size()));
    file.write(reinterpret_cast<const char*>(head.data()), 4);
    file.write(reinterpret_cast<const char*>(body.data()), static_cast<std::streamsize>(body.size()));
    file.write(reinterpret_cast<const char*>(tail.data()), 4);
}

}  // namespace

bool PngWriter::writeImage(const std::filesystem::path& path, const Image& image, const std::string& title)
{
    std::cout << "---" << std::endl;
    std::cout << "Write image " << path.generic_string() << std::endl;
    const std::filesystem::path parent = path.parent_path();
    if (!parent.empty())
    {
        std::error_code ec;
        std::filesystem::create_directories(parent, ec);
        if (ec)
        {
            std::cout << "Failed to create output directory " << parent.generic_string() << ": " << ec.message()
                      << std::endl;
            return false;
        }
    }
    std::ofstream file(path, std::ios::binary);
    if (!file)
    {
        std::cout << "Failed to initialize png writer for " << path.generic_string() << std::endl;
        return false;
    }
    const std::uint32_t width = static_cast<std::uint32_t>(image.width());
    const std::uint32_t height = static_cast<std::uint32_t>(image.height());

    // raw scanlines: filter byte 0, then big-endian 16-bit R G B per pixel
    std::vector<std::uint8_t> raw;
    raw.reserve(static_cast<size_t>(height) * (1 + static_cast<size_t>(width) * 6));
    for (std::uint32_t y = 0; y < height; ++y)
    {
        raw.push_back(0);
        for (std::uint32_t x = 0; x < width; ++x)
        {
            const Pixel& p = image.getPixel(x, y);
            for (const std::uint16_t v : {p.red, p.green, p.blue})
            {
                raw.push_back(static_cast<std::uint8_t>(v >> 8));
                raw.push_back(static_cast<std::uint8_t>(v & 0xFF));
            }
        }
    }
    // zlib stream of stored deflate blocks
    std::vector<std::uint8_t> z = {0x78, 0x01};
    std::uint32_t a = 1, b = 0;
    for (const std::uint8_t v : raw)
    {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    size_t at = 0;
    do
    {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        const bool last = at + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(static_cast<std::uint8_t>(n & 0xFF));
        z.push_back(static_cast<std::uint8_t>(n >> 8));
        z.push_back(static_cast<std::uint8_t>(~n & 0xFF));
        z.push_back(static_cast<std::uint8_t>((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + static_cast<std::ptrdiff_t>(at), raw.begin() + static_cast<std::ptrdiff_t>(at + n));
        at += n;
    } while (at < raw.size());
    put32(z, (b << 16) | a);

    const std::uint8_t signature[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    file.write(reinterpret_cast<const char*>(signature), 8);
    std::vector<std::uint8_t> ihdr;
    put32(ihdr, width);
    put32(ihdr, height);
    ihdr.insert(ihdr.end(), {16, 2, 0, 0, 0});  // depth 16, colour type RGB, deflate, adaptive, no interlace
    chunk(file, "IHDR", ihdr);
    if (!title.empty())
    {
        std::vector<std::uint8_t> text = {'T', 'i', 't', 'l', 'e', 0};
        text.insert(text.end(), title.begin(), title.end());
        chunk(file, "tEXt", text);
    }
    chunk(file, "IDAT", z);
    chunk(file, "IEND", {});
    file.flush();
    if (!file)
    {
        std::cout << "Failed to write png " << path.generic_string() << std::endl;
        return false;
    }
    return true;
}
