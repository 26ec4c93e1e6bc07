#include "PngWriter.h"

#include <cstdint>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <thread>
#include <vector>

namespace
{

// CRC-32 (ISO 3309, as PNG asks), eight bytes per step through eight tables.
std::uint32_t crcTable[8][256];
bool crcReady = false;

void prepareCrc()
{
    if (crcReady)
    {
        return;
    }
    for (std::uint32_t i = 0; i < 256; ++i)
    {
        std::uint32_t c = i;
        for (int k = 0; k < 8; ++k)
        {
            c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        }
        crcTable[0][i] = c;
    }
    for (std::uint32_t i = 0; i < 256; ++i)
    {
        for (int t = 1; t < 8; ++t)
        {
            crcTable[t][i] = crcTable[0][crcTable[t - 1][i] & 0xFFu] ^ (crcTable[t - 1][i] >> 8);
        }
    }
    crcReady = true;
}

// running CRC: start from 0xFFFFFFFF, invert at the end
std::uint32_t crcUpdate(std::uint32_t crc, const std::uint8_t* data, size_t n)
{
    prepareCrc();
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
    {
        const std::uint32_t lo = crc ^ (static_cast<std::uint32_t>(data[i]) | (static_cast<std::uint32_t>(data[i + 1]) << 8) |
                                        (static_cast<std::uint32_t>(data[i + 2]) << 16) |
                                        (static_cast<std::uint32_t>(data[i + 3]) << 24));
        crc = crcTable[7][lo & 0xFFu] ^ crcTable[6][(lo >> 8) & 0xFFu] ^ crcTable[5][(lo >> 16) & 0xFFu] ^
              crcTable[4][lo >> 24] ^ crcTable[3][data[i + 4]] ^ crcTable[2][data[i + 5]] ^ crcTable[1][data[i + 6]] ^
              crcTable[0][data[i + 7]];
    }
    for (; i < n; ++i)
    {
        crc = crcTable[0][(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    }
    return crc;
}

void put32(std::vector<std::uint8_t>& out, std::uint32_t v)
{
    out.insert(out.end(), {static_cast<std::uint8_t>(v >> 24), static_cast<std::uint8_t>(v >> 16),
                           static_cast<std::uint8_t>(v >> 8), static_cast<std::uint8_t>(v)});
}

void chunk(std::ofstream& file, const char type[4], const std::vector<std::uint8_t>& payload)
{
    std::vector<std::uint8_t> head;
    put32(head, static_cast<std::uint32_t>(payload.size()));
    std::uint32_t crc = crcUpdate(0xFFFFFFFFu, reinterpret_cast<const std::uint8_t*>(type), 4);
    crc = ~crcUpdate(crc, payload.data(), payload.size());
    std::vector<std::uint8_t> tail;
    put32(tail, crc);
    file.write(reinterpret_cast<const char*>(head.data()), 4);
    file.write(type, 4);
    file.write(reinterpret_cast<const char*>(payload.data()), static_cast<std::streamsize>(payload.size()));
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
    const size_t rowBytes = 1 + static_cast<size_t>(width) * 6;
    std::vector<std::uint8_t> raw(static_cast<size_t>(height) * rowBytes);
    {
        const size_t workers = std::max<size_t>(1, std::min<size_t>({16, std::thread::hardware_concurrency(), height}));
        const size_t band = (height + workers - 1) / workers;
        auto fill = [&](size_t firstRow, size_t lastRow) {
            for (size_t y = firstRow; y < lastRow; ++y)
            {
                std::uint8_t* row = raw.data() + y * rowBytes;
                *row++ = 0;
                for (std::uint32_t x = 0; x < width; ++x)
                {
                    const Pixel& p = image.getPixel(x, static_cast<std::uint32_t>(y));
                    for (const std::uint16_t v : {p.red, p.green, p.blue})
                    {
                        *row++ = static_cast<std::uint8_t>(v >> 8);
                        *row++ = static_cast<std::uint8_t>(v & 0xFF);
                    }
                }
            }
        };
        std::vector<std::thread> threads;
        for (size_t w = 1; w < workers; ++w)
        {
            const size_t first = w * band, last = std::min<size_t>(height, first + band);
            if (first < last)
            {
                threads.emplace_back(fill, first, last);
            }
        }
        fill(0, std::min<size_t>(height, band));
        for (std::thread& t : threads)
        {
            t.join();
        }
    }
    // zlib stream of stored deflate blocks; Adler-32 with the modulo taken once per 5552 bytes
    // (the longest run for which the sums stay below 2^32)
    std::vector<std::uint8_t> z;
    z.reserve(raw.size() + 5 * (raw.size() / 65535 + 1) + 6);
    z.push_back(0x78);
    z.push_back(0x01);
    std::uint32_t a = 1, b = 0;
    for (size_t at = 0; at < raw.size();)
    {
        const size_t n = std::min<size_t>(5552, raw.size() - at);
        for (size_t i = 0; i < n; ++i)
        {
            a += raw[at + i];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        at += n;
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
