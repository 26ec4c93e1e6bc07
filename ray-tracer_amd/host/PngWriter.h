// 16-bit RGB PNG output (reference src/PngWriter.cpp, src/PngWriteSession.cpp:118-173:
// bit depth 16, colour type RGB, no interlace, a "Title" text chunk).  The reference
// hands the rows to libpng; this writer emits the same image as a valid PNG with its own
// chunk, CRC-32 and Adler-32 code and stored (uncompressed) deflate blocks, so it needs no
// library.  Pixel values, dimensions and row order are identical; file bytes are not
// (libpng compresses).
#pragma once

#include "Renderer.h"

#include <filesystem>
#include <string>

namespace PngWriter
{
bool writeImage(const std::filesystem::path& path, const Image& image, const std::string& title);
}
