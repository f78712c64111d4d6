// images.h -- 24-bit BMP dump of a scalar image, the role of the reference's
// tests/helper/images.hpp:14-99 (used by tests/project_gadget/project_gadget.cu:98-112).
// Pixel (i, j) of a row-major width x height image maps to the colour
// (v - min) / (max - min) * (r_max, g_max, b_max), clamped; rows are stored bottom-up, BGR,
// each row padded to a multiple of four bytes.  Host-only.
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

template <typename T>
inline void make_bitmap(const T* img_src, size_t width, size_t height, T img_min, T img_max,
                        const std::string& img_fname, unsigned char r_max = 150,
                        unsigned char g_max = 210, unsigned char b_max = 255)
{
    const size_t row_bytes = (3 * width + 3) / 4 * 4;
    std::vector<unsigned char> file(54 + row_bytes * height, 0);
    auto put32 = [&](size_t at, uint32_t v) {
        for (int k = 0; k < 4; ++k) file[at + k] = (v >> (8 * k)) & 0xff;
    };
    file[0] = 'B'; file[1] = 'M';
    put32(2, uint32_t(file.size()));
    put32(10, 54);                     // offset of the pixel array
    put32(14, 40);                     // BITMAPINFOHEADER
    put32(18, uint32_t(width));
    put32(22, uint32_t(height));
    file[26] = 1;                      // colour planes
    file[28] = 24;                     // bits per pixel
    const double range = double(img_max) - double(img_min);
    const unsigned char top[3] = { b_max, g_max, r_max };
    for (size_t j = 0; j < height; ++j) {
        unsigned char* row = &file[54 + row_bytes * (height - 1 - j)];
        for (size_t i = 0; i < width; ++i) {
            double f = (double(img_src[i + j * width]) - double(img_min)) / range;
            f = f < 0 ? 0 : (f > 1 ? 1 : f);          // also maps NaN / -inf (log of 0) to 0
            if (!(f == f)) f = 0;
            for (int c = 0; c < 3; ++c) row[3 * i + c] = (unsigned char)(f * top[c]);
        }
    }
    std::FILE* fp = std::fopen(img_fname.c_str(), "wb");
    if (!fp) throw std::runtime_error("cannot write " + img_fname);
    std::fwrite(file.data(), 1, file.size(), fp);
    std::fclose(fp);
}
