// png_writer.cc -- RGB8 PNG encoder (stored deflate blocks, CRC-32, Adler-32).
// Stands in for `img.save(output_file)` of the reference (src/main.rs:116); any valid PNG with the same
// pixels is equivalent.  Unlike the reference, write failures are reported.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrt_ffi.h"
#include "../csrc/rrt_internal.h"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0xFFFFFFFFu) {
    if (!crc_ready) crc_init();
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    put_be32(out, crc32(td.data(), td.size()) ^ 0xFFFFFFFFu);
}

}  // namespace

extern "C" rrt_status rrt_png_write(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8 || width == 0 || height == 0) {
        rrt::set_thread_error("rrt_png_write: bad argument");
        return RRT_ERR_INVALID;
    }
    // raw scanlines: filter byte 0 + RGB
    const size_t row = 1 + (size_t)width * 3;
    std::vector<uint8_t> raw(row * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw[y * row] = 0;
        std::memcpy(&raw[y * row + 1], rgb8 + (size_t)y * width * 3, (size_t)width * 3);
    }
    // zlib stream of stored blocks
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size()) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
    }
    put_be32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(2);  // colour type: RGB
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});

    FILE* f = std::fopen(path, "wb");
    if (!f) {
        rrt::set_thread_error(std::string("rrt_png_write: cannot open ") + path);
        return RRT_ERR_IO;
    }
    const size_t w = std::fwrite(out.data(), 1, out.size(), f);
    const int rc = std::fclose(f);
    if (w != out.size() || rc != 0) {
        rrt::set_thread_error(std::string("rrt_png_write: short write to ") + path);
        return RRT_ERR_IO;
    }
    return RRT_OK;
}
