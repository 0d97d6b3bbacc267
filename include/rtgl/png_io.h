// png_io.h -- minimal PNG reader / writer on top of zlib (the reference uses stb_image / stb_image_write
// through src/gfx/image.cpp:60-78; neither is available here).  Reads 8-bit, non-interlaced greyscale,
// grey+alpha, RGB and RGBA files into 3- or 4-channel images (what stbi_load returns for them and what
// the cube-map upload consumes, src/gfx/gl.cpp:246-252); writes 8-bit RGB / RGBA.  Link with -lz.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rtgl {

struct Image8 { int width = 0, height = 0, channels = 0; std::vector<uint8_t> pixels; };

namespace detail {
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
inline int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace detail

inline bool read_png(const std::string &path, Image8 &out, bool flip_vertically = false)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (pos + 12 <= file.size()) {
        uint32_t len = detail::be32(&file[pos]);
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) return false;
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = (int)detail::be32(data); h = (int)detail::be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    int src_ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (w <= 0 || h <= 0 || depth != 8 || src_ch == 0 || interlace != 0) return false;
    const size_t stride = (size_t)w * src_ch;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; ++y) {                              // undo the per-row filters (PNG spec section 9)
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *in = &raw[(stride + 1) * y + 1];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)src_ch ? cur[i - src_ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)src_ch) ? up[i - src_ch] : 0;
            int v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += detail::paeth(a, b, c); break;
            default: return false;
            }
            cur[i] = (uint8_t)v;
        }
    }
    out.width = w; out.height = h; out.channels = (src_ch == 1 || src_ch == 3) ? 3 : 4;
    out.pixels.resize((size_t)w * h * out.channels);
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = &img[stride * (flip_vertically ? h - 1 - y : y)];
        uint8_t *d = &out.pixels[(size_t)w * out.channels * y];
        for (int x = 0; x < w; ++x, s += src_ch, d += out.channels) {
            if (src_ch >= 3) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; if (src_ch == 4) d[3] = s[3]; }
            else { d[0] = d[1] = d[2] = s[0]; if (src_ch == 2) d[3] = s[1]; }
        }
    }
    return true;
}

inline bool write_png(const std::string &path, const uint8_t *pixels, int w, int h, int channels)
{
    if (w <= 0 || h <= 0 || (channels != 3 && channels != 4)) return false;
    const size_t stride = (size_t)w * channels;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    for (int y = 0; y < h; ++y) {
        raw[(stride + 1) * y] = 0;
        std::memcpy(&raw[(stride + 1) * y + 1], pixels + stride * y, stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char *type, const uint8_t *data, size_t len) {
        detail::put32(out, (uint32_t)len);
        size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        if (len) out.insert(out.end(), data, data + len);
        detail::put32(out, (uint32_t)crc32(0L, &out[start], (uInt)(len + 4)));
    };
    uint8_t ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w; ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = 8; ihdr[9] = channels == 3 ? 2 : 6; ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), clen);
    chunk("IEND", nullptr, 0);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

}  // namespace rtgl
