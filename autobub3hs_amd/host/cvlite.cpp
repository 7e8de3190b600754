// cvlite.cpp -- BMP reader for the mask images (cam_masks/<series>/camN[_bellows]_mask.bmp are 1-bit
// or 8-bit uncompressed BMPs).  Grey conversion of palette / BGR pixels uses OpenCV's fixed-point
// BGR2GRAY weights (B 1868, G 9617, R 4899, >> 14).
#include "cvlite.hpp"

#ifndef ABUB_USE_OPENCV
#include <cstdio>
#include <vector>

namespace cv {

static inline uchar grey(int b, int g, int r) { return (uchar)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14); }

Mat imread(const std::string &path, int)
{
    Mat out;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return out;
    std::vector<uchar> buf;
    uchar tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0)
        buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    auto u16 = [&](size_t o) { return (unsigned)buf[o] | ((unsigned)buf[o + 1] << 8); };
    auto u32 = [&](size_t o) { return (unsigned)buf[o] | ((unsigned)buf[o + 1] << 8) | ((unsigned)buf[o + 2] << 16) | ((unsigned)buf[o + 3] << 24); };
    if (buf.size() < 54 || buf[0] != 'B' || buf[1] != 'M')
        return out;
    const unsigned dataOff = u32(10), hdr = u32(14);
    if (hdr < 40)
        return out;
    const int w = (int)u32(18);
    int h = (int)u32(22);
    const unsigned bpp = u16(28), comp = u32(30);
    unsigned ncol = u32(46);
    bool topDown = false;
    if (h < 0) {
        h = -h;
        topDown = true;
    }
    if (w <= 0 || h <= 0 || (comp != 0 && !(comp == 3 && bpp == 32)) ||
        !(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32))
        return out;
    if (ncol == 0 && bpp <= 8)
        ncol = 1u << bpp;
    uchar pal[256];
    for (unsigned i = 0; i < 256; ++i)
        pal[i] = (uchar)i;
    if (bpp <= 8) {
        size_t po = 14 + hdr;
        for (unsigned i = 0; i < ncol && po + 4 * i + 3 < buf.size(); ++i)
            pal[i] = grey(buf[po + 4 * i], buf[po + 4 * i + 1], buf[po + 4 * i + 2]);
    }
    const size_t stride = ((size_t)w * bpp + 31) / 32 * 4;
    if ((size_t)dataOff + stride * h > buf.size())
        return out;
    out.create(h, w, CV_8U);
    for (int y = 0; y < h; ++y) {
        const uchar *src = buf.data() + dataOff + stride * (topDown ? y : h - 1 - y);
        uchar *dst = out.ptr<uchar>(y);
        for (int x = 0; x < w; ++x) {
            switch (bpp) {
            case 1: dst[x] = pal[(src[x >> 3] >> (7 - (x & 7))) & 1]; break;
            case 4: dst[x] = pal[(src[x >> 1] >> ((x & 1) ? 0 : 4)) & 15]; break;
            case 8: dst[x] = pal[src[x]]; break;
            case 24: dst[x] = grey(src[3 * x], src[3 * x + 1], src[3 * x + 2]); break;
            default: dst[x] = grey(src[4 * x], src[4 * x + 1], src[4 * x + 2]); break;
            }
        }
    }
    return out;
}

} // namespace cv
#endif
