// pngwalk.hpp -- what a host thread finds out about a PNG file before its bytes go to the GPU decoder
// (abub_png_decode_dev, abub_png.hip): the IDAT chunks and, for a palette image, the palette -> grey table.
#ifndef ABUB3HS_PNGWALK_HPP
#define ABUB3HS_PNGWALK_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "abub_hip.h"

namespace abub {

// What a host thread finds out about a PNG file before its bytes go to the GPU decoder: the IDAT chunks and, for a
// palette image, the palette -> grey table (the same rounding as cv::imdecode's, cvlite.cpp).  false = not an 8-bit grey
// or palette image of W x H without interlace (or not a PNG at all): such a file is decoded on the host.
struct PngInfo {
    std::vector<abub_png_seg> segs; // offsets relative to the file's first byte
    bool palette = false;
    uint8_t lut[256];
    uint64_t zlen = 0;
};
inline bool pngWalk(const uint8_t *buf, size_t size, int W, int H, PngInfo &out)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 + 25 || memcmp(buf, sig, 8) != 0)
        return false;
    auto be32 = [&](size_t o) { return ((uint32_t)buf[o] << 24) | ((uint32_t)buf[o + 1] << 16) | ((uint32_t)buf[o + 2] << 8) | (uint32_t)buf[o + 3]; };
    out.segs.clear();
    out.palette = false;
    out.zlen = 0;
    uint8_t pal[256][3];
    int npal = 0;
    bool haveHdr = false, end = false;
    size_t o = 8;
    uint32_t w = 0, h = 0, depth = 0, ctype = 0, interlace = 1;
    while (!end && o + 12 <= size) {
        const uint32_t len = be32(o);
        const uint8_t *type = buf + o + 4, *data = buf + o + 8;
        if (o + 12 + (size_t)len > size)
            return false;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(o + 8);
            h = be32(o + 12);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
            haveHdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            npal = std::min<int>((int)(len / 3), 256);
            memcpy(pal, data, (size_t)npal * 3);
        } else if (!memcmp(type, "IDAT", 4)) {
            out.segs.push_back(abub_png_seg{(uint32_t)(o + 8), len});
            out.zlen += len;
        } else if (!memcmp(type, "IEND", 4))
            end = true;
        o += 12 + (size_t)len;
    }
    if (!haveHdr || (int)w != W || (int)h != H || depth != 8 || (ctype != 0 && ctype != 3) || interlace != 0 || out.segs.empty() ||
        out.zlen >= ((uint64_t)1 << 28) || size >= ((size_t)1 << 31))
        return false;
    if (ctype == 3) {
        out.palette = true;
        for (int v = 0; v < 256; ++v)
            out.lut[v] = v < npal ? (uint8_t)((pal[v][0] * 9797 + pal[v][1] * 19234 + pal[v][2] * 3737 + 16384) >> 15) : 0;
    }
    return true;
}


} // namespace abub
#endif
