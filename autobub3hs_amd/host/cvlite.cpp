// cvlite.cpp -- grey decoders for the two container formats PICO data comes in: PNG (camera frames,
// bellows templates) and BMP (older series, mask files).  Decode only; all image processing is on the GPU.
//  * BGR/palette -> grey for BMP uses OpenCV's fixed-point BGR2GRAY weights (B 1868, G 9617, R 4899, >> 14).
//  * PNG colour -> grey follows libpng's png_set_rgb_to_gray(0.299, 0.587) as OpenCV's PNG reader requests it:
//    (r*9797 + g*19234 + b*3737 + 16384) >> 15.  Both are the identity on grey palettes / grey pixels, which is
//    what PICO frames are (palette-grey PNG); exactness on coloured inputs is unpinned (no OpenCV here).
#include "cvlite.hpp"

#ifndef ABUB_USE_OPENCV
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <dlfcn.h>
#include <zlib.h>

namespace cv {

// Optional fast path for the inflate of PNG / zip payloads: libdeflate (2-3x zlib's speed), when the system has the
// shared library (no headers needed: three functions of its stable C ABI, looked up at run time).  Absent -> zlib.
namespace {
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*zlibDecompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*freeD)(void *) = nullptr;
    LibDeflate()
    {
        const char *off = getenv("ABUB_NO_LIBDEFLATE");
        if (off && atoi(off))
            return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        zlibDecompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_zlib_decompress");
        freeD = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !zlibDecompress || !freeD)
            alloc = nullptr;
    }
};
const LibDeflate &libDeflate()
{
    static LibDeflate l;
    return l;
}
struct ThreadDecompressor {
    void *d = nullptr;
    ~ThreadDecompressor()
    {
        if (d)
            libDeflate().freeD(d);
    }
};
// zlib-wrapped deflate stream -> exactly `outLen` bytes; false on any error
bool inflateZlibStream(const uchar *in, size_t inLen, uchar *out, size_t outLen)
{
    const LibDeflate &l = libDeflate();
    if (l.alloc) {
        static thread_local ThreadDecompressor td;
        if (!td.d)
            td.d = l.alloc();
        if (td.d) {
            size_t got = 0;
            if (l.zlibDecompress(td.d, in, inLen, out, outLen, &got) == 0 && got == outLen)
                return true;
            // (fall through: let zlib have a look, e.g. trailing data libdeflate refuses)
        }
    }
    uLongf rawLen = (uLongf)outLen;
    return uncompress(out, &rawLen, in, (uLong)inLen) == Z_OK && rawLen == outLen;
}
} // namespace

static inline uchar bgr2grey(int b, int g, int r) { return (uchar)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14); }
static inline uchar png_rgb2grey(int r, int g, int b) { return (uchar)((r * 9797 + g * 19234 + b * 3737 + 16384) >> 15); }

// ---------------------------------------------------------------------------------------------
// BMP
// ---------------------------------------------------------------------------------------------
static Mat decodeBMP(const uchar *buf, size_t size)
{
    Mat out;
    auto u16 = [&](size_t o) { return (unsigned)buf[o] | ((unsigned)buf[o + 1] << 8); };
    auto u32 = [&](size_t o) { return (unsigned)buf[o] | ((unsigned)buf[o + 1] << 8) | ((unsigned)buf[o + 2] << 16) | ((unsigned)buf[o + 3] << 24); };
    if (size < 54 || buf[0] != 'B' || buf[1] != 'M')
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
        for (unsigned i = 0; i < ncol && i < 256 && po + 4 * i + 3 < size; ++i)
            pal[i] = bgr2grey(buf[po + 4 * i], buf[po + 4 * i + 1], buf[po + 4 * i + 2]);
    }
    const size_t stride = ((size_t)w * bpp + 31) / 32 * 4;
    if ((size_t)dataOff + stride * h > size)
        return out;
    out.create(h, w, CV_8U);
    for (int y = 0; y < h; ++y) {
        const uchar *src = buf + dataOff + stride * (topDown ? y : h - 1 - y);
        uchar *dst = out.ptr<uchar>(y);
        for (int x = 0; x < w; ++x) {
            switch (bpp) {
            case 1: dst[x] = pal[(src[x >> 3] >> (7 - (x & 7))) & 1]; break;
            case 4: dst[x] = pal[(src[x >> 1] >> ((x & 1) ? 0 : 4)) & 15]; break;
            case 8: dst[x] = pal[src[x]]; break;
            case 24: dst[x] = bgr2grey(src[3 * x], src[3 * x + 1], src[3 * x + 2]); break;
            default: dst[x] = bgr2grey(src[4 * x], src[4 * x + 1], src[4 * x + 2]); break;
            }
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// PNG (non-interlaced)
// ---------------------------------------------------------------------------------------------
static inline int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Where a decoder puts its pixels: asked once, after the header, for a w x h byte image (row-major, pitch w); nullptr =
// the caller does not want an image of that size.
struct PixelSink {
    uchar *(*get)(void *ctx, unsigned w, unsigned h);
    void *ctx;
};

// PNG -> 8-bit grey.  The two large temporaries (concatenated IDAT data, inflated scanlines) are thread-local and keep
// their capacity: a decode thread of the batched ingestion path allocates nothing per frame (per-frame megabyte
// allocations -- mmap / munmap / page faults under one mm lock -- are what kept 128 decode threads at the rate of 16).
static bool decodePNGTo(const uchar *buf, size_t size, PixelSink sink)
{
    static const uchar sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 + 25 || memcmp(buf, sig, 8) != 0)
        return false;
    auto be32 = [&](size_t o) { return ((unsigned)buf[o] << 24) | ((unsigned)buf[o + 1] << 16) | ((unsigned)buf[o + 2] << 8) | (unsigned)buf[o + 3]; };
    unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    uchar pal[256][3];
    int npal = 0;
    static thread_local std::vector<uchar> idat, raw;
    idat.clear();
    size_t o = 8;
    bool haveHdr = false, end = false;
    while (!end && o + 12 <= size) {
        const unsigned len = be32(o);
        const uchar *type = buf + o + 4;
        const uchar *data = buf + o + 8;
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
            npal = (int)(len / 3);
            if (npal > 256)
                npal = 256;
            for (int i = 0; i < npal; ++i) {
                pal[i][0] = data[3 * i];
                pal[i][1] = data[3 * i + 1];
                pal[i][2] = data[3 * i + 2];
            }
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            end = true;
        }
        o += 12 + (size_t)len;
    }
    if (!haveHdr || w == 0 || h == 0 || interlace != 0 || idat.empty())
        return false;
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return false;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16))
        return false;
    if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8)
        return false;
    if (ctype == 3 && depth == 16)
        return false;
    const size_t bitsPerPixel = (size_t)channels * depth;
    const size_t rowBytes = ((size_t)w * bitsPerPixel + 7) / 8;
    const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1; // filter distance
    if (raw.size() < (rowBytes + 1) * (size_t)h)
        raw.resize((rowBytes + 1) * (size_t)h);
    if (!inflateZlibStream(idat.data(), idat.size(), raw.data(), (rowBytes + 1) * (size_t)h))
        return false;
    uchar *const base = sink.get(sink.ctx, w, h);
    if (!base)
        return false;
    // grey / palette lookup tables (the frames of a run are 8-bit grey or 8-bit palettised: one table look-up per pixel)
    uchar lut[256];
    for (int v = 0; v < 256; ++v)
        lut[v] = ctype == 3 ? (v < npal ? png_rgb2grey(pal[v][0], pal[v][1], pal[v][2]) : 0) : (uchar)v;
    const bool direct8 = ctype == 0 && depth == 8; // rows can be unfiltered straight into the output image
    std::vector<uchar> prev(rowBytes, 0), cur(rowBytes);
    std::vector<uchar> zero(rowBytes, 0);
    for (unsigned y = 0; y < h; ++y) {
        const uchar *src = raw.data() + (rowBytes + 1) * (size_t)y;
        const int ft = src[0];
        ++src;
        // the filter type is constant along a row: one tight loop per type (the row above is `up`, zeros for row 0)
        uchar *rowOut = direct8 ? (base + (size_t)y * w) : cur.data();
        const uchar *up = y == 0 ? zero.data() : (direct8 ? (base + (size_t)(y - 1) * w) : prev.data());
        switch (ft) {
        case 0:
            std::memcpy(rowOut, src, rowBytes);
            break;
        case 1:
            for (size_t i = 0; i < bpp && i < rowBytes; ++i)
                rowOut[i] = src[i];
            for (size_t i = bpp; i < rowBytes; ++i)
                rowOut[i] = (uchar)(src[i] + rowOut[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < rowBytes; ++i)
                rowOut[i] = (uchar)(src[i] + up[i]);
            break;
        case 3:
            for (size_t i = 0; i < bpp && i < rowBytes; ++i)
                rowOut[i] = (uchar)(src[i] + (up[i] >> 1));
            for (size_t i = bpp; i < rowBytes; ++i)
                rowOut[i] = (uchar)(src[i] + ((rowOut[i - bpp] + up[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < bpp && i < rowBytes; ++i)
                rowOut[i] = (uchar)(src[i] + up[i]); // paeth(0, b, 0) = b
            if (bpp == 1) { // the common case, with a and c carried in registers
                int a = rowBytes ? rowOut[0] : 0, c = rowBytes ? up[0] : 0;
                for (size_t i = 1; i < rowBytes; ++i) {
                    const int b = up[i];
                    a = (uchar)(src[i] + paeth(a, b, c));
                    rowOut[i] = (uchar)a;
                    c = b;
                }
            } else {
                for (size_t i = bpp; i < rowBytes; ++i)
                    rowOut[i] = (uchar)(src[i] + paeth(rowOut[i - bpp], up[i], up[i - bpp]));
            }
            break;
        default:
            return false;
        }
        if (direct8)
            continue;
        if ((ctype == 0 || ctype == 3) && depth == 8) { // 8-bit palette (or grey through the identity table)
            uchar *d8 = (base + (size_t)y * w);
            for (unsigned x = 0; x < w; ++x)
                d8[x] = lut[cur[x]];
            prev.swap(cur);
            continue;
        }
        uchar *dst = (base + (size_t)y * w);
        for (unsigned x = 0; x < w; ++x) {
            if (ctype == 0 || ctype == 3) {
                unsigned v;
                if (depth == 8)
                    v = cur[x];
                else if (depth == 16)
                    v = cur[2 * x]; // high byte (16 -> 8 bit strip)
                else {
                    const unsigned per = 8 / depth, shift = (per - 1 - (x % per)) * depth;
                    v = (cur[x / per] >> shift) & ((1u << depth) - 1);
                    if (ctype == 0)
                        v = v * 255u / ((1u << depth) - 1); // expand_gray_1_2_4_to_8
                }
                if (ctype == 3) {
                    dst[x] = (int)v < npal ? png_rgb2grey(pal[v][0], pal[v][1], pal[v][2]) : 0;
                } else
                    dst[x] = (uchar)v;
            } else if (ctype == 4) {
                dst[x] = depth == 8 ? cur[2 * x] : cur[4 * x];
            } else { // RGB / RGBA
                const size_t step = (size_t)channels * (depth / 8), k = depth / 8;
                dst[x] = png_rgb2grey(cur[x * step], cur[x * step + k], cur[x * step + 2 * k]);
            }
        }
        prev.swap(cur);
    }
    return true;
}

static uchar *matSink(void *ctx, unsigned w, unsigned h)
{
    Mat *m = (Mat *)ctx;
    m->create((int)h, (int)w, CV_8U);
    return m->data;
}
static Mat decodePNG(const uchar *buf, size_t size)
{
    Mat out;
    PixelSink sink = {matSink, &out};
    if (!decodePNGTo(buf, size, sink))
        out.release();
    return out;
}

struct FixedDst {
    uchar *dst;
    unsigned w, h;
};
static uchar *fixedSink(void *ctx, unsigned w, unsigned h)
{
    FixedDst *f = (FixedDst *)ctx;
    return (w == f->w && h == f->h) ? f->dst : nullptr;
}
// decodes straight into `dst` (W * H bytes); false when the data is not a W x H image this decoder reads
bool imdecodeInto(const uchar *data, size_t size, uchar *dst, int W, int H)
{
    if (!data || !dst || W <= 0 || H <= 0)
        return false;
    if (size >= 8 && data[0] == 0x89 && data[1] == 'P') {
        FixedDst f = {dst, (unsigned)W, (unsigned)H};
        PixelSink sink = {fixedSink, &f};
        return decodePNGTo(data, size, sink);
    }
    const Mat m = imdecode(data, size, 0); // (BMP: small files, decoded the ordinary way)
    if (m.empty() || m.cols != W || m.rows != H)
        return false;
    std::memcpy(dst, m.data, (size_t)W * H);
    return true;
}

Mat imdecode(const uchar *data, size_t size, int)
{
    if (size >= 8 && data[0] == 0x89 && data[1] == 'P')
        return decodePNG(data, size);
    if (size >= 2 && data[0] == 'B' && data[1] == 'M')
        return decodeBMP(data, size);
    return Mat();
}

Mat imdecode(const std::vector<uchar> &buf, int flags) { return imdecode(buf.data(), buf.size(), flags); }

namespace {
void put32be(std::vector<uchar> &v, uint32_t x)
{
    v.push_back((uchar)(x >> 24));
    v.push_back((uchar)(x >> 16));
    v.push_back((uchar)(x >> 8));
    v.push_back((uchar)x);
}
void pngChunk(std::vector<uchar> &out, const char *type, const std::vector<uchar> &data)
{
    put32be(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32be(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(out.size() - at)));
}
} // namespace

bool imwrite(const std::string &path, const Mat &img)
{
    if (img.empty())
        return false;
    const int W = img.cols, H = img.rows;
    std::vector<uchar> out;
    const bool bmp = path.size() >= 4 && (path.compare(path.size() - 4, 4, ".bmp") == 0 || path.compare(path.size() - 4, 4, ".BMP") == 0);
    if (bmp) {
        const uint32_t stride = (uint32_t)(W + 3) / 4 * 4, off = 14 + 40 + 1024, size = off + stride * (uint32_t)H;
        auto le16 = [&](uint32_t x) { out.push_back((uchar)x); out.push_back((uchar)(x >> 8)); };
        auto le32 = [&](uint32_t x) { le16(x & 0xffff); le16(x >> 16); };
        out.push_back('B');
        out.push_back('M');
        le32(size); le32(0); le32(off);
        le32(40); le32((uint32_t)W); le32((uint32_t)H); le16(1); le16(8); le32(0); le32(stride * (uint32_t)H); le32(2835); le32(2835);
        le32(256); le32(0);
        for (int i = 0; i < 256; ++i) {
            out.push_back((uchar)i); out.push_back((uchar)i); out.push_back((uchar)i); out.push_back(0);
        }
        for (int y = H - 1; y >= 0; --y) {
            out.insert(out.end(), img.ptr<uchar>(y), img.ptr<uchar>(y) + W);
            out.insert(out.end(), stride - (uint32_t)W, 0);
        }
    } else {
        static const uchar sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        out.insert(out.end(), sig, sig + 8);
        std::vector<uchar> ihdr;
        put32be(ihdr, (uint32_t)W);
        put32be(ihdr, (uint32_t)H);
        const uchar tail[5] = {8, 0, 0, 0, 0}; // 8-bit grey, deflate, adaptive filtering, no interlace
        ihdr.insert(ihdr.end(), tail, tail + 5);
        pngChunk(out, "IHDR", ihdr);
        std::vector<uchar> raw((size_t)(W + 1) * H);
        for (int y = 0; y < H; ++y) {
            raw[(size_t)y * (W + 1)] = 0; // filter type None
            std::memcpy(&raw[(size_t)y * (W + 1) + 1], img.ptr<uchar>(y), (size_t)W);
        }
        uLongf clen = compressBound((uLong)raw.size());
        std::vector<uchar> z(clen);
        if (compress2(z.data(), &clen, raw.data(), (uLong)raw.size(), 1) != Z_OK)
            return false;
        z.resize(clen);
        pngChunk(out, "IDAT", z);
        pngChunk(out, "IEND", std::vector<uchar>());
    }
    FILE *f = fopen(path.c_str(), "wb");
    if (!f)
        return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

void rectangle(Mat &img, const Rect &r, const Scalar &color, int, int, int)
{
    if (img.empty() || r.width <= 0 || r.height <= 0)
        return;
    const uchar v = (uchar)std::min(255.0, std::max(0.0, color.val[0]));
    const int x0 = r.x, x1 = r.x + r.width - 1, y0 = r.y, y1 = r.y + r.height - 1;
    for (int x = std::max(x0, 0); x <= std::min(x1, img.cols - 1); ++x) {
        if (y0 >= 0 && y0 < img.rows)
            img.ptr<uchar>(y0)[x] = v;
        if (y1 >= 0 && y1 < img.rows)
            img.ptr<uchar>(y1)[x] = v;
    }
    for (int y = std::max(y0, 0); y <= std::min(y1, img.rows - 1); ++y) {
        if (x0 >= 0 && x0 < img.cols)
            img.ptr<uchar>(y)[x0] = v;
        if (x1 >= 0 && x1 < img.cols)
            img.ptr<uchar>(y)[x1] = v;
    }
}

Mat imread(const std::string &path, int flags)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return Mat();
    std::vector<uchar> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        long n = ftell(f);
        if (n > 0) {
            buf.resize((size_t)n);
            rewind(f);
            if (fread(buf.data(), 1, buf.size(), f) != buf.size())
                buf.clear();
        }
    }
    fclose(f);
    return buf.empty() ? Mat() : imdecode(buf, flags);
}

} // namespace cv
#endif
