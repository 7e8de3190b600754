// ZipParser.cpp -- zip-archive frame source.  Index semantics of the reference's ParseFolder/ZipParser.cpp:
// image entries match ^.*/(\d+)/.*/?(cam\d.*image\s*\d+.*(png|bmp)) -> [event][frame name] (:110-141),
// event list = entries ^.*/(\d+)/$ (:252-262), frame list per camera = names matching
// cam<c>.*image.*(png|bmp), lexicographic (std::map order, :292-300), GetImage -1 = missing / undecodable,
// 1 = ok (:194-238).  The container itself is read here: end-of-central-directory (+ zip64 locator),
// central directory walk, local header skip, stored or raw-deflate payload.
#include "ParseFolder/ZipParser.hpp"

#include <algorithm>
#include <cstring>
#include <iostream>
#include <regex>
#include <sstream>

#include <zlib.h>

namespace {
inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

bool preadAll(FILE *fp, uint64_t off, void *dst, size_t n)
{
    if (fseeko(fp, (off_t)off, SEEK_SET) != 0)
        return false;
    return fread(dst, 1, n, fp) == n;
}
} // namespace

ZipParser::ZipParser(std::string RunFolder_, std::string ImageFolder_, std::string ImageFormat_)
    : Parser(RunFolder_, ImageFolder_, ImageFormat_)
{
    // strip a trailing separator, add ".zip" when there is no extension (ZipParser.cpp:40-44)
    std::string p = RunFolder;
    while (p.size() > 1 && p.back() == '/')
        p.pop_back();
    const size_t slash = p.find_last_of('/');
    const size_t dot = p.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash))
        p += ".zip";
    RunFolder = p;
    fp = fopen(RunFolder.c_str(), "rb");
    if (!fp) {
        std::cerr << "Error initializing zip file; cannot continue" << std::endl;
        throw -10;
    }
    index = std::make_shared<Index>();
}

ZipParser::~ZipParser()
{
    if (fp)
        fclose(fp);
}

ZipParser *ZipParser::clone()
{
    ZipParser *o = new ZipParser(RunFolder, ImageFolder, ImageFormat);
    o->index = index; // shared, immutable once built
    return o;
}

void ZipParser::BuildFileList()
{
    if (index->built)
        return;
    Index &ix = *index;
    // ---- locate the end of central directory record -------------------------------------------
    if (fseeko(fp, 0, SEEK_END) != 0)
        throw -10;
    const uint64_t fsize = (uint64_t)ftello(fp);
    const uint64_t tail = std::min<uint64_t>(fsize, 65536 + 22);
    std::vector<unsigned char> buf(tail);
    if (!preadAll(fp, fsize - tail, buf.data(), tail))
        throw -10;
    int64_t eocd = -1;
    for (int64_t i = (int64_t)tail - 22; i >= 0; --i)
        if (rd32(&buf[i]) == 0x06054b50u) {
            eocd = i;
            break;
        }
    if (eocd < 0)
        throw -10;
    uint64_t nEntries = rd16(&buf[eocd + 10]), cdSize = rd32(&buf[eocd + 12]), cdOff = rd32(&buf[eocd + 16]);
    if (nEntries == 0xffff || cdSize == 0xffffffffu || cdOff == 0xffffffffu) {
        // zip64: locator sits right before the EOCD
        if (eocd >= 20 && rd32(&buf[eocd - 20]) == 0x07064b50u) {
            const uint64_t z64 = rd64(&buf[eocd - 20 + 8]);
            unsigned char r[56];
            if (!preadAll(fp, z64, r, 56) || rd32(r) != 0x06064b50u)
                throw -10;
            nEntries = rd64(r + 32);
            cdSize = rd64(r + 40);
            cdOff = rd64(r + 48);
        } else
            throw -10;
    }
    std::vector<unsigned char> cd(cdSize);
    if (!preadAll(fp, cdOff, cd.data(), cdSize))
        throw -10;
    // ---- walk the central directory -----------------------------------------------------------
    const std::regex imgRe("^.*/(\\d+)/.*/?(cam\\d.*image\\s*\\d+.*(png|bmp))");
    const std::regex runIdRe(".*(\\d{8}_\\d+).*");
    std::regex runFileRe;
    size_t o = 0;
    for (uint64_t k = 0; k < nEntries; ++k) {
        if (o + 46 > cd.size() || rd32(&cd[o]) != 0x02014b50u)
            throw -10;
        Entry e;
        e.method = rd16(&cd[o + 10]);
        e.compressedSize = rd32(&cd[o + 20]);
        e.uncompressedSize = rd32(&cd[o + 24]);
        const size_t nameLen = rd16(&cd[o + 28]), extraLen = rd16(&cd[o + 30]), commentLen = rd16(&cd[o + 32]);
        e.localHeaderOffset = rd32(&cd[o + 42]);
        if (o + 46 + nameLen + extraLen + commentLen > cd.size())
            throw -10;
        e.name.assign((const char *)&cd[o + 46], nameLen);
        // zip64 extended information (header id 0x0001): only the saturated fields are present, in order
        size_t x = o + 46 + nameLen;
        const size_t xEnd = x + extraLen;
        while (x + 4 <= xEnd) {
            const unsigned id = rd16(&cd[x]), sz = rd16(&cd[x + 2]);
            if (id == 0x0001) {
                size_t q = x + 4;
                if (e.uncompressedSize == 0xffffffffu && q + 8 <= xEnd) {
                    e.uncompressedSize = rd64(&cd[q]);
                    q += 8;
                }
                if (e.compressedSize == 0xffffffffu && q + 8 <= xEnd) {
                    e.compressedSize = rd64(&cd[q]);
                    q += 8;
                }
                if (e.localHeaderOffset == 0xffffffffu && q + 8 <= xEnd)
                    e.localHeaderOffset = rd64(&cd[q]);
            }
            x += 4 + sz;
        }
        o += 46 + nameLen + extraLen + commentLen;

        const int id = (int)ix.entries.size();
        ix.entries.push_back(e);
        ix.FileContents.push_back(e.name);
        std::smatch m;
        if (std::regex_match(e.name, m, imgRe))
            ix.ImageLocs[m[1].str()][m[2].str()] = id;
        if (ix.runID.empty() && std::regex_match(e.name, m, runIdRe)) {
            ix.runID = m[1].str();
            runFileRe = std::regex(".*" + ix.runID + ".txt");
        }
        if (!ix.runID.empty() && ix.runFileLoc < 0 && std::regex_match(e.name, runFileRe))
            ix.runFileLoc = id;
    }
    ix.built = true;
}

bool ZipParser::readEntry(int entry, std::vector<unsigned char> &out)
{
    const Entry &e = index->entries[entry];
    unsigned char lh[30];
    if (!preadAll(fp, e.localHeaderOffset, lh, 30) || rd32(lh) != 0x04034b50u)
        return false;
    const uint64_t dataOff = e.localHeaderOffset + 30 + rd16(lh + 26) + rd16(lh + 28);
    if (e.method == 0) { // stored: straight into `out` (callers on the batched path pass a thread-local vector that keeps its capacity)
        out.resize(e.compressedSize);
        return !e.compressedSize || preadAll(fp, dataOff, out.data(), out.size());
    }
    static thread_local std::vector<unsigned char> comp;
    comp.resize(e.compressedSize);
    if (e.compressedSize && !preadAll(fp, dataOff, comp.data(), comp.size()))
        return false;
    if (e.method != 8)
        return false;
    out.resize(e.uncompressedSize);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -MAX_WBITS) != Z_OK)
        return false;
    zs.next_in = comp.data();
    zs.avail_in = (uInt)comp.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.total_out == out.size();
}

int ZipParser::GetImage(std::string EventID, std::string FrameName, cv::Mat &Image)
{
    BuildFileList();
    auto ev = index->ImageLocs.find(EventID);
    if (ev == index->ImageLocs.end())
        return -1;
    auto fr = ev->second.find(FrameName);
    if (fr == ev->second.end())
        return -1;
    std::vector<unsigned char> data;
    if (!readEntry(fr->second, data))
        return 0; // container-level error (the reference returns !err of the minizip calls)
    Image = cv::imdecode(data, 0);
    if (Image.empty()) {
        std::cerr << "Failed to decode image " << FrameName << std::endl;
        return -1;
    }
    return 1;
}

int ZipParser::GetImageInto(std::string EventID, std::string FrameName, unsigned char *dst, int W, int H)
{
#ifdef ABUB_USE_OPENCV
    return Parser::GetImageInto(EventID, FrameName, dst, W, H);
#else
    BuildFileList();
    auto ev = index->ImageLocs.find(EventID);
    if (ev == index->ImageLocs.end())
        return -1;
    auto fr = ev->second.find(FrameName);
    if (fr == ev->second.end())
        return -1;
    static thread_local std::vector<unsigned char> data; // keeps its capacity from frame to frame
    if (!readEntry(fr->second, data))
        return 0;
    return cv::imdecodeInto(data.data(), data.size(), dst, W, H) ? 1 : -1;
#endif
}

long long ZipParser::GetImageFileSize(std::string EventID, std::string FrameName)
{
    BuildFileList();
    auto ev = index->ImageLocs.find(EventID);
    if (ev == index->ImageLocs.end())
        return -1;
    auto fr = ev->second.find(FrameName);
    if (fr == ev->second.end())
        return -1;
    const Entry &e = index->entries[fr->second];
    return (long long)(e.method == 0 ? e.compressedSize : e.uncompressedSize);
}

long long ZipParser::ReadImageFile(std::string EventID, std::string FrameName, unsigned char *dst, size_t cap)
{
    BuildFileList();
    auto ev = index->ImageLocs.find(EventID);
    if (ev == index->ImageLocs.end())
        return -1;
    auto fr = ev->second.find(FrameName);
    if (fr == ev->second.end())
        return -1;
    const Entry &e = index->entries[fr->second];
    if (e.method == 0) { // stored (what a run archive of PNGs is): straight from the archive into dst
        if (e.compressedSize > cap)
            return -1;
        unsigned char lh[30];
        if (!preadAll(fp, e.localHeaderOffset, lh, 30) || rd32(lh) != 0x04034b50u)
            return -1;
        const uint64_t dataOff = e.localHeaderOffset + 30 + rd16(lh + 26) + rd16(lh + 28);
        if (e.compressedSize && !preadAll(fp, dataOff, dst, e.compressedSize))
            return -1;
        return (long long)e.compressedSize;
    }
    static thread_local std::vector<unsigned char> data; // a deflated entry: the zip layer is inflated on the host
    if (!readEntry(fr->second, data) || data.size() > cap)
        return -1;
    memcpy(dst, data.data(), data.size());
    return (long long)data.size();
}

void ZipParser::GetFileLists(const char *, std::vector<std::string> &, const char *) {} // empty upstream too (:241-244)

void ZipParser::GetEventDirLists(std::vector<std::string> &EventList)
{
    BuildFileList();
    const std::regex re("^.*/(\\d+)/$");
    std::smatch m;
    for (const std::string &name : index->FileContents)
        if (std::regex_match(name, m, re))
            EventList.push_back(m[1].str());
}

void ZipParser::ParseAndSortFramesInFolder(std::string EventID, int camera, std::vector<std::string> &Contents)
{
    BuildFileList();
    const std::regex re("cam" + std::to_string(camera) + ".*image.*(png|bmp)");
    auto ev = index->ImageLocs.find(EventID);
    if (ev == index->ImageLocs.end())
        return;
    for (auto &kv : ev->second) // std::map: already in lexicographic order
        if (std::regex_match(kv.first, re))
            Contents.push_back(kv.first);
    std::sort(Contents.begin(), Contents.end());
}

void ZipParser::GetRunFileInfo(std::vector<std::string> &EventListFromFile)
{
    BuildFileList();
    if (index->runFileLoc < 0)
        return;
    std::vector<unsigned char> data;
    if (!readEntry(index->runFileLoc, data))
        return;
    std::istringstream ifs(std::string(data.begin(), data.end()));
    std::string skip;
    int eventNum;
    while (ifs >> skip >> eventNum >> skip >> skip >> skip >> skip >> skip >> skip >> skip >> skip >> skip)
        EventListFromFile.push_back(std::to_string(eventNum));
}
