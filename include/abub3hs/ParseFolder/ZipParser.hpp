// ZipParser.hpp -- frames of a run stored in one zip archive (interface of the reference's
// ParseFolder/ZipParser.hpp).  The reference reads the archive through minizip-ng (an empty submodule in the
// reference checkout); here a small central-directory reader (zip64 aware, stored + deflate via zlib) does it.
#ifndef ABUB3HS_ZIPPARSER_HPP
#define ABUB3HS_ZIPPARSER_HPP

#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>

#include "Parser.hpp"

class ZipParser : public Parser {
public:
    struct Entry {
        std::string name;
        uint64_t localHeaderOffset, compressedSize, uncompressedSize;
        int method; // 0 stored, 8 deflate
    };
    struct Index {
        std::vector<Entry> entries;                                          // archive order
        std::vector<std::string> FileContents;                               // all entry names
        std::map<std::string, std::map<std::string, int>> ImageLocs;         // event -> frame name -> entry
        std::string runID;
        int runFileLoc = -1;
        bool built = false;
    };

    // throws int(-10) if the archive cannot be opened, like the reference (ZipParser.cpp:66-70)
    ZipParser(std::string RunFolder, std::string ImageFolder, std::string ImageFormat);
    ~ZipParser() override;

    ZipParser *clone() override; // own file handle, shared (immutable once built) index

    int GetImage(std::string EventID, std::string FrameName, cv::Mat &Image) override;
    int GetImageInto(std::string EventID, std::string FrameName, unsigned char *dst, int W, int H) override;
    long long GetImageFileSize(std::string EventID, std::string FrameName) override;
    long long ReadImageFile(std::string EventID, std::string FrameName, unsigned char *dst, size_t cap) override;
    void GetEventDirLists(std::vector<std::string> &EventList) override;
    void GetFileLists(const char *EventFolder, std::vector<std::string> &FileList, const char *camera_out_name) override;
    void ParseAndSortFramesInFolder(std::string EventID, int camera, std::vector<std::string> &Contents) override;
    void GetRunFileInfo(std::vector<std::string> &EventListFromFile) override;

private:
    void BuildFileList();
    bool readEntry(int entry, std::vector<unsigned char> &out);
    FILE *fp = nullptr;
    std::shared_ptr<Index> index;
};

#endif
