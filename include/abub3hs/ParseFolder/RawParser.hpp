// RawParser.hpp -- frames of a run stored as a directory tree <run>/<event>/<ImageFolder>/<frame file>
// (interface of the reference's ParseFolder/RawParser.hpp; no boost: dirent + <regex>).
#ifndef ABUB3HS_RAWPARSER_HPP
#define ABUB3HS_RAWPARSER_HPP

#include "Parser.hpp"

class RawParser : public Parser {
public:
    RawParser(std::string RunFolder, std::string ImageFolder, std::string ImageFormat);
    ~RawParser() override;

    RawParser *clone() override;

    int GetImage(std::string EventID, std::string FrameName, cv::Mat &Image) override;
    int GetImageInto(std::string EventID, std::string FrameName, unsigned char *dst, int W, int H) override;
    long long GetImageFileSize(std::string EventID, std::string FrameName) override;
    long long ReadImageFile(std::string EventID, std::string FrameName, unsigned char *dst, size_t cap) override;
    void GetEventDirLists(std::vector<std::string> &EventList) override;
    void GetFileLists(const char *EventFolder, std::vector<std::string> &FileList, const char *camera_out_name) override;
    void ParseAndSortFramesInFolder(std::string EventID, int camera, std::vector<std::string> &Contents) override;
    void GetRunFileInfo(std::vector<std::string> &EventListFromFile) override;
};

#endif
