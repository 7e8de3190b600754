// RawParser.cpp -- directory-tree frame source.  Behaviour of the reference's ParseFolder/RawParser.cpp:
// GetImage 1 = ok / 0 = undecodable (never -1, :46), event list = sub-directories of the run folder
// (:90-135), frame list = files matching ^.*cam<c>.*(png|bmp) sorted lexicographically (:137-157).
#include "ParseFolder/RawParser.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <regex>

#include <dirent.h>
#include <sys/stat.h>

static std::string joinPath(const std::string &a, const std::string &b)
{
    if (a.empty())
        return b;
    if (b.empty())
        return a;
    if (a.back() == '/')
        return b.front() == '/' ? a + b.substr(1) : a + b;
    return b.front() == '/' ? a + b : a + "/" + b;
}

RawParser::RawParser(std::string RunFolder, std::string ImageFolder, std::string ImageFormat)
    : Parser(RunFolder, ImageFolder, ImageFormat)
{
}

RawParser::~RawParser() {}

RawParser *RawParser::clone() { return new RawParser(RunFolder, ImageFolder, ImageFormat); }

int RawParser::GetImage(std::string EventID, std::string FrameName, cv::Mat &Image)
{
    const std::string path = joinPath(joinPath(joinPath(RunFolder, EventID), ImageFolder), FrameName);
    Image = cv::imread(path, 0);
    return !Image.empty();
}

int RawParser::GetImageInto(std::string EventID, std::string FrameName, unsigned char *dst, int W, int H)
{
#ifdef ABUB_USE_OPENCV
    return Parser::GetImageInto(EventID, FrameName, dst, W, H);
#else
    const std::string path = joinPath(joinPath(joinPath(RunFolder, EventID), ImageFolder), FrameName);
    static thread_local std::vector<unsigned char> data; // keeps its capacity from frame to frame
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return -1;
    size_t n = 0;
    if (fseeko(f, 0, SEEK_END) == 0) {
        const off_t sz = ftello(f);
        if (sz > 0 && fseeko(f, 0, SEEK_SET) == 0) {
            data.resize((size_t)sz);
            n = fread(data.data(), 1, (size_t)sz, f);
        }
    }
    fclose(f);
    return n && cv::imdecodeInto(data.data(), n, dst, W, H) ? 1 : -1;
#endif
}

long long RawParser::GetImageFileSize(std::string EventID, std::string FrameName)
{
    const std::string path = joinPath(joinPath(joinPath(RunFolder, EventID), ImageFolder), FrameName);
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode))
        return -1;
    return (long long)st.st_size;
}

long long RawParser::ReadImageFile(std::string EventID, std::string FrameName, unsigned char *dst, size_t cap)
{
    const std::string path = joinPath(joinPath(joinPath(RunFolder, EventID), ImageFolder), FrameName);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return -1;
    const size_t n = fread(dst, 1, cap, f);
    const bool more = n == cap && fgetc(f) != EOF; // (the file grew since its size was asked: not what was planned for)
    fclose(f);
    return more ? -1 : (long long)n;
}

void RawParser::GetFileLists(const char *EventFolder, std::vector<std::string> &FileList, const char *camera_out_name)
{
    DIR *dir = opendir(EventFolder);
    if (!dir) {
        StatusCode = 1;
        return;
    }
    while (struct dirent *f = readdir(dir)) {
        if (f->d_name[0] == '.')
            continue;
        if (strstr(f->d_name, camera_out_name))
            FileList.push_back(f->d_name);
    }
    closedir(dir);
}

void RawParser::GetEventDirLists(std::vector<std::string> &EventList)
{
    DIR *dir = opendir(RunFolder.c_str());
    if (!dir) {
        StatusCode = 1;
        return;
    }
    while (struct dirent *f = readdir(dir)) {
        if (f->d_name[0] == '.')
            continue;
        struct stat sb;
        if (stat(joinPath(RunFolder, f->d_name).c_str(), &sb) == 0 && S_ISDIR(sb.st_mode))
            EventList.push_back(f->d_name);
    }
    closedir(dir);
}

void RawParser::ParseAndSortFramesInFolder(std::string EventID, int Camera, std::vector<std::string> &Contents)
{
    const std::string eventDir = joinPath(joinPath(RunFolder, EventID), ImageFolder);
    const std::regex re("^.*cam" + std::to_string(Camera) + ".*(png|bmp)");
    DIR *dir = opendir(eventDir.c_str());
    if (!dir)
        return;
    while (struct dirent *f = readdir(dir)) {
        if (!strcmp(f->d_name, ".") || !strcmp(f->d_name, ".."))
            continue;
        if (std::regex_match(joinPath(eventDir, f->d_name), re))
            Contents.push_back(f->d_name);
    }
    closedir(dir);
    std::sort(Contents.begin(), Contents.end());
}

// <run>/<runID>.txt: one line per event, the second column is the event number (:160-175)
void RawParser::GetRunFileInfo(std::vector<std::string> &EventListFromFile)
{
    std::string folder = RunFolder;
    while (folder.size() > 1 && folder.back() == '/')
        folder.pop_back();
    const size_t slash = folder.find_last_of('/');
    const std::string runID = slash == std::string::npos ? folder : folder.substr(slash + 1);
    std::ifstream ifs(joinPath(folder, runID + ".txt"));
    std::string skip;
    int eventNum;
    while (ifs >> skip >> eventNum >> skip >> skip >> skip >> skip >> skip >> skip >> skip >> skip >> skip)
        EventListFromFile.push_back(std::to_string(eventNum));
}
