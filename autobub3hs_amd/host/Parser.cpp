// Parser base + MemParser (in-memory frame source).  See include/abub3hs/ParseFolder/Parser.hpp.
#include "ParseFolder/Parser.hpp"

#include <cstring>

#include <algorithm>
#include <cstdio>

Parser::~Parser() {}

long long Parser::GetImageFileSize(std::string, std::string) { return -1; }
long long Parser::ReadImageFile(std::string, std::string, unsigned char *, size_t) { return -1; }

// keep only the events the run file knows about (reference ParseFolder/Parser.cpp:14-26)
void Parser::VerifyEventList(std::vector<std::string> &EventList)
{
    std::vector<std::string> known;
    GetRunFileInfo(known);
    EventList.erase(std::remove_if(EventList.begin(), EventList.end(),
                                   [&](const std::string &e) { return std::find(known.begin(), known.end(), e) == known.end(); }),
                    EventList.end());
}

void MemParser::AddFrames(const std::string &EventID, int camera, const std::vector<cv::Mat> &frames, int firstIndex)
{
    std::vector<Frame> &v = (*events_)[EventID][camera];
    v.clear();
    for (size_t k = 0; k < frames.size(); ++k) {
        char name[64];
        snprintf(name, sizeof name, "cam%d_image%u.png", camera, (unsigned)(firstIndex + k));
        v.push_back(Frame{name, frames[k]});
    }
    std::sort(v.begin(), v.end(), [](const Frame &a, const Frame &b) { return a.name < b.name; });
}

void MemParser::AddNamedFrames(const std::string &EventID, int camera, const std::vector<std::string> &names,
                               const std::vector<cv::Mat> *images)
{
    std::vector<Frame> &v = (*events_)[EventID][camera];
    v.clear();
    for (size_t k = 0; k < names.size(); ++k)
        v.push_back(Frame{names[k], images && k < images->size() ? (*images)[k] : cv::Mat()});
    std::sort(v.begin(), v.end(), [](const Frame &a, const Frame &b) { return a.name < b.name; });
}

int Parser::GetImageInto(std::string EventID, std::string FrameName, unsigned char *dst, int W, int H)
{
    cv::Mat img;
    const int rc = GetImage(EventID, FrameName, img);
    if (rc == -1 || img.empty() || img.cols != W || img.rows != H)
        return -1;
    std::memcpy(dst, img.data, (size_t)W * H);
    return 1;
}

int MemParser::GetImage(std::string EventID, std::string FrameName, cv::Mat &out)
{
    auto ev = events_->find(EventID);
    if (ev == events_->end())
        return -1;
    for (auto &cam : ev->second)
        for (Frame &f : cam.second)
            if (f.name == FrameName) {
                if (f.image.empty())
                    return -1;
                out = f.image; // shares the buffer, like cv::Mat assignment
                return 1;
            }
    return -1;
}

void MemParser::GetEventDirLists(std::vector<std::string> &EventList)
{
    for (auto &kv : *events_)
        EventList.push_back(kv.first);
}

void MemParser::GetFileLists(const char *EventFolder, std::vector<std::string> &FileList, const char *camera_out_name)
{
    auto ev = events_->find(EventFolder);
    if (ev == events_->end())
        return;
    for (auto &cam : ev->second)
        for (Frame &f : cam.second)
            if (f.name.find(camera_out_name) != std::string::npos)
                FileList.push_back(f.name);
}

void MemParser::ParseAndSortFramesInFolder(std::string EventID, int camera, std::vector<std::string> &Contents)
{
    auto ev = events_->find(EventID);
    if (ev == events_->end())
        return;
    auto cam = ev->second.find(camera);
    if (cam == ev->second.end())
        return;
    for (Frame &f : cam->second)
        Contents.push_back(f.name);
    std::sort(Contents.begin(), Contents.end()); // lexicographic, RawParser.cpp:155 / ZipParser.cpp:310
}

void MemParser::GetRunFileInfo(std::vector<std::string> &) {}
