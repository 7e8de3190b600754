// Parser.hpp -- frame-source interface of the analysis classes (mirrors the reference's abstract
// ParseFolder/Parser.hpp:10-34 so that Trainer / AnalyzerUnit keep their constructor signatures),
// plus MemParser, an in-memory implementation used by the tests and the synthetic benchmarks.
// RawParser / ZipParser (directory and zip ingestion, PNG/BMP decode) are the "next" row of the
// scope table.
#ifndef ABUB3HS_PARSER_HPP
#define ABUB3HS_PARSER_HPP

#include <map>
#include <string>
#include <vector>

#include "../cvlite.hpp"

class Parser {
protected:
    std::string RunFolder;
    std::string ImageFolder;
    std::string ImageFormat;
    int StatusCode;

public:
    Parser(std::string runFolder, std::string imageFolder, std::string imageFormat)
        : RunFolder(runFolder), ImageFolder(imageFolder), ImageFormat(imageFormat), StatusCode(0) {}
    virtual ~Parser() = 0;

    virtual Parser *clone() = 0;

    // 1 = ok, -1 = missing / undecodable (the value the callers test, AnalyzerUnit.cpp:208, Trainer.cpp:262)
    virtual int GetImage(std::string EventID, std::string FrameName, cv::Mat &out) = 0;
    // not in the reference: the same frame as 8-bit grey pixels written straight into `dst` (W * H bytes) -- what the batched
    // ingestion path needs (no cv::Mat, no copy).  1 = ok; anything else (missing, undecodable, another size) = not written.
    // The base implementation goes through GetImage(); RawParser / ZipParser decode in place with thread-local scratch.
    virtual int GetImageInto(std::string EventID, std::string FrameName, unsigned char *dst, int W, int H);
    // not in the reference either: the frame's FILE as it is stored (still a PNG / BMP), for the batched path's decoder on
    // the GPU (abub_png_decode_dev).  GetImageFileSize: its size in bytes, or -1 where the parser cannot hand files out (the
    // caller then uses GetImageInto); ReadImageFile: the bytes into dst (room for cap), returns their count or -1.
    virtual long long GetImageFileSize(std::string EventID, std::string FrameName);
    virtual long long ReadImageFile(std::string EventID, std::string FrameName, unsigned char *dst, size_t cap);
    virtual void GetEventDirLists(std::vector<std::string> &EventList) = 0;
    virtual void GetFileLists(const char *EventFolder, std::vector<std::string> &FileList, const char *camera_out_name) = 0;
    // frame names of camera `camera` in event `EventID`, sorted lexicographically (RawParser.cpp:155)
    virtual void ParseAndSortFramesInFolder(std::string EventID, int camera, std::vector<std::string> &Contents) = 0;

    virtual void GetRunFileInfo(std::vector<std::string> &info) = 0;
    void VerifyEventList(std::vector<std::string> &EventList);
};

// Events held in memory: event id -> camera -> ordered (frame name, image).  Shares the image buffers
// between clones (cv::Mat is reference counted), so cloning per analyzer is cheap.
class MemParser : public Parser {
public:
    struct Frame {
        std::string name;
        cv::Mat image; // empty == undecodable (GetImage returns -1)
    };
    typedef std::map<int, std::vector<Frame>> CameraMap;

    MemParser() : Parser("", "", "cam%d_image%u.png"), events_(new std::map<std::string, CameraMap>()) {}

    // frames are named cam<c>_image<k>.png with zero-free numbering starting at `firstIndex`;
    // the list is kept in lexicographic name order like the real parsers.
    void AddFrames(const std::string &EventID, int camera, const std::vector<cv::Mat> &frames, int firstIndex = 30);
    // the same with the names given (a batched driver registers the real frame names of a run; `images` may be
    // NULL when the pixels only live in HBM)
    void AddNamedFrames(const std::string &EventID, int camera, const std::vector<std::string> &names,
                        const std::vector<cv::Mat> *images = nullptr);

    Parser *clone() override { return new MemParser(*this); }
    int GetImage(std::string EventID, std::string FrameName, cv::Mat &out) override;
    void GetEventDirLists(std::vector<std::string> &EventList) override;
    void GetFileLists(const char *EventFolder, std::vector<std::string> &FileList, const char *camera_out_name) override;
    void ParseAndSortFramesInFolder(std::string EventID, int camera, std::vector<std::string> &Contents) override;
    void GetRunFileInfo(std::vector<std::string> &info) override;

private:
    std::shared_ptr<std::map<std::string, CameraMap>> events_;
};

#endif
