// devctx.hpp -- glue between the analysis classes and the C-ABI (include/abub_hip.h layer B):
// a per-host-thread GPU context and the HBM-resident image of one (event, camera).
#ifndef ABUB3HS_DEVCTX_HPP
#define ABUB3HS_DEVCTX_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "abub_hip.h"
#include "cvlite.hpp"

class Parser;
class Trainer;

namespace abub {

// Throws std::runtime_error carrying abub_last_error(): AnyCamAnalysis maps exceptions to status -6
// (reference AutoBubStart3.cpp:114-117).
void check(int rc, const char *what);

// One context per host thread, recreated when the frame geometry changes or more frames are needed.
class DeviceContext {
public:
    static DeviceContext &forThread(int W, int H, int minFrames);
    static void releaseThread();
    abub_ctx *ctx = nullptr;
    int W = 0, H = 0, maxF = 0;
    unsigned long long residentModel = 0; // Trainer::ModelId currently in HBM
    const void *residentEvent = nullptr;  // EventOnDevice currently in the frame slab
    void ensureModel(const Trainer &t);
    ~DeviceContext();
};

// Decoded frames of one (event, camera) plus everything the trigger search needs from the GPU.
class EventOnDevice {
public:
    EventOnDevice(Parser *parser, const std::string &eventID, const std::vector<std::string> &frameNames,
                  const Trainer *model);
    int F = 0, W = 0, H = 0;
    std::vector<cv::Mat> frames; // empty Mat == Parser::GetImage returned -1
    bool frameOk(int i) const { return i >= 0 && i < F && !frames[i].empty(); }

    // histogram of D(frame[i]; frame[max(i-refOffset,0)]); all frames are evaluated in one batched
    // launch the first time any of them is asked for
    const uint32_t *diffHist(int i, int refOffset);
    // D(frame[i]; frame[ref]) becomes the context's current image; returns its histogram
    const uint32_t *diffFrame(int i, int ref, cv::Mat *out = nullptr);
    const uint32_t *diffFrameROI(int i, int ref, cv::Rect roi, cv::Mat *out = nullptr);
    // post-trigger image of frame i becomes the current image; returns its histogram
    const uint32_t *postTrig(int i, cv::Mat *out = nullptr);
    // foreground (v > thr) raster indices of the current image
    void foreground(int thr, std::vector<uint32_t> &idx);

private:
    DeviceContext &resident();
    const Trainer *model_;
    std::vector<uint32_t> hists_[3]; // per refOffset (1,2): [F][256], empty until computed
    uint32_t lastHist_[256];
};

} // namespace abub
#endif
