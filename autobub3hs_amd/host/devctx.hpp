// devctx.hpp -- glue between the analysis classes and the C-ABI (include/abub_hip.h layer B):
// a per-host-thread GPU context and the HBM-resident image of one (event, camera).
#ifndef ABUB3HS_DEVCTX_HPP
#define ABUB3HS_DEVCTX_HPP

#include <cstdint>
#include <stdexcept>
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
    unsigned long long residentEvent = 0; // serial of the EventOnDevice currently in the frame slab (0: none)
    void ensureModel(const Trainer &t);
    ~DeviceContext();
};

// What the analysis classes need from the GPU for one (event, camera).  Two providers exist:
// EventOnDevice (below: decodes through the Parser, uploads into the per-thread context, one call at
// a time -- the drop-in path) and the batched provider of RunPipeline (pipeline.cpp: every stack of a
// run is already resident in HBM and all launches are shared across events).
class EventData {
public:
    virtual ~EventData() {}
    int F = 0, W = 0, H = 0;
    virtual bool frameOk(int i) const = 0;
    // host copy of frame i, or an empty Mat when the frames only exist in HBM
    virtual cv::Mat hostFrame(int i) const = 0;
    // histogram of D(frame[i]; frame[max(i-refOffset,0)])
    virtual const uint32_t *diffHist(int i, int refOffset) = 0;
    // D(frame[i]; frame[ref]) becomes the current image; returns its histogram
    virtual const uint32_t *diffFrame(int i, int ref, cv::Mat *out = nullptr) = 0;
    virtual const uint32_t *diffFrameROI(int i, int ref, cv::Rect roi, cv::Mat *out = nullptr) = 0;
    // post-trigger image of frame i becomes the current image; returns its histogram
    virtual const uint32_t *postTrig(int i, cv::Mat *out = nullptr) = 0;
    // foreground (v > thr) raster indices of the current image
    virtual void foreground(int thr, std::vector<uint32_t> &idx) = 0;
    // bellows veto: exact correlation terms of frame i against a template ((H-th+1) x (W-tw+1) placements)
    virtual void matchTerms(int i, const cv::Mat &templ, std::vector<unsigned long long> &num,
                            std::vector<unsigned long long> &wsum2) = 0;
    // current image = saturate(current image - sub); returns the new histogram
    virtual const uint32_t *subtractFromCurrent(const cv::Mat &sub) = 0;
};

// thrown by a provider that cannot serve a request (the batched pipeline re-runs such a stack one at a time)
struct NeedsDropInPath : public std::runtime_error {
    explicit NeedsDropInPath(const char *what) : std::runtime_error(what) {}
};

// Decoded frames of one (event, camera), pushed through the per-thread context on demand.
class EventOnDevice : public EventData {
public:
    EventOnDevice(Parser *parser, const std::string &eventID, const std::vector<std::string> &frameNames,
                  const Trainer *model);
    ~EventOnDevice() override;
    std::vector<cv::Mat> frames; // empty Mat == Parser::GetImage returned -1
    bool frameOk(int i) const override { return i >= 0 && i < F && !frames[i].empty(); }
    cv::Mat hostFrame(int i) const override { return frames[i]; }

    // all frames are evaluated in one batched launch the first time any histogram is asked for
    const uint32_t *diffHist(int i, int refOffset) override;
    const uint32_t *diffFrame(int i, int ref, cv::Mat *out = nullptr) override;
    const uint32_t *diffFrameROI(int i, int ref, cv::Rect roi, cv::Mat *out = nullptr) override;
    const uint32_t *postTrig(int i, cv::Mat *out = nullptr) override;
    void foreground(int thr, std::vector<uint32_t> &idx) override;
    void matchTerms(int i, const cv::Mat &templ, std::vector<unsigned long long> &num,
                    std::vector<unsigned long long> &wsum2) override;
    const uint32_t *subtractFromCurrent(const cv::Mat &sub) override;

private:
    DeviceContext &resident();
    const Trainer *model_;
    unsigned long long serial_; // process-wide, never reused (what DeviceContext::residentEvent remembers)
    std::vector<uint32_t> hists_[3]; // per refOffset (1,2): [F][256], empty until computed
    uint32_t lastHist_[256];
};

} // namespace abub
#endif
