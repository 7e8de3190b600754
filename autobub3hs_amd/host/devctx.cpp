#include "devctx.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "AlgorithmTraining/Trainer.hpp"
#include "ParseFolder/Parser.hpp"

namespace abub {

void check(int rc, const char *what)
{
    if (rc != ABUB_OK && rc != ABUB_E_OVERFLOW)
        throw std::runtime_error(std::string(what) + ": " + abub_last_error());
}

static int pickDevice()
{
    const char *e = getenv("ABUB_DEVICE");
    if (!e)
        e = getenv("LOCAL_RANK");
    int d = e ? atoi(e) : 0;
    int n = abub_device_count();
    if (n <= 0)
        throw std::runtime_error("abub: no HIP device available (the hot path has no CPU fallback)");
    return d % n;
}

// One context per host thread, owned by a thread_local holder: a worker thread that ends (the CLI's per-camera
// training threads, its event workers) gives its HBM slab and pinned staging buffers back.
struct ContextHolder {
    DeviceContext *p = nullptr;
    ~ContextHolder() { delete p; }
};
static thread_local ContextHolder t_holder;
#define t_ctx (t_holder.p)

// Resident stacks are identified by a process-wide serial number, not by the address of their EventOnDevice: an
// event made resident on thread A and destroyed on thread B cannot leave A's context believing that a later object
// at the same address is already uploaded.
static std::atomic<unsigned long long> g_eventSerial{1};

DeviceContext::~DeviceContext()
{
    if (ctx)
        abub_ctx_destroy(ctx);
}

void DeviceContext::releaseThread()
{
    delete t_ctx;
    t_ctx = nullptr;
}

DeviceContext &DeviceContext::forThread(int W, int H, int minFrames)
{
    if (t_ctx && (t_ctx->W != W || t_ctx->H != H || t_ctx->maxF < minFrames)) {
        delete t_ctx;
        t_ctx = nullptr;
    }
    if (!t_ctx) {
        DeviceContext *c = new DeviceContext();
        int cap = minFrames < 64 ? 64 : minFrames;
        int rc = abub_ctx_create(&c->ctx, pickDevice(), W, H, cap);
        if (rc != ABUB_OK) {
            delete c;
            throw std::runtime_error(std::string("abub_ctx_create: ") + abub_last_error());
        }
        c->W = W;
        c->H = H;
        c->maxF = cap;
        t_ctx = c;
    }
    return *t_ctx;
}

void DeviceContext::ensureModel(const Trainer &t)
{
    if (t.ModelId != 0 && t.ModelId == residentModel)
        return;
    if (t.TrainedAvgImage.empty() || t.TrainedSigmaImage.empty())
        throw std::runtime_error("abub: analyzer used with an untrained Trainer");
    check(abub_ctx_set_model(ctx, t.TrainedAvgImage.data, t.TrainedSigmaImage.data), "abub_ctx_set_model");
    residentModel = t.ModelId;
}

EventOnDevice::EventOnDevice(Parser *parser, const std::string &eventID,
                             const std::vector<std::string> &frameNames, const Trainer *model)
    : model_(model), serial_(g_eventSerial.fetch_add(1))
{
    F = (int)frameNames.size();
    frames.resize(F);
    for (int i = 0; i < F; ++i) {
        cv::Mat m;
        int err = parser->GetImage(eventID, frameNames[i], m);
        if (err == -1 || m.empty())
            continue;
        if (W == 0) {
            W = m.cols;
            H = m.rows;
        }
        if (m.cols != W || m.rows != H)
            continue; // treated as undecodable
        frames[i] = m;
    }
    std::memset(lastHist_, 0, sizeof lastHist_);
}

EventOnDevice::~EventOnDevice()
{
    // the context must forget a stack whose owner goes away
    if (t_ctx && t_ctx->residentEvent == serial_)
        t_ctx->residentEvent = 0; // (a context of another thread may keep the stale serial: it can never match again)
}

DeviceContext &EventOnDevice::resident()
{
    if (W == 0)
        throw std::runtime_error("abub: event has no decodable frame");
    DeviceContext &dc = DeviceContext::forThread(W, H, F);
    if (dc.residentEvent != serial_) {
        // undecodable frames travel as zeros; the state machines never read their results
        static thread_local std::vector<uint8_t> zeros;
        if (zeros.size() < (size_t)W * H)
            zeros.assign((size_t)W * H, 0);
        std::vector<const uint8_t *> ptrs(F);
        for (int i = 0; i < F; ++i)
            ptrs[i] = frames[i].empty() ? zeros.data() : frames[i].data;
        check(abub_ctx_upload_stack(dc.ctx, ptrs.data(), F), "abub_ctx_upload_stack");
        dc.residentEvent = serial_;
    }
    dc.ensureModel(*model_);
    return dc;
}

const uint32_t *EventOnDevice::diffHist(int i, int refOffset)
{
    if (refOffset < 1 || refOffset > 2 || i < 1 || i >= F)
        throw std::runtime_error("abub: diffHist index out of range");
    std::vector<uint32_t> &h = hists_[refOffset];
    if (h.empty()) {
        DeviceContext &dc = resident();
        h.resize((size_t)F * 256);
        check(abub_ctx_diff_hist_batch(dc.ctx, refOffset, 1, F - 1, h.data() + 256), "abub_ctx_diff_hist_batch");
    }
    return h.data() + (size_t)i * 256;
}

const uint32_t *EventOnDevice::diffFrame(int i, int ref, cv::Mat *out)
{
    DeviceContext &dc = resident();
    if (out)
        out->create(H, W, CV_8U);
    check(abub_ctx_diff_frame(dc.ctx, i, ref, out ? out->data : nullptr, lastHist_), "abub_ctx_diff_frame");
    return lastHist_;
}

const uint32_t *EventOnDevice::diffFrameROI(int i, int ref, cv::Rect roi, cv::Mat *out)
{
    DeviceContext &dc = resident();
    if (out)
        out->create(H, W, CV_8U);
    check(abub_ctx_diff_frame_roi(dc.ctx, i, ref, roi.x, roi.y, roi.width, roi.height,
                                  out ? out->data : nullptr, lastHist_),
          "abub_ctx_diff_frame_roi");
    return lastHist_;
}

const uint32_t *EventOnDevice::postTrig(int i, cv::Mat *out)
{
    DeviceContext &dc = resident();
    if (out)
        out->create(H, W, CV_8U);
    check(abub_ctx_posttrig(dc.ctx, i, out ? out->data : nullptr, lastHist_), "abub_ctx_posttrig");
    return lastHist_;
}

void EventOnDevice::matchTerms(int i, const cv::Mat &templ, std::vector<unsigned long long> &num,
                               std::vector<unsigned long long> &wsum2)
{
    DeviceContext &dc = resident();
    const size_t n = (size_t)(W - templ.cols + 1) * (H - templ.rows + 1);
    num.resize(n);
    wsum2.resize(n);
    check(abub_ctx_match_template(dc.ctx, i, templ.data, templ.cols, templ.rows, num.data(), wsum2.data()),
          "abub_ctx_match_template");
}

const uint32_t *EventOnDevice::subtractFromCurrent(const cv::Mat &sub)
{
    DeviceContext &dc = DeviceContext::forThread(W, H, F);
    check(abub_ctx_subtract_image(dc.ctx, sub.data, lastHist_), "abub_ctx_subtract_image");
    return lastHist_;
}

void EventOnDevice::foreground(int thr, std::vector<uint32_t> &idx)
{
    DeviceContext &dc = DeviceContext::forThread(W, H, F);
    const int cap = 1 << 16;
    idx.resize(cap);
    int n = 0;
    int rc = abub_ctx_foreground(dc.ctx, thr, idx.data(), cap, &n);
    if (rc == ABUB_E_OVERFLOW) {
        // dense foreground (rare: a huge blob or a broken model): take the whole image once and
        // build the list from it -- the pixel arithmetic still happened on the GPU
        std::vector<uint8_t> img((size_t)W * H);
        check(abub_ctx_fetch_image(dc.ctx, img.data()), "abub_ctx_fetch_image");
        idx.clear();
        for (size_t p = 0; p < img.size(); ++p)
            if ((int)img[p] > thr)
                idx.push_back((uint32_t)p);
        return;
    }
    check(rc, "abub_ctx_foreground");
    idx.resize(n);
}

} // namespace abub
