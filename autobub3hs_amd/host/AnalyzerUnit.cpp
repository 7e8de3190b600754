// AnalyzerUnit.cpp -- trigger-frame search.  The state machine follows the reference's
// AnalyzerUnit::FindTriggerFrame (AnalyzerUnit.cpp:119-324) decision for decision, but instead of
// decoding and differencing frames one by one it reads 256-bin histograms that ONE batched GPU pass
// over the whole frame stack produced (the look-ahead diffs are the main-loop diffs of
// frames i+1, i+2, so the same histograms serve both).
#include "AnalyzerUnit.hpp"

#include <cassert>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <iostream>
#include <stdexcept>

#include "devctx.hpp"
#include "hostlogic.hpp"

AnalyzerUnit::AnalyzerUnit(std::string EventID, std::string ImageDir, int CameraNumber, Trainer **TrainedData,
                           std::string MaskDir, Parser *Parser)
{
    this->ImageDir = ImageDir;
    this->MaskDir = MaskDir;
    this->CameraNumber = CameraNumber;
    this->EventID = EventID;
    this->TrainedData = new Trainer(**TrainedData); // the analyzer owns a copy (AnalyzerUnit.cpp:27)
    this->MatTrigFrame = 0;
    this->loc_thres = 3;
    this->pix_counts.resize(256);
    this->FileParser = Parser; // ownership taken (AnalyzerUnit.cpp:31,44)
    this->FileParser->ParseAndSortFramesInFolder(this->EventID, this->CameraNumber, this->CameraFrames);
}

AnalyzerUnit::~AnalyzerUnit(void)
{
    for (bubble *b : BubbleList)
        delete b;
    if (dev && ownsDev)
        delete dev;
    delete TrainedData;
    delete FileParser;
}

abub::EventData &AnalyzerUnit::device()
{
    if (!dev) {
        dev = new abub::EventOnDevice(FileParser, EventID, CameraFrames, TrainedData);
        ownsDev = true;
    }
    return *dev;
}

void AnalyzerUnit::AttachEventData(abub::EventData *data)
{
    if (dev && ownsDev)
        delete dev;
    dev = data;
    ownsDev = false;
}

// The frame list comes from the Parser in the constructor; this legacy entry point re-reads it.
void AnalyzerUnit::ParseAndSortFramesInFolder(void)
{
    CameraFrames.clear();
    FileParser->ParseAndSortFramesInFolder(EventID, CameraNumber, CameraFrames);
}

void AnalyzerUnit::ProduceOutput(void) {} // declared but never defined upstream (AnalyzerUnit.hpp:55)

void AnalyzerUnit::gammaCorrection(const cv::Mat &, cv::Mat &, const float) { return; } // disabled upstream (:329)

// Public ProcessFrame on caller-provided images: they are staged as a two-frame stack.
void AnalyzerUnit::ProcessFrame(cv::Mat &workingFrame, cv::Mat &prevFrame, cv::Mat &diff_frame, int blur_diam, int img_num)
{
    cv::Rect ROI(0, 0, workingFrame.cols, workingFrame.rows);
    ProcessFrame(workingFrame, prevFrame, diff_frame, blur_diam, ROI, img_num);
}

void AnalyzerUnit::ProcessFrame(cv::Mat &workingFrame, cv::Mat &prevFrame, cv::Mat &diff_frame, int blur_diam, cv::Rect ROI, int)
{
    if (blur_diam != 5)
        throw std::runtime_error("AnalyzerUnit::ProcessFrame: only the 5x5 kernel of the reference is implemented");
    if (workingFrame.empty() || prevFrame.empty() || workingFrame.rows != prevFrame.rows || workingFrame.cols != prevFrame.cols)
        throw std::runtime_error("AnalyzerUnit::ProcessFrame: empty or mismatching frames");
    abub::DeviceContext &dc = abub::DeviceContext::forThread(workingFrame.cols, workingFrame.rows, 2);
    const uint8_t *fr[2] = {workingFrame.data, prevFrame.data};
    abub::check(abub_ctx_upload_stack(dc.ctx, fr, 2), "abub_ctx_upload_stack");
    dc.residentEvent = 0;
    dc.ensureModel(*TrainedData);
    diff_frame.create(workingFrame.rows, workingFrame.cols, CV_8U);
    const bool full = ROI.x == 0 && ROI.y == 0 && ROI.width == workingFrame.cols && ROI.height == workingFrame.rows;
    if (full)
        abub::check(abub_ctx_diff_frame(dc.ctx, 0, 1, diff_frame.data, nullptr), "abub_ctx_diff_frame");
    else
        abub::check(abub_ctx_diff_frame_roi(dc.ctx, 0, 1, ROI.x, ROI.y, ROI.width, ROI.height, diff_frame.data, nullptr),
                    "abub_ctx_diff_frame_roi");
}

// 128-bin entropy and its z-score: compiled but unused upstream (AnalyzerUnit.cpp:386-433).
float AnalyzerUnit::calculateEntropyFrame(cv::Mat &img, bool)
{
    abub::DeviceContext &dc = abub::DeviceContext::forThread(img.cols, img.rows, 2);
    std::vector<uint8_t> zeros(img.total(), 0);
    uint32_t h[256];
    abub::check(abub_ctx_pair_hist(dc.ctx, zeros.data(), img.data, h), "abub_ctx_pair_hist");
    dc.residentEvent = 0;
    return abub::entropyFromHist(h, 128, img.total());
}

double AnalyzerUnit::calculateEntropySignificance(cv::Mat &img, bool store, bool debug)
{
    double e = calculateEntropyFrame(img, debug);
    if (store)
        entropies.push_back(e);
    double mean = CalcMean(entropies);
    double sigma = CalcStdDev(entropies, mean);
    return (e - mean) / sigma;
}

double AnalyzerUnit::calculateSignificanceFrame(cv::Mat &img, bool store, bool)
{
    abub::DeviceContext &dc = abub::DeviceContext::forThread(img.cols, img.rows, 2);
    std::vector<uint8_t> zeros(img.total(), 0);
    uint32_t h[256];
    abub::check(abub_ctx_pair_hist(dc.ctx, zeros.data(), img.data, h), "abub_ctx_pair_hist");
    dc.residentEvent = 0;
    return abub::significanceFromHist(pix_counts, h, img.total(), store, TrainedData->TrainingSetSize, loc_thres_max, loc_thres);
}

// ---- debug image write-out of the trigger search (AnalyzerUnit.cpp:233-238, 353-365): only with the analyzer debug
// digit (nonStopMode == false), into $HOME/test/abub_debug/ which must exist.  The pos / neg planes and their
// filtered versions are intermediate images of the fused kernel that never exist on the GPU; for the dump they are
// formed on the host from the decoded frames (reference formulas; results of the analysis never come from here).
static void debugPlanes(const cv::Mat &cur, const cv::Mat &ref, const cv::Mat &sigma, cv::Mat out[4])
{
    const int W = cur.cols, H = cur.rows;
    for (int k = 0; k < 4; ++k)
        out[k].create(H, W, CV_8U);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const int c = cur.data[i], r = ref.data[i], s6 = 6 * (int)sigma.data[i];
        out[0].data[i] = (uchar)std::max(0, c - r - s6); // pos_diff (:351)
        out[1].data[i] = (uchar)std::max(0, r - c - s6); // neg_diff (:352)
    }
    auto refl = [](int p, int n) { return n == 1 ? 0 : (p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p)); };
    static const int wt[5] = {1, 4, 6, 4, 1};
    std::vector<int> tmp((size_t)W * H);
    for (int k = 0; k < 2; ++k) { // cv::GaussianBlur 5x5, sigma 0, BORDER_REFLECT_101: (S + 128) >> 8 (:359-360)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int acc = 0;
                for (int j = -2; j <= 2; ++j)
                    acc += wt[j + 2] * out[k].data[(size_t)y * W + refl(refl(x + j, W), W)];
                tmp[(size_t)y * W + x] = acc;
            }
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int acc = 0;
                for (int j = -2; j <= 2; ++j)
                    acc += wt[j + 2] * tmp[(size_t)refl(refl(y + j, H), H) * W + x];
                out[2 + k].data[(size_t)y * W + x] = (uchar)((acc + 128) >> 8);
            }
    }
}

void AnalyzerUnit::FindTriggerFrame(bool nonStopMode, int startframe)
{
    const int n = (int)CameraFrames.size();
    if (n < 5) { // malformed sequence
        okToProceed = false;
        TriggerFrameIdentificationStatus = -9;
        return;
    }
    float entropyThreshold = 3.5f;
    if (startframe < 1)
        startframe = 1;
    if (startframe == 1) {
        pix_counts.clear();
        pix_counts.resize(256);
        entropies.clear();
    }
    TriggerFrameIdentificationStatus = -3;

    // with a small training set the search compares against the previous frame only and asks for more
    bool twoFrameOffset = true;
    if (TrainedData->TrainingSetSize < 6) {
        twoFrameOffset = false;
        entropyThreshold *= 5 / 3.5;
    }
    const int refOffset = twoFrameOffset ? 2 : 1;
    if (!nonStopMode)
        std::cout << "twoFrameOffset: " << twoFrameOffset << "; this->TrainedData->TrainingSetSize: " << TrainedData->TrainingSetSize << std::endl;

    abub::EventData &ev = device();
    const size_t P = (size_t)ev.W * ev.H;
    auto significance = [&](int frame, bool store) {
        return abub::significanceFromHist(pix_counts, ev.diffHist(frame, refOffset), P, store,
                                          TrainedData->TrainingSetSize, loc_thres_max, loc_thres);
    };

    const int frame_num_offset = 50 - (n - 1) / 2; // (:130, used by the debug prints only)
    const char *home = getenv("HOME");
    const std::string dbgDir = std::string(home ? home : ".") + "/test/abub_debug/ev_" + EventID + "_";
    auto debugDump = [&](int frame, double sig, bool print) {
        if (print)
            std::cout << "Entropy of BkgSub " << frame + frame_num_offset << " image: " << (float)sig << "\n";
        const int ref = std::max(frame - refOffset, 0);
        cv::Mat cur = ev.hostFrame(frame), prv = ev.hostFrame(ref);
        if (!cur.empty() && !prv.empty()) {
            cv::Mat pl[4];
            debugPlanes(cur, prv, TrainedData->TrainedSigmaImage, pl);
            cv::imwrite(dbgDir + "pos_" + CameraFrames[frame], pl[0]);
            cv::imwrite(dbgDir + "neg_" + CameraFrames[frame], pl[1]);
            cv::imwrite(dbgDir + "pos_filter_" + CameraFrames[frame], pl[2]);
            cv::imwrite(dbgDir + "neg_filter_" + CameraFrames[frame], pl[3]);
        }
        if (print) {
            cv::Mat D;
            ev.diffFrame(frame, ref, &D);
            cv::imwrite(dbgDir + CameraFrames[frame], D);
        }
    };

    for (int i = startframe; i < n; i++) {
        if (!ev.frameOk(i)) { // Parser::GetImage == -1 on the frame under evaluation
            std::cout << "Image " << CameraFrames[i] << " is corrupted/empty of camera " << CameraNumber << " for the event " << EventID << "." << std::endl;
            okToProceed = false;
            TriggerFrameIdentificationStatus = -9;
            return;
        }
        float singleEntropy = (float)significance(i, true);
        if (!nonStopMode)
            debugDump(i, singleEntropy, true);
        if (singleEntropy > entropyThreshold && i >= minEvalFrameNumber) {
            // LED-flicker veto: the next two frames must stay significant
            if (i != n - 1) {
                const int numFramesCheck = 2;
                double max_so_far = singleEntropy;
                for (int ii = 1; ii <= numFramesCheck && ii + i < n; ii++) {
                    if (!ev.frameOk(i + ii))
                        throw std::runtime_error("AnalyzerUnit::FindTriggerFrame: undecodable look-ahead frame");
                    singleEntropy = (float)significance(i + ii, false);
                    if (!nonStopMode)
                        debugDump(i + ii, singleEntropy, false); // (the look-ahead only dumps the planes, :286)
                    if (singleEntropy / (entropyThreshold / 3.5 * 5) + singleEntropy / max_so_far <= 3)
                        break;
                    else if (ii == numFramesCheck) {
                        TriggerFrameIdentificationStatus = 0;
                        MatTrigFrame = i;
                    }
                    if (singleEntropy > max_so_far)
                        max_so_far = singleEntropy;
                }
                if (TriggerFrameIdentificationStatus == 0)
                    break;
            }
        }
    }
    if (TriggerFrameIdentificationStatus == -3)
        okToProceed = false;
}

template <typename num>
double CalcMean(std::vector<num> &vec, int size)
{
    if (size == -1)
        size = (int)vec.size();
    double sum = 0;
    for (num &val : vec)
        sum += val;
    return sum / size;
}

// `val * val` in the element type (AnalyzerUnit.cpp:529): for int it wraps above 46340 on the reference's platform;
// spelled out in unsigned arithmetic so that the wrap is defined behaviour here
static double squareLikeUpstream(double v) { return v * v; }
static double squareLikeUpstream(int v) { return (double)(int)((unsigned)v * (unsigned)v); }

template <typename num>
double CalcStdDev(std::vector<num> &vec, double mean, int size)
{
    if (size == -1)
        size = (int)vec.size();
    double sum = 0;
    for (num &val : vec)
        sum += squareLikeUpstream(val);
    return std::sqrt(sum / size - mean * mean);
}
template double CalcMean<double>(std::vector<double> &, int);
template double CalcStdDev<double>(std::vector<double> &, double, int);
template double CalcMean<int>(std::vector<int> &, int);
template double CalcStdDev<int>(std::vector<int> &, double, int);

void sqrt_mat(cv::Mat &M)
{
    for (int i = 0; i < M.rows; i++) {
        uchar *Mi = M.ptr<uchar>(i);
        for (int j = 0; j < M.cols; j++) {
            double r = std::sqrt((double)Mi[j]);
            Mi[j] = (uchar)(r >= 1 ? r : 1);
        }
    }
}

bool frameSortFunc(std::string i, std::string j)
{
    unsigned int si = 0, sj = 0;
    int ci = 0, cj = 0;
    int gi = sscanf(i.c_str(), "cam%d_image%u.png", &ci, &si);
    int gj = sscanf(j.c_str(), "cam%d_image%u.png", &cj, &sj);
    assert(gi == 2 && gj == 2);
    (void)gi;
    (void)gj;
    return si < sj;
}
