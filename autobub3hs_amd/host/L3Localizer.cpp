// L3Localizer.cpp -- genesis-frame blob extraction and <=10-frame tracking.  Decisions follow the
// reference's BubbleLocalizer/L3Localizer.cpp (cited per function); images never leave HBM: the GPU
// returns a histogram (for the Otsu threshold) and the compacted foreground pixel list, from which
// the host traces the contours.
#include "BubbleLocalizer/L3Localizer.hpp"

#include <cmath>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>
#include <stdexcept>

#include <sys/stat.h>

#include "common/CommonParameters.h"
#include "devctx.hpp"
#include "hostlogic.hpp"

L3Localizer::L3Localizer(std::string EventID, std::string ImageDir, int CameraNumber, bool nonStopPref,
                         Trainer **TrainedData, std::string MaskDir, Parser *Parser)
    : AnalyzerUnit(EventID, ImageDir, CameraNumber, TrainedData, MaskDir, Parser)
{
    nonStopMode = nonStopPref;
    color = cv::Scalar(255, 255, 255);
    color_red = cv::Scalar(0, 0, 255);
    color_orange = cv::Scalar(0, 140, 255);
    color_green = cv::Scalar(0, 255, 0);
    Level1SuspicionFlag = false;
    numBubbleMultiplicity = 0;
    topCutCornerX = topCutCornerY = 0;
    blur_diam = 5;
}

namespace abub {
bool g_quietAnalyzers = false; // batched drivers switch the per-analyzer chatter off
}

L3Localizer::~L3Localizer()
{
    if (!abub::g_quietAnalyzers)
        std::cout << "Releasing memory\n"; // L3Localizer.cpp:79
}

// threshold (TOZERO tozeroThr, then BINARY|OTSU) + external contours of the context's current image
static cv::Mat cachedMask(const std::string &path);
static thread_local abub::ContourFinder t_finder;

static void contoursOfCurrentImage(abub::EventData &ev, const uint32_t *hist, int tozeroThr,
                                   std::vector<std::vector<cv::Point>> &contours, cv::Mat *debugMask = nullptr)
{
    const int thr = abub::binarizeThresholdFromHist(hist, (size_t)ev.W * ev.H, tozeroThr);
    std::vector<uint32_t> fg;
    ev.foreground(thr, fg);
    if (debugMask) { // the thresholded image of L3Localizer.cpp:254 (debug write-out only)
        *debugMask = cv::Mat::zeros(ev.H, ev.W, CV_8U);
        for (uint32_t i : fg)
            debugMask->data[i] = 255;
    }
    t_finder.find(fg, ev.W, ev.H, contours);
}

static BubbleImageFrame describe(const std::vector<cv::Point> &contour, const cv::Rect &box, bool genesisFallback)
{
    BubbleImageFrame f;
    f.ContArea = abub::contourAreaOf(contour);
    f.newPosition = box;
    f.moments = abub::momentsOf(contour);
    f.ContRadius = std::sqrt(f.ContArea / 3.14159);
    if (!genesisFallback || f.moments.m00 > 0) {
        f.MassCentres = cv::Point2f((float)(f.moments.m10 / f.moments.m00), (float)(f.moments.m01 / f.moments.m00));
    } else {
        // degenerate polygon: mean of the vertices (L3Localizer.cpp:409-418)
        double x = 0, y = 0, n = 0;
        for (const cv::Point &p : contour) {
            x += p.x;
            y += p.y;
            n++;
        }
        f.MassCentres = cv::Point2f((float)(x / n), (float)(y / n));
    }
    return f;
}

static void trackResident(abub::EventData &ev, int frame, const cv::Mat &templ, cv::Point2f &best)
{
    if (templ.cols > ev.W || templ.rows > ev.H)
        throw std::runtime_error("L3Localizer: bellows template larger than the frame");
    std::vector<unsigned long long> num, w2;
    ev.matchTerms(frame, templ, num, w2);
    abub::bestMatchFromTerms(num.data(), w2.data(), ev.W - templ.cols + 1, ev.H - templ.rows + 1, templ, best.x, best.y);
}

// L3Localizer.cpp:215-460, including the bellows-movement veto (:292-390): when every genesis contour lies in the
// bellows mask, the bellows template is located in the trigger and pre-trigger frames, its motion is rendered as a
// ProcessFrame of two synthetic frames on the overlap ROI and subtracted from D before thresholding again.
void L3Localizer::CalculateInitialBubbleParams(void)
{
    abub::EventData &ev = device();
    const int prevOffset = (TrainedData->TrainingSetSize < 6) ? 1 : 2;
    int pre = MatTrigFrame - prevOffset;
    if (pre < 0)
        pre = 0;
    // debug write-out (L3Localizer.cpp:222-257, 448-449): only with the localizer debug digit; DebugPeek/ must exist
    const bool dbg = !nonStopMode;
    const int frame_num_offset = 50 - ((int)CameraFrames.size() - 1) / 2;
    const std::string peek = "DebugPeek/ev" + EventID + "_cam" + std::to_string(CameraNumber);
    cv::Mat overTheSigma, thresFrame;
    if (dbg)
        std::cout << "-----Start ev " << EventID << ", cam " << CameraNumber << ", frame " << MatTrigFrame + frame_num_offset
                  << "-----" << std::endl;
    const uint32_t *hist = ev.diffFrame(MatTrigFrame, pre, dbg ? &overTheSigma : nullptr);
    if (dbg) {
        cv::imwrite(peek + "_000_AvgImage.png", TrainedData->TrainedAvgImage);
        cv::imwrite(peek + "_00_PreTrigFrame.png", preTrigFrame);
        cv::imwrite(peek + "_0_TrigFrame.png", triggerFrame);
        cv::imwrite(peek + "_02_OvrThe6Sigma.png", overTheSigma);
        std::cout << "this->loc_thres: " << loc_thres << std::endl;
    }
    std::vector<std::vector<cv::Point>> contours;
    contoursOfCurrentImage(ev, hist, loc_thres, contours, dbg ? &thresFrame : nullptr);
    if (dbg)
        cv::imwrite(peek + "_3_OtsuThresholded.png", thresFrame);

    std::vector<cv::Rect> minRect(contours.size());
    int largestBoxArea = 0;
    bool allInBellowsMask = !contours.empty();
    {
        std::vector<std::vector<cv::Point>> kept;
        std::vector<cv::Rect> keptRect;
        for (size_t i = 0; i < contours.size(); i++) {
            cv::Rect r = abub::boundingRectOf(contours[i]);
            if (!isInMask(&r, true)) {
                allInBellowsMask = false;
                largestBoxArea = std::max(largestBoxArea, r.width * r.height);
                kept.push_back(contours[i]);
                keptRect.push_back(r);
            } else if (!nonStopMode)
                std::cout << "Found bubble in bellows mask." << std::endl;
        }
        if (allInBellowsMask) {
            const std::string path = MaskDir + "/cam" + std::to_string(CameraNumber) + "_bellows_template.png";
            cv::Mat TemplateImage = cachedMask(path);
            if (TemplateImage.empty()) {
                if (!abub::g_quietAnalyzers)
                    std::cout << "Template image not loadable for event " << EventID << " camera " << CameraNumber
                              << "; cannot veto bellows movement triggers" << std::endl;
            } else {
                const int W = ev.W, H = ev.H, tw = TemplateImage.cols, th = TemplateImage.rows;
                cv::Point2f pt, pp;
                trackResident(ev, MatTrigFrame, TemplateImage, pt);
                trackResident(ev, pre, TemplateImage, pp);
                if (pt.x >= pp.x) { // nudge the two copies one pixel apart (:318-325)
                    pt.x++;
                    pp.x--;
                } else {
                    pt.x--;
                    pp.x++;
                }
                const cv::Rect rt((int)pt.x, (int)pt.y, tw, th), rp((int)pp.x, (int)pp.y, tw, th);
                auto inside = [&](const cv::Rect &r) { return r.x >= 0 && r.y >= 0 && r.x + r.width <= W && r.y + r.height <= H; };
                if (!inside(rt) || !inside(rp))
                    throw std::runtime_error("L3Localizer: bellows template position outside the frame");
                cv::Mat trig_copy = cv::Mat::zeros(H, W, CV_8U), preTrig_copy = cv::Mat::zeros(H, W, CV_8U);
                for (int r = 0; r < th; ++r) {
                    std::memcpy(trig_copy.ptr<uchar>(rt.y + r) + rt.x, TemplateImage.ptr<uchar>(r), (size_t)tw);
                    std::memcpy(preTrig_copy.ptr<uchar>(rp.y + r) + rp.x, TemplateImage.ptr<uchar>(r), (size_t)tw);
                }
                cv::Rect diffROI = GetDiffROI(pt, pp, TemplateImage);
                if (diffROI.width < 0 || diffROI.height < 0 || !inside(diffROI))
                    throw std::runtime_error("L3Localizer: bellows ROI outside the frame");
                cv::Mat diff_frame;
                ProcessFrame(trig_copy, preTrig_copy, diff_frame, 5, diffROI); // uses the frame slab: D is recomputed below
                ev.diffFrame(MatTrigFrame, pre);
                hist = ev.subtractFromCurrent(diff_frame); // overTheSigma -= diff_frame (:362)
                contoursOfCurrentImage(ev, hist, loc_thres, contours);
            }
            largestBoxArea = 0;
            minRect.clear();
            for (auto &c : contours) {
                minRect.push_back(abub::boundingRectOf(c));
                largestBoxArea = std::max(largestBoxArea, minRect.back().width * minRect.back().height);
            }
        } else {
            contours.swap(kept);
            minRect.swap(keptRect);
        }
    }
    for (size_t i = 0; i < contours.size(); i++) {
        const int BoxArea = minRect[i].width * minRect[i].height;
        if (BoxArea > 10 || BoxArea >= largestBoxArea) {
            bubbleRects.push_back(minRect[i]);
            if (dbg && !presentationFrame.empty())
                cv::rectangle(presentationFrame, minRect[i], cv::Scalar(255, 255, 255), 1, 8, 0); // (:397)
            BubbleImageFrame f = describe(contours[i], minRect[i], true);
            if (!isInMask(&f.newPosition))
                continue; // genesis outside the fiducial mask
            BubbleList.push_back(new bubble(f));
        }
    }
    if (dbg) {
        cv::imwrite(peek + "_4_BubbleDetected.png", presentationFrame);
        std::cout << "-----End ev " << EventID << ", cam " << CameraNumber << ", frame " << MatTrigFrame + frame_num_offset
                  << "-----" << std::endl;
    }
}

// L3Localizer.cpp:764-869
void L3Localizer::CalculatePostTriggerFrameParams(int postTrigFrameNumber)
{
    abub::EventData &ev = device();
    const int frame = MatTrigFrame + postTrigFrameNumber;
    if (!ev.frameOk(frame))
        throw std::runtime_error("L3Localizer: undecodable post-trigger frame");
    const uint32_t *hist = ev.postTrig(frame);
    std::vector<std::vector<cv::Point>> contours;
    contoursOfCurrentImage(ev, hist, 3, contours);

    std::vector<BubbleImageFrame> sightings;
    for (auto &c : contours) {
        cv::Rect r = abub::boundingRectOf(c);
        if (r.width * r.height > 10) {
            BubbleImageFrame f = describe(c, r, false);
            if (!isInMask(&f.newPosition))
                continue;
            sightings.push_back(f);
        }
    }
    for (bubble *b : BubbleList)
        b->lockThisIteration = false;
    // first bubble whose last position is close enough takes the sighting; the search stops there
    // even if that bubble was already served this frame (L3Localizer.cpp:847-863)
    for (BubbleImageFrame &s : sightings) {
        const float x = s.MassCentres.x, y = s.MassCentres.y;
        for (bubble *b : BubbleList) {
            const float bx = b->last_x, by = b->last_y;
            if ((bx - x < 5) && (std::fabs(by - y) < 5)) {
                *b << s;
                break;
            }
        }
    }
}

void L3Localizer::printBubbleList(void)
{
    for (bubble *b : BubbleList)
        b->printAllXY();
}

// L3Localizer.cpp:881-968
void L3Localizer::LocalizeOMatic(std::string)
{
    if (CameraFrames.size() <= 5)
        okToProceed = false;
    if (!okToProceed)
        return;
    abub::EventData &ev = device();
    const int prevOffset = (TrainedData->TrainingSetSize < 6) ? 1 : 2;
    int preTrigNum = MatTrigFrame - prevOffset;
    if (preTrigNum < 0)
        preTrigNum = 0;
    if (!ev.frameOk(MatTrigFrame) || !ev.frameOk(preTrigNum) || !ev.frameOk(0))
        throw std::runtime_error("L3Localizer::LocalizeOMatic: undecodable trigger / pre-trigger frame");
    triggerFrame = ev.hostFrame(MatTrigFrame); // empty when the frames only live in HBM
    preTrigFrame = ev.hostFrame(preTrigNum);
    presentationFrame = triggerFrame.clone();
    ComparisonFrame = ev.hostFrame(0);

    CalculateInitialBubbleParams();

    const int last = (MatTrigFrame < 29) ? NumFramesBubbleTrack : (39 - MatTrigFrame);
    for (int k = 1; k <= last; k++) {
        if ((size_t)(MatTrigFrame + k) >= CameraFrames.size())
            break;
        CalculatePostTriggerFrameParams(k);
    }
}

// mask files are immutable during a run: decode each once per process instead of once per analyzer
static cv::Mat cachedMask(const std::string &path)
{
    static std::mutex mu;
    static std::map<std::string, cv::Mat> cache;
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0)
        return cv::Mat();
    const std::string key = path + "#" + std::to_string((long long)sb.st_mtim.tv_sec) + "." +
                            std::to_string((long long)sb.st_mtim.tv_nsec) + "#" + std::to_string((long long)sb.st_size);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end())
        return it->second;
    cv::Mat m = cv::imread(path, cv::IMREAD_GRAYSCALE);
    if (!m.empty())
        cache[key] = m;
    return m;
}

// L3Localizer.cpp:971-1012.  Lookups outside the mask image (unchecked upstream) count as "outside".
bool L3Localizer::isInMask(cv::Rect *genesis_coords, bool bellows)
{
    const int xpix = (int)(genesis_coords->x + genesis_coords->width / 2.);
    const int ypix = genesis_coords->y + genesis_coords->height / 2;
    if (MaskDir == "")
        return !bellows;
    std::string path = MaskDir + "/cam" + std::to_string(CameraNumber);
    if (bellows)
        path += "_bellows";
    path += "_mask.bmp";
    if (bellows && bellows_mask.empty() && !bellows_mask_tried) {
        bellows_mask = cachedMask(path);
        bellows_mask_tried = true;
    } else if (!bellows && cam_mask.empty() && !cam_mask_tried) {
        cam_mask = cachedMask(path);
        cam_mask_tried = true;
    }
    const cv::Mat &mask = bellows ? bellows_mask : cam_mask;
    if (mask.empty()) {
        // (cameras without a bellows mask file -- cam0 / cam2 of 40l-19 -- get this line for every contour upstream; the
        // batched pipeline runs its analyzers quietly: hundreds of stacks per step would serialise on the stream's lock)
        if (!abub::g_quietAnalyzers)
            std::cout << "Mask image not loadable for event " << EventID << " camera " << CameraNumber << "; skipping mask check" << std::endl;
        return !bellows;
    }
    if (xpix < 0 || ypix < 0 || xpix >= mask.cols || ypix >= mask.rows)
        return false;
    return (int)mask.at<uchar>(ypix, xpix) > 0;
}

// ---- entry points that exist upstream but belong to rows outside the current scope --------------
cv::Rect L3Localizer::GetDiffROI(cv::Point2f p1, cv::Point2f p2, cv::Mat &frame)
{
    int sx = (int)std::max(p1.x, p2.x), sy = (int)std::max(p1.y, p2.y);
    int dx = (int)(std::min(p1.x, p2.x) + frame.cols - sx), dy = (int)(std::min(p1.y, p2.y) + frame.rows - sy);
    return cv::Rect(sx, sy, dx, dy);
}
// public form on an arbitrary host frame (L3Localizer.cpp:473-543): the frame is staged as a one-frame stack
void L3Localizer::TrackAFeature(cv::Mat &frame, cv::Mat TemplateImage, cv::Point2f &BestMatchLoc)
{
    if (frame.empty() || TemplateImage.empty() || TemplateImage.cols > frame.cols || TemplateImage.rows > frame.rows)
        throw std::runtime_error("L3Localizer::TrackAFeature: empty image or template larger than the frame");
    abub::DeviceContext &dc = abub::DeviceContext::forThread(frame.cols, frame.rows, 2);
    const uint8_t *fr[1] = {frame.data};
    abub::check(abub_ctx_upload_stack(dc.ctx, fr, 1), "abub_ctx_upload_stack");
    dc.residentEvent = 0;
    const int rw = frame.cols - TemplateImage.cols + 1, rh = frame.rows - TemplateImage.rows + 1;
    std::vector<unsigned long long> num((size_t)rw * rh), w2((size_t)rw * rh);
    abub::check(abub_ctx_match_template(dc.ctx, 0, TemplateImage.data, TemplateImage.cols, TemplateImage.rows, num.data(),
                                        w2.data()),
                "abub_ctx_match_template");
    abub::bestMatchFromTerms(num.data(), w2.data(), rw, rh, TemplateImage, BestMatchLoc.x, BestMatchLoc.y);
}
void L3Localizer::CalculateInitialBubbleParamsCam2(void) {}      // dead upstream (L3Localizer.cpp:547)
void L3Localizer::CalculatePostTriggerFrameParamsCam2(int) {}    // dead upstream
bool bubbleBRectSort(cv::RotatedRect a, cv::RotatedRect b) { return a.center.y < b.center.y; }
