// hostlogic.hpp -- the sequential, per-event host logic that sits between the HIP kernels:
// 256-entry histogram statistics, Otsu, and contour extraction from the (sparse) foreground list.
// None of this touches full frames: the kernels hand over histograms and compacted pixel indices.
#ifndef ABUB3HS_HOSTLOGIC_HPP
#define ABUB3HS_HOSTLOGIC_HPP

#include <cstddef>
#include <cstdint>
#include <vector>

#include "cvlite.hpp"

namespace abub {

// AnalyzerUnit::calculateSignificanceFrame (reference AnalyzerUnit.cpp:435-504) evaluated on the
// 256-bin histogram of the diff frame.  pix_counts / loc_thres are the analyzer's public fields.
double significanceFromHist(std::vector<std::vector<int>> &pix_counts, const uint32_t hist[256],
                            size_t totalPixels, bool store, int trainingSetSize, int locThresMax,
                            int &loc_thres);

// Shannon entropy of an nbins-bin histogram folded from 256 bins (nbins = 16: Trainer.cpp:341-376 and
// ImageEntropyMethods.cpp:32-57; nbins = 128: AnalyzerUnit.cpp:386-423).  float32 like the reference.
float entropyFromHist(const uint32_t hist[256], int nbins, size_t totalPixels);

// Threshold that cv::threshold(THRESH_TOZERO, tozeroThr) followed by THRESH_BINARY|THRESH_OTSU
// applies (L3Localizer.cpp:252-254, 786-787): returns t such that mask = (v > t).
int binarizeThresholdFromHist(const uint32_t hist[256], size_t totalPixels, int tozeroThr);

// cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_TC89_L1) driven by the foreground pixel list instead
// of a dense scan (L3Localizer.cpp:264, 374, 793).  One finder per thread: it keeps a zeroed,
// padded scratch plane between calls and touches only the listed pixels.
class ContourFinder {
public:
    ContourFinder() : w_(0), h_(0) {}
    // idx: raster indices (y*W+x) of the foreground pixels, in any order.
    void find(std::vector<uint32_t> &idx, int W, int H, std::vector<std::vector<cv::Point>> &contours);

private:
    void traceBorder(size_t start, std::vector<signed char> &codes);
    std::vector<signed char> plane_;
    std::vector<int> rowLo_, rowHi_; // x-span of the foreground per touched row
    int w_, h_;
};

// Teh-Chin (L1 curvature) dominant points of a closed Freeman chain.
void approxChainTC89L1(cv::Point origin, const std::vector<signed char> &codes, std::vector<cv::Point> &out);

// L3Localizer::TrackAFeature (L3Localizer.cpp:497-540) from the exact correlation terms the GPU returns:
// CCORR_NORMED normalisation (double), cv::normalize to [0,1] (float32), first maximum, 3x3 centre of mass.
void bestMatchFromTerms(const unsigned long long *num, const unsigned long long *wsum2, int rw, int rh,
                        const cv::Mat &templ, float &bx, float &by);

cv::Rect boundingRectOf(const std::vector<cv::Point> &pts);
double contourAreaOf(const std::vector<cv::Point> &pts);
cv::Moments momentsOf(const std::vector<cv::Point> &pts);

} // namespace abub
#endif
