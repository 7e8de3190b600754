// ImageEntropyMethods.cpp -- see the header; semantics of ImageEntropyMethods.cpp:32-57 (16 bins over [0,256),
// p = count / (rows*cols) in float32, E = -sum p*log2(p)).
#include "ImageEntropyMethods/ImageEntropyMethods.hpp"

#include <vector>

#include "devctx.hpp"
#include "hostlogic.hpp"

float calculateEntropyFrame(cv::Mat &ImageFrame)
{
    abub::DeviceContext &dc = abub::DeviceContext::forThread(ImageFrame.cols, ImageFrame.rows, 2);
    std::vector<uint8_t> zeros(ImageFrame.total(), 0);
    uint32_t h[256];
    abub::check(abub_ctx_pair_hist(dc.ctx, zeros.data(), ImageFrame.data, h), "abub_ctx_pair_hist");
    dc.residentEvent = 0;
    return abub::entropyFromHist(h, 16, ImageFrame.total());
}
