// ImageEntropyMethods.hpp -- free-function 16-bin Shannon entropy of an image (API of the reference's
// ImageEntropyMethods/ImageEntropyMethods.hpp; built upstream as libAbubEntropySubsystem.so but no longer called:
// Trainer and AnalyzerUnit carry their own copies).  Here: 256-bin histogram on the GPU, folded on the host.
// Unlike the reference's version (global cv::Mat state, ImageEntropyMethods.cpp:21-26) this one is re-entrant.
#ifndef ABUB3HS_IMAGEENTROPYMETHODS_HPP
#define ABUB3HS_IMAGEENTROPYMETHODS_HPP

#include "../cvlite.hpp"

float calculateEntropyFrame(cv::Mat &ImageFrame);

#endif
