// L3Localizer.hpp -- concrete analyzer: genesis-frame blob extraction, mask checks, <=10-frame
// tracking.  Public surface of the reference's BubbleLocalizer/L3Localizer.hpp:28-90.
#ifndef ABUB3HS_L3LOCALIZER_HPP
#define ABUB3HS_L3LOCALIZER_HPP

#include <string>
#include <vector>

#include "../AlgorithmTraining/Trainer.hpp"
#include "../AnalyzerUnit.hpp"
#include "../bubble/bubble.hpp"
#include "../cvlite.hpp"

#define SEARCH_LEVEL_1 0
#define SEARCH_LEVEL_2 1

class L3Localizer : public AnalyzerUnit {
    bool nonStopMode;
    cv::Scalar color, color_orange, color_green, color_red;

    int topCutCornerX;
    int topCutCornerY;

    cv::Mat presentationFrame;
    cv::Mat ComparisonFrame, triggerFrame, preTrigFrame;
    cv::Mat PostTrigWorkingFrame;
    cv::Mat cam_mask, bellows_mask;
    bool cam_mask_tried = false, bellows_mask_tried = false;

    int blur_diam;

public:
    L3Localizer(std::string EventID, std::string ImageDir, int CameraNumber, bool nonStopPref,
                Trainer **TrainedData, std::string MaskDir, Parser *Parser);
    ~L3Localizer();

    void CalculateInitialBubbleParams(void);
    cv::Rect GetDiffROI(cv::Point2f point1, cv::Point2f point2, cv::Mat &frame);
    void TrackAFeature(cv::Mat &frame, cv::Mat TemplateImage, cv::Point2f &BestMatchLoc);
    void CalculateInitialBubbleParamsCam2(void);
    void CalculatePostTriggerFrameParams(int postTrigFrameNumber);
    void CalculatePostTriggerFrameParamsCam2(int);
    void printBubbleList(void);

    int numBubbleMultiplicity = 0;
    bool Level1SuspicionFlag;

    void LocalizeOMatic(std::string) override;
    bool isInMask(cv::Rect *, bool bellows = false);
};

bool bubbleBRectSort(cv::RotatedRect, cv::RotatedRect);

#endif
