// AnalyzerUnit.hpp -- abstract per-(event, camera) analyzer: owns the frame list, finds the trigger
// frame.  Public surface of the reference's AnalyzerUnit.hpp:16-98; the pixel work is done by the
// HIP library through a per-thread context (one batched pass over the whole frame stack).
#ifndef ABUB3HS_ANALYZERUNIT_HPP
#define ABUB3HS_ANALYZERUNIT_HPP

#include <string>
#include <vector>

#include "AlgorithmTraining/Trainer.hpp"
#include "ParseFolder/Parser.hpp"
#include "bubble/bubble.hpp"
#include "cvlite.hpp"

#define MIN_IMAGE_SIZE 100000

namespace abub {
class EventData;       // GPU-side view of this analyzer's frame stack (host/devctx.hpp)
struct AnalyzerProbe;  // test hook (host/capi.cpp): reaches the private per-frame statistics, dead upstream
}

class AnalyzerUnit {
    friend struct abub::AnalyzerProbe;
    int StatusCode = 0;

    int minEvalFrameNumber = 2; // frames 0,1 are training frames
    int firstTrainingFrames = 1;
    int loc_thres_max = 3;

    float calculateEntropyFrame(cv::Mat &, bool debug = false);
    double calculateSignificanceFrame(cv::Mat &ImageFrame, bool store, bool debug = false);
    double calculateEntropySignificance(cv::Mat &ImageFrame, bool store, bool debug = false);

protected:
    std::string ImageDir;
    std::string EventID;
    std::string MaskDir;

    std::vector<std::string> CameraFrames;
    int CameraNumber;

    Parser *FileParser;

    // GPU-side data of this (event, camera): created lazily from the Parser (decode + upload + one
    // batched histogram pass), or injected by a run-level pipeline that already holds the frames in HBM
    abub::EventData *dev = nullptr;
    bool ownsDev = true;
    abub::EventData &device();

public:
    AnalyzerUnit(std::string EventID, std::string ImageDir, int CameraNumber, Trainer **TrainedData,
                 std::string MaskDir, Parser *Parser);
    virtual ~AnalyzerUnit(void);

    void ParseAndSortFramesInFolder(void);
    void ProduceOutput(void);
    void gammaCorrection(const cv::Mat &src, cv::Mat &dst, const float gamma);

    // sigma-suppressed, blurred two-sided frame difference (full frame and ROI overloads)
    void ProcessFrame(cv::Mat &workingFrame, cv::Mat &prevFrame, cv::Mat &subtr_frame, int blur_diam = 5, int img_num = -1);
    void ProcessFrame(cv::Mat &workingFrame, cv::Mat &prevFrame, cv::Mat &subtr_frame, int blur_diam, cv::Rect ROI, int img_num = -1);

    std::vector<cv::RotatedRect> BubblePixelPos;
    int MatTrigFrame;

    int loc_thres;

    std::vector<std::vector<int>> pix_counts;
    std::vector<double> entropies;

    void FindTriggerFrame(bool nonStopMode, int startframe);

    // not in the reference: lets a batched driver supply the GPU-side data (not owned)
    void AttachEventData(abub::EventData *data);

    virtual void LocalizeOMatic(std::string) = 0;
    std::vector<cv::Rect> bubbleRects;

    Trainer *TrainedData;
    std::vector<bubble *> BubbleList;

    bool okToProceed = true;
    int TriggerFrameIdentificationStatus = 0; // 0 ok, -3 no trigger, -9 malformed sequence
};

template <typename num>
double CalcMean(std::vector<num> &vec, int size = -1);
template <typename num>
double CalcStdDev(std::vector<num> &vec, double mean, int size = -1);
bool frameSortFunc(std::string, std::string);
void sqrt_mat(cv::Mat &M);

#endif
