// Trainer.hpp -- per-camera background model (mean / sigma image), API of the reference's
// AlgorithmTraining/Trainer.hpp:13-66.  The Welford pass and the training-veto histogram run on the
// GPU (abub_ctx_train / abub_ctx_pair_hist); this class keeps the reference's public fields.
#ifndef ABUB3HS_TRAINER_HPP
#define ABUB3HS_TRAINER_HPP

#include <string>
#include <vector>

#include "../ParseFolder/Parser.hpp"
#include "../cvlite.hpp"

class Trainer {
    void ParseAndSortFramesInFolder(std::string, std::string);
    float calculateEntropyFrame(cv::Mat &);

    std::vector<std::string> CameraFrames;
    int camera;
    std::string EventDir;
    std::string ImageFolder;
    bool debug;
    Parser *FileParser;

public:
    std::string ImageFormat;
    std::string SearchPattern;

    cv::Mat TrainedAvgImage;
    cv::Mat TrainedSigmaImage;
    cv::Mat TrainedLBPAvg;   // LBP branch is dead in the reference (MakeAvgSigmaImage(false) only)
    cv::Mat TrainedLBPSigma;

    bool isLBPApplied = false;

    int StatusCode;       // 0 ok, -7 no usable training frames
    int TrainingSetSize;  // number of frames that went into the model

    std::vector<std::string> EventList;
    std::vector<int> TrainingSequence{0, 1}; // frame indices of each event used for training

    // identifies the (mu, sigma) pair so that a GPU context re-uploads it only when it changes;
    // copied by the copy constructor together with the images
    unsigned long long ModelId = 0;

    Trainer(int camera, std::vector<std::string> EventList, std::string EventDir, std::string ImageFormat,
            std::string ImageFolder, Parser *FileParser, bool debug = false);
    Trainer(const Trainer &other);
    ~Trainer(void);

    void MakeAvgSigmaImage(bool PerformLBPOnImages);
    void CalculateMeanSigmaImageVector(std::vector<cv::Mat> &images, cv::Mat &mean, cv::Mat &sigma);
};

#endif
