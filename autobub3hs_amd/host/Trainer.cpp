// Trainer.cpp -- per-camera background model.  Control flow of the reference's
// AlgorithmTraining/Trainer.cpp:221-331 (which events train, the entropy veto, StatusCode -7); the
// two pixel passes run on the GPU: histogram of sat(f1-f0) (abub_ctx_pair_hist) and the float32
// Welford mean/sigma (abub_ctx_train).
#include "AlgorithmTraining/Trainer.hpp"

#include <atomic>
#include <cstdio>
#include <exception>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <thread>

#include "devctx.hpp"
#include "hostlogic.hpp"

static std::atomic<unsigned long long> g_modelCounter{1};

Trainer::Trainer(int camera, std::vector<std::string> EventList, std::string EventDir, std::string ImageFormat,
                 std::string ImageFolder, Parser *FileParser, bool debug)
{
    this->camera = camera;
    this->EventList = EventList;
    this->EventDir = EventDir;
    this->ImageFormat = ImageFormat;
    this->ImageFolder = ImageFolder;
    this->debug = debug;
    this->FileParser = FileParser;
    // "cam%d" + the following characters, used to pick a camera's files (Trainer.cpp:35-38)
    const std::string searchCode = "cam%d";
    const size_t at = ImageFormat.find(searchCode);
    this->SearchPattern = at == std::string::npos ? searchCode : ImageFormat.substr(at, at + searchCode.size() + 5);
    this->StatusCode = 0;
    this->TrainingSetSize = 0;
}

// Deep copy, as the reference does per analyzer (Trainer.cpp:45-73, AnalyzerUnit.cpp:27); ModelId
// travels with the images so the GPU copy of the model is shared.
Trainer::Trainer(const Trainer &o)
{
    camera = o.camera;
    EventList = o.EventList;
    EventDir = o.EventDir;
    o.TrainedAvgImage.copyTo(TrainedAvgImage);
    o.TrainedSigmaImage.copyTo(TrainedSigmaImage);
    o.TrainedLBPAvg.copyTo(TrainedLBPAvg);
    o.TrainedLBPSigma.copyTo(TrainedLBPSigma);
    isLBPApplied = o.isLBPApplied;
    ImageFormat = o.ImageFormat;
    ImageFolder = o.ImageFolder;
    SearchPattern = o.SearchPattern;
    debug = o.debug;
    FileParser = o.FileParser ? o.FileParser->clone() : nullptr;
    TrainingSetSize = o.TrainingSetSize;
    StatusCode = o.StatusCode;
    TrainingSequence = o.TrainingSequence;
    ModelId = o.ModelId;
}

Trainer::~Trainer(void)
{
    delete FileParser;
}

void Trainer::ParseAndSortFramesInFolder(std::string, std::string) {} // superseded by Parser (Trainer.cpp:248)

// 16-bin Shannon entropy of an image (Trainer.cpp:341-376): 256-bin histogram on the GPU, folded on the host.
float Trainer::calculateEntropyFrame(cv::Mat &img)
{
    abub::DeviceContext &dc = abub::DeviceContext::forThread(img.cols, img.rows, 2);
    static thread_local std::vector<uint8_t> zeros;
    zeros.assign(img.total(), 0);
    uint32_t h[256];
    abub::check(abub_ctx_pair_hist(dc.ctx, zeros.data(), img.data, h), "abub_ctx_pair_hist");
    dc.residentEvent = 0;
    return abub::entropyFromHist(h, 16, img.total());
}

void Trainer::CalculateMeanSigmaImageVector(std::vector<cv::Mat> &images, cv::Mat &mean, cv::Mat &sigma)
{
    if (images.empty())
        throw std::runtime_error("Trainer::CalculateMeanSigmaImageVector: empty image list");
    const int rows = images[0].rows, cols = images[0].cols;
    TrainingSetSize = (int)images.size(); // Trainer.cpp:161
    std::vector<const uint8_t *> ptrs;
    for (cv::Mat &m : images) {
        if (m.rows != rows || m.cols != cols)
            throw std::runtime_error("Trainer: training frames differ in size");
        ptrs.push_back(m.data);
    }
    mean.create(rows, cols, CV_8U);
    sigma.create(rows, cols, CV_8U);
    abub::DeviceContext &dc = abub::DeviceContext::forThread(cols, rows, 2);
    abub::check(abub_ctx_train(dc.ctx, ptrs.data(), (int)ptrs.size(), mean.data, sigma.data), "abub_ctx_train");
    dc.residentEvent = 0;
    dc.residentModel = 0; // abub_ctx_train leaves ITS result resident; force a keyed upload on next use
}

void Trainer::MakeAvgSigmaImage(bool PerformLBPOnImages)
{
    isLBPApplied = PerformLBPOnImages; // LBP path is dead code upstream (only `false` is ever passed)
    std::vector<cv::Mat> training;
    printf("Camera %d training ... ", camera);
    // The training frames of every event (Trainer.cpp:248-276) are listed first and decoded by a few threads, each with its
    // own Parser clone (the reference decodes them one after the other; the cameras already train side by side,
    // AutoBubStart3.cpp:304-307); everything after that -- messages, entropy veto, the order of the training set -- runs
    // in event order as upstream.
    struct Wanted {
        std::vector<std::string> frames;
        std::vector<cv::Mat> img;
        std::vector<int> err;
        std::exception_ptr thrown; // re-thrown in event order below, where the sequential loop would have met it
    };
    std::vector<Wanted> wanted(EventList.size());
    for (size_t e = 0; e < EventList.size(); ++e) {
        FileParser->ParseAndSortFramesInFolder(EventList[e], camera, wanted[e].frames);
        wanted[e].img.resize(TrainingSequence.size());
        wanted[e].err.assign(TrainingSequence.size(), -1);
    }
    {
        unsigned nthr = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 4));
        if (const char *t = getenv("ABUB_TRAIN_THREADS"))
            nthr = (unsigned)std::max(1, atoi(t));
        nthr = (unsigned)std::min<size_t>(nthr, std::max<size_t>(1, EventList.size()));
        std::atomic<size_t> next{0};
        auto work = [&](Parser *p) {
            for (;;) {
                const size_t e = next.fetch_add(1);
                if (e >= EventList.size())
                    break;
                for (size_t q = 0; q < TrainingSequence.size(); ++q) {
                    const int which = TrainingSequence[q];
                    try {
                        wanted[e].err[q] = which < (int)wanted[e].frames.size()
                                               ? p->GetImage(EventList[e], wanted[e].frames[which], wanted[e].img[q])
                                               : -1;
                    } catch (...) {
                        wanted[e].thrown = std::current_exception();
                        break;
                    }
                }
            }
        };
        std::vector<std::thread> th;
        std::vector<std::unique_ptr<Parser>> clones;
        for (unsigned t = 1; t < nthr; ++t) {
            clones.emplace_back(FileParser->clone());
            th.emplace_back(work, clones.back().get());
        }
        work(FileParser);
        for (auto &t : th)
            t.join();
    }
    for (size_t e = 0; e < EventList.size(); ++e) {
        if (wanted[e].thrown)
            std::rethrow_exception(wanted[e].thrown);
        std::vector<cv::Mat> pair;
        bool good = true;
        if (!wanted[e].frames.empty()) {
            for (size_t q = 0; q < TrainingSequence.size(); ++q) {
                if (wanted[e].err[q] != -1 && !wanted[e].img[q].empty())
                    pair.push_back(wanted[e].img[q]);
                else {
                    std::cout << "Skipping corrupted image for training.\n";
                    good = false;
                }
            }
        } else {
            std::cout << "Event " << EventList[e] << " is nonexistant on the disk. Skipping training on this event\n";
            good = false;
        }
        float entropy = 0.f;
        if (good) {
            // entropy of the saturating difference frame1 - frame0 (Trainer.cpp:279-280)
            abub::DeviceContext &dc = abub::DeviceContext::forThread(pair[0].cols, pair[0].rows, 2);
            uint32_t h[256];
            abub::check(abub_ctx_pair_hist(dc.ctx, pair[0].data, pair[1].data, h), "abub_ctx_pair_hist");
            dc.residentEvent = 0;
            entropy = abub::entropyFromHist(h, 16, pair[0].total());
        }
        if (entropy <= 0.0005 && good)
            for (cv::Mat &m : pair)
                training.push_back(m);
        wanted[e].img.clear(); // (vetoed frames are released at once)
    }
    if (training.empty()) {
        std::cout << "Training image set for camera " << camera << " has 0 frames. This means that the event is malformed." << std::endl;
        StatusCode = -7;
        return;
    }
    CalculateMeanSigmaImageVector(training, TrainedAvgImage, TrainedSigmaImage);
    ModelId = g_modelCounter.fetch_add(1);
    printf("complete.\n");
}
