// capi.cpp -- small C surface over the C++ API mirror so that the Python test-suite and bench.py can
// drive Trainer / L3Localizer exactly the way the reference's main() does (AutoBubStart3.cpp:294-307,
// :363-364 and AnyCamAnalysis :67-125).  Not part of the drop-in boundary.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "AlgorithmTraining/Trainer.hpp"
#include "AnalyzerUnit.hpp"
#include "ImageEntropyMethods/ImageEntropyMethods.hpp"
#include "BubbleLocalizer/L3Localizer.hpp"
#include "PICOFormatWriter/PICOFormatWriterV4.hpp"
#include "ParseFolder/Parser.hpp"
#include "ParseFolder/RawParser.hpp"
#include "ParseFolder/ZipParser.hpp"
#include "devctx.hpp"
#include "driver.hpp"
#include "hostlogic.hpp"
#include "pngwalk.hpp"
#include "runbatch.hpp"

namespace {

struct BubbleOut {
    std::vector<BubbleImageFrame> desc;
    std::vector<float> dz;
    float dzdt, drdt;
};

struct Run {
    Parser *parser = nullptr; // MemParser (tests / synthetic), RawParser or ZipParser
    MemParser *mem = nullptr;
    std::string frames_of_last_query;
    std::map<int, Trainer *> trainers;
    // last analysis
    int staged = 0, trig = 0, status = 0, loc_thres = 0, ok = 0;
    std::vector<BubbleOut> bubbles;
    std::string error;
};

// AnyCamAnalysis (AutoBubStart3.cpp:87-117): retry until a bubble is found or the search fails.
int anyCamAnalysis(AnalyzerUnit *A, Run *run)
{
    int staged = 0;
    try {
        do {
            A->FindTriggerFrame(true, A->MatTrigFrame + 1);
            if (A->okToProceed) {
                A->LocalizeOMatic("");
                if (A->okToProceed)
                    staged = A->BubbleList.empty() ? -1 : 0; // stageCameraOutput (PICOFormatWriterV4.cpp:99-110)
                else {
                    staged = -8;
                    break;
                }
            } else {
                staged = A->TriggerFrameIdentificationStatus;
                break;
            }
        } while (A->BubbleList.size() == 0);
    } catch (std::exception &e) {
        run->error = e.what();
        std::cout << e.what() << '\n';
        staged = -6;
    }
    return staged;
}

} // namespace

// test hook for the per-frame statistics that are private in AnalyzerUnit and dead upstream (128-bin entropy and its
// z-score, AnalyzerUnit.cpp:386-433; the image overload of calculateSignificanceFrame :435-504)
namespace abub {
struct AnalyzerProbe {
    static float entropy(AnalyzerUnit &A, cv::Mat &m) { return A.calculateEntropyFrame(m, false); }
    static double entropySignificance(AnalyzerUnit &A, cv::Mat &m, bool store) { return A.calculateEntropySignificance(m, store, false); }
    static double significance(AnalyzerUnit &A, cv::Mat &m, bool store) { return A.calculateSignificanceFrame(m, store, false); }
};
} // namespace abub

extern "C" {

// feeds `n` images [n][H][W] through an analyzer of (ev, cam): out[3*k] = 128-bin entropy of image k, out[3*k+1] =
// its z-score over the history so far (stored), out[3*k+2] = the 256-bin significance (stored)
int abh_probe_frame_stats(void *r, const char *ev, int cam, const uint8_t *imgs, int n, int W, int H, double *out)
{
    Run *run = (Run *)r;
    auto it = run->trainers.find(cam);
    if (it == run->trainers.end() || !it->second)
        return -100;
    Trainer *t = it->second;
    try {
        L3Localizer A(ev, "", cam, true, &t, "", run->parser->clone());
        for (int k = 0; k < n; ++k) {
            cv::Mat m(H, W, CV_8U);
            std::memcpy(m.data, imgs + (size_t)k * W * H, (size_t)W * H);
            out[3 * k] = abub::AnalyzerProbe::entropy(A, m);
            out[3 * k + 1] = abub::AnalyzerProbe::entropySignificance(A, m, true);
            out[3 * k + 2] = abub::AnalyzerProbe::significance(A, m, true);
        }
    } catch (std::exception &e) {
        run->error = e.what();
        return -1;
    }
    abub::DeviceContext::releaseThread();
    return 0;
}

void *abh_run_new()
{
    Run *r = new Run();
    r->mem = new MemParser();
    r->parser = r->mem;
    return r;
}

// kind 0: directory tree (RawParser), 1: zip archive (ZipParser); NULL if the source cannot be opened
void *abh_run_open(int kind, const char *runFolder, const char *imageFolder, const char *imageFormat)
{
    try {
        Run *r = new Run();
        if (kind == 0)
            r->parser = new RawParser(runFolder, imageFolder, imageFormat);
        else
            r->parser = new ZipParser(runFolder, imageFolder, imageFormat);
        return r;
    } catch (int code) {
        return nullptr;
    } catch (std::exception &) {
        return nullptr;
    }
}

// '\n'-joined list into the run's scratch string; returns its c_str (valid until the next query)
const char *abh_run_events(void *r)
{
    Run *run = (Run *)r;
    std::vector<std::string> ev;
    run->parser->GetEventDirLists(ev);
    std::sort(ev.begin(), ev.end(), [](const std::string &a, const std::string &b) { return std::stoi(a) < std::stoi(b); });
    run->frames_of_last_query.clear();
    for (auto &e : ev)
        run->frames_of_last_query += e + "\n";
    return run->frames_of_last_query.c_str();
}
const char *abh_run_frames(void *r, const char *ev, int cam)
{
    Run *run = (Run *)r;
    std::vector<std::string> fr;
    run->parser->ParseAndSortFramesInFolder(ev, cam, fr);
    run->frames_of_last_query.clear();
    for (auto &f : fr)
        run->frames_of_last_query += f + "\n";
    return run->frames_of_last_query.c_str();
}
// decode one frame through the parser: returns GetImage's code, fills w/h and (if cap suffices) the pixels
int abh_run_image(void *r, const char *ev, const char *frame, uint8_t *out, int cap, int *w, int *h)
{
    Run *run = (Run *)r;
    cv::Mat m;
    int rc = run->parser->GetImage(ev, frame, m);
    *w = m.cols;
    *h = m.rows;
    if (!m.empty() && (int)m.total() <= cap)
        std::memcpy(out, m.data, m.total());
    return rc;
}
int abh_imdecode(const uint8_t *data, int n, uint8_t *out, int cap, int *w, int *h)
{
    cv::Mat m = cv::imdecode(data, (size_t)n, 0);
    *w = m.cols;
    *h = m.rows;
    if (m.empty())
        return -1;
    if ((int)m.total() <= cap)
        std::memcpy(out, m.data, m.total());
    return 0;
}

// cv::imwrite of the debug write-out (PNG, or BMP by extension)
// pngWalk (host/pngwalk.hpp: what RunBatched learns about a file before the GPU decodes it): 1 = a W x H 8-bit grey / palette PNG
// (segs_out: up to cap (offset, length) pairs of its IDAT chunks, *nsegs their number, *palette, lut_out[256]), 0 = a file
// for the host decoder
int abh_png_walk(const uint8_t *data, int n, int W, int H, uint32_t *segs_out, int cap, int *nsegs, int *palette, uint8_t *lut_out)
{
    abub::PngInfo info;
    if (!abub::pngWalk(data, (size_t)n, W, H, info))
        return 0;
    *nsegs = (int)info.segs.size();
    for (int i = 0; i < (int)info.segs.size() && i < cap; ++i) {
        segs_out[2 * i] = info.segs[i].off;
        segs_out[2 * i + 1] = info.segs[i].len;
    }
    *palette = info.palette ? 1 : 0;
    if (info.palette)
        std::memcpy(lut_out, info.lut, 256);
    return 1;
}

int abh_imwrite(const char *path, const uint8_t *img, int W, int H)
{
    cv::Mat m(H, W, CV_8U);
    std::memcpy(m.data, img, (size_t)W * H);
    return cv::imwrite(path, m) ? 0 : -1;
}

void abh_run_free(void *r)
{
    Run *run = (Run *)r;
    for (auto &kv : run->trainers)
        delete kv.second;
    delete run->parser;
    delete run;
    abub::DeviceContext::releaseThread();
}

// frames: [F][H][W]; ok: F flags or NULL
int abh_run_add_event(void *r, const char *ev, int cam, const uint8_t *frames, int F, int W, int H, const uint8_t *ok)
{
    Run *run = (Run *)r;
    std::vector<cv::Mat> v;
    for (int k = 0; k < F; ++k) {
        cv::Mat m;
        if (!ok || ok[k]) {
            m.create(H, W, CV_8U);
            std::memcpy(m.data, frames + (size_t)k * W * H, (size_t)W * H);
        }
        v.push_back(m);
    }
    if (!run->mem)
        return -1;
    run->mem->AddFrames(ev, cam, v);
    return 0;
}

// Trainer over every event of the run (numeric order like AutoBubStart3.cpp:284)
int abh_train(void *r, int cam, int *status, int *tss, uint8_t *mu_out, uint8_t *sigma_out)
{
    Run *run = (Run *)r;
    try {
        std::vector<std::string> events;
        run->parser->GetEventDirLists(events);
        std::sort(events.begin(), events.end(), [](const std::string &a, const std::string &b) { return std::stoi(a) < std::stoi(b); });
        delete run->trainers[cam];
        Trainer *t = new Trainer(cam, events, "", "cam%d_image%u.png", "", run->parser->clone(), false);
        run->trainers[cam] = t;
        t->MakeAvgSigmaImage(false);
        *status = t->StatusCode;
        *tss = t->TrainingSetSize;
        if (t->StatusCode == 0 && mu_out && sigma_out) {
            std::memcpy(mu_out, t->TrainedAvgImage.data, t->TrainedAvgImage.total());
            std::memcpy(sigma_out, t->TrainedSigmaImage.data, t->TrainedSigmaImage.total());
        }
        return 0;
    } catch (std::exception &e) {
        run->error = e.what();
        return -1;
    }
}

// install a model directly (tests that want a specific mu/sigma/TrainingSetSize)
int abh_set_model(void *r, int cam, const uint8_t *mu, const uint8_t *sigma, int W, int H, int tss)
{
    Run *run = (Run *)r;
    static unsigned long long next = 1ull << 40;
    delete run->trainers[cam];
    Trainer *t = new Trainer(cam, {}, "", "cam%d_image%u.png", "", run->parser->clone(), false);
    t->TrainedAvgImage.create(H, W, CV_8U);
    t->TrainedSigmaImage.create(H, W, CV_8U);
    std::memcpy(t->TrainedAvgImage.data, mu, (size_t)W * H);
    std::memcpy(t->TrainedSigmaImage.data, sigma, (size_t)W * H);
    t->TrainingSetSize = tss;
    t->ModelId = next++;
    run->trainers[cam] = t;
    return 0;
}

int abh_analyze(void *r, const char *ev, int cam, const char *maskdir)
{
    Run *run = (Run *)r;
    run->bubbles.clear();
    run->error.clear();
    auto it = run->trainers.find(cam);
    if (it == run->trainers.end() || !it->second) {
        run->error = "no trainer for camera";
        return -100;
    }
    Trainer *t = it->second;
    AnalyzerUnit *A = nullptr;
    try {
        A = new L3Localizer(ev, "", cam, true, &t, maskdir ? maskdir : "", run->parser->clone());
    } catch (std::exception &e) {
        run->error = e.what();
        return -100;
    }
    run->staged = anyCamAnalysis(A, run);
    run->trig = A->MatTrigFrame;
    run->status = A->TriggerFrameIdentificationStatus;
    run->loc_thres = A->loc_thres;
    run->ok = A->okToProceed;
    for (bubble *b : A->BubbleList) {
        BubbleOut o;
        o.desc = b->KnownDescriptors;
        o.dz = b->dz;
        o.dzdt = b->dZdT();
        o.drdt = b->dRdT();
        run->bubbles.push_back(o);
    }
    delete A;
    return run->staged;
}

// A whole run (every event of the Run's parser, cameras 0..ncams-1, the Run's trained models) through the batched
// pipeline into <outdir>abub3hs_<run>.txt; stats: [total_s, list_s, decode_s, gpu_s, write_s, frames, failed,
// batches, events_per_batch, gpus, frames decoded on the GPU, on host threads, gpudecode_s].  Returns 0, 1 when the batched path declines the run, -1 on errors.
int abh_run_batched(void *r, int ncams, const char *maskdir, const char *outdir, const char *run_number, int frameOffset,
                    int ngpus, int nthreads, int decodeThreads, int batchMB, int shardRank, int shardWorld, double *statsOut)
{
    Run *run = (Run *)r;
    try {
        std::vector<std::string> events;
        run->parser->GetEventDirLists(events);
        std::sort(events.begin(), events.end(), [](const std::string &a, const std::string &b) { return std::stoi(a) < std::stoi(b); });
        std::vector<Trainer *> trainers;
        for (int c = 0; c < ncams; ++c) {
            auto it = run->trainers.find(c);
            if (it == run->trainers.end() || !it->second) {
                run->error = "no trainer for camera";
                return -1;
            }
            trainers.push_back(it->second);
        }
        abub::BatchedRunOptions bo;
        bo.ngpus = std::max(1, ngpus);
        bo.hostThreads = std::max(1, nthreads);
        bo.decodeThreads = std::max(1, decodeThreads);
        if (batchMB > 0)
            bo.batchBytes = (size_t)batchMB << 20;
        bo.shardRank = shardRank;
        bo.shardWorld = std::max(1, shardWorld);
        bo.maskDir = maskdir ? maskdir : "";
        abub::BatchedRunStats bs;
        std::string why;
        const int rc = abub::RunBatched(run->parser, events, trainers, ncams, outdir, run_number, frameOffset, bo, &bs, &why);
        if (rc != 0)
            run->error = why;
        if (statsOut) {
            const double v[13] = {bs.total_s, bs.list_s, bs.decode_s, bs.gpu_s, bs.write_s, (double)bs.frames, (double)bs.framesFailed,
                                  (double)bs.batches, (double)bs.eventsPerBatch, (double)bs.gpus, (double)bs.framesGpuDecoded,
                                  (double)bs.framesHostDecoded, bs.gpudecode_s};
            std::memcpy(statsOut, v, sizeof v);
        }
        return rc;
    } catch (std::exception &e) {
        run->error = e.what();
        return -1;
    }
}

// reference main(): header once per run (AutoBubStart3.cpp:250-251)
void abh_write_header(const char *outdir, const char *run_number, int frameOffset, int ncams)
{
    OutputWriter w(outdir, run_number, frameOffset, ncams);
    w.writeHeader();
}

// reference main() loop body for one event (AutoBubStart3.cpp:352-387): a writer per event, one
// L3Localizer per camera driven by AnyCamAnalysis, then the block is appended to abub3hs_<run>.txt
int abh_event_to_file(void *r, const char *ev, int actualEventNumber, int ncams, const char *maskdir,
                      const char *outdir, const char *run_number, int frameOffset)
{
    Run *run = (Run *)r;
    OutputWriter writer(outdir, run_number, frameOffset, ncams);
    std::vector<AnalyzerUnit *> analyzers;
    for (int cam = 0; cam < ncams; ++cam) {
        auto it = run->trainers.find(cam);
        if (it == run->trainers.end() || !it->second)
            return -100;
        Trainer *t = it->second;
        AnalyzerUnit *A = new L3Localizer(ev, "", cam, true, &t, maskdir ? maskdir : "", run->parser->clone());
        analyzers.push_back(A);
        abub::AnyCamAnalysis(A, cam, true, &writer, "", actualEventNumber);
    }
    writer.writeCameraOutput(); // before the analyzers (owners of the bubbles) go away
    for (AnalyzerUnit *A : analyzers)
        delete A;
    return 0;
}

// writer probe for CPU unit tests: per camera a status, a trigger frame and bubbles given as descriptor
// rows (x,y,w,h,area,radius,m00,m10,m01,cx,cy); first row of a bubble = genesis
void abh_writer_probe(const char *outdir, const char *run_number, int frameOffset, int ncams, int event,
                      const int *status, const int *frame0, const int *nbub, const int *ndesc, const double *desc)
{
    OutputWriter w(outdir, run_number, frameOffset, ncams);
    std::vector<std::vector<bubble *>> lists(ncams);
    int bi = 0, di = 0;
    for (int c = 0; c < ncams; ++c) {
        for (int k = 0; k < nbub[c]; ++k, ++bi) {
            bubble *b = nullptr;
            for (int d = 0; d < ndesc[bi]; ++d, ++di) {
                const double *r = desc + 11 * (size_t)di;
                BubbleImageFrame f;
                f.newPosition = cv::Rect((int)r[0], (int)r[1], (int)r[2], (int)r[3]);
                f.ContArea = r[4];
                f.ContRadius = r[5];
                f.moments.m00 = r[6];
                f.moments.m10 = r[7];
                f.moments.m01 = r[8];
                f.MassCentres = cv::Point2f((float)r[9], (float)r[10]);
                if (!b)
                    b = new bubble(f);
                else {
                    b->lockThisIteration = false;
                    *b << f;
                }
            }
            lists[c].push_back(b);
        }
        if (status[c] == 0)
            w.stageCameraOutput(lists[c], c, frame0[c], event);
        else
            w.stageCameraOutputError(c, status[c], event);
    }
    w.writeCameraOutput();
    for (auto &l : lists)
        for (bubble *b : l)
            delete b;
}

int abh_last_trig(void *r) { return ((Run *)r)->trig; }
int abh_last_status(void *r) { return ((Run *)r)->status; }
int abh_last_loc_thres(void *r) { return ((Run *)r)->loc_thres; }
int abh_last_ok(void *r) { return ((Run *)r)->ok; }
const char *abh_last_error(void *r) { return ((Run *)r)->error.c_str(); }
int abh_last_nbubbles(void *r) { return (int)((Run *)r)->bubbles.size(); }
int abh_last_ndesc(void *r, int b) { return (int)((Run *)r)->bubbles[b].desc.size(); }
// out: x,y,w,h,area,radius,m00,m10,m01,cx,cy
void abh_last_desc(void *r, int b, int d, double *out)
{
    const BubbleImageFrame &f = ((Run *)r)->bubbles[b].desc[d];
    out[0] = f.newPosition.x;
    out[1] = f.newPosition.y;
    out[2] = f.newPosition.width;
    out[3] = f.newPosition.height;
    out[4] = f.ContArea;
    out[5] = f.ContRadius;
    out[6] = f.moments.m00;
    out[7] = f.moments.m10;
    out[8] = f.moments.m01;
    out[9] = f.MassCentres.x;
    out[10] = f.MassCentres.y;
}
int abh_last_ndz(void *r, int b) { return (int)((Run *)r)->bubbles[b].dz.size(); }
float abh_last_dz(void *r, int b, int i) { return ((Run *)r)->bubbles[b].dz[i]; }
float abh_last_dzdt(void *r, int b) { return ((Run *)r)->bubbles[b].dzdt; }
float abh_last_drdt(void *r, int b) { return ((Run *)r)->bubbles[b].drdt; }

// host-logic probes for CPU-side unit tests (no GPU needed)
int abh_contours(const uint32_t *idx, int n, int W, int H, int *npts_out, int *xy_out, int cap_contours, int cap_pts)
{
    std::vector<uint32_t> v(idx, idx + n);
    std::vector<std::vector<cv::Point>> cs;
    abub::ContourFinder f;
    f.find(v, W, H, cs);
    int k = 0, used = 0;
    for (auto &c : cs) {
        if (k >= cap_contours || used + (int)c.size() > cap_pts)
            return -1;
        npts_out[k++] = (int)c.size();
        for (auto &p : c) {
            xy_out[2 * used] = p.x;
            xy_out[2 * used + 1] = p.y;
            ++used;
        }
    }
    return k;
}
int abh_binarize_threshold(const uint32_t *hist, int P, int tozero) { return abub::binarizeThresholdFromHist(hist, (size_t)P, tozero); }
float abh_entropy(const uint32_t *hist, int nbins, int P) { return abub::entropyFromHist(hist, nbins, (size_t)P); }
void abh_blob_stats(const int *xy, int n, double *out /*x,y,w,h,area,m00,m10,m01*/)
{
    std::vector<cv::Point> pts;
    for (int i = 0; i < n; ++i)
        pts.push_back(cv::Point(xy[2 * i], xy[2 * i + 1]));
    cv::Rect r = abub::boundingRectOf(pts);
    cv::Moments m = abub::momentsOf(pts);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.width;
    out[3] = r.height;
    out[4] = abub::contourAreaOf(pts);
    out[5] = m.m00;
    out[6] = m.m10;
    out[7] = m.m01;
}
void abh_best_match(const unsigned long long *num, const unsigned long long *wsum2, int rw, int rh, const uint8_t *tmpl,
                    int tw, int th, float *bx, float *by)
{
    cv::Mat t(th, tw, CV_8U);
    std::memcpy(t.data, tmpl, (size_t)tw * th);
    abub::bestMatchFromTerms(num, wsum2, rw, rh, t, *bx, *by);
}
float abh_entropy_frame(const uint8_t *img, int W, int H)
{
    cv::Mat m(H, W, CV_8U);
    std::memcpy(m.data, img, (size_t)W * H);
    return calculateEntropyFrame(m);
}
// significance state machine probe: feeds histograms sequentially
void *abh_sig_new() { return new std::vector<std::vector<int>>(256); }
void abh_sig_free(void *s) { delete (std::vector<std::vector<int>> *)s; }
double abh_sig_eval(void *s, const uint32_t *hist, int P, int store, int tss, int *loc_thres)
{
    return abub::significanceFromHist(*(std::vector<std::vector<int>> *)s, hist, (size_t)P, store != 0, tss, 3, *loc_thres);
}
}
