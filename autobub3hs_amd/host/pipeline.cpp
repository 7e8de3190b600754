// pipeline.cpp -- run-level batched driver: every (event, camera) stack of a run is resident in HBM
// as one slab [E][C][F][H][W]; the GPU work of ALL stacks is issued in a handful of launches per
// stage, the per-event state machines (the same AnalyzerUnit / L3Localizer code as the drop-in path)
// run on host threads in between.  This is what bench.py times as "end-to-end detect".
//
//   stage 1  K2 trigger-only over every frame of every stack        -> [S][F-1][256] histograms
//   stage 2  host: FindTriggerFrame per stack (AnyCamAnalysis loop, AutoBubStart3.cpp:87-110)
//   stage 3  K2 store for the genesis pairs, K3 for the post-trigger frames, Otsu thresholds on the
//            host from the histograms, K4 compaction of all foreground pixels into one list
//   stage 4  host: LocalizeOMatic per stack (contours, blobs, tracking)
//   stacks whose trigger produced no accepted bubble go round again from the next frame.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "AlgorithmTraining/Trainer.hpp"
#include "AnalyzerUnit.hpp"
#include "BubbleLocalizer/L3Localizer.hpp"
#include "ParseFolder/Parser.hpp"
#include "common/CommonParameters.h"
#include "devctx.hpp"
#include "hostlogic.hpp"

namespace abub {
extern bool g_quietAnalyzers;

namespace {

#define HIPOK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess)                                                             \
            throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct PlannedImage {
    int kind; // 0 = D(i; ref) (genesis), 1 = post-trigger image of frame i
    int i, ref;
    int slot; // index into the image / histogram slabs of this round
    int tozero;
    int thr;
    const uint32_t *fg = nullptr; // raster indices of the candidate pixels (value > tozero), grouped on the GPU
    const uint8_t *fgv = nullptr; // their values: the Otsu cut is applied on the host
    uint32_t nfg = 0;
};

// EventData served from the pipeline's batched results
class BatchEventData : public EventData {
public:
    const uint32_t *hists = nullptr; // [F-1][256] of this stack (frame i at (i-1)*256)
    int refOffset = 2;
    const uint32_t *roundHists = nullptr; // [nslots][256] of the current round
    std::vector<PlannedImage> planned;
    int cur = -1;

    bool frameOk(int i) const override { return i >= 0 && i < F; }
    cv::Mat hostFrame(int) const override { return cv::Mat(); }
    const uint32_t *diffHist(int i, int off) override
    {
        if (off != refOffset || i < 1 || i >= F)
            throw std::runtime_error("BatchEventData::diffHist: unplanned request");
        return hists + (size_t)(i - 1) * 256;
    }
    const uint32_t *find(int kind, int i, int ref)
    {
        for (size_t k = 0; k < planned.size(); ++k)
            if (planned[k].kind == kind && planned[k].i == i && (kind == 1 || planned[k].ref == ref)) {
                cur = (int)k;
                return roundHists + (size_t)planned[k].slot * 256;
            }
        throw std::runtime_error("BatchEventData: image was not planned for this round");
    }
    const uint32_t *diffFrame(int i, int ref, cv::Mat *) override { return find(0, i, ref); }
    const uint32_t *diffFrameROI(int, int, cv::Rect, cv::Mat *) override
    {
        throw std::runtime_error("BatchEventData: ROI diff is not available in the batched path");
    }
    const uint32_t *postTrig(int i, cv::Mat *) override { return find(1, i, 0); }
    void foreground(int thr, std::vector<uint32_t> &idx) override
    {
        if (cur < 0 || planned[cur].thr != thr)
            throw std::runtime_error("BatchEventData::foreground: threshold differs from the planned one");
        const PlannedImage &p = planned[cur];
        idx.clear();
        for (uint32_t k = 0; k < p.nfg; ++k)
            if ((int)p.fgv[k] > thr)
                idx.push_back(p.fg[k]);
    }
};

struct BubbleOut {
    std::vector<BubbleImageFrame> desc;
    std::vector<float> dz;
    float dzdt, drdt;
};

struct StackState {
    std::unique_ptr<L3Localizer> analyzer;
    BatchEventData data;
    int staged = 0;
    bool done = false;
    bool localize = false;
    std::string error;
    std::vector<BubbleOut> bubbles;
    int trig = 0, status = 0, loc_thres = 3, ok = 1;
};

template <typename Fn>
void parallelFor(int n, int nthreads, Fn fn)
{
    if (nthreads <= 1 || n <= 1) {
        for (int i = 0; i < n; ++i)
            fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    int nt = std::min(nthreads, n);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&]() {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= n)
                    break;
                fn(i);
            }
        });
    for (auto &t : th)
        t.join();
}

double nowMs()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

class RunPipeline {
public:
    int device, W, H, F, E, C, S, nthreads;
    size_t P;
    std::vector<int> tss;
    std::string maskDir;
    // device scratch
    abub_job *d_jobs1 = nullptr;   // stage-1 job list [S*(F-1)]
    uint32_t *d_hist1 = nullptr;   // [S*(F-1)][256]
    abub_job *d_jobs3 = nullptr;   // per-round jobs [S*11]
    uint32_t *d_hist3 = nullptr;   // [S*11][256]
    uint8_t *d_img = nullptr;      // [S*11][H][W]
    int32_t *d_thr = nullptr;      // [S*11]
    uint32_t *d_pairs = nullptr;   // [cap][2]
    uint32_t *d_count = nullptr;
    uint32_t *d_gscratch = nullptr, *d_goff = nullptr, *d_gidx = nullptr; // grouped list
    uint8_t *d_gval = nullptr;
    uint32_t *h_goff = nullptr, *h_gidx = nullptr;
    uint8_t *h_gval = nullptr;
    uint32_t pairCap = 0;
    // pinned host
    uint32_t *h_hist1 = nullptr, *h_hist3 = nullptr, *h_pairs = nullptr, *h_count = nullptr;
    abub_job *h_jobs3 = nullptr;
    int32_t *h_thr = nullptr;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    std::vector<StackState> stacks;
    std::vector<Trainer *> trainers;
    MemParser parser;
    double tms[8] = {0};
    int rounds = 0;
    uint32_t lastPairs = 0;

    RunPipeline(int device_, int W_, int H_, int F_, int E_, int C_, const int *tss_, int nthreads_, const char *maskdir)
        : device(device_), W(W_), H(H_), F(F_), E(E_), C(C_), S(E_ * C_), nthreads(nthreads_), P((size_t)W_ * H_),
          tss(tss_, tss_ + C_), maskDir(maskdir ? maskdir : "")
    {
        if (W <= 0 || H <= 0 || F <= 0 || E <= 0 || C <= 0)
            throw std::runtime_error("RunPipeline: bad geometry");
        HIPOK(hipSetDevice(device));
        const size_t n1 = (size_t)S * std::max(F - 1, 1), n3 = (size_t)S * (NumFramesBubbleTrack + 1);
        pairCap = 8u << 20;
        HIPOK(hipMalloc((void **)&d_jobs1, n1 * sizeof(abub_job)));
        HIPOK(hipMalloc((void **)&d_hist1, n1 * 1024));
        HIPOK(hipMalloc((void **)&d_jobs3, n3 * sizeof(abub_job)));
        HIPOK(hipMalloc((void **)&d_hist3, n3 * 1024));
        HIPOK(hipMalloc((void **)&d_img, n3 * P));
        HIPOK(hipMalloc((void **)&d_thr, n3 * sizeof(int32_t)));
        HIPOK(hipMalloc((void **)&d_pairs, (size_t)pairCap * 8));
        HIPOK(hipMalloc((void **)&d_count, sizeof(uint32_t)));
        HIPOK(hipMalloc((void **)&d_gscratch, 2 * n3 * sizeof(uint32_t)));
        HIPOK(hipMalloc((void **)&d_goff, (n3 + 1) * sizeof(uint32_t)));
        HIPOK(hipMalloc((void **)&d_gidx, (size_t)pairCap * 4));
        HIPOK(hipMalloc((void **)&d_gval, (size_t)pairCap));
        HIPOK(hipHostMalloc((void **)&h_goff, (n3 + 1) * sizeof(uint32_t), hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_gidx, (size_t)pairCap * 4, hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_gval, (size_t)pairCap, hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_hist1, n1 * 1024, hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_hist3, n3 * 1024, hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_pairs, (size_t)pairCap * 8, hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_count, sizeof(uint32_t), hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_jobs3, n3 * sizeof(abub_job), hipHostMallocDefault));
        HIPOK(hipHostMalloc((void **)&h_thr, n3 * sizeof(int32_t), hipHostMallocDefault));
        // stage-1 jobs: FindTriggerFrame's pairing, ref = max(i - off, 0) with off = 1 when the model was
        // trained on fewer than 6 frames (AnalyzerUnit.cpp:185-188)
        std::vector<abub_job> j1(n1);
        for (int s = 0; s < S; ++s) {
            const int c = s % C, off = tss[c] < 6 ? 1 : 2;
            for (int i = 1; i < F; ++i) {
                abub_job &j = j1[(size_t)s * (F - 1) + (i - 1)];
                j.cur = (uint32_t)(s * F + i);
                j.ref = (uint32_t)(s * F + std::max(i - off, 0));
                j.model = (uint32_t)c;
                j.out = (uint32_t)((size_t)s * (F - 1) + (i - 1));
            }
        }
        if (F > 1)
            HIPOK(hipMemcpy(d_jobs1, j1.data(), n1 * sizeof(abub_job), hipMemcpyHostToDevice));
        // frame names only: the images live in HBM
        for (int c = 0; c < C; ++c) {
            Trainer *t = new Trainer(c, {}, "", "cam%d_image%u.png", "", parser.clone(), false);
            t->TrainingSetSize = tss[c];
            t->ModelId = 0;
            trainers.push_back(t);
        }
        std::vector<cv::Mat> none((size_t)F);
        for (int e = 0; e < E; ++e)
            for (int c = 0; c < C; ++c)
                parser.AddFrames(std::to_string(e), c, none, 10000); // 5-digit numbers: lexicographic == numeric
    }

    ~RunPipeline()
    {
        (void)hipSetDevice(device);
        (void)hipFree(d_jobs1);
        (void)hipFree(d_hist1);
        (void)hipFree(d_jobs3);
        (void)hipFree(d_hist3);
        (void)hipFree(d_img);
        (void)hipFree(d_thr);
        (void)hipFree(d_pairs);
        (void)hipFree(d_count);
        (void)hipFree(d_gscratch);
        (void)hipFree(d_goff);
        (void)hipFree(d_gidx);
        (void)hipFree(d_gval);
        (void)hipHostFree(h_goff);
        (void)hipHostFree(h_gidx);
        (void)hipHostFree(h_gval);
        (void)hipHostFree(h_hist1);
        (void)hipHostFree(h_hist3);
        (void)hipHostFree(h_pairs);
        (void)hipHostFree(h_count);
        (void)hipHostFree(h_jobs3);
        (void)hipHostFree(h_thr);
        for (Trainer *t : trainers)
            delete t;
    }

    void run(const uint8_t *d_frames, const uint8_t *d_mu, const uint8_t *d_sigma6, hipStream_t st)
    {
        HIPOK(hipSetDevice(device));
        g_quietAnalyzers = true;
        stream = st;
        std::fill(tms, tms + 8, 0.0);
        rounds = 0;
        double t0 = nowMs();
        // ---- stage 1 -------------------------------------------------------------------------
        const int n1 = S * (F - 1);
        if (n1 > 0) {
            check(abub_diff_hist_dev(d_frames, d_sigma6, d_jobs1, n1, W, H, d_hist1, nullptr, 0, stream), "stage1 K2");
            HIPOK(hipMemcpyAsync(h_hist1, d_hist1, (size_t)n1 * 1024, hipMemcpyDeviceToHost, stream));
        }
        // analyzers are (re)built while the GPU works
        stacks.clear();
        stacks.resize(S);
        parallelFor(S, nthreads, [&](int s) {
            StackState &st_ = stacks[s];
            const int e = s / C, c = s % C;
            Trainer *t = trainers[c];
            st_.analyzer.reset(new L3Localizer(std::to_string(e), "", c, true, &t, maskDir, parser.clone()));
            st_.data.F = F;
            st_.data.W = W;
            st_.data.H = H;
            st_.data.refOffset = tss[c] < 6 ? 1 : 2;
            st_.data.hists = h_hist1 + (size_t)s * (F - 1) * 256;
            st_.analyzer->AttachEventData(&st_.data);
        });
        HIPOK(hipStreamSynchronize(stream));
        tms[0] = nowMs() - t0;

        std::vector<int> pending(S);
        for (int s = 0; s < S; ++s)
            pending[s] = s;
        while (!pending.empty()) {
            ++rounds;
            // ---- stage 2: trigger search + plan ------------------------------------------------
            double t2 = nowMs();
            parallelFor((int)pending.size(), nthreads, [&](int k) { triggerAndPlan(stacks[pending[k]]); });
            tms[1] += nowMs() - t2;
            // ---- stage 3: batched images, thresholds, foreground ---------------------------------
            double t3 = nowMs();
            std::vector<int> loc;
            for (int s : pending)
                if (stacks[s].localize)
                    loc.push_back(s);
            if (!loc.empty())
                batchImages(loc, d_frames, d_mu, d_sigma6);
            tms[2] += nowMs() - t3;
            // ---- stage 4: localize + track -------------------------------------------------------
            double t4 = nowMs();
            parallelFor((int)loc.size(), nthreads, [&](int k) { localize(stacks[loc[k]]); });
            tms[3] += nowMs() - t4;
            std::vector<int> next;
            for (int s : pending)
                if (!stacks[s].done)
                    next.push_back(s);
            pending.swap(next);
        }
        // results out, analyzers released
        parallelFor(S, nthreads, [&](int s) {
            StackState &st_ = stacks[s];
            AnalyzerUnit *A = st_.analyzer.get();
            st_.trig = A->MatTrigFrame;
            st_.status = A->TriggerFrameIdentificationStatus;
            st_.loc_thres = A->loc_thres;
            st_.ok = A->okToProceed;
            for (bubble *b : A->BubbleList) {
                BubbleOut o;
                o.desc = b->KnownDescriptors;
                o.dz = b->dz;
                o.dzdt = b->dZdT();
                o.drdt = b->dRdT();
                st_.bubbles.push_back(std::move(o));
            }
            st_.analyzer.reset();
        });
        tms[4] = nowMs() - t0;
    }

private:
    // AnyCamAnalysis body up to LocalizeOMatic (AutoBubStart3.cpp:87-107)
    void triggerAndPlan(StackState &st_)
    {
        st_.localize = false;
        AnalyzerUnit *A = st_.analyzer.get();
        try {
            A->FindTriggerFrame(true, A->MatTrigFrame + 1);
            if (!A->okToProceed) {
                st_.staged = A->TriggerFrameIdentificationStatus;
                st_.done = true;
                return;
            }
            if (F <= 5) { // LocalizeOMatic refuses (L3Localizer.cpp:889) -> -8
                A->LocalizeOMatic("");
                st_.staged = -8;
                st_.done = true;
                return;
            }
            const int t = A->MatTrigFrame;
            const int off = A->TrainedData->TrainingSetSize < 6 ? 1 : 2;
            st_.data.planned.clear();
            st_.data.cur = -1;
            PlannedImage g;
            g.kind = 0;
            g.i = t;
            g.ref = std::max(t - off, 0);
            g.tozero = A->loc_thres;
            g.slot = -1;
            g.thr = 0;
            st_.data.planned.push_back(g);
            const int last = (t < 29) ? NumFramesBubbleTrack : (39 - t);
            for (int k = 1; k <= last; ++k) {
                if (t + k >= F)
                    break;
                PlannedImage p;
                p.kind = 1;
                p.i = t + k;
                p.ref = 0;
                p.tozero = 3;
                p.slot = -1;
                p.thr = 0;
                st_.data.planned.push_back(p);
            }
            st_.localize = true;
        } catch (std::exception &e) {
            st_.error = e.what();
            st_.staged = -6;
            st_.done = true;
        }
    }

    void batchImages(const std::vector<int> &loc, const uint8_t *d_frames, const uint8_t *d_mu, const uint8_t *d_sigma6)
    {
        // slots: all genesis images first (K2 store), then all post-trigger images (K3)
        int nd = 0, np = 0;
        for (int s : loc)
            for (PlannedImage &p : stacks[s].data.planned)
                (p.kind == 0 ? nd : np)++;
        int di = 0, pi = nd;
        for (int s : loc) {
            const int c = s % C;
            for (PlannedImage &p : stacks[s].data.planned) {
                p.slot = p.kind == 0 ? di++ : pi++;
                abub_job &j = h_jobs3[p.slot];
                j.cur = (uint32_t)(s * F + p.i);
                j.ref = (uint32_t)(s * F + p.ref);
                j.model = (uint32_t)c;
                j.out = (uint32_t)(p.kind == 0 ? p.slot : p.slot - nd);
            }
        }
        const int nimg = nd + np;
        double ta = nowMs();
        std::vector<PlannedImage *> bySlot((size_t)nimg);
        for (int s : loc) {
            stacks[s].data.roundHists = h_hist3;
            for (PlannedImage &p : stacks[s].data.planned) {
                bySlot[p.slot] = &p;
                h_thr[p.slot] = p.tozero; // candidate cut = TOZERO threshold, known before the launch
            }
        }
        HIPOK(hipMemcpyAsync(d_jobs3, h_jobs3, (size_t)nimg * sizeof(abub_job), hipMemcpyHostToDevice, stream));
        HIPOK(hipMemcpyAsync(d_thr, h_thr, (size_t)nimg * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIPOK(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
        const bool fused = abub_fast_path(W) != 0;
        if (fused) {
            // images are never materialised: histogram + candidate list come out of the same pass
            check(abub_diff_hist_compact_dev(d_frames, d_sigma6, d_jobs3, nd, W, H, d_hist3, nullptr, d_thr, d_pairs,
                                             pairCap, d_count, 0, stream),
                  "stage3 K2 compact");
            if (np > 0)
                check(abub_posttrig_compact_dev(d_frames, d_mu, d_sigma6, d_jobs3 + nd, np, W, H,
                                                d_hist3 + (size_t)nd * 256, nullptr, d_thr + nd, d_pairs, pairCap,
                                                d_count, (uint32_t)nd, stream),
                      "stage3 K3 compact");
        } else {
            check(abub_diff_hist_dev(d_frames, d_sigma6, d_jobs3, nd, W, H, d_hist3, d_img, 0, stream), "stage3 K2 store");
            if (np > 0)
                check(abub_posttrig_dev(d_frames, d_mu, d_sigma6, d_jobs3 + nd, np, W, H, d_hist3 + (size_t)nd * 256,
                                        d_img + (size_t)nd * P, stream),
                      "stage3 K3");
            check(abub_fg_compact_pairs_dev(d_img, nimg, W, H, d_thr, d_pairs, pairCap, d_count, stream), "stage3 K4");
        }
        // group the list by image on the device; the host gets contiguous runs and never re-buckets
        check(abub_pairs_group_dev(d_pairs, d_count, pairCap, nimg, d_gscratch, d_goff, d_gidx, d_gval, stream),
              "stage3 group");
        HIPOK(hipMemcpyAsync(h_hist3, d_hist3, (size_t)nimg * 1024, hipMemcpyDeviceToHost, stream));
        HIPOK(hipMemcpyAsync(h_count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPOK(hipMemcpyAsync(h_goff, d_goff, (size_t)(nimg + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPOK(hipStreamSynchronize(stream));
        tms[5] += nowMs() - ta; // launches + kernels + hist/count D2H
        ta = nowMs();
        const uint32_t cnt = *h_count;
        lastPairs = cnt;
        if (cnt > pairCap)
            throw std::runtime_error("RunPipeline: foreground list overflow (dense foreground in too many images)");
        if (cnt) {
            HIPOK(hipMemcpyAsync(h_gidx, d_gidx, (size_t)cnt * 4, hipMemcpyDeviceToHost, stream));
            HIPOK(hipMemcpyAsync(h_gval, d_gval, (size_t)cnt, hipMemcpyDeviceToHost, stream));
        }
        // thresholds (TOZERO + Otsu) on the host from the histograms, while the list travels
        parallelFor(nimg, nthreads, [&](int k) {
            PlannedImage *p = bySlot[k];
            p->thr = binarizeThresholdFromHist(h_hist3 + (size_t)k * 256, P, p->tozero);
            p->fg = h_gidx + h_goff[k];
            p->fgv = h_gval + h_goff[k];
            p->nfg = h_goff[k + 1] - h_goff[k];
        });
        HIPOK(hipStreamSynchronize(stream));
        tms[6] += nowMs() - ta; // list D2H (+ thresholds)
    }

    // AnyCamAnalysis body from LocalizeOMatic on (AutoBubStart3.cpp:94-110)
    void localize(StackState &st_)
    {
        AnalyzerUnit *A = st_.analyzer.get();
        try {
            A->LocalizeOMatic("");
            if (!A->okToProceed) {
                st_.staged = -8;
                st_.done = true;
                return;
            }
            st_.staged = A->BubbleList.empty() ? -1 : 0;
            st_.done = !A->BubbleList.empty(); // no accepted bubble: search on from the next frame
        } catch (std::exception &e) {
            st_.error = e.what();
            st_.staged = -6;
            st_.done = true;
        }
    }
};

} // namespace abub

// ---- C surface (bench.py / tests) ---------------------------------------------------------------
extern "C" {

void *abh_pipe_new(int device, int W, int H, int F, int E, int C, const int *tss, int nthreads, const char *maskdir)
{
    try {
        return new abub::RunPipeline(device, W, H, F, E, C, tss, nthreads, maskdir);
    } catch (std::exception &e) {
        fprintf(stderr, "abh_pipe_new: %s\n", e.what());
        return nullptr;
    }
}

void abh_pipe_free(void *p) { delete (abub::RunPipeline *)p; }

static thread_local std::string g_pipeErr;
const char *abh_pipe_error() { return g_pipeErr.c_str(); }

int abh_pipe_run(void *p, const void *frames_dev, const void *mu_dev, const void *sigma6_dev, void *stream)
{
    try {
        ((abub::RunPipeline *)p)->run((const uint8_t *)frames_dev, (const uint8_t *)mu_dev, (const uint8_t *)sigma6_dev,
                                      (hipStream_t)stream);
        return 0;
    } catch (std::exception &e) {
        g_pipeErr = e.what();
        return -1;
    }
}

// out: staged, trig, status, loc_thres, ok, nbubbles
void abh_pipe_result(void *p, int s, int *out)
{
    abub::StackState &st = ((abub::RunPipeline *)p)->stacks[s];
    out[0] = st.staged;
    out[1] = st.trig;
    out[2] = st.status;
    out[3] = st.loc_thres;
    out[4] = st.ok;
    out[5] = (int)st.bubbles.size();
}
int abh_pipe_ndesc(void *p, int s, int b) { return (int)((abub::RunPipeline *)p)->stacks[s].bubbles[b].desc.size(); }
void abh_pipe_desc(void *p, int s, int b, int d, double *out)
{
    const BubbleImageFrame &f = ((abub::RunPipeline *)p)->stacks[s].bubbles[b].desc[d];
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
float abh_pipe_dzdt(void *p, int s, int b) { return ((abub::RunPipeline *)p)->stacks[s].bubbles[b].dzdt; }
float abh_pipe_drdt(void *p, int s, int b) { return ((abub::RunPipeline *)p)->stacks[s].bubbles[b].drdt; }
const char *abh_pipe_stack_error(void *p, int s) { return ((abub::RunPipeline *)p)->stacks[s].error.c_str(); }
// out[0..4]: stage1, stage2, stage3, stage4, total (ms) of the last run; returns the number of rounds
int abh_pipe_timing(void *p, double *out)
{
    abub::RunPipeline *r = (abub::RunPipeline *)p;
    for (int k = 0; k < 8; ++k)
        out[k] = r->tms[k];
    out[8] = r->lastPairs;
    return r->rounds;
}
}
