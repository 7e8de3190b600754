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
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
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
#include "PICOFormatWriter/PICOFormatWriterV4.hpp"
#include "devctx.hpp"
#include "hostlogic.hpp"
#include "pngwalk.hpp"
#include "runbatch.hpp"

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

// thrown by the batched provider when the trigger search asks for a frame whose block has not been evaluated yet: the
// pipeline evaluates that block for the stack and runs the search again (see RunPipeline::runGroup)
struct NeedMoreFrames : public std::runtime_error {
    NeedMoreFrames() : std::runtime_error("trigger search needs the next block of frames") {}
};

// EventData served from the pipeline's batched results
class BatchEventData : public EventData {
public:
    // The trigger search's histograms arrive in blocks of frames: block k covers frames [bstart[k], bstart[k + 1]) and
    // bh[k] points at its histograms once that block has been evaluated for this stack (frame i at (i - bstart[k]) * 256).
    static constexpr int MAXB = 32;
    int nblocks = 0;
    int bstart[MAXB + 1] = {0};
    const uint32_t *bh[MAXB] = {nullptr};
    // inc[k] (optional): per frame of block k, 1 while the frame's histogram is not final -- its dense rows were handed over
    // by the bound scan and the row machine has not run on them yet (deferred pieces); cleared when they are completed
    uint8_t *inc[MAXB] = {nullptr};
    int fetchOf[MAXB] = {0}, slotOf[MAXB] = {0}; // which launch of block k served this stack, and its position in it
    int needBlock = -1;    // set when diffHist() throws NeedMoreFrames ...
    bool needPieces = false; // ... true: block needBlock is there, but frame needFrame (and later ones) must be completed
    int needFrame = 0;
    int refOffset = 2;
    const uint32_t *roundHists = nullptr; // [nslots][256] of the current round
    std::vector<PlannedImage> planned;
    int cur = -1;
    const uint8_t *ok = nullptr; // per-frame "decoded" flags of a stack that came from disk (NULL: all good)

    bool frameOk(int i) const override { return i >= 0 && i < F && (!ok || ok[i]); }
    cv::Mat hostFrame(int) const override { return cv::Mat(); }
    const uint32_t *diffHist(int i, int off) override
    {
        if (off != refOffset || i < 1 || i >= F)
            throw std::runtime_error("BatchEventData::diffHist: unplanned request");
        int k = 0;
        while (k + 1 < nblocks && i >= bstart[k + 1])
            ++k;
        if (!bh[k]) {
            needBlock = k;
            needPieces = false;
            throw NeedMoreFrames();
        }
        if (inc[k] && inc[k][i - bstart[k]]) {
            needBlock = k;
            needPieces = true;
            needFrame = i;
            throw NeedMoreFrames();
        }
        return bh[k] + (size_t)(i - bstart[k]) * 256;
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
    void matchTerms(int, const cv::Mat &, std::vector<unsigned long long> &, std::vector<unsigned long long> &) override
    {
        throw NeedsDropInPath("bellows veto requested in the batched path");
    }
    const uint32_t *subtractFromCurrent(const cv::Mat &) override
    {
        throw NeedsDropInPath("bellows veto requested in the batched path");
    }
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
    bool dropIn = false; // must be re-run through the one-at-a-time path (bellows veto)
    bool needMore = false; // the trigger search stopped at a frame block that is not evaluated yet (data.needBlock)
    std::string error;
    std::vector<BubbleOut> bubbles;
    int trig = 0, status = 0, loc_thres = 3, ok = 1;
};

} // namespace

// What a batched driver knows about one (event, camera) stack that came from a Parser: the reference's event id,
// the real frame names in the Parser's order and which of them decoded (host/runbatch: RunBatched)
struct StackMeta {
    std::string eventID;
    std::vector<std::string> names;
    std::vector<uint8_t> ok;
};

namespace {

// Persistent worker pool shared by all stack groups: a group that is waiting for the GPU lends its
// threads to the groups that are in their host stages (no per-call thread creation either).
class WorkerPool {
public:
    explicit WorkerPool(int nthreads)
    {
        for (int t = 0; t < nthreads; ++t)
            workers.emplace_back([this]() { workerLoop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : workers)
            t.join();
    }
    // runs fn(0..n-1); the caller takes part, returns when every index is done
    void parallelFor(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0)
            return;
        if (n == 1 || workers.empty()) {
            for (int i = 0; i < n; ++i)
                fn(i);
            return;
        }
        auto batch = std::make_shared<Batch>();
        batch->n = n;
        batch->fn = &fn;
        batch->remaining = n;
        {
            std::lock_guard<std::mutex> lock(mu);
            batches.push_back(batch);
        }
        cv.notify_all();
        runBatch(*batch); // help
        std::unique_lock<std::mutex> lock(mu);
        batch->done.wait(lock, [&] { return batch->remaining == 0; });
        for (auto it = batches.begin(); it != batches.end(); ++it)
            if (it->get() == batch.get()) {
                batches.erase(it);
                break;
            }
    }

private:
    struct Batch {
        int n = 0;
        const std::function<void(int)> *fn = nullptr;
        std::atomic<int> next{0};
        int remaining = 0; // guarded by mu
        std::condition_variable done;
    };
    void runBatch(Batch &b)
    {
        int finished = 0;
        for (;;) {
            const int i = b.next.fetch_add(1);
            if (i >= b.n)
                break;
            (*b.fn)(i);
            ++finished;
        }
        if (finished) {
            std::lock_guard<std::mutex> lock(mu);
            b.remaining -= finished;
            if (b.remaining == 0)
                b.done.notify_all();
        }
    }
    void workerLoop()
    {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            std::shared_ptr<Batch> pick;
            for (auto &b : batches)
                if (b->next.load() < b->n) {
                    pick = b;
                    break;
                }
            if (pick) {
                lock.unlock();
                runBatch(*pick);
                lock.lock();
                continue;
            }
            if (stop)
                return;
            cv.wait(lock);
        }
    }
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Batch>> batches;
    std::vector<std::thread> workers;
    bool stop = false;
};

double nowMs()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

// One slice of the run with its own stream and scratch: groups run concurrently on host threads, so the
// GPU work of one group overlaps the host state machines of another (no group waits on another).
struct Group {
    int s0 = 0, s1 = 0; // stacks [s0, s1)
    hipStream_t stream = nullptr;
    hipEvent_t stage1Done = nullptr, kernelsDone = nullptr;
    // trigger search, per frame block: job lists and histograms (capacity: every stack of the group once)
    std::vector<abub_job *> d_jobsB, h_jobsB;
    std::vector<uint32_t *> d_histB, h_histB;
    std::vector<int> usedB; // stacks already served per block in this run (each stack fetches a block at most once)
    // deferred pieces, per block: handed-over row ranges of every launch of the block, per-launch counters, per-job
    // "incomplete" / "wanted" flags
    struct Fetch {
        int first = 0, n = 0;   // stacks [first, first + n) of the block's buffers
        size_t pieceOff = 0;    // its range of the block's piece list
        bool deferred = false;
    };
    std::vector<std::vector<Fetch>> fetches;
    std::vector<void *> d_piecesB;
    std::vector<uint32_t *> d_pcountB;
    std::vector<uint8_t *> d_incB, h_incB, d_wantB, h_wantB;
    std::vector<size_t> pieceCapB, pieceUsedB;
    hipEvent_t blockDone = nullptr;
    abub_job *d_jobs3 = nullptr;
    uint32_t *d_hist3 = nullptr;
    uint8_t *d_img = nullptr;
    int32_t *d_thr = nullptr;
    uint32_t *d_pairs = nullptr, *d_count = nullptr, *d_gscratch = nullptr, *d_goff = nullptr, *d_gidx = nullptr;
    uint8_t *d_gval = nullptr;
    uint32_t *h_hist3 = nullptr, *h_count = nullptr, *h_goff = nullptr, *h_gidx = nullptr;
    uint8_t *h_gval = nullptr;
    abub_job *h_jobs3 = nullptr;
    int32_t *h_thr = nullptr;
    uint32_t pairCap = 0;
    int nthreads = 1;
    double tms[8] = {0};
    int rounds = 0;
    uint32_t lastPairs = 0;
    std::string error;
};

class RunPipeline {
public:
    int device, W, H, F, E, C, S, nthreads, ngroups;
    size_t P;
    std::vector<int> tss;
    std::string maskDir;
    std::vector<Group> groups;
    std::unique_ptr<WorkerPool> pool;
    std::vector<int> blocks;            // frame blocks of the trigger search: block k = frames [blocks[k], blocks[k + 1])
    bool deferPieces = false;           // trigger search: dense frames' rows are evaluated on demand (see the constructor)
    long long jobsCompleted = 0;        // ... jobs of the last run that were completed that way
    long long jobsLaunched = 0;         // trigger-search jobs of the last run (F - 1 per stack when nothing is lazy)
    int dropIns = 0;                    // stacks of the last run that went through the one-at-a-time path (bellows veto)
    hipStream_t stage1Stream = nullptr; // all trigger-search launches, in group order (see run())
    int chainStride = 0;                // FindTriggerFrame's frame offset when every camera shares it, else 0
    bool ordered = true;                // localisation kernels queue on stage1Stream too (see batchImages())
    std::vector<void *> devAllocs, hostAllocs;
    std::mutex allocMu, launchMu;
    std::vector<StackState> stacks;
    std::vector<Trainer *> trainers;
    MemParser parser;
    // optional (runs ingested from a Parser): real ids / names / decode flags per stack; a stack may be shorter than F
    std::vector<StackMeta> meta;
    MemParser metaParser;
    void setStackMeta(std::vector<StackMeta> &&m)
    {
        if ((int)m.size() != S)
            throw std::runtime_error("RunPipeline::setStackMeta: one entry per stack expected");
        meta = std::move(m);
        metaParser = MemParser();
        for (int s = 0; s < S; ++s) {
            if ((int)meta[s].names.size() > F || meta[s].ok.size() != meta[s].names.size())
                throw std::runtime_error("RunPipeline::setStackMeta: stack longer than the pipeline's frame count");
            metaParser.AddNamedFrames(meta[s].eventID, s % C, meta[s].names);
        }
    }
    // trigger-search job of frame i of stack s (FindTriggerFrame's pairing: ref = max(i - off, 0), off = 1 when the model
    // was trained on fewer than 6 frames, AnalyzerUnit.cpp:185-188).  Frames a shorter stack does not have are replaced
    // by its last one on both sides (D = 0: a quiet job that keeps the chain structure the scan relies on).
    abub_job triggerJob(int s, int i, uint32_t out) const
    {
        const int c = s % C, off = tss[c] < 6 ? 1 : 2;
        const int Fs = meta.empty() ? F : (int)meta[s].names.size();
        const int last = std::max(Fs - 1, 0);
        abub_job j;
        j.cur = (uint32_t)(s * F + std::min(i, last));
        j.ref = (uint32_t)(s * F + std::min(std::max(i - off, 0), last));
        j.model = (uint32_t)c;
        j.out = out;
        return j;
    }
    // after the launch of block k has been waited for: the stack at position q of fetch `f` gets its histograms (and flags)
    void bindBlock(Group &G, StackState &st_, int k, int f, int q)
    {
        const Group::Fetch &fe = G.fetches[k][f];
        const size_t blen = (size_t)(blocks[k + 1] - blocks[k]), slot = (size_t)fe.first + q;
        st_.data.bh[k] = G.h_histB[k] + slot * blen * 256;
        st_.data.inc[k] = fe.deferred ? G.h_incB[k] + slot * blen : nullptr;
        st_.data.fetchOf[k] = f;
        st_.data.slotOf[k] = q;
    }
    // K2 over block k for the listed stacks (all of them on the same block): jobs -> device, launch, histograms -> host.
    // Returns the first slot (in stacks) of the block's histogram buffer the results go to.
    int launchBlock(Group &G, int k, const std::vector<int> &list, const uint8_t *d_frames, const uint8_t *d_sigma6,
                    hipStream_t stream)
    {
        const int a = blocks[k], blen = blocks[k + 1] - blocks[k];
        const int first = G.usedB[k], n = (int)list.size();
        if (blen <= 0 || n == 0)
            return first;
        if (first + n > G.s1 - G.s0)
            throw std::runtime_error("RunPipeline: a frame block was requested twice for one stack");
        abub_job *hj = G.h_jobsB[k] + (size_t)first * blen;
        for (int q = 0; q < n; ++q)
            for (int i = 0; i < blen; ++i)
                hj[(size_t)q * blen + i] = triggerJob(list[q], a + i, (uint32_t)((size_t)q * blen + i));
        abub_job *dj = G.d_jobsB[k] + (size_t)first * blen;
        uint32_t *dh = G.d_histB[k] + (size_t)first * blen * 256, *hh = G.h_histB[k] + (size_t)first * blen * 256;
        const int nj = n * blen;
        HIPOK(hipMemcpyAsync(dj, hj, (size_t)nj * sizeof(abub_job), hipMemcpyHostToDevice, stream));
        Group::Fetch fe;
        fe.first = first;
        fe.n = n;
        const size_t pcap = deferPieces ? abub_k2_pieces_cap(nj, W, H) : 0;
        if (deferPieces && G.fetches[k].size() < 64 && G.pieceUsedB[k] + pcap <= G.pieceCapB[k]) {
            // the scan alone: dense frames' rows go to this launch's range of the block's piece list, their jobs are flagged
            fe.deferred = true;
            fe.pieceOff = G.pieceUsedB[k];
            G.pieceUsedB[k] += pcap;
            uint8_t *dinc = G.d_incB[k] + (size_t)first * blen, *hinc = G.h_incB[k] + (size_t)first * blen;
            check(abub_diff_hist_chained_deferred_dev(d_frames, d_sigma6, dj, nj, W, H, dh, blen, chainStride,
                                                      (uint64_t *)G.d_piecesB[k] + fe.pieceOff, (uint32_t)pcap,
                                                      G.d_pcountB[k] + G.fetches[k].size(), dinc, stream),
                  "trigger search K2 (deferred pieces)");
            HIPOK(hipMemcpyAsync(hinc, dinc, (size_t)nj, hipMemcpyDeviceToHost, stream));
        } else {
            // all cameras on the same frame offset: per stack the jobs are a chain (job i refs the cur frame of job i - off)
            // and the scan loads every frame row once for both of its jobs
            check(chainStride > 0 ? abub_diff_hist_chained_dev(d_frames, d_sigma6, dj, nj, W, H, dh, blen, chainStride, stream)
                                  : abub_diff_hist_dev(d_frames, d_sigma6, dj, nj, W, H, dh, nullptr, 0, stream),
                  "trigger search K2");
        }
        HIPOK(hipMemcpyAsync(hh, dh, (size_t)nj * 1024, hipMemcpyDeviceToHost, stream));
        G.fetches[k].push_back(fe);
        G.usedB[k] = first + n;
        jobsLaunched += nj;
        return first;
    }
    double tms[8] = {0};
    int rounds = 0;
    uint32_t lastPairs = 0;

    template <typename T>
    T *dalloc(size_t n)
    {
        void *p = nullptr;
        HIPOK(hipMalloc(&p, n * sizeof(T) + 256));
        std::lock_guard<std::mutex> lock(allocMu); // group threads may grow their lists concurrently
        devAllocs.push_back(p);
        return (T *)p;
    }
    template <typename T>
    T *halloc(size_t n)
    {
        void *p = nullptr;
        HIPOK(hipHostMalloc(&p, n * sizeof(T) + 256, hipHostMallocDefault));
        std::lock_guard<std::mutex> lock(allocMu);
        hostAllocs.push_back(p);
        return (T *)p;
    }

    RunPipeline(int device_, int W_, int H_, int F_, int E_, int C_, const int *tss_, int nthreads_, const char *maskdir)
        : device(device_), W(W_), H(H_), F(F_), E(E_), C(C_), S(E_ * C_), nthreads(nthreads_), P((size_t)W_ * H_),
          tss(tss_, tss_ + C_), maskDir(maskdir ? maskdir : "")
    {
        try {
            if (W <= 0 || H <= 0 || F <= 0 || E <= 0 || C <= 0)
                throw std::runtime_error("RunPipeline: bad geometry");
            HIPOK(hipSetDevice(device));
            const char *eg = getenv("ABUB_PIPE_GROUPS");
            ngroups = eg ? atoi(eg) : 1; // >1 overlaps host stages of one group with the GPU work of the next
            const char *eo = getenv("ABUB_PIPE_ORDERED");
            ordered = eo ? atoi(eo) != 0 : true;
            int prLow = 0, prHigh = 0; // (numerically lower = higher priority)
            HIPOK(hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
            HIPOK(hipStreamCreateWithPriority(&stage1Stream, hipStreamNonBlocking, prLow));
            if (ngroups < 1)
                ngroups = 1;
            if (ngroups > S)
                ngroups = S;
            const int K = NumFramesBubbleTrack + 1;
            chainStride = tss[0] < 6 ? 1 : 2;
            for (int c = 1; c < C; ++c)
                if ((tss[c] < 6 ? 1 : 2) != chainStride)
                    chainStride = 0;
            groups.resize(ngroups);
            pool.reset(new WorkerPool(std::max(0, nthreads - ngroups))); // the group driver threads take part too
            // Frame blocks of the trigger search.  The reference walks the frames in order and stops at the trigger
            // (AnalyzerUnit.cpp:191, break at :307); it never differences the frames behind it unless the localizer finds no
            // bubble and the search goes on (AutoBubStart3.cpp:87-110).  So the histograms are produced block by block: block 0
            // for every stack up front, later blocks only for the stacks whose search reaches them.  ABUB_PIPE_LAZY=0: one
            // block (every frame of every stack up front, what round 2 did).
            {
                const char *el = getenv("ABUB_PIPE_LAZY");
                const bool lazy = el ? atoi(el) != 0 : true;
                const char *e0 = getenv("ABUB_PIPE_BLOCK0"), *e1 = getenv("ABUB_PIPE_BLOCK");
                // first block: up to the frame the cameras' own trigger puts the bubble at (the middle of the stack) plus the
                // two look-ahead frames and a margin; then blocks of about a fifth of the stack
                int first = e0 && atoi(e0) > 0 ? atoi(e0) : F / 2 + 4, step = e1 && atoi(e1) > 0 ? atoi(e1) : std::max(4, F / 5);
                blocks.clear();
                blocks.push_back(1);
                if (lazy && F > 8)
                    for (int b = std::min(first + 1, F); b < F && (int)blocks.size() < BatchEventData::MAXB; b += step)
                        blocks.push_back(b);
                blocks.push_back(std::max(F, 1)); // block k = frames [blocks[k], blocks[k + 1])
                // Deferred pieces: inside a block the bound scan still covers every frame, but the row machine runs only on the
                // dense frames a search actually reaches (ABUB_PIPE_DEFER=0: at once, for every frame of the block).
                const char *ed = getenv("ABUB_PIPE_DEFER");
                deferPieces = (ed ? atoi(ed) != 0 : true) && chainStride > 0 && abub_fast_path(W) != 0;
            }
            const int nB = (int)blocks.size() - 1;
            for (int g = 0; g < ngroups; ++g) {
                Group &G = groups[g];
                G.s0 = (int)((long long)S * g / ngroups);
                G.s1 = (int)((long long)S * (g + 1) / ngroups);
                const size_t ns = (size_t)(G.s1 - G.s0), n3 = ns * K;
                G.nthreads = std::max(1, nthreads / ngroups);
                const char *ec = getenv("ABUB_PIPE_PAIRCAP"); // initial candidate-list capacity (grows on demand)
                G.pairCap = ec && atoi(ec) > 0 ? (uint32_t)atoi(ec) : (8u << 20) / ngroups;
                // the short localisation launches of a finished group must not queue behind the next group's
                // chip-filling trigger search
                HIPOK(hipStreamCreateWithPriority(&G.stream, hipStreamNonBlocking, prHigh));
                HIPOK(hipEventCreateWithFlags(&G.stage1Done, hipEventDisableTiming));
                HIPOK(hipEventCreateWithFlags(&G.kernelsDone, hipEventDisableTiming));
                HIPOK(hipEventCreateWithFlags(&G.blockDone, hipEventDisableTiming));
                G.usedB.assign(nB, 0);
                for (int k = 0; k < nB; ++k) {
                    const size_t nj = ns * (size_t)std::max(blocks[k + 1] - blocks[k], 1);
                    G.d_jobsB.push_back(dalloc<abub_job>(nj));
                    G.h_jobsB.push_back(halloc<abub_job>(nj));
                    G.d_histB.push_back(dalloc<uint32_t>(nj * 256));
                    G.h_histB.push_back(halloc<uint32_t>(nj * 256));
                    // deferred pieces: at most H / 16 + 8 row ranges per job (chunks are at least 16 rows), 64 launches per block
                    const size_t pc = deferPieces ? nj * (size_t)(H / 16 + 8) : 1;
                    G.pieceCapB.push_back(pc);
                    G.d_piecesB.push_back((void *)dalloc<uint64_t>(pc));
                    G.d_pcountB.push_back(dalloc<uint32_t>(64));
                    G.d_incB.push_back(dalloc<uint8_t>(nj));
                    G.h_incB.push_back(halloc<uint8_t>(nj));
                    G.d_wantB.push_back(dalloc<uint8_t>(nj));
                    G.h_wantB.push_back(halloc<uint8_t>(nj));
                }
                G.fetches.assign(nB, std::vector<Group::Fetch>());
                G.pieceUsedB.assign(nB, 0);
                G.d_jobs3 = dalloc<abub_job>(n3);
                G.d_hist3 = dalloc<uint32_t>(n3 * 256);
                G.d_img = dalloc<uint8_t>(abub_fast_path(W) ? 256 : n3 * P); // only the unfused fallback stores images
                G.d_thr = dalloc<int32_t>(n3);
                G.d_pairs = dalloc<uint32_t>((size_t)G.pairCap * 2);
                G.d_count = dalloc<uint32_t>(1);
                G.d_gscratch = dalloc<uint32_t>(2 * n3);
                G.d_goff = dalloc<uint32_t>(n3 + 1);
                G.d_gidx = dalloc<uint32_t>(G.pairCap);
                G.d_gval = dalloc<uint8_t>(G.pairCap);
                G.h_hist3 = halloc<uint32_t>(n3 * 256);
                G.h_count = halloc<uint32_t>(1);
                G.h_goff = halloc<uint32_t>(n3 + 1);
                G.h_gidx = halloc<uint32_t>(G.pairCap);
                G.h_gval = halloc<uint8_t>(G.pairCap);
                G.h_jobs3 = halloc<abub_job>(n3);
                G.h_thr = halloc<int32_t>(n3);
            }
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
        } catch (...) { // (e.g. a hipMalloc that fails: give back what was allocated so far)
            release();
            throw;
        }
    }

    ~RunPipeline() { release(); }

    // everything the object owns (also called by the constructor when it fails half-way: a destructor would not run)
    void release() noexcept
    {
        (void)hipSetDevice(device);
        for (Group &G : groups) {
            if (G.stream) {
                (void)abub_scratch_release(G.stream);
                (void)hipStreamDestroy(G.stream);
            }
            if (G.stage1Done)
                (void)hipEventDestroy(G.stage1Done);
            if (G.kernelsDone)
                (void)hipEventDestroy(G.kernelsDone);
            if (G.blockDone)
                (void)hipEventDestroy(G.blockDone);
        }
        if (stage1Stream) {
            (void)abub_scratch_release(stage1Stream); // the trigger search's work list lives in library scratch
            (void)hipStreamDestroy(stage1Stream);
        }
        if (copyStream)
            (void)hipStreamDestroy(copyStream);
        for (auto &e : copied)
            (void)hipEventDestroy(e);
        for (void *p : devAllocs)
            (void)hipFree(p);
        for (void *p : hostAllocs)
            (void)hipHostFree(p);
        for (Trainer *t : trainers)
            delete t;
        groups.clear();
        copied.clear();
        devAllocs.clear();
        hostAllocs.clear();
        trainers.clear();
        stage1Stream = copyStream = nullptr;
    }

    // `callerStream`: work already queued there (e.g. the upload of the frames) is waited for first
    const uint8_t *d_sigmaRaw = nullptr; // optional: sigma (not 6*sigma) for stacks that need the drop-in path
    uint8_t *d_ownFrames = nullptr;      // frame slab owned by the pipeline (streamed mode only)
    hipStream_t copyStream = nullptr;
    std::vector<hipEvent_t> copied;

    // Streamed mode (BASELINE configs[4]): the run sits in HOST memory (ideally pinned).  Stack groups are
    // uploaded in order on a copy stream; the trigger search of group g waits only for its own upload, so it
    // overlaps the transfer of group g+1 (double buffering in time; the slab itself stays resident for the
    // localisation stages).
    void runFromHost(const uint8_t *h_frames, const uint8_t *d_mu, const uint8_t *d_sigma6)
    {
        HIPOK(hipSetDevice(device));
        if (!d_ownFrames) {
            d_ownFrames = dalloc<uint8_t>((size_t)S * F * P);
            HIPOK(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
            copied.resize(ngroups);
            for (auto &e : copied)
                HIPOK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        for (int g = 0; g < ngroups; ++g) {
            const Group &G = groups[g];
            const size_t off = (size_t)G.s0 * F * P, n = (size_t)(G.s1 - G.s0) * F * P;
            HIPOK(hipMemcpyAsync(d_ownFrames + off, h_frames + off, n, hipMemcpyHostToDevice, copyStream));
            HIPOK(hipEventRecord(copied[g], copyStream));
        }
        waitCopies = true;
        run(d_ownFrames, d_mu, d_sigma6, nullptr);
        waitCopies = false;
    }
    bool waitCopies = false;

    void run(const uint8_t *d_frames, const uint8_t *d_mu, const uint8_t *d_sigma6, hipStream_t callerStream)
    {
        HIPOK(hipSetDevice(device));
        if (!waitCopies)
            HIPOK(hipStreamSynchronize(callerStream));
        g_quietAnalyzers = true;
        double t0 = nowMs();
        stacks.clear();
        stacks.resize(S);
        // Stage 1 of every group goes to ONE stream in group order: the trigger search of group g+1 runs
        // on the GPU while the host threads of group g are in their state machines (two kernels launched on
        // different streams would simply share the chip and finish together, leaving nothing to overlap).
        jobsLaunched = jobsCompleted = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            Group &G = groups[gi];
            if (waitCopies)
                HIPOK(hipStreamWaitEvent(stage1Stream, copied[gi], 0));
            std::fill(G.usedB.begin(), G.usedB.end(), 0);
            std::fill(G.pieceUsedB.begin(), G.pieceUsedB.end(), 0);
            for (auto &fv : G.fetches)
                fv.clear();
            std::vector<int> all;
            for (int sI = G.s0; sI < G.s1; ++sI)
                all.push_back(sI);
            if (F > 1)
                launchBlock(G, 0, all, d_frames, d_sigma6, stage1Stream); // block 0: every stack, slot == index in the group
            HIPOK(hipEventRecord(G.stage1Done, stage1Stream));
        }
        std::vector<std::thread> th;
        for (int g = 1; g < ngroups; ++g)
            th.emplace_back([&, g]() { runGroupNoThrow(groups[g], d_frames, d_mu, d_sigma6); });
        runGroupNoThrow(groups[0], d_frames, d_mu, d_sigma6);
        for (auto &t : th)
            t.join();
        std::fill(tms, tms + 8, 0.0);
        rounds = 0;
        lastPairs = 0;
        // stacks the batched providers could not serve (bellows veto): one at a time through the drop-in path
        dropIns = 0;
        for (int s = 0; s < S; ++s)
            if (stacks[s].dropIn) {
                ++dropIns;
                runDropIn(s, d_frames, d_mu);
            }
        for (Group &G : groups) {
            if (!G.error.empty())
                throw std::runtime_error(G.error);
            for (int k = 0; k < 8; ++k)
                tms[k] = std::max(tms[k], G.tms[k]);
            rounds = std::max(rounds, G.rounds);
            lastPairs += G.lastPairs;
        }
        tms[4] = nowMs() - t0;
    }

private:
    // AnyCamAnalysis of one stack with host copies of its frames and model (the rare bellows-veto case)
    void runDropIn(int s, const uint8_t *d_frames, const uint8_t *d_mu)
    {
        StackState &st_ = stacks[s];
        st_.bubbles.clear();
        if (!d_sigmaRaw) {
            st_.staged = -6;
            st_.error = "bellows veto needs the drop-in path, but no sigma image was given to the pipeline";
            return;
        }
        const int e = s / C, c = s % C;
        try {
            const int Fs = meta.empty() ? F : (int)meta[s].names.size();
            std::vector<cv::Mat> frames((size_t)Fs);
            for (int i = 0; i < Fs; ++i) {
                if (!meta.empty() && !meta[s].ok[i])
                    continue; // undecodable on disk: stays an empty Mat (GetImage == -1)
                frames[i].create(H, W, CV_8U);
                HIPOK(hipMemcpy(frames[i].data, d_frames + ((size_t)s * F + i) * P, P, hipMemcpyDeviceToHost));
            }
            MemParser mp;
            const std::string evName = meta.empty() ? std::to_string(e) : meta[s].eventID;
            if (meta.empty())
                mp.AddFrames(evName, c, frames, 10000);
            else
                mp.AddNamedFrames(evName, c, meta[s].names, &frames);
            Trainer t(c, {}, "", "cam%d_image%u.png", "", mp.clone(), false);
            t.TrainedAvgImage.create(H, W, CV_8U);
            t.TrainedSigmaImage.create(H, W, CV_8U);
            HIPOK(hipMemcpy(t.TrainedAvgImage.data, d_mu + (size_t)c * P, P, hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(t.TrainedSigmaImage.data, d_sigmaRaw + (size_t)c * P, P, hipMemcpyDeviceToHost));
            t.TrainingSetSize = tss[c];
            t.ModelId = 0; // always (re)uploaded
            Trainer *tp = &t;
            L3Localizer A(evName, "", c, true, &tp, maskDir, mp.clone());
            int staged = 0;
            do {
                A.FindTriggerFrame(true, A.MatTrigFrame + 1);
                if (A.okToProceed) {
                    A.LocalizeOMatic("");
                    if (A.okToProceed)
                        staged = A.BubbleList.empty() ? -1 : 0;
                    else {
                        staged = -8;
                        break;
                    }
                } else {
                    staged = A.TriggerFrameIdentificationStatus;
                    break;
                }
            } while (A.BubbleList.size() == 0);
            st_.staged = staged;
            st_.trig = A.MatTrigFrame;
            st_.status = A.TriggerFrameIdentificationStatus;
            st_.loc_thres = A.loc_thres;
            st_.ok = A.okToProceed;
            for (bubble *b : A.BubbleList) {
                BubbleOut o;
                o.desc = b->KnownDescriptors;
                o.dz = b->dz;
                o.dzdt = b->dZdT();
                o.drdt = b->dRdT();
                st_.bubbles.push_back(std::move(o));
            }
        } catch (std::exception &ex) {
            st_.error = ex.what();
            st_.staged = -6;
        }
        DeviceContext::releaseThread();
    }

    void runGroupNoThrow(Group &G, const uint8_t *d_frames, const uint8_t *d_mu, const uint8_t *d_sigma6)
    {
        G.error.clear();
        try {
            (void)hipSetDevice(device);
            runGroup(G, d_frames, d_mu, d_sigma6);
        } catch (std::exception &e) {
            G.error = e.what();
        }
    }

    void runGroup(Group &G, const uint8_t *d_frames, const uint8_t *d_mu, const uint8_t *d_sigma6)
    {
        std::fill(G.tms, G.tms + 8, 0.0);
        G.rounds = 0;
        const int ns = G.s1 - G.s0;
        double t0 = nowMs();
        // ---- stage 1 (already queued by run()) -----------------------------------------------------
        // analyzers are (re)built while the GPU works
        pool->parallelFor(ns, [&](int k) {
            const int s = G.s0 + k;
            StackState &st_ = stacks[s];
            const int e = s / C, c = s % C;
            Trainer *t = trainers[c];
            if (meta.empty()) {
                st_.analyzer.reset(new L3Localizer(std::to_string(e), "", c, true, &t, maskDir, parser.clone()));
                st_.data.F = F;
                st_.data.ok = nullptr;
            } else {
                st_.analyzer.reset(new L3Localizer(meta[s].eventID, "", c, true, &t, maskDir, metaParser.clone()));
                st_.data.F = (int)meta[s].names.size();
                st_.data.ok = meta[s].ok.data();
            }
            st_.data.W = W;
            st_.data.H = H;
            st_.data.refOffset = tss[c] < 6 ? 1 : 2;
            st_.data.nblocks = (int)blocks.size() - 1;
            for (size_t b = 0; b < blocks.size(); ++b)
                st_.data.bstart[b] = blocks[b];
            for (int b = 0; b < st_.data.nblocks; ++b) {
                st_.data.bh[b] = nullptr;
                st_.data.inc[b] = nullptr;
            }
            // block 0 was launched for every stack of the group, in group order (fetch 0, position k)
            if (F > 1)
                bindBlock(G, st_, 0, 0, k);
            st_.analyzer->AttachEventData(&st_.data);
        });
        HIPOK(hipEventSynchronize(G.stage1Done));
        G.tms[0] = nowMs() - t0;

        std::vector<int> pending(ns);
        for (int k = 0; k < ns; ++k)
            pending[k] = G.s0 + k;
        while (!pending.empty()) {
            ++G.rounds;
            // ---- stage 2: trigger search + plan ------------------------------------------------
            // A search that runs into a frame block which has not been evaluated for its stack stops there
            // (NeedMoreFrames); those blocks are evaluated -- one launch per block index -- and the searches run again.
            double t2 = nowMs();
            std::vector<int> todo(pending);
            while (!todo.empty()) {
                pool->parallelFor((int)todo.size(), [&](int k) { triggerAndPlan(stacks[todo[k]]); });
                std::vector<int> need;
                for (int sI : todo)
                    if (stacks[sI].needMore)
                        need.push_back(sI);
                if (need.empty())
                    break;
                fetchBlocks(G, need, d_frames, d_sigma6);
                todo.swap(need);
            }
            G.tms[1] += nowMs() - t2;
            // ---- stage 3: batched images, thresholds, foreground ---------------------------------
            double t3 = nowMs();
            std::vector<int> loc;
            for (int s : pending)
                if (stacks[s].localize)
                    loc.push_back(s);
            if (!loc.empty())
                batchImages(G, loc, d_frames, d_mu, d_sigma6);
            G.tms[2] += nowMs() - t3;
            // ---- stage 4: localize + track -------------------------------------------------------
            double t4 = nowMs();
            pool->parallelFor((int)loc.size(), [&](int k) { localize(stacks[loc[k]]); });
            G.tms[3] += nowMs() - t4;
            std::vector<int> next;
            for (int s : pending)
                if (!stacks[s].done)
                    next.push_back(s);
            pending.swap(next);
        }
        // results out, analyzers released
        pool->parallelFor(ns, [&](int k) {
            StackState &st_ = stacks[G.s0 + k];
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
    }

    // What the stopped searches asked for: the next frame block of a stack (one launch per block index), or the dense rows
    // of frames of a block that is already there (one row-machine launch per launch of the block that holds them).
    void fetchBlocks(Group &G, const std::vector<int> &need, const uint8_t *d_frames, const uint8_t *d_sigma6)
    {
        hipStream_t stream = ordered ? stage1Stream : G.stream;
        const int nB = (int)blocks.size() - 1;
        std::vector<std::vector<int>> byBlock((size_t)nB);
        std::vector<std::vector<std::vector<int>>> byFetch((size_t)nB); // [block][fetch] -> stacks that need pieces
        for (int sI : need) {
            const BatchEventData &d = stacks[sI].data;
            if (!d.needPieces)
                byBlock[(size_t)d.needBlock].push_back(sI);
            else {
                auto &v = byFetch[(size_t)d.needBlock];
                if (v.size() <= (size_t)d.fetchOf[d.needBlock])
                    v.resize((size_t)d.fetchOf[d.needBlock] + 1);
                v[(size_t)d.fetchOf[d.needBlock]].push_back(sI);
            }
        }
        std::vector<int> newFetch((size_t)nB, -1);
        {
            std::lock_guard<std::mutex> lock(launchMu); // (the counters, and one launch sequence at a time per pipeline)
            for (int k = 0; k < nB; ++k) {
                if (!byBlock[k].empty()) {
                    launchBlock(G, k, byBlock[k], d_frames, d_sigma6, stream);
                    newFetch[k] = (int)G.fetches[k].size() - 1;
                }
                const size_t blen = (size_t)(blocks[k + 1] - blocks[k]);
                for (size_t f = 0; f < byFetch[k].size(); ++f) {
                    if (byFetch[k][f].empty())
                        continue;
                    // complete, for every asking stack, the flagged frames from the one it stopped at to the end of the block
                    const Group::Fetch &fe = G.fetches[k][f];
                    const size_t nj = (size_t)fe.n * blen, base = (size_t)fe.first * blen;
                    uint8_t *hw = G.h_wantB[k] + base;
                    std::memset(hw, 0, nj);
                    for (int sI : byFetch[k][f]) {
                        BatchEventData &d = stacks[sI].data;
                        for (size_t j = (size_t)(d.needFrame - blocks[k]); j < blen; ++j)
                            if (d.inc[k][j]) {
                                hw[(size_t)d.slotOf[k] * blen + j] = 1;
                                ++jobsCompleted;
                            }
                    }
                    HIPOK(hipMemcpyAsync(G.d_wantB[k] + base, hw, nj, hipMemcpyHostToDevice, stream));
                    check(abub_diff_hist_pieces_dev(d_frames, d_sigma6, G.d_jobsB[k] + base, (int)nj, W, H, G.d_histB[k] + base * 256,
                                                    (uint64_t *)G.d_piecesB[k] + fe.pieceOff, G.d_pcountB[k] + f, G.d_wantB[k] + base,
                                                    stream),
                          "trigger search K2 (pieces)");
                    for (int sI : byFetch[k][f]) {
                        BatchEventData &d = stacks[sI].data;
                        const size_t j0 = (size_t)(d.needFrame - blocks[k]), o = (base + (size_t)d.slotOf[k] * blen + j0) * 256;
                        HIPOK(hipMemcpyAsync(G.h_histB[k] + o, G.d_histB[k] + o, (blen - j0) * 1024, hipMemcpyDeviceToHost, stream));
                    }
                }
            }
            HIPOK(hipEventRecord(G.blockDone, stream));
        }
        HIPOK(hipEventSynchronize(G.blockDone));
        for (int k = 0; k < nB; ++k) {
            for (size_t q = 0; q < byBlock[k].size(); ++q) {
                StackState &st_ = stacks[byBlock[k][q]];
                bindBlock(G, st_, k, newFetch[k], (int)q);
                st_.needMore = false;
            }
            const size_t blen = (size_t)(blocks[k + 1] - blocks[k]);
            for (auto &v : byFetch[k])
                for (int sI : v) {
                    StackState &st_ = stacks[sI];
                    for (size_t j = (size_t)(st_.data.needFrame - blocks[k]); j < blen; ++j)
                        st_.data.inc[k][j] = 0; // final now
                    st_.needMore = false;
                }
        }
    }

    // AnyCamAnalysis body up to LocalizeOMatic (AutoBubStart3.cpp:87-107)
    void triggerAndPlan(StackState &st_)
    {
        st_.localize = false;
        st_.needMore = false;
        AnalyzerUnit *A = st_.analyzer.get();
        try {
            // what the search may change before it runs out of evaluated frames (it only ever appends to pix_counts)
            size_t pixSize[256];
            const bool havePix = A->pix_counts.size() == 256;
            for (int b = 0; havePix && b < 256; ++b)
                pixSize[b] = A->pix_counts[b].size();
            const int trig0 = A->MatTrigFrame, loc0 = A->loc_thres, status0 = A->TriggerFrameIdentificationStatus;
            const bool ok0 = A->okToProceed;
            try {
                A->FindTriggerFrame(true, A->MatTrigFrame + 1);
            } catch (NeedMoreFrames &) {
                for (int b = 0; havePix && b < 256 && A->pix_counts.size() == 256; ++b)
                    A->pix_counts[b].resize(pixSize[b]);
                A->MatTrigFrame = trig0;
                A->loc_thres = loc0;
                A->TriggerFrameIdentificationStatus = status0;
                A->okToProceed = ok0;
                st_.needMore = true;
                return;
            }
            if (!A->okToProceed) {
                st_.staged = A->TriggerFrameIdentificationStatus;
                st_.done = true;
                return;
            }
            const int F = st_.data.F;       // this stack's frame count (<= the pipeline's)
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

    void batchImages(Group &G, const std::vector<int> &loc, const uint8_t *d_frames, const uint8_t *d_mu,
                     const uint8_t *d_sigma6)
    {
        // Kernels go to `stream`, results come back on the group's own stream.  Ordered mode (default): every
        // group's kernels queue on the one trigger-search stream, so the GPU sees K2(g0) K2(g1) .. S3(g0) S3(g1) ..
        // strictly one kernel at a time -- chip-filling kernels launched on different streams only slow each
        // other down -- while the host stages of group g run under the kernels of group g+1.
        hipStream_t stream = ordered ? stage1Stream : G.stream;
        hipStream_t back = G.stream;
        // slots: all genesis images first (K2), then all post-trigger images (K3), camera-major: consecutive K3 jobs
        // then share their model, which is what lets one scanning wave serve several tracking frames (k3_zero_scan)
        int nd = 0, np = 0;
        for (int s : loc)
            for (PlannedImage &p : stacks[s].data.planned)
                (p.kind == 0 ? nd : np)++;
        std::vector<int> order(loc);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return a % C < b % C; });
        int di = 0, pi = nd;
        for (int s : order) {
            const int c = s % C;
            for (PlannedImage &p : stacks[s].data.planned) {
                p.slot = p.kind == 0 ? di++ : pi++;
                abub_job &j = G.h_jobs3[p.slot];
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
            stacks[s].data.roundHists = G.h_hist3;
            for (PlannedImage &p : stacks[s].data.planned) {
                bySlot[p.slot] = &p;
                G.h_thr[p.slot] = p.tozero; // candidate cut = TOZERO threshold, known before the launch
            }
        }
        uint32_t cnt = 0;
        for (int attempt = 0;; ++attempt) {
            HIPOK(hipMemcpyAsync(G.d_jobs3, G.h_jobs3, (size_t)nimg * sizeof(abub_job), hipMemcpyHostToDevice, stream));
            HIPOK(hipMemcpyAsync(G.d_thr, G.h_thr, (size_t)nimg * sizeof(int32_t), hipMemcpyHostToDevice, stream));
            HIPOK(hipMemsetAsync(G.d_count, 0, sizeof(uint32_t), stream));
            const bool fused = abub_fast_path(W) != 0;
            if (fused) {
                // images are never materialised: histogram + candidate list come out of the same pass
                check(abub_diff_hist_compact_dev(d_frames, d_sigma6, G.d_jobs3, nd, W, H, G.d_hist3, nullptr, G.d_thr,
                                                 G.d_pairs, G.pairCap, G.d_count, 0, stream),
                      "stage3 K2 compact");
                if (np > 0)
                    check(abub_posttrig_compact_dev(d_frames, d_mu, d_sigma6, G.d_jobs3 + nd, np, W, H,
                                                    G.d_hist3 + (size_t)nd * 256, nullptr, G.d_thr + nd, G.d_pairs,
                                                    G.pairCap, G.d_count, (uint32_t)nd, stream),
                          "stage3 K3 compact");
            } else {
                check(abub_diff_hist_dev(d_frames, d_sigma6, G.d_jobs3, nd, W, H, G.d_hist3, G.d_img, 0, stream),
                      "stage3 K2 store");
                if (np > 0)
                    check(abub_posttrig_dev(d_frames, d_mu, d_sigma6, G.d_jobs3 + nd, np, W, H, G.d_hist3 + (size_t)nd * 256,
                                            G.d_img + (size_t)nd * P, stream),
                          "stage3 K3");
                check(abub_fg_compact_pairs_dev(G.d_img, nimg, W, H, G.d_thr, G.d_pairs, G.pairCap, G.d_count, stream),
                      "stage3 K4");
            }
            // group the list by image on the device; the host gets contiguous runs and never re-buckets
            // (per-slot counts come from the histograms the same launches produced: no counting pass)
            check(abub_pairs_group_hist_dev(G.d_pairs, G.d_count, G.pairCap, nimg, G.d_gscratch, G.d_goff, G.d_gidx, G.d_gval,
                                            G.d_hist3, G.d_thr, stream),
                  "stage3 group");
            HIPOK(hipEventRecord(G.kernelsDone, stream));
            HIPOK(hipStreamWaitEvent(back, G.kernelsDone, 0));
            HIPOK(hipMemcpyAsync(G.h_hist3, G.d_hist3, (size_t)nimg * 1024, hipMemcpyDeviceToHost, back));
            HIPOK(hipMemcpyAsync(G.h_count, G.d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, back));
            HIPOK(hipMemcpyAsync(G.h_goff, G.d_goff, (size_t)(nimg + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, back));
            HIPOK(hipStreamSynchronize(back));
            G.tms[5] += nowMs() - ta; // launches + kernels + hist/count D2H
            ta = nowMs();
            cnt = *G.h_count;
            G.lastPairs = cnt;
            if (cnt <= G.pairCap)
                break;
            // dense foreground (e.g. a flash frame): the kernels kept counting past the capacity, so the needed
            // size is known -- grow the lists once and redo the batch
            if (attempt > 0 || cnt > (1u << 30))
                throw std::runtime_error("RunPipeline: foreground list overflow (dense foreground in too many images)");
            growLists(G, cnt + cnt / 4 + 1024);
        }
        if (cnt) {
            HIPOK(hipMemcpyAsync(G.h_gidx, G.d_gidx, (size_t)cnt * 4, hipMemcpyDeviceToHost, back));
            HIPOK(hipMemcpyAsync(G.h_gval, G.d_gval, (size_t)cnt, hipMemcpyDeviceToHost, back));
        }
        // thresholds (TOZERO + Otsu) on the host from the histograms, while the list travels
        pool->parallelFor(nimg, [&](int k) {
            PlannedImage *p = bySlot[k];
            p->thr = binarizeThresholdFromHist(G.h_hist3 + (size_t)k * 256, P, p->tozero);
            p->fg = G.h_gidx + G.h_goff[k];
            p->fgv = G.h_gval + G.h_goff[k];
            p->nfg = G.h_goff[k + 1] - G.h_goff[k];
        });
        HIPOK(hipStreamSynchronize(back));
        G.tms[6] += nowMs() - ta; // list D2H (+ thresholds)
    }

    // the old buffers stay on the allocation lists and are released with the pipeline
    void growLists(Group &G, uint32_t cap)
    {
        G.pairCap = cap;
        G.d_pairs = dalloc<uint32_t>((size_t)cap * 2);
        G.d_gidx = dalloc<uint32_t>(cap);
        G.d_gval = dalloc<uint8_t>(cap);
        G.h_gidx = halloc<uint32_t>(cap);
        G.h_gval = halloc<uint8_t>(cap);
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
        } catch (NeedsDropInPath &) {
            st_.dropIn = true;
            st_.done = true;
        } catch (std::exception &e) {
            st_.error = e.what();
            st_.staged = -6;
            st_.done = true;
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// RunBatched: a run from a Parser through the batched pipeline (see runbatch.hpp).  Replaces the detect loop of the
// reference's main program (AutoBubStart3.cpp:338-388): same per-(event, camera) analyses, same output blocks in the
// same order, but the frames of a whole batch of events are decoded once (the reference decodes a frame up to three
// times: main loop, look-ahead, localizer), uploaded once and processed with a handful of launches.
// ------------------------------------------------------------------------------------------------------------------
int RunBatched(Parser *parser, const std::vector<std::string> &EventList, const std::vector<Trainer *> &Trainers,
               int numCams, const std::string &out_dir, const std::string &run_number, int frameOffset,
               const BatchedRunOptions &opt, BatchedRunStats *stats, std::string *why)
{
    const double tAll = nowMs();
    auto refuse = [&](const char *msg) {
        if (why)
            *why = msg;
        return 1;
    };
    const int C = numCams;
    if (C <= 0 || (int)Trainers.size() != C)
        return refuse("one trainer per camera expected");
    const int W = Trainers[0]->TrainedAvgImage.cols, H = Trainers[0]->TrainedAvgImage.rows;
    if (W <= 0 || H <= 0)
        return refuse("untrained model");
    for (Trainer *t : Trainers)
        if (t->TrainedAvgImage.cols != W || t->TrainedAvgImage.rows != H || t->TrainedSigmaImage.cols != W ||
            t->TrainedSigmaImage.rows != H)
            return refuse("cameras with different image sizes");
    const size_t P = (size_t)W * H;
    std::vector<int> mine; // indices into EventList handled by this process
    for (int i = 0; i < (int)EventList.size(); ++i)
        if (opt.shardWorld <= 1 || i % opt.shardWorld == opt.shardRank)
            mine.push_back(i);
    BatchedRunStats st;
    st.W = W;
    st.H = H;
    st.events = (int)mine.size();
    if (mine.empty()) {
        if (stats)
            *stats = st;
        return 0;
    }
    const int ndec = std::max(1, opt.decodeThreads);

    // ---- frame lists of every (event, camera), in the Parser's (lexicographic) order -------------------------------
    double t0 = nowMs();
    std::vector<std::vector<std::vector<std::string>>> lists(mine.size(), std::vector<std::vector<std::string>>(C));
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < std::min<int>(ndec, (int)mine.size()); ++t)
            th.emplace_back([&]() {
                std::unique_ptr<Parser> p(parser->clone());
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= mine.size())
                        break;
                    for (int c = 0; c < C; ++c)
                        p->ParseAndSortFramesInFolder(EventList[mine[k]], c, lists[k][c]);
                }
            });
        for (auto &t : th)
            t.join();
    }
    int Fmax = 1;
    for (auto &ev : lists)
        for (auto &l : ev)
            Fmax = std::max(Fmax, (int)l.size());
    st.list_s = (nowMs() - t0) * 1e-3;
    if (Fmax > 1024)
        return refuse("more than 1024 frames in one stack");
    st.Fmax = Fmax;
    const size_t perEvent = (size_t)C * Fmax * P;
    int G = (int)std::max<size_t>(1, std::min<size_t>(opt.batchBytes / perEvent, mine.size()));
    // A run that would fit a few batches is cut into at least twelve per GPU (of at least four events): decoding batch
    // b + 1 then overlaps the GPU work of batch b, and the two pinned slabs stay small -- page-locking 2.6 GB takes about
    // as long as decoding it on 16 cores (measured: 1.0 s of a 2.8 s run of 96 events with batches of 24).
    {
        const int ng = std::max(1, opt.ngpus);
        const int want = std::max(4, (int)((mine.size() + (size_t)12 * ng - 1) / ((size_t)12 * ng)));
        G = std::max(1, std::min(G, want));
    }
    G = std::min(G, 512);
    // ---- where the frames are decoded --------------------------------------------------------------------------------
    // On the GPU (abub_png.hip) when the parser hands out the files as they are stored and the first frame is a PNG the
    // kernels take; a host thread still reads each file and walks its chunks, and decodes the odd frame the GPU path
    // refuses.  Otherwise host threads decode every frame (GetImageInto) as before.
    bool devDecode = opt.gpuDecode != 0 && (W & 3) == 0 && W >= 4 && W <= 2048;
    if (const char *e = getenv("ABUB_GPU_DECODE"))
        devDecode = devDecode && atoi(e) != 0;
    if (devDecode) {
        devDecode = false;
        for (size_t k = 0; k < lists.size() && !devDecode; ++k)
            for (int c = 0; c < C && !devDecode; ++c)
                if (!lists[k][c].empty()) {
                    std::unique_ptr<Parser> p(parser->clone());
                    const long long sz = p->GetImageFileSize(EventList[mine[k]], lists[k][c][0]);
                    if (sz > 0 && sz < ((long long)1 << 30)) {
                        std::vector<unsigned char> buf((size_t)sz);
                        PngInfo info;
                        devDecode = p->ReadImageFile(EventList[mine[k]], lists[k][c][0], buf.data(), buf.size()) == sz &&
                                    pngWalk(buf.data(), buf.size(), W, H, info);
                    }
                    k = lists.size(); // (one probe decides)
                    break;
                }
    }
    int Ggpu = 0; // device-decode mode: the first Ggpu events of a batch are decoded on the GPU, the others by the host threads
    if (devDecode) {
        // The inflate kernel runs four streams per CU at a time (1024 on an MI355X) and a batch takes as long as its longest
        // stream: batches carry that many frames for the GPU, not one more.
        int ncu = 256;
        {
            int dev0 = opt.firstDevice, v = 0, nd = 0;
            if (hipGetDeviceCount(&nd) == hipSuccess && nd > 0 &&
                hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev0 % nd) == hipSuccess && v > 0)
                ncu = v;
        }
        const int perEv = std::max(1, C * Fmax);
        const int capG = (int)std::max<size_t>(1, opt.batchBytes / perEvent);
        Ggpu = std::max(1, std::min((4 * ncu) / perEv, std::min(capG, (int)mine.size())));
        if (const char *e = getenv("ABUB_GPU_DECODE_EVENTS")) // (tests: a small GPU share)
            Ggpu = std::max(1, std::min(Ggpu, atoi(e)));
        // The host threads can decode the frames of a few more events per batch while the GPU works (they only READ the
        // files otherwise): ABUB_HOST_DECODE_EVENTS=n.  Off by default -- measured on a 96-event run (16 threads): 8.1 k
        // frames/s without, 7.5 k with n = 4, 7.3 k with n = 8: the first batch's host share is not overlapped with
        // anything, the GPU decode slows by 10 - 15 % beside 16 busy cores, and a run of eight batches never makes that up.
        int Ghost = 0;
        if (const char *e = getenv("ABUB_HOST_DECODE_EVENTS"))
            Ghost = std::max(0, atoi(e));
        Ghost = std::max(0, std::min(Ghost, std::min(capG, (int)mine.size()) - Ggpu));
        G = Ggpu + Ghost;
    }
    const int nb = ((int)mine.size() + G - 1) / G;
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    if (ndev <= 0)
        throw std::runtime_error("RunBatched: no GPU");
    const int ngpus = std::max(1, std::min(opt.ngpus, nb));
    st.gpus = ngpus;
    st.batches = nb;
    st.eventsPerBatch = G;

    std::mutex turnMu;
    std::condition_variable turnCv;
    int turn = 0;       // next batch to be written (output is in event order, AutoBubStart3.cpp:380-383)
    bool failed = false;
    std::vector<std::string> errors(ngpus);
    std::mutex statMu;

    struct Decoded {
        std::vector<StackMeta> meta;
        double ms = 0;
        long long ok = 0, bad = 0;
    };
    auto decodeBatch = [&](int b, uint8_t *h, Decoded &out, int nthreads) {
        const double td = nowMs();
        const int e0 = b * G, nEv = std::min(G, (int)mine.size() - e0);
        out.meta.assign((size_t)G * C, StackMeta());
        std::vector<std::pair<int, int>> tasks;
        for (int k = 0; k < G; ++k)
            for (int c = 0; c < C; ++c) {
                StackMeta &m = out.meta[(size_t)k * C + c];
                if (k < nEv) {
                    m.eventID = EventList[mine[e0 + k]];
                    m.names = lists[e0 + k][c];
                    m.ok.assign(m.names.size(), 0);
                    for (int f = 0; f < (int)m.names.size(); ++f)
                        tasks.emplace_back(k * C + c, f);
                } else
                    m.eventID = "_pad" + std::to_string(k); // filler of the last batch: no frames -> -9, never written
            }
        std::atomic<size_t> next{0};
        std::atomic<long long> good{0}, bad{0};
        std::vector<std::thread> th;
        for (int t = 0; t < std::max(1, std::min<int>(nthreads, (int)tasks.size())); ++t)
            th.emplace_back([&]() {
                std::unique_ptr<Parser> p(parser->clone());
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= tasks.size())
                        break;
                    const int s = tasks[i].first, f = tasks[i].second;
                    StackMeta &m = out.meta[s];
                    uint8_t *dst = h + ((size_t)s * Fmax + f) * P;
                    int rc = -1;
                    try {
                        rc = p->GetImageInto(m.eventID, m.names[f], dst, W, H); // decoded in place, no per-frame allocation
                    } catch (...) {
                        rc = -1;
                    }
                    // (anything but 1 = undecodable, like EventOnDevice; so is a frame of another size)
                    if (rc != 1) {
                        // (its slot would otherwise keep the bytes of an earlier batch or half a decode: results never use
                        // them, but dense garbage costs the trigger search's kernels time that varies from run to run)
                        std::memset(dst, 0, P);
                        ++bad;
                        continue;
                    }
                    m.ok[f] = 1;
                    ++good;
                }
            });
        for (auto &t : th)
            t.join();
        out.ok = good;
        out.bad = bad;
        out.ms = nowMs() - td;
    };

    // ---- device-decode mode: a batch's FILES into one pinned buffer, with what the GPU decoder needs to know about each ----
    struct Encoded {
        std::vector<StackMeta> meta;
        uint8_t *h_files = nullptr; // pinned; grown on demand by the reading thread (its worker's device is current there)
        size_t cap = 0, bytes = 0;
        uint8_t *d_files = nullptr; // the reading thread uploads them as soon as they are read: the copy runs beside the GPU
        size_t dcap = 0;            // work of the batch before
        hipEvent_t uploaded = nullptr;
        std::vector<abub_png_frame> desc;        // the frames the GPU decodes
        std::vector<std::pair<int, int>> where;  // (stack, frame) of desc[i]
        std::vector<uint32_t> fileOff, fileLen;  // of desc[i] inside h_files
        std::vector<abub_png_seg> segs;
        std::vector<uint8_t> luts;               // 256 bytes each
        size_t zbytes = 0;
        // frames a host thread decoded while reading (files the GPU path does not take): pixels + (stack, frame)
        std::vector<std::vector<uint8_t>> hostPix;
        std::vector<std::pair<int, int>> hostWhere;
        double ms = 0;
        long long bad = 0, hostGood = 0;
    };
    auto readBatch = [&](int b, Encoded &out, uint8_t *h_host, int nthreads, int dev, hipStream_t upStream) {
        const double td = nowMs();
        (void)hipSetDevice(dev);
        const int e0 = b * G, nEv = std::min(G, (int)mine.size() - e0);
        out.meta.assign((size_t)G * C, StackMeta());
        struct Task {
            int s, f;
            long long size;
            size_t off;
            int state; // 0 = for the GPU, 1 = decoded here, 2 = missing / undecodable
            PngInfo info;
            std::vector<uint8_t> pix;
        };
        std::vector<Task> tasks, hostTasks; // (the files for the GPU are read first: its work can start before the host's is done)
        std::unique_ptr<Parser> sizer(parser->clone());
        size_t total = 0;
        for (int k = 0; k < G; ++k)
            for (int c = 0; c < C; ++c) {
                StackMeta &m = out.meta[(size_t)k * C + c];
                if (k < nEv) {
                    m.eventID = EventList[mine[e0 + k]];
                    m.names = lists[e0 + k][c];
                    m.ok.assign(m.names.size(), 0);
                    for (int f = 0; f < (int)m.names.size(); ++f) {
                        Task t;
                        t.s = k * C + c;
                        t.f = f;
                        if (k >= Ggpu) { // the host threads' share of the batch: decoded straight into the pinned slab
                            t.size = 0;
                            t.off = 0;
                            t.state = 3;
                            hostTasks.push_back(std::move(t));
                            continue;
                        }
                        t.size = sizer->GetImageFileSize(m.eventID, m.names[f]);
                        t.state = (t.size > 0 && t.size < ((long long)1 << 30)) ? 0 : 2;
                        t.off = total;
                        if (t.state == 0)
                            total += ((size_t)t.size + 15) & ~(size_t)15;
                        tasks.push_back(std::move(t));
                    }
                } else
                    m.eventID = "_pad" + std::to_string(k);
            }
        total += 16;
        if (total > out.cap) {
            if (out.h_files)
                (void)hipHostFree(out.h_files);
            out.h_files = nullptr;
            out.cap = total + total / 4;
            if (hipHostMalloc((void **)&out.h_files, out.cap, hipHostMallocDefault) != hipSuccess) {
                out.cap = 0;
                throw std::runtime_error("RunBatched: hipHostMalloc of the file staging buffer failed");
            }
        }
        out.bytes = total;
        const size_t nGpuTasks = tasks.size();
        for (Task &t : hostTasks)
            tasks.push_back(std::move(t));
        std::atomic<size_t> next{0};
        std::atomic<long long> hostGood{0}, hostBad{0};
        std::vector<std::thread> th;
        for (int t = 0; t < std::max(1, std::min<int>(nthreads, (int)tasks.size())); ++t)
            th.emplace_back([&]() {
                std::unique_ptr<Parser> p(parser->clone());
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= tasks.size())
                        break;
                    Task &t = tasks[i];
                    if (t.state == 3) {
                        StackMeta &hm = out.meta[t.s];
                        uint8_t *hd = h_host + ((size_t)(t.s - Ggpu * C) * Fmax + t.f) * P;
                        int rc = -1;
                        try {
                            rc = p->GetImageInto(hm.eventID, hm.names[t.f], hd, W, H);
                        } catch (...) {
                            rc = -1;
                        }
                        if (rc != 1) {
                            std::memset(hd, 0, P);
                            ++hostBad;
                        } else {
                            hm.ok[t.f] = 1;
                            ++hostGood;
                        }
                        continue;
                    }
                    if (t.state != 0)
                        continue;
                    const StackMeta &m = out.meta[t.s];
                    uint8_t *dst = out.h_files + t.off;
                    long long got = -1;
                    try {
                        got = p->ReadImageFile(m.eventID, m.names[t.f], dst, (size_t)t.size);
                    } catch (...) {
                        got = -1;
                    }
                    if (got != t.size) {
                        t.state = 2;
                        continue;
                    }
                    try {
                        if (!pngWalk(dst, (size_t)t.size, W, H, t.info)) {
                            // not a file for the GPU decoder (BMP, 16-bit, colour, interlaced, another size): the host decoder's answer
                            t.pix.resize(P);
                            t.state = cv::imdecodeInto(dst, (size_t)t.size, t.pix.data(), W, H) ? 1 : 2;
                        }
                    } catch (...) { // (an allocation that fails inside a pool thread must not end the process)
                        t.state = 2;
                    }
                }
            });
        for (auto &t : th)
            t.join();
        out.desc.clear();
        out.where.clear();
        out.fileOff.clear();
        out.fileLen.clear();
        out.segs.clear();
        out.luts.clear();
        out.hostPix.clear();
        out.hostWhere.clear();
        out.bad = hostBad;
        out.hostGood = hostGood;
        size_t zoff = 0;
        for (size_t ti = 0; ti < nGpuTasks; ++ti) {
            Task &t = tasks[ti];
            if (t.state == 2) {
                ++out.bad;
                continue;
            }
            if (t.state == 1) {
                out.hostPix.push_back(std::move(t.pix));
                out.hostWhere.emplace_back(t.s, t.f);
                continue;
            }
            abub_png_frame d;
            d.seg_begin = (uint32_t)out.segs.size();
            d.seg_count = (uint32_t)t.info.segs.size();
            d.zoff = (uint32_t)zoff;
            d.zlen = (uint32_t)t.info.zlen;
            d.lut = 0xffffffffu;
            d.reserved = 0;
            d.dst = ((uint64_t)t.s * Fmax + t.f) * P;
            if (t.info.palette) { // (the frames of a run share their palette: look for the table among those already kept)
                size_t nl = out.luts.size() / 256, l = 0;
                for (; l < nl; ++l)
                    if (!memcmp(&out.luts[l * 256], t.info.lut, 256))
                        break;
                if (l == nl)
                    out.luts.insert(out.luts.end(), t.info.lut, t.info.lut + 256);
                d.lut = (uint32_t)l;
            }
            for (const abub_png_seg &sg : t.info.segs)
                out.segs.push_back(abub_png_seg{(uint32_t)(t.off + sg.off), sg.len});
            zoff += (((size_t)d.zlen + 15) & ~(size_t)15) + 16;
            out.desc.push_back(d);
            out.where.emplace_back(t.s, t.f);
            out.fileOff.push_back((uint32_t)t.off);
            out.fileLen.push_back((uint32_t)t.size);
        }
        out.zbytes = zoff + 16;
        if (out.bytes >= ((size_t)1 << 32) || out.zbytes >= ((size_t)1 << 32))
            throw std::runtime_error("RunBatched: a batch of more than 4 GB of files (lower the batch size)");
        if (!out.desc.empty()) {
            if (out.bytes > out.dcap) {
                if (out.d_files)
                    (void)hipFree(out.d_files);
                out.d_files = nullptr;
                out.dcap = out.bytes + out.bytes / 4;
                if (hipMalloc((void **)&out.d_files, out.dcap) != hipSuccess) {
                    out.dcap = 0;
                    throw std::runtime_error("RunBatched: hipMalloc of the uploaded files failed");
                }
            }
            if (!out.uploaded && hipEventCreateWithFlags(&out.uploaded, hipEventDisableTiming) != hipSuccess)
                throw std::runtime_error("RunBatched: hipEventCreate failed");
            if (hipMemcpyAsync(out.d_files, out.h_files, out.bytes, hipMemcpyHostToDevice, upStream) != hipSuccess ||
                hipEventRecord(out.uploaded, upStream) != hipSuccess)
                throw std::runtime_error("RunBatched: upload of the files failed");
        }
        out.ms = nowMs() - td;
    };

    auto writeBatch = [&](RunPipeline &pipe, int b) {
        const int e0 = b * G, nEv = std::min(G, (int)mine.size() - e0);
        for (int k = 0; k < nEv; ++k) {
            OutputWriter out(out_dir, run_number, frameOffset, C);
            const int actualEventNumber = atoi(EventList[mine[e0 + k]].c_str());
            std::vector<std::vector<bubble *>> owned(C);
            for (int c = 0; c < C; ++c) {
                StackState &ss = pipe.stacks[(size_t)k * C + c];
                if (!ss.error.empty())
                    std::cout << ss.error << '\n'; // (AnyCamAnalysis prints the exception text, then stages -6)
                if (ss.staged == 0) {
                    for (BubbleOut &bo : ss.bubbles) { // the analyzers are gone: rebuild the track records
                        bubble *bb = new bubble(bo.desc[0]);
                        for (size_t d = 1; d < bo.desc.size(); ++d) {
                            bb->lockThisIteration = false;
                            *bb << bo.desc[d];
                        }
                        owned[c].push_back(bb);
                    }
                    out.stageCameraOutput(owned[c], c, ss.trig, actualEventNumber);
                } else
                    out.stageCameraOutputError(c, ss.staged, actualEventNumber);
            }
            out.writeCameraOutput();
            for (auto &l : owned)
                for (bubble *bb : l)
                    delete bb;
        }
    };

    auto worker = [&](int g) {
        uint8_t *h_slab[2] = {nullptr, nullptr}, *d_slab[2] = {nullptr, nullptr}, *d_model = nullptr;
        hipStream_t copyStream = nullptr;
        // The look-ahead decode thread writes decd[] and reads nthr: both live OUTSIDE the try block, so that they outlive
        // it on every failure path (the thread is joined below, after the catch).
        Decoded decd[2];
        Encoded encd[2]; // device-decode mode
        // device-decode mode: the uploaded files, the decoder's scratch, its descriptors (grown on demand)
        uint8_t *d_z = nullptr, *d_raw = nullptr, *d_luts = nullptr;
        hipStream_t upStream = nullptr;
        abub_png_frame *d_desc = nullptr;
        abub_png_seg *d_segs = nullptr;
        int32_t *d_status = nullptr, *h_status = nullptr;
        size_t capZ = 0, capRaw = 0, capLuts = 0, capDesc = 0, capSegs = 0, capStatus = 0;
        const int nthr = std::max(1, ndec / ngpus);
        std::thread dec;
        std::exception_ptr decErr[2];
        try {
            const int dev = (opt.firstDevice + g) % ndev;
            HIPOK(hipSetDevice(dev));
            const size_t slabBytes = (size_t)G * perEvent;
            const int nslots = g + ngpus < nb ? 2 : 1; // a worker with a single batch needs no second buffer
            for (int k = 0; k < nslots; ++k) {
                // (device-decode mode uploads files; only the host threads' share of a batch comes as frames)
                const size_t pinned = devDecode ? (size_t)(G - Ggpu) * perEvent : slabBytes;
                if (pinned)
                    HIPOK(hipHostMalloc((void **)&h_slab[k], pinned, hipHostMallocDefault));
                HIPOK(hipMalloc((void **)&d_slab[k], slabBytes));
            }
            auto grow = [&](void **ptr, size_t &cap, size_t need) {
                if (need <= cap)
                    return;
                if (*ptr)
                    HIPOK(hipFree(*ptr));
                *ptr = nullptr;
                cap = need + need / 4 + 256;
                HIPOK(hipMalloc(ptr, cap));
            };
            if (devDecode) {
                HIPOK(hipHostMalloc((void **)&h_status, (size_t)G * C * Fmax * sizeof(int32_t) + 64, hipHostMallocDefault));
                HIPOK(hipStreamCreateWithFlags(&upStream, hipStreamNonBlocking));
            }
            HIPOK(hipMalloc((void **)&d_model, 3 * (size_t)C * P)); // mu | sigma | sigma6
            uint8_t *d_mu = d_model, *d_sigma = d_model + (size_t)C * P, *d_s6 = d_model + 2 * (size_t)C * P;
            HIPOK(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
            std::vector<int> tss(C);
            for (int c = 0; c < C; ++c) {
                HIPOK(hipMemcpy(d_mu + (size_t)c * P, Trainers[c]->TrainedAvgImage.data, P, hipMemcpyHostToDevice));
                HIPOK(hipMemcpy(d_sigma + (size_t)c * P, Trainers[c]->TrainedSigmaImage.data, P, hipMemcpyHostToDevice));
                tss[c] = Trainers[c]->TrainingSetSize;
            }
            check(abub_sigma6_dev(d_sigma, d_s6, (size_t)C * P, copyStream), "abub_sigma6_dev");
            HIPOK(hipStreamSynchronize(copyStream));
            const bool trace = getenv("ABUB_INGEST_TRACE") != nullptr;
            if (trace)
                fprintf(stderr, "worker %d: buffers and model on the device at %.1f ms\n", g, nowMs() - tAll);
            RunPipeline pipe(dev, W, H, Fmax, G, C, tss.data(), std::max(1, opt.hostThreads), opt.maskDir.c_str());
            pipe.d_sigmaRaw = d_sigma;
            if (trace)
                fprintf(stderr, "worker %d: pipeline of %d events ready at %.1f ms\n", g, G, nowMs() - tAll);
            int slot = 0;
            // (an exception of the look-ahead thread -- a failed allocation, a throwing parser -- is carried over and re-thrown here)
            auto startDecode = [&](int bb, int sl) {
                decErr[sl] = nullptr;
                dec = std::thread([&, bb, sl, dev]() { // (dev by value: it lives inside the try block)
                    try {
                        if (devDecode)
                            readBatch(bb, encd[sl], h_slab[sl], nthr, dev, upStream);
                        else
                            decodeBatch(bb, h_slab[sl], decd[sl], nthr);
                    } catch (...) {
                        decErr[sl] = std::current_exception();
                    }
                });
            };
            if (g < nb)
                startDecode(g, slot);
            if (devDecode) {
                // (the decoder's scratch while the first batch's files are being read: sizes from the batch's frame count; the
                // stream buffer from a guess that regrows if a batch proves it wrong)
                const size_t nfMax = (size_t)std::max(1, Ggpu) * C * Fmax;
                grow((void **)&d_raw, capRaw, nfMax * abub_png_raw_stride(W, H));
                grow((void **)&d_desc, capDesc, nfMax * sizeof(abub_png_frame));
                grow((void **)&d_status, capStatus, nfMax * sizeof(int32_t));
                grow((void **)&d_z, capZ, nfMax * (P / 4 * 3));
                if (trace)
                    fprintf(stderr, "worker %d: decoder scratch ready at %.1f ms\n", g, nowMs() - tAll);
            }
            for (int b = g; b < nb; b += ngpus) {
                const double tj = nowMs();
                dec.join();
                if (trace)
                    fprintf(stderr, "batch %d: waited %.1f ms for its files at %.1f ms\n", b, nowMs() - tj, nowMs() - tAll);
                if (decErr[slot])
                    std::rethrow_exception(decErr[slot]);
                const int bn = b + ngpus;
                if (bn < nb)
                    startDecode(bn, slot ^ 1);
                const int nEv = std::min(G, (int)mine.size() - b * G);
                const double tg = nowMs();
                double dms = 0, pngms = 0;
                long long good = 0, bad = 0, onGpu = 0, onHost = 0;
                if (!devDecode) {
                    HIPOK(hipMemcpyAsync(d_slab[slot], h_slab[slot], (size_t)nEv * perEvent, hipMemcpyHostToDevice, copyStream));
                    dms = decd[slot].ms;
                    good = decd[slot].ok;
                    bad = decd[slot].bad;
                    onHost = good;
                    pipe.setStackMeta(std::move(decd[slot].meta));
                } else {
                    Encoded &E = encd[slot];
                    const int nf = (int)E.desc.size();
                    // frames nobody decodes (missing, undecodable) stay zero: results never use them, but dense garbage would
                    // cost the trigger search's kernels time that varies from run to run
                    const int nEvGpu = std::min(nEv, Ggpu);
                    HIPOK(hipMemsetAsync(d_slab[slot], 0, (size_t)nEvGpu * perEvent, copyStream));
                    if (nEv > nEvGpu) // the events the host threads decoded
                        HIPOK(hipMemcpyAsync(d_slab[slot] + (size_t)nEvGpu * perEvent, h_slab[slot], (size_t)(nEv - nEvGpu) * perEvent,
                                             hipMemcpyHostToDevice, copyStream));
                    const double tp = nowMs();
                    std::vector<uint8_t> okGpu((size_t)nf, 0);
                    if (nf) {
                        const size_t stride = abub_png_raw_stride(W, H);
                        grow((void **)&d_z, capZ, E.zbytes);
                        grow((void **)&d_raw, capRaw, (size_t)nf * stride);
                        grow((void **)&d_desc, capDesc, (size_t)nf * sizeof(abub_png_frame));
                        grow((void **)&d_segs, capSegs, E.segs.size() * sizeof(abub_png_seg) + 8);
                        grow((void **)&d_luts, capLuts, E.luts.size() + 256);
                        grow((void **)&d_status, capStatus, (size_t)nf * sizeof(int32_t));
                        HIPOK(hipStreamWaitEvent(copyStream, E.uploaded, 0)); // (the reading thread's upload of the files)
                        HIPOK(hipMemcpyAsync(d_desc, E.desc.data(), (size_t)nf * sizeof(abub_png_frame), hipMemcpyHostToDevice, copyStream));
                        HIPOK(hipMemcpyAsync(d_segs, E.segs.data(), E.segs.size() * sizeof(abub_png_seg), hipMemcpyHostToDevice, copyStream));
                        if (!E.luts.empty())
                            HIPOK(hipMemcpyAsync(d_luts, E.luts.data(), E.luts.size(), hipMemcpyHostToDevice, copyStream));
                        check(abub_png_decode_dev(E.d_files, E.bytes, d_desc, nf, d_segs, (int)E.segs.size(), d_luts, (int)(E.luts.size() / 256), W,
                                                  H, d_z, capZ, d_raw, capRaw, d_slab[slot], (size_t)nEv * perEvent, d_status, copyStream),
                              "abub_png_decode_dev");
                        HIPOK(hipMemcpyAsync(h_status, d_status, (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, copyStream));
                        HIPOK(hipStreamSynchronize(copyStream));
                        for (int i = 0; i < nf; ++i)
                            okGpu[i] = h_status[i] == 0;
                    }
                    // a frame the kernels refused: the host decoder's answer (the same image, or the same failure)
                    std::vector<uint8_t> pix;
                    for (int i = 0; i < nf; ++i) {
                        StackMeta &m = E.meta[E.where[i].first];
                        if (okGpu[i]) {
                            m.ok[E.where[i].second] = 1;
                            ++onGpu;
                            continue;
                        }
                        pix.resize(P);
                        if (cv::imdecodeInto(E.h_files + E.fileOff[i], E.fileLen[i], pix.data(), W, H)) {
                            HIPOK(hipMemcpy(d_slab[slot] + E.desc[i].dst, pix.data(), P, hipMemcpyHostToDevice));
                            m.ok[E.where[i].second] = 1;
                            ++onHost;
                        } else {
                            HIPOK(hipMemset(d_slab[slot] + E.desc[i].dst, 0, P)); // (a refused frame may be half written)
                            ++E.bad;
                        }
                    }
                    for (size_t i = 0; i < E.hostPix.size(); ++i) {
                        const size_t at = ((size_t)E.hostWhere[i].first * Fmax + E.hostWhere[i].second) * P;
                        HIPOK(hipMemcpy(d_slab[slot] + at, E.hostPix[i].data(), P, hipMemcpyHostToDevice));
                        E.meta[E.hostWhere[i].first].ok[E.hostWhere[i].second] = 1;
                        ++onHost;
                    }
                    onHost += E.hostGood;
                    pngms = nowMs() - tp;
                    if (trace)
                        fprintf(stderr, "batch %d: %d frames for the GPU (%zu MB of files), %lld decoded by host threads, read + host decode %.1f ms, "
                                        "upload + GPU decode %.1f ms\n", b, nf, E.bytes >> 20, E.hostGood, E.ms, pngms);
                    dms = E.ms;
                    good = onGpu + onHost;
                    bad = E.bad;
                    pipe.setStackMeta(std::move(E.meta));
                }
                if (const char *tf = getenv("ABUB_TEST_FAIL_BATCH")) // test hook: a batch fails while the next one decodes
                    if (atoi(tf) == b)
                        throw std::runtime_error("injected failure of batch " + std::to_string(b) + " (ABUB_TEST_FAIL_BATCH)");
                pipe.run(d_slab[slot], d_mu, d_s6, copyStream); // waits for the upload first
                const double gms = nowMs() - tg;
                double wms = 0;
                {
                    std::unique_lock<std::mutex> lock(turnMu);
                    turnCv.wait(lock, [&] { return turn == b || failed; });
                    if (failed)
                        break;
                    const double tw = nowMs();
                    writeBatch(pipe, b);
                    wms = nowMs() - tw;
                    ++turn;
                }
                turnCv.notify_all();
                {
                    std::lock_guard<std::mutex> lock(statMu);
                    st.decode_s += dms * 1e-3;
                    st.gpu_s += gms * 1e-3;
                    st.write_s += wms * 1e-3;
                    st.frames += good;
                    st.framesFailed += bad;
                    st.framesGpuDecoded += onGpu;
                    st.framesHostDecoded += onHost;
                    st.gpudecode_s += pngms * 1e-3;
                }
                slot ^= 1;
            }
        } catch (std::exception &e) {
            errors[g] = e.what();
            {
                std::lock_guard<std::mutex> lock(turnMu);
                failed = true;
            }
            turnCv.notify_all();
        }
        if (dec.joinable())
            dec.join();
        if (copyStream)
            (void)hipStreamDestroy(copyStream);
        for (int k = 0; k < 2; ++k) {
            if (h_slab[k])
                (void)hipHostFree(h_slab[k]);
            if (d_slab[k])
                (void)hipFree(d_slab[k]);
        }
        if (d_model)
            (void)hipFree(d_model);
        if (upStream) {
            (void)hipStreamSynchronize(upStream);
            (void)hipStreamDestroy(upStream);
        }
        for (Encoded &E : encd) {
            if (E.d_files)
                (void)hipFree(E.d_files);
            if (E.uploaded)
                (void)hipEventDestroy(E.uploaded);
        }
        for (void *q : {(void *)d_z, (void *)d_raw, (void *)d_luts, (void *)d_desc, (void *)d_segs, (void *)d_status})
            if (q)
                (void)hipFree(q);
        if (h_status)
            (void)hipHostFree(h_status);
        for (Encoded &E : encd)
            if (E.h_files)
                (void)hipHostFree(E.h_files);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < ngpus; ++g)
        th.emplace_back(worker, g);
    worker(0);
    for (auto &t : th)
        t.join();
    for (const std::string &e : errors)
        if (!e.empty())
            throw std::runtime_error("RunBatched: " + e);
    st.total_s = (nowMs() - tAll) * 1e-3;
    if (stats)
        *stats = st;
    return 0;
}

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

void abh_pipe_set_sigma(void *p, const void *sigma_dev) { ((abub::RunPipeline *)p)->d_sigmaRaw = (const uint8_t *)sigma_dev; }

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

int abh_pipe_run_host(void *p, const void *frames_host, const void *mu_dev, const void *sigma6_dev)
{
    try {
        ((abub::RunPipeline *)p)->runFromHost((const uint8_t *)frames_host, (const uint8_t *)mu_dev, (const uint8_t *)sigma6_dev);
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
// out[0..11]: stage1 .. stage4, total, stage-3 details (ms), candidate pairs, trigger-search jobs, drop-in stacks,
// jobs completed on demand of the last run;
// returns the number of rounds
int abh_pipe_timing(void *p, double *out)
{
    abub::RunPipeline *r = (abub::RunPipeline *)p;
    for (int k = 0; k < 8; ++k)
        out[k] = r->tms[k];
    out[8] = r->lastPairs;
    out[9] = (double)r->jobsLaunched; // trigger-search jobs the run evaluated (S * (F - 1) when nothing is lazy)
    out[10] = (double)r->dropIns;     // stacks re-run one at a time (bellows veto)
    out[11] = (double)r->jobsCompleted; // trigger-search jobs whose dense rows were evaluated on demand (deferred pieces)
    return r->rounds;
}
}
