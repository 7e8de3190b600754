// runbatch.hpp -- a whole run from a Parser (directory tree or zip archive) through the batched GPU pipeline:
// the replacement of the reference's detect loop (AutoBubStart3.cpp:338-388) when the per-event drop-in path is
// not asked for.  Events are decoded on host threads straight into pinned buffers, uploaded batch by batch
// (decode of batch b+1 overlaps the GPU work of batch b), analysed by RunPipeline, and written through
// OutputWriter in event order.  With several GPUs the batches are dealt round-robin to one worker thread per GPU.
#ifndef ABUB3HS_RUNBATCH_HPP
#define ABUB3HS_RUNBATCH_HPP

#include <cstddef>
#include <string>
#include <vector>

class Parser;
class Trainer;

namespace abub {

struct BatchedRunOptions {
    int ngpus = 1;                  // worker threads, one per GPU: device (firstDevice + g) % hipGetDeviceCount
    int firstDevice = 0;
    int hostThreads = 16;           // state machines / contour tracing per worker (RunPipeline's pool)
    int decodeThreads = 16;         // PNG / BMP decode threads, shared by the workers
    size_t batchBytes = (size_t)3 << 30; // frame bytes per batch; each worker holds two pinned and two HBM slabs of it
    int shardRank = 0, shardWorld = 1;   // this process takes events with index % shardWorld == shardRank
    std::string maskDir;
    int gpuDecode = -1;             // PNG frames decoded on the GPU (abub_png_decode_dev): 1 on, 0 off (host threads decode),
                                    // -1 = on where the parser hands out the files and the frames are 8-bit grey / palette
                                    // PNGs of a width the kernels take (ABUB_GPU_DECODE=0/1 overrides)
};

struct BatchedRunStats {
    double list_s = 0, decode_s = 0, gpu_s = 0, write_s = 0, total_s = 0; // decode/gpu: summed over batches (they overlap)
    long long frames = 0, framesFailed = 0;
    long long framesGpuDecoded = 0, framesHostDecoded = 0; // of `frames`: by abub_png_decode_dev / by a host thread
    double gpudecode_s = 0;                                // upload of the files + the decode kernels, summed over batches
    int events = 0, batches = 0, eventsPerBatch = 0, W = 0, H = 0, Fmax = 0, gpus = 0;
};

// Detects every event of `EventList` (already in output order) with the trained `Trainers` (one per camera) and
// appends the blocks to <out_dir>abub3hs_<run_number>.txt.  Returns 0; or a positive code when the batched path
// cannot take this run (different image sizes per camera, no frames, ...) and nothing was written -- the caller
// then runs the per-event loop; throws std::runtime_error on GPU / IO failures.
int RunBatched(Parser *parser, const std::vector<std::string> &EventList, const std::vector<Trainer *> &Trainers,
               int numCams, const std::string &out_dir, const std::string &run_number, int frameOffset,
               const BatchedRunOptions &opt, BatchedRunStats *stats, std::string *why);

} // namespace abub
#endif
