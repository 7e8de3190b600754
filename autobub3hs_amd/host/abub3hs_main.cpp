// abub3hs_main.cpp -- command-line driver with the reference's argument surface and run flow
// (AutoBubStart3.cpp:127-405): -d data_dir -r run_ID -o out_dir [-z] [-c mask_dir] [-m] [-D series] [-e event]
// [--debug code]; per-series image naming / frameOffset / camera count (:220-245); header, event list, training
// (-7 rows if it fails), event loop with ordered output, -5 if the run cannot be read.  boost::program_options and
// OpenMP are replaced by a small parser and a std::thread pool (ABUB_THREADS, default min(16, cores)).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

#include "AlgorithmTraining/Trainer.hpp"
#include "BubbleLocalizer/L3Localizer.hpp"
#include "PICOFormatWriter/PICOFormatWriterV4.hpp"
#include "ParseFolder/RawParser.hpp"
#include "ParseFolder/ZipParser.hpp"
#include "driver.hpp"
#include "runbatch.hpp"

static std::string usage()
{
    return "Usage: abub3hs [-hzme] [-D data_series] [-c cam_mask_dir] [--debug code] -d data_dir -r run_ID -o out_dir\n"
           "Run the AutoBub3hs bubble finding algorithm on a PICO run (MI355X hot path)\n\n"
           "Required arguments:\n"
           "  -d, --data_dir = Dir\t\tpath to the directory in which the run folder/file is stored\n"
           "  -r, --run_id = Str\t\trun ID, formatted as YYYYMMDD_*\n"
           "  -o, --out_dir = Dir\t\tdirectory to write the output file to\n\n"
           "Optional arguments:\n"
           "  -h, --help\t\t\tgive this help message\n"
           "  -z, --zip\t\t\tindicate the run is stored as a zip file; otherwise assumed to be in a directory\n"
           "  -c, --cam_mask_dir = Dir\tdirectory containing the camera mask images\n"
           "  -m, --mask_check\t\tuse camera masks in default directory. Not needed if directory specified\n"
           "  -D, --data_series = Str\tname of the data series, e.g. 40l-19, 30l-16, etc.\n"
           "  -e, --event = Int\t\tspecify a single event to process. Mostly just useful for debugging and testing\n"
           "  --debug = Int\t\t\t3 digit int; eg: 101: first digit = localizer debug; second digit = multithread off; third digit = analyzer debug\n"
           "MI355X options:\n"
           "  --gpus = Int\t\t\tGPUs to spread the event batches over (one worker thread per GPU; default 1)\n"
           "  --gpu-shard = r/N\t\tprocess only the events whose index in the sorted event list is r modulo N;\n"
           "\t\t\t\twrites abub3hs_<run>.part<r>of<N>.txt (N > 1)\n"
           "  --merge = N\t\t\tassemble abub3hs_<run>.txt in --out_dir from the N part files of a sharded run\n"
           "  --per-event\t\t\tone analyzer at a time like the reference's loop (also chosen by -e and --debug);\n"
           "\t\t\t\tdefault: whole batches of events decoded into pinned memory and analysed together\n";
}

static bool eventNameOrderSort(const std::string &a, const std::string &b) { return std::stoi(a) < std::stoi(b); }

// cores this process may use: the affinity mask (taskset, cpuset) AND the cgroup's CPU quota (cpu.max: a container may see
// 256 cores and be allowed 16 of them) -- the reference's omp_get_max_threads() (AutoBubStart3.cpp:338) sees only the former
static int usableCores()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
        n = CPU_COUNT(&set);
    std::ifstream q("/sys/fs/cgroup/cpu.max"); // cgroup v2: "<quota> <period>" or "max <period>"
    std::string quota;
    long long period = 0;
    if (q >> quota >> period && quota != "max" && period > 0) {
        const long long c = (atoll(quota.c_str()) + period - 1) / period;
        if (c >= 1 && c < n)
            n = (int)c;
    }
    return n;
}

// --merge N: the reference writes ONE file in event order (the `ordered` clause, AutoBubStart3.cpp:380-383).  Shard r of N
// holds the events r, r + N, .. of the sorted event list, one block of rows per event, in order: the header comes from
// part 0, then the blocks are dealt back round-robin.  A block = consecutive rows with the same event number (column 2).
static int mergeParts(const std::string &out_dir, const std::string &run_number, int N)
{
    std::vector<std::vector<std::string>> blocks((size_t)N);
    std::string header;
    for (int r = 0; r < N; ++r) {
        const std::string path = out_dir + "abub3hs_" + run_number + ".part" + std::to_string(r) + "of" + std::to_string(N) + ".txt";
        std::ifstream in(path);
        if (!in) {
            std::cerr << "--merge: cannot read " << path << std::endl;
            return -1;
        }
        std::string line, head, lastEv;
        for (int k = 0; k < 6 && std::getline(in, line); ++k) // 3 header lines, "8", two blank lines (writeHeader)
            head += line + "\n";
        if (r == 0)
            header = head;
        else if (head != header) {
            std::cerr << "--merge: " << path << " has a different header" << std::endl;
            return -1;
        }
        while (std::getline(in, line)) {
            std::istringstream ls(line);
            std::string run, ev;
            ls >> run >> ev;
            if (blocks[r].empty() || ev != lastEv)
                blocks[r].push_back(std::string());
            blocks[r].back() += line + "\n";
            lastEv = ev;
        }
    }
    std::ofstream out(out_dir + "abub3hs_" + run_number + ".txt");
    if (!out) {
        std::cerr << "--merge: cannot write into " << out_dir << std::endl;
        return -1;
    }
    out << header;
    size_t longest = 0;
    for (auto &b : blocks)
        longest = std::max(longest, b.size());
    for (size_t j = 0; j < longest; ++j)
        for (int r = 0; r < N; ++r)
            if (j < blocks[r].size())
                out << blocks[r][j];
    return out.good() ? 0 : -1;
}

int main(int argc, char **argv)
{
    std::string dataLoc, run_number, out_dir, mask_dir, data_series;
    int event_user = -1, debug_mode = 0, ngpus = 1, shardRank = 0, shardWorld = 1, mergeN = 0;
    bool zipped = false, mask_check = false, help = argc == 1, perEvent = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], v;
        auto value = [&](std::string &dst) {
            size_t eq = a.find('=');
            if (a.rfind("--", 0) == 0 && eq != std::string::npos)
                dst = a.substr(eq + 1);
            else if (i + 1 < argc)
                dst = argv[++i];
        };
        auto is = [&](const char *s, const char *l) { return a == s || a == l || a.rfind(std::string(l) + "=", 0) == 0; };
        if (is("-h", "--help"))
            help = true;
        else if (is("-z", "--zip"))
            zipped = true;
        else if (is("-m", "--mask_check"))
            mask_check = true;
        else if (is("-d", "--data_dir"))
            value(dataLoc);
        else if (is("-r", "--run_num") || a == "--run_id")
            value(run_number);
        else if (is("-o", "--out_dir"))
            value(out_dir);
        else if (is("-c", "--cam_mask_dir"))
            value(mask_dir);
        else if (is("-D", "--data_series"))
            value(data_series);
        else if (is("-e", "--event")) {
            value(v);
            event_user = atoi(v.c_str());
        } else if (a.rfind("--debug", 0) == 0) {
            value(v);
            debug_mode = atoi(v.c_str());
        } else if (a.rfind("--gpus", 0) == 0) {
            value(v);
            ngpus = std::max(1, atoi(v.c_str()));
        } else if (a.rfind("--gpu-shard", 0) == 0) {
            value(v);
            if (sscanf(v.c_str(), "%d/%d", &shardRank, &shardWorld) != 2 || shardWorld < 1 || shardRank < 0 || shardRank >= shardWorld) {
                std::cerr << "--gpu-shard expects r/N with 0 <= r < N" << std::endl;
                return -1;
            }
        } else if (a.rfind("--merge", 0) == 0) {
            value(v);
            mergeN = atoi(v.c_str());
            if (mergeN < 1) {
                std::cerr << "--merge expects the number of shards" << std::endl;
                return -1;
            }
        } else if (a == "--per-event") {
            perEvent = true;
        } else {
            std::cerr << "unrecognised option '" << a << "'" << std::endl;
            return -1;
        }
    }
    if (help) {
        std::cout << usage() << std::endl;
        return 1;
    }
    if (mergeN > 0) {
        if (run_number.empty() || out_dir.empty()) {
            std::cerr << "--merge needs --run_id and --out_dir" << std::endl;
            return -1;
        }
        if (out_dir[out_dir.length() - 1] != '/')
            out_dir += "/";
        return mergeParts(out_dir, run_number, mergeN);
    }
    if (dataLoc.empty() || run_number.empty() || out_dir.empty()) {
        std::cerr << "Insufficient required arguments; use \"autobub3hs -h\" to view required arguments" << std::endl;
        return -1;
    }
    printf("This is AutoBub v3, the automatic unified bubble finder code for all chambers\n");

    std::string this_path = argv[0];
    std::string abub_dir = this_path.substr(0, this_path.find_last_of("/") + 1);
    if (mask_check && mask_dir == "")
        mask_dir = abub_dir + "cam_masks/" + data_series;
    else if (!mask_check && mask_dir == "")
        std::cout << "Not performing mask check on this run." << std::endl;

    std::string eventDir = dataLoc + "/" + run_number + "/";
    if (out_dir[out_dir.length() - 1] != '/')
        out_dir += "/";

    // how the different experiments stored their images (AutoBubStart3.cpp:216-245)
    std::string imageFormat, imageFolder;
    int frameOffset, numCams;
    if (data_series == "01l-21" || data_series == "2l-16") {
        imageFormat = "cam%dimage %u.bmp";
        imageFolder = "/";
        frameOffset = 0;
        numCams = 2;
    } else if (data_series == "40l-19") {
        imageFormat = "cam%d_image%u.png";
        imageFolder = "/Images/";
        frameOffset = 30;
        numCams = run_number >= "20200713_7" ? 4 : 2;
        if (run_number < "20200501_1")
            frameOffset = 20;
    } else {
        imageFormat = "cam%d_image%u.png";
        imageFolder = "/Images/";
        frameOffset = 30;
        numCams = 4;
    }
    if (const char *nc = getenv("ABUB_NUM_CAMS")) // synthetic runs with fewer cameras
        numCams = atoi(nc);

    // Threads.  The reference runs omp_get_max_threads() events at once (AutoBubStart3.cpp:338-342): the per-event loop and
    // the decoders of the batched path take every core this process may use (at most 128); the host stages of the batched
    // pipeline (state machines, contours) saturate at about 16.  ABUB_THREADS / ABUB_DECODE_THREADS override.
    const int cores = usableCores();
    int nthreads = std::min(cores, 128), hostThreads = std::min(cores, 16), decodeThreads = std::min(cores, 128);
    if (const char *t = getenv("ABUB_THREADS"))
        nthreads = hostThreads = decodeThreads = std::max(1, atoi(t));
    if (debug_mode % 100 / 10)
        nthreads = hostThreads = decodeThreads = 1;
    if (shardWorld > 1) // every shard writes its own part file (--merge N assembles the run's file)
        OutputWriter::PartSuffix = ".part" + std::to_string(shardRank) + "of" + std::to_string(shardWorld);

    OutputWriter *header = new OutputWriter(out_dir, run_number, frameOffset, numCams);
    header->writeHeader();

    std::vector<std::string> EventList;
    Parser *FileParser = nullptr;
    try {
        if (zipped)
            FileParser = new ZipParser(eventDir, imageFolder, imageFormat);
        else
            FileParser = new RawParser(eventDir, imageFolder, imageFormat);
        FileParser->GetEventDirLists(EventList);
    } catch (...) {
        std::cout << "Failed to read the images from run " << run_number << ". Autobub cannot continue.\n";
        if (shardRank == 0) { // (one block for the run: it goes into part 0 of a sharded run)
            for (int icam = 0; icam < numCams; icam++)
                header->stageCameraOutputError(icam, -5, -1);
            header->writeCameraOutput();
        }
        return -5;
    }
    std::sort(EventList.begin(), EventList.end(), eventNameOrderSort);

    printf("**Starting training. AutoBub is in learn mode**\n");
    std::vector<Trainer *> Trainers;
    for (int icam = 0; icam < numCams; icam++)
        Trainers.push_back(new Trainer(icam, EventList, eventDir, imageFormat, imageFolder, FileParser->clone(), debug_mode / 100));
    {
        std::vector<std::thread> th; // one thread per camera, like `#pragma omp parallel for` (:304-307)
        for (int icam = 0; icam < numCams; icam++)
            th.emplace_back([&, icam]() {
                try {
                    Trainers[icam]->MakeAvgSigmaImage(false);
                } catch (std::exception &e) {
                    std::cout << e.what() << '\n';
                    Trainers[icam]->StatusCode = -7;
                }
            });
        for (auto &t : th)
            t.join();
    }
    bool succeeded = true;
    for (Trainer *t : Trainers)
        if (t->StatusCode)
            succeeded = false;
    if (!succeeded) {
        std::cout << "Failed to train on images from run " << run_number << ". Autobub cannot continue.\n";
        for (size_t evi = 0; evi < EventList.size(); evi++) {
            if (shardWorld > 1 && (int)(evi % (size_t)shardWorld) != shardRank)
                continue;
            for (int icam = 0; icam < numCams; icam++)
                header->stageCameraOutputError(icam, -7, atoi(EventList[evi].c_str()));
            header->writeCameraOutput();
        }
        return -7;
    }
    printf("***Training complete. AutoBub is now in detect mode***\n");
    delete header;

    // ---- batched detect (default): every event of a batch decoded once into pinned memory, one set of launches per
    // batch, output in event order.  The per-event loop below stays for -e / --debug / --per-event and as the
    // fallback when the batched path declines the run.
    if (const char *e = getenv("ABUB_PER_EVENT"))
        perEvent = perEvent || atoi(e) != 0;
    if (!perEvent && event_user < 0 && debug_mode == 0) {
        abub::BatchedRunOptions bo;
        bo.ngpus = ngpus;
        if (const char *d = getenv("ABUB_DEVICE"))
            bo.firstDevice = atoi(d);
        bo.hostThreads = hostThreads;
        bo.decodeThreads = decodeThreads;
        if (const char *t = getenv("ABUB_DECODE_THREADS"))
            bo.decodeThreads = std::max(1, atoi(t));
        if (const char *b = getenv("ABUB_BATCH_MB"))
            bo.batchBytes = (size_t)std::max(1, atoi(b)) << 20;
        bo.shardRank = shardRank;
        bo.shardWorld = shardWorld;
        bo.maskDir = mask_dir;
        abub::BatchedRunStats bs;
        std::string why;
        int rc = 1;
        try {
            rc = abub::RunBatched(FileParser, EventList, Trainers, numCams, out_dir, run_number, frameOffset, bo, &bs, &why);
        } catch (std::exception &e) {
            std::cout << "batched detect failed: " << e.what() << std::endl;
            return -6;
        }
        if (rc == 0) {
            printf("batched detect: %d events in %d batches of <= %d on %d GPU(s), %lld frames %dx%d decoded (%lld on the GPU, %lld on "
                   "host threads; %lld undecodable), %.2f s total (list %.2f, read/decode %.2f, upload+GPU+host stages %.2f of which "
                   "GPU decode %.2f, write %.2f) = %.1f frames/s ingest-inclusive\n",
                   bs.events, bs.batches, bs.eventsPerBatch, bs.gpus, bs.frames, bs.W, bs.H, bs.framesGpuDecoded, bs.framesHostDecoded,
                   bs.framesFailed, bs.total_s, bs.list_s, bs.decode_s, bs.gpu_s, bs.gpudecode_s, bs.write_s,
                   bs.total_s > 0 ? (bs.frames + bs.framesFailed) / bs.total_s : 0.0);
            printf("run complete.\n");
            for (Trainer *t : Trainers)
                delete t;
            delete FileParser;
            printf("AutoBub done analyzing this run. Thank you.\n");
            return 0;
        }
        std::cout << "batched detect not used (" << why << "): falling back to the per-event loop" << std::endl;
    }

    // events in parallel, output appended in event order (the `ordered` clause :380-383)
    std::cout << "Total threads: " << nthreads << std::endl;
    std::atomic<int> next{0};
    std::mutex turnMutex;
    std::condition_variable turnCv;
    int turn = 0;
    auto worker = [&]() {
        for (;;) {
            const int evi = next.fetch_add(1);
            if (evi >= (int)EventList.size())
                break;
            const bool skip = (event_user >= 0 && evi != event_user) // compares the loop index, like upstream (:350)
                              || (shardWorld > 1 && evi % shardWorld != shardRank);
            OutputWriter *out = nullptr;
            std::vector<AnalyzerUnit *> Analyzers;
            if (!skip) {
                out = new OutputWriter(out_dir, run_number, frameOffset, numCams);
                const std::string imageDir = eventDir + EventList[evi] + "/Images/";
                const int actualEventNumber = atoi(EventList[evi].c_str());
                for (int icam = 0; icam < numCams; icam++) {
                    Analyzers.push_back(new L3Localizer(EventList[evi], imageDir, icam, debug_mode / 100 ? false : true,
                                                        &Trainers[icam], mask_dir, FileParser->clone()));
                    abub::AnyCamAnalysis(Analyzers[icam], icam, debug_mode % 10 ? false : true, out, out_dir, actualEventNumber);
                }
            }
            {
                std::unique_lock<std::mutex> lock(turnMutex);
                turnCv.wait(lock, [&] { return turn == evi; });
                if (out)
                    out->writeCameraOutput();
                ++turn;
            }
            turnCv.notify_all();
            delete out;
            for (AnalyzerUnit *A : Analyzers)
                delete A;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back(worker);
        for (auto &t : th)
            t.join();
    }
    printf("run complete.\n");
    for (Trainer *t : Trainers)
        delete t;
    delete FileParser;
    printf("AutoBub done analyzing this run. Thank you.\n");
    return 0;
}
