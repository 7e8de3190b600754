// PICOFormatWriterV4.hpp -- "PICO recon format" text output (API of the reference's
// PICOFormatWriter/PICOFormatWriterV4.hpp:13-63; file abub3hs_<run>.txt).  Host-only: this is where the
// results of the GPU path become the bytes downstream tools diff.
#ifndef ABUB3HS_PICOFORMATWRITERV4_HPP
#define ABUB3HS_PICOFORMATWRITERV4_HPP

#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../bubble/bubble.hpp"
#include "../cvlite.hpp"

class OutputWriter {
    int camera;
    int frameOffset;
    int StatusCode;
    std::string OutputDir;
    std::string run_number;
    std::string abubOutFilename;

    std::ofstream OutFile;
    std::stringstream _StreamOutput;

    int NumCams;

public:
    // not in the reference: inserted before ".txt" in the output file name of every writer of this process (a sharded run
    // -- `--gpu-shard r/N` -- writes abub3hs_<run>.part<r>of<N>.txt; `--merge N` assembles abub3hs_<run>.txt)
    static std::string PartSuffix;

    struct BubbleData {
        std::vector<bubble *> BubbleObjectData; // borrowed: the analyzer owns the bubbles
        int StatusCode;
        int frame0;
        int event;
        float dzdt;
        float drdt;
        BubbleData();
    };
    std::vector<BubbleData *> AllBubbleData; // one slot per camera

    OutputWriter(std::string OutDir, std::string run_number, int frameOffset, int NumCams);
    ~OutputWriter(void);

    void writeHeader(void);
    void stageCameraOutput(std::vector<bubble *> bubbles, int camera, int frame0, int event);
    void stageCameraOutputError(int camera, int error, int event);

    void formEachBubbleOutput(int camera, int &ibubImageStart, int nBubTotal);
    void writeCameraOutput(void);
    int CalculateNBubCamera(int);
};

#endif
