// PICOFormatWriterV4.cpp -- recon-format writer.  Byte layout of the reference's
// PICOFormatWriter/PICOFormatWriterV4.cpp: header :54-88, error row :146-174, bubble row :180-276,
// per-event assembly :287-302.  Two-space separators, std::fixed with precision 2 for every float,
// missing track slots as integer -1, rows end with "1  \n".
#include "PICOFormatWriter/PICOFormatWriterV4.hpp"

#include "common/CommonParameters.h"

OutputWriter::BubbleData::BubbleData() : StatusCode(0), frame0(0), event(0), dzdt(0), drdt(0) {}

std::string OutputWriter::PartSuffix;

OutputWriter::OutputWriter(std::string OutDir, std::string run_number, int frameOffset, int NumCams)
{
    this->OutputDir = OutDir;
    this->run_number = run_number;
    this->abubOutFilename = this->OutputDir + "abub3hs_" + this->run_number + PartSuffix + ".txt";
    this->frameOffset = frameOffset;
    this->NumCams = NumCams;
    this->camera = 0;
    this->StatusCode = 0;
    for (int i = 0; i < NumCams; i++)
        AllBubbleData.push_back(new BubbleData());
}

OutputWriter::~OutputWriter(void)
{
    for (BubbleData *d : AllBubbleData)
        delete d;
}

void OutputWriter::writeHeader(void)
{
    const int n = NumFramesBubbleTrack;
    OutFile.open(abubOutFilename);
    OutFile << "Output of AutoBub v3 - the automatic unified bubble finder code by Pitam, using OpenCV.\n";
    OutFile << "run  ev  ibubimage  TotalBub4CamImg  camera  frame0  hori  vert  GenesisW  GenesisH  dZdt  dRdt  ";
    OutFile << "TrkFrame(" << n << ")  TrkHori(" << n << ")  TrkVert(" << n << ")  ";
    OutFile << "TrkBubW(" << n << ")  TrkBubH(" << n << ")  TrkBubRadius(" << n << ")  FakeValue\n";
    OutFile << "%12s  %5d  %d  %d  %d  %d  %.02f  %.02f  %d  %d  %.02f  %.02f  ";
    for (int j = 0; j < n; j++)
        OutFile << "%d " << " ";
    for (int block = 0; block < 5; block++)
        for (int j = 0; j < n; j++)
            OutFile << "%.02f " << " ";
    OutFile << "%d";
    OutFile << "\n8\n\n\n";
    OutFile.close();
}

// An empty bubble list is staged as status -1 (:99-110); AnyCamAnalysis later overwrites it with the
// status of the retry that found nothing.
void OutputWriter::stageCameraOutput(std::vector<bubble *> bubbles, int camera, int frame0, int event)
{
    BubbleData *d = AllBubbleData[camera];
    d->BubbleObjectData = bubbles;
    d->StatusCode = bubbles.empty() ? -1 : 0;
    d->frame0 = frame0;
    d->event = event;
}

void OutputWriter::stageCameraOutputError(int camera, int error, int event)
{
    AllBubbleData[camera]->StatusCode = error;
    AllBubbleData[camera]->event = event;
}

void OutputWriter::formEachBubbleOutput(int camera, int &ibubImageStart, int nBubTotal)
{
    const int n = NumFramesBubbleTrack;
    std::stringstream &o = _StreamOutput;
    o.clear();
    o.precision(2);
    o.setf(std::ios::fixed, std::ios::floatfield);
    const char *sep = "  ";
    BubbleData *d = AllBubbleData[camera];

    if (d->StatusCode != 0) {
        // run ev 0 0 cam <code> 0.00 0.00 0 0 0.00 0.00 | 10 x 0 | 50 x 0.00 | 1
        o << run_number << sep << d->event << sep << 0 << sep << 0 << sep << camera << sep << d->StatusCode << sep << 0.0
          << sep << 0.0 << sep << 0 << sep << 0;
        o << sep << 0.0 << sep << 0.0 << sep;
        for (int j = 0; j < n; j++)
            o << 0 << sep;
        for (int j = 0; j < 5 * n; j++)
            o << 0.0 << sep;
        o << "1  \n";
        return;
    }

    for (size_t i = 0; i < d->BubbleObjectData.size(); i++) {
        bubble *b = d->BubbleObjectData[i];
        const int first = d->frame0 + frameOffset;
        o << run_number << sep << d->event << sep << ibubImageStart + (int)i << sep << nBubTotal << sep << camera << sep;
        o << first << sep;
        const float width = (float)b->GenesisPosition.width, height = (float)b->GenesisPosition.height;
        const float x = b->GenesisPositionCentroid.x, y = b->GenesisPositionCentroid.y;
        const float dzdt = b->dZdT(), drdt = b->dRdT();
        const int tracked = (int)b->KnownDescriptors.size() - 1;
        const int missing = n > tracked ? n - tracked : 0;
        o << x << sep << y << sep << (int)width << sep << (int)height << sep << dzdt << sep << drdt << sep;

        // frame numbers: the tracked ones, then `missing` more starting again at the last tracked one
        for (int j = 1; j <= tracked; j++)
            o << first + j << sep;
        for (int j = 0; j < missing; j++)
            o << first + tracked + j << sep;
        for (int j = 1; j <= tracked; j++)
            o << b->KnownDescriptors[j].MassCentres.x << sep;
        for (int j = 0; j < missing; j++)
            o << -1 << sep;
        for (int j = 1; j <= tracked; j++)
            o << b->KnownDescriptors[j].MassCentres.y << sep;
        for (int j = 0; j < missing; j++)
            o << -1 << sep;
        for (int j = 1; j <= tracked; j++)
            o << b->KnownDescriptors[j].newPosition.width << sep;
        for (int j = 0; j < missing; j++)
            o << -1 << sep;
        for (int j = 1; j <= tracked; j++)
            o << b->KnownDescriptors[j].newPosition.height << sep;
        for (int j = 0; j < missing; j++)
            o << -1 << sep;
        for (int j = 1; j <= tracked; j++)
            o << b->KnownDescriptors[j].ContRadius << sep;
        for (int j = 0; j < missing; j++)
            o << -1 << sep;
        o << "1  \n";
    }
    ibubImageStart += (int)d->BubbleObjectData.size();
}

// ibubimage runs across the cameras of the event from 1; TotalBub4CamImg counts status-0 cameras only
void OutputWriter::writeCameraOutput(void)
{
    int ibubImageStart = 1;
    int nBubTotal = 0;
    for (int i = 0; i < NumCams; i++)
        nBubTotal += AllBubbleData[i]->StatusCode != 0 ? 0 : (int)AllBubbleData[i]->BubbleObjectData.size();
    for (int i = 0; i < NumCams; i++)
        formEachBubbleOutput(i, ibubImageStart, nBubTotal);
    OutFile.open(abubOutFilename, std::fstream::out | std::fstream::app);
    OutFile << _StreamOutput.rdbuf();
    OutFile.close();
}

int OutputWriter::CalculateNBubCamera(int cam)
{
    return AllBubbleData[cam]->StatusCode != 0 ? 0 : (int)AllBubbleData[cam]->BubbleObjectData.size();
}
