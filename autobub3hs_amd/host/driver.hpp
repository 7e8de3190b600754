// driver.hpp -- the per-camera retry loop of the reference's main program (AnyCamAnalysis,
// AutoBubStart3.cpp:67-125), shared by the CLI (abub3hs_main.cpp) and the test C-API (capi.cpp).
#ifndef ABUB3HS_DRIVER_HPP
#define ABUB3HS_DRIVER_HPP

#include <exception>
#include <iostream>
#include <mutex>
#include <string>

#include "AnalyzerUnit.hpp"
#include "PICOFormatWriter/PICOFormatWriterV4.hpp"

namespace abub {

// Runs FindTriggerFrame / LocalizeOMatic until a bubble is found or the search fails, staging the
// outcome in `writer`.  Returns the final staged status (0, -3, -9, -8, -6).
inline int AnyCamAnalysis(AnalyzerUnit *A, int camera, bool nonStopPref, OutputWriter *writer, const std::string &out_dir,
                          int actualEventNumber)
{
    static std::mutex stageMutex; // `#pragma omp critical` around the staging upstream (:96-97)
    int staged = 0;
    try {
        do {
            A->FindTriggerFrame(nonStopPref, A->MatTrigFrame + 1);
            if (A->okToProceed) {
                A->LocalizeOMatic(out_dir);
                if (A->okToProceed) {
                    std::lock_guard<std::mutex> lock(stageMutex);
                    writer->stageCameraOutput(A->BubbleList, camera, A->MatTrigFrame, actualEventNumber);
                    staged = A->BubbleList.empty() ? -1 : 0;
                } else {
                    writer->stageCameraOutputError(camera, -8, actualEventNumber);
                    staged = -8;
                    break;
                }
            } else {
                writer->stageCameraOutputError(camera, A->TriggerFrameIdentificationStatus, actualEventNumber);
                staged = A->TriggerFrameIdentificationStatus;
                break;
            }
        } while (A->BubbleList.size() == 0);
    } catch (std::exception &e) {
        std::cout << e.what() << '\n';
        writer->stageCameraOutputError(camera, -6, actualEventNumber);
        staged = -6;
    }
    return staged;
}

} // namespace abub
#endif
