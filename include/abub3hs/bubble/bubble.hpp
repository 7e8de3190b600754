// bubble.hpp -- per-bubble track record handed to the recon writer (mirrors the reference's
// bubble/bubble.hpp:22-82: BubbleImageFrame, class bubble with KnownDescriptors, operator<<, dZdT, dRdT).
#ifndef ABUB3HS_BUBBLE_HPP
#define ABUB3HS_BUBBLE_HPP

#include <utility>
#include <vector>

#include "../cvlite.hpp"

// one sighting of a bubble in one frame
struct BubbleImageFrame {
    cv::Rect newPosition;   // bounding box of the contour polygon
    double ContArea;        // polygon area
    double ContRadius;      // sqrt(ContArea / 3.14159)
    cv::Moments moments;
    cv::Point2f MassCentres; // polygon centroid
};

class bubble {
    int _dZdT;

public:
    explicit bubble(BubbleImageFrame genesis);
    ~bubble();

    std::vector<BubbleImageFrame> KnownDescriptors; // [0] = genesis, then one per tracked frame

    float last_x;
    float last_y;

    cv::Rect GenesisPosition;
    cv::Point2f GenesisPositionCentroid;

    std::vector<float> dz;

    void dSizedT(std::vector<std::pair<float, float>> &);
    float dRdT(void);
    float dZdT(void);

    bool lockThisIteration; // at most one sighting is appended per frame

    bool isNewPositionProbable(int &x, int &y);
    void printAllXY(void);

    void operator<<(BubbleImageFrame sighting);
};

#endif
