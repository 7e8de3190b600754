// bubble.cpp -- track record of one bubble (semantics of the reference's bubble/bubble.cpp:34-118).
#include "bubble/bubble.hpp"

#include <cmath>
#include <iostream>

// A new bubble starts locked: the genesis sighting already counts for its frame (bubble.cpp:34-48).
bubble::bubble(BubbleImageFrame genesis) : _dZdT(0)
{
    KnownDescriptors.push_back(genesis);
    last_x = genesis.MassCentres.x;
    last_y = genesis.MassCentres.y;
    GenesisPositionCentroid = genesis.MassCentres;
    GenesisPosition = genesis.newPosition;
    lockThisIteration = true;
}

bubble::~bubble() {}

// Append a sighting unless one was already taken this frame (bubble.cpp:54-72).  dz records the
// distance between the previous centroid x and the new bounding-box x.
void bubble::operator<<(BubbleImageFrame sighting)
{
    if (lockThisIteration)
        return;
    KnownDescriptors.push_back(sighting);
    dz.push_back(last_x - sighting.newPosition.x);
    last_x = sighting.MassCentres.x;
    last_y = sighting.MassCentres.y;
    lockThisIteration = true;
}

void bubble::printAllXY(void)
{
    for (const BubbleImageFrame &d : KnownDescriptors)
        std::cout << "X: " << d.MassCentres.x << " ";
    std::cout << "\n";
    for (const BubbleImageFrame &d : KnownDescriptors)
        std::cout << "Y: " << d.MassCentres.y << " ";
    std::cout << "\n";
}

bool bubble::isNewPositionProbable(int &x, int &y)
{
    return std::fabs(last_y - y) <= 4 && (last_x - x) < 5;
}

void bubble::dSizedT(std::vector<std::pair<float, float>> &) {}

// mean drift of the bounding-box x per tracked frame; 0/0 = NaN for an untracked bubble (bubble.cpp:101-108)
float bubble::dZdT(void)
{
    const int n = (int)KnownDescriptors.size();
    const float total = (float)(KnownDescriptors[0].newPosition.x - KnownDescriptors[n - 1].newPosition.x);
    return (float)(total / ((float)n - 1.0));
}

// growth of the bounding box per tracked frame (bubble.cpp:110-118)
float bubble::dRdT(void)
{
    const int n = (int)KnownDescriptors.size();
    const float dx = (float)(KnownDescriptors[0].newPosition.width - KnownDescriptors[n - 1].newPosition.width);
    const float dy = (float)(KnownDescriptors[0].newPosition.height - KnownDescriptors[n - 1].newPosition.height);
    const float dr = (float)std::sqrt(dx * dx + dy * dy);
    return (float)(dr / ((float)n - 1.0));
}
