// hostlogic.cpp -- see include/abub3hs/hostlogic.hpp.  Built with -ffp-contract=off: the double
// statistics must round exactly like the reference's unfused arithmetic.
#include "hostlogic.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace abub {

// ---------------------------------------------------------------------------------------------
// significance (AnalyzerUnit.cpp:435-504, CalcMean/CalcStdDev :514-532)
// ---------------------------------------------------------------------------------------------
static inline double meanOf(const std::vector<int> &v, int n)
{
    double s = 0;
    for (int x : v)
        s += x;
    return s / n;
}
static inline double stddevOf(const std::vector<int> &v, double mean, int n)
{
    double s = 0;
    for (int x : v) // `val*val` is evaluated in int in the reference: keep the 32-bit wrap
        s += (int)((unsigned)x * (unsigned)x);
    return std::sqrt(s / n - mean * mean);
}

double significanceFromHist(std::vector<std::vector<int>> &pix_counts, const uint32_t hist[256],
                            size_t totalPixels, bool store, int trainingSetSize, int locThresMax,
                            int &loc_thres)
{
    double sig = 0;
    int remaining = (int)totalPixels;
    int firstOver = -1, maxAdc = 0;
    for (int bin = 0; bin < 256 && remaining > 0; ++bin) {
        const float count = (float)hist[bin]; // cv::calcHist output is CV_32F
        if (store)
            pix_counts[bin].push_back((int)count);
        if (bin > 1) {
            const int n0 = (int)pix_counts[0].size();
            const double mean = meanOf(pix_counts[bin], n0);
            const double sd = stddevOf(pix_counts[bin], mean, n0);
            if ((double)count != mean || sd > 0)
                sig += ((double)count - mean) / sd;
            if (sig < 0)
                sig = 0;
        }
        if (sig > 3.5 && firstOver < 0)
            firstOver = bin;
        if (bin > maxAdc)
            maxAdc = bin;
        if (store) {
            int t = std::max(firstOver - 1, maxAdc - 1);
            if (t < 2)
                t = 2;
            if (t > locThresMax || trainingSetSize < 6)
                t = locThresMax;
            loc_thres = t;
        }
        remaining = (int)((float)remaining - count);
    }
    return sig;
}

float entropyFromHist(const uint32_t hist[256], int nbins, size_t totalPixels)
{
    const int per = 256 / nbins;
    const float scale = (float)(1.0 / (double)(int)totalPixels);
    float e = 0.f;
    for (int b = 0; b < nbins; ++b) {
        uint32_t c = 0;
        for (int k = 0; k < per; ++k)
            c += hist[b * per + k];
        const float p = (float)c * scale;
        if (p != 0) {
            const float t = p * log2f(p);
            e = e - t;
        }
    }
    return e;
}

// ---------------------------------------------------------------------------------------------
// Otsu on the TOZERO'd image's histogram
// ---------------------------------------------------------------------------------------------
int binarizeThresholdFromHist(const uint32_t hist[256], size_t totalPixels, int tozeroThr)
{
    // histogram of the image after v = (v > tozeroThr ? v : 0)
    double h[256];
    double folded = 0;
    for (int i = 0; i < 256; ++i) {
        if (i <= tozeroThr) {
            folded += hist[i];
            h[i] = 0;
        } else
            h[i] = hist[i];
    }
    h[0] = folded;
    const double scale = 1. / (double)totalPixels;
    double mu = 0;
    for (int i = 0; i < 256; ++i)
        mu += i * h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, best = 0;
    int T = 0;
    for (int i = 0; i < 256; ++i) {
        const double p = h[i] * scale;
        mu1 *= q1;
        q1 += p;
        const double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON)
            continue;
        mu1 = (mu1 + i * p) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double between = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (between > best) {
            best = between;
            T = i;
        }
    }
    // mask = (tozero(v) > T)  <=>  v > max(tozeroThr, T)
    return std::max(tozeroThr, T);
}

// ---------------------------------------------------------------------------------------------
// contours
// ---------------------------------------------------------------------------------------------
static const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1}; // Freeman codes, y pointing down
static const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// Suzuki-Abe outer border following with the pixel marking OpenCV uses: 2 on ordinary border pixels,
// 2|-128 where the east neighbour was probed and found empty ("right edge").
void ContourFinder::traceBorder(size_t start, std::vector<signed char> &codes)
{
    const int step = w_ + 2;
    int off[16];
    for (int k = 0; k < 16; ++k)
        off[k] = kDy[k & 7] * step + kDx[k & 7];
    signed char *img = plane_.data();
    codes.clear();
    const size_t p0 = start;
    int s = 4;
    size_t p1 = 0;
    bool found = false;
    for (int n = 0; n < 7; ++n) { // clockwise from NW to SW
        s = (s - 1) & 7;
        if (img[p0 + off[s]] != 0) {
            p1 = p0 + off[s];
            found = true;
            break;
        }
    }
    if (!found) {
        img[p0] = (signed char)(2 | -128);
        return;
    }
    size_t p3 = p0, p4 = 0;
    for (;;) {
        const int sEnd = s;
        while (s < 15) {
            ++s;
            p4 = p3 + off[s];
            if (img[p4] != 0)
                break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)sEnd)
            img[p3] = (signed char)(2 | -128);
        else if (img[p3] == 1)
            img[p3] = 2;
        codes.push_back((signed char)s);
        if (p4 == p0 && p3 == p1)
            break;
        p3 = p4;
        s = (s + 4) & 7;
    }
}

void ContourFinder::find(std::vector<uint32_t> &idx, int W, int H,
                         std::vector<std::vector<cv::Point>> &contours)
{
    contours.clear();
    if (idx.empty())
        return;
    if (W != w_ || H != h_) {
        w_ = W;
        h_ = H;
        plane_.assign((size_t)(W + 2) * (H + 2), 0);
    }
    // No sort: the raster scan only needs, per image row, the x-span that contains foreground.  One pass
    // marks the pixels and records [min x, max x] per touched row; rows are then visited in increasing y.
    const int step = W + 2;
    signed char *img = plane_.data();
    int yMin = H, yMax = -1;
    for (uint32_t i : idx) {
        const int y = (int)(i / W);
        yMin = std::min(yMin, y);
        yMax = std::max(yMax, y);
    }
    rowLo_.assign((size_t)(yMax - yMin + 1), W);
    rowHi_.assign((size_t)(yMax - yMin + 1), -1);
    for (uint32_t i : idx) {
        const int y = (int)(i / W), x = (int)(i % W);
        img[(size_t)(y + 1) * step + (x + 1)] = 1;
        rowLo_[y - yMin] = std::min(rowLo_[y - yMin], x);
        rowHi_[y - yMin] = std::max(rowHi_[y - yMin], x);
    }

    std::vector<signed char> codes;
    std::vector<std::vector<cv::Point>> found; // discovery order
    for (int y = yMin; y <= yMax; ++y) {
        if (rowHi_[y - yMin] < 0)
            continue;
        const int xFirst = rowLo_[y - yMin] + 1, xLast = rowHi_[y - yMin] + 1; // padded coordinates
        signed char *row = img + (size_t)(y + 1) * step;
        int lnbd = 0, prev = 0;
        for (int x = xFirst; x <= xLast + 1; ++x) {
            int p = row[x];
            if (p == prev)
                continue;
            if (prev == 0 && p == 1) {
                if (!(row[lnbd] > 0)) { // not inside an already traced outer border
                    traceBorder((size_t)(y + 1) * step + x, codes);
                    found.emplace_back();
                    approxChainTC89L1(cv::Point(x - 1, y), codes, found.back());
                    p = row[x];
                }
            } else if (p == 0 && prev >= 1) {
                if (prev & -2)
                    lnbd = x - 1; // a hole would start here; RETR_EXTERNAL never follows it
            }
            prev = p;
            if (prev & -2)
                lnbd = x;
        }
    }
    for (uint32_t i : idx)
        img[(size_t)(i / W + 1) * step + (i % W + 1)] = 0;
    contours.assign(found.rbegin(), found.rend()); // OpenCV returns the last discovered first
}

// Teh-Chin dominant point detection with the L1 (1-curvature) measure, as OpenCV's
// CHAIN_APPROX_TC89_L1 performs it on the Freeman chain of a traced border.
void approxChainTC89L1(cv::Point origin, const std::vector<signed char> &codes, std::vector<cv::Point> &out)
{
    static const int kTurn[15] = {1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1};
    out.clear();
    const int len = (int)codes.size();
    if (len == 0) {
        out.push_back(origin);
        return;
    }
    struct Node {
        cv::Point pt;
        int k, s, next;
    };
    std::vector<Node> a((size_t)len + 8);
    const int HEAD = len + 7; // list head lives in the spare tail of the array
    a[HEAD].next = -1;
    int tail = HEAD;
    {
        cv::Point pt = origin;
        int prevCode = codes[len - 1];
        for (int i = 0; i < len; ++i) {
            const int c = codes[i];
            a[i].pt = pt;
            a[i].s = kTurn[c - prevCode + 7];
            a[i].k = 0;
            a[i].next = -1;
            if (a[i].s != 0) {
                a[tail].next = i;
                tail = i;
            }
            pt.x += kDx[c];
            pt.y += kDy[c];
            prevCode = c;
        }
        a[tail].next = -1;
    }
    if (a[HEAD].next < 0) {
        out.push_back(origin);
        return;
    }
    auto wrapDown = [len](int i) { return i < 0 ? i + len : i; };
    auto wrapUp = [len](int i) { return i >= len ? i - len : i; };

    // support regions
    for (int cur = a[HEAD].next; cur >= 0; cur = a[cur].next) {
        const cv::Point p0 = a[cur].pt;
        int k, l = 0, dNum = 0;
        for (k = 1;; ++k) {
            const int i1 = wrapDown(cur - k), i2 = wrapUp(cur + k);
            const int dx = a[i2].pt.x - a[i1].pt.x, dy = a[i2].pt.y - a[i1].pt.y;
            const int lk = dx * dx + dy * dy;
            const int dkNum = (p0.x - a[i1].pt.x) * dy - (p0.y - a[i1].pt.y) * dx;
            const float d = (float)(((double)dNum) * lk - ((double)dkNum) * l);
            int32_t bits;
            std::memcpy(&bits, &d, sizeof bits);
            if (k > 1 && (l >= lk || (dNum > 0 && bits <= 0) || (dNum < 0 && bits >= 0)))
                break;
            dNum = dkNum;
            l = lk;
            if (k >= len) {
                ++k;
                break;
            }
        }
        a[cur].k = k - 1;
    }
    // non-maxima suppression inside half the support region
    for (int prev = HEAD, cur = a[HEAD].next; cur >= 0;) {
        const int half = a[cur].k >> 1, s = a[cur].s;
        int j = 1;
        for (; j <= half; ++j) {
            if (a[wrapDown(cur - j)].s > s)
                break;
            if (a[wrapUp(cur + j)].s > s)
                break;
        }
        const int nxt = a[cur].next;
        if (j <= half) {
            a[prev].next = nxt;
            a[cur].s = 0;
        } else
            prev = cur;
        cur = nxt;
    }
    // drop weak points whose support region has length one
    for (int prev = HEAD, cur = a[HEAD].next; cur >= 0;) {
        const int nxt = a[cur].next;
        bool drop = false;
        if (a[cur].k == 1) {
            const int s = a[cur].s;
            drop = s <= a[wrapDown(cur - 1)].s || s <= a[wrapUp(cur + 1)].s;
        }
        if (drop) {
            a[prev].next = nxt;
            a[cur].s = 0;
        } else
            prev = cur;
        cur = nxt;
    }
    // clean runs of adjacent survivors (L1 variant)
    bool allSurvived = false;
    if (a[0].s != 0 && a[len - 1].s != 0) { // a run wraps around the start of the chain
        int i1 = 1;
        for (; i1 < len && a[i1].s != 0; ++i1)
            a[i1 - 1].s = 0;
        if (i1 == len)
            allSurvived = true;
        else {
            --i1;
            int i2 = len - 2;
            for (; i2 > 0 && a[i2].s != 0; --i2) {
                a[i2].next = -1;
                a[i2 + 1].s = 0;
            }
            ++i2;
            if (i1 == 0 && i2 == len - 1) { // only two points in the run
                i1 = a[0].next;
                a[len] = a[0];
                a[len].next = -1;
                a[len - 1].next = len;
            }
            a[HEAD].next = i1;
        }
    }
    if (!allSurvived) {
        int first = HEAD, prev = HEAD, run = 1;
        for (int cur = a[HEAD].next; cur >= 0;) {
            const int nxt = a[cur].next;
            if (nxt < 0 || nxt - cur != 1) {
                if (run >= 2) {
                    if (run == 2) {
                        const int s1 = a[prev].s, s2 = a[cur].s;
                        if (s1 > s2 || (s1 == s2 && a[prev].k <= a[cur].k))
                            a[prev].next = nxt; // second of the couple goes
                        else
                            a[first].next = cur; // first of the couple goes
                    } else
                        a[a[first].next].next = cur; // keep only the ends of a longer run
                }
                first = cur;
                run = 1;
            } else
                ++run;
            prev = cur;
            cur = nxt;
        }
    }
    for (int cur = a[HEAD].next; cur >= 0; cur = a[cur].next)
        out.push_back(a[cur].pt);
}

void bestMatchFromTerms(const unsigned long long *num, const unsigned long long *wsum2, int rw, int rh,
                        const cv::Mat &templ, float &bx, float &by)
{
    const size_t n = (size_t)rw * rh;
    // template norm the way cv::matchTemplate derives it (meanStdDev, then sqrt(sdv^2 + mean^2) / sqrt(1/N))
    const double N = (double)templ.total();
    double sum = 0, sq = 0;
    for (size_t i = 0; i < templ.total(); ++i) {
        const double v = templ.data[i];
        sum += v;
        sq += v * v;
    }
    const double invArea = 1. / N, mean = sum * invArea;
    const double var = sq * invArea - mean * mean;
    const double sdv = std::sqrt(var > 0 ? var : 0);
    double templNorm = std::sqrt(sdv * sdv + mean * mean);
    templNorm /= std::sqrt(invArea);
    std::vector<float> res(n);
    for (size_t i = 0; i < n; ++i) {
        double v = (double)(float)(double)num[i]; // the correlation plane is CV_32F
        const double w2 = (double)wsum2[i];
        const double lim = 10 * FLT_EPSILON * w2;
        const double t = (w2 <= std::min(0.5, lim)) ? 0 : std::sqrt(w2) * templNorm;
        if (std::fabs(v) < t)
            v /= t;
        else if (std::fabs(v) < t * 1.125)
            v = v > 0 ? 1 : -1;
        else
            v = 0;
        res[i] = (float)v;
    }
    double smin = res[0], smax = res[0];
    for (size_t i = 1; i < n; ++i) {
        smin = std::min(smin, (double)res[i]);
        smax = std::max(smax, (double)res[i]);
    }
    const double scale = (smax - smin > DBL_EPSILON) ? 1. / (smax - smin) : 0;
    const float a = (float)scale, b = (float)(0.0 - smin * scale);
    size_t best = 0;
    for (size_t i = 0; i < n; ++i) {
        const float v = res[i] * a;
        res[i] = v + b;
        if (res[i] > res[best])
            best = i;
    }
    // strict '>' above keeps the first maximum; the loop order is raster order like cv::minMaxLoc
    const int mx = (int)(best % rw), my = (int)(best / rw);
    float sx = 0.f, sy = 0.f, mass = 0.f;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j) {
            const int x = mx + i, y = my + j;
            if (x < 0 || y < 0 || x >= rw || y >= rh)
                continue; // unchecked upstream
            const float pv = res[(size_t)y * rw + x];
            const float px = (float)x * pv, py = (float)y * pv;
            sx = sx + px;
            sy = sy + py;
            mass = mass + pv;
        }
    const float inv = (float)(1.0 / mass);
    bx = sx * inv;
    by = sy * inv;
}

cv::Rect boundingRectOf(const std::vector<cv::Point> &pts)
{
    if (pts.empty())
        return cv::Rect();
    int x0 = pts[0].x, x1 = pts[0].x, y0 = pts[0].y, y1 = pts[0].y;
    for (const cv::Point &p : pts) {
        x0 = std::min(x0, p.x);
        x1 = std::max(x1, p.x);
        y0 = std::min(y0, p.y);
        y1 = std::max(y1, p.y);
    }
    return cv::Rect(x0, y0, x1 - x0 + 1, y1 - y0 + 1);
}

double contourAreaOf(const std::vector<cv::Point> &pts)
{
    if (pts.empty())
        return 0.;
    double a = 0;
    float px = (float)pts.back().x, py = (float)pts.back().y;
    for (const cv::Point &p : pts) {
        const float qx = (float)p.x, qy = (float)p.y;
        a += (double)px * qy - (double)py * qx;
        px = qx;
        py = qy;
    }
    return std::fabs(a * 0.5);
}

cv::Moments momentsOf(const std::vector<cv::Point> &pts)
{
    cv::Moments m;
    if (pts.empty())
        return m;
    double a00 = 0, a10 = 0, a01 = 0;
    double xp = pts.back().x, yp = pts.back().y;
    for (const cv::Point &p : pts) {
        const double x = p.x, y = p.y;
        const double cross = xp * y - x * yp;
        a00 += cross;
        a10 += cross * (xp + x);
        a01 += cross * (yp + y);
        xp = x;
        yp = y;
    }
    if (std::fabs(a00) > FLT_EPSILON) {
        const double half = a00 > 0 ? 0.5 : -0.5;
        const double sixth = a00 > 0 ? 0.16666666666666666666666666666667 : -0.16666666666666666666666666666667;
        m.m00 = a00 * half;
        m.m10 = a10 * sixth;
        m.m01 = a01 * sixth;
    }
    return m;
}

} // namespace abub
