/*
 * abub_oracle.c -- CPU ORACLE (test infrastructure only, see abub_oracle.h).
 *
 * Plain-C restatement of the AutoBub3hs hot path.  Reference citations are relative to
 * /root/reference (picoexperiment/AutoBub3hs @ 2024-11-25).  OpenCV primitives are restated from
 * OpenCV's published algorithms (imgproc: smooth/box_filter/thresh/contours/approx/moments/shapedescr),
 * see oracle/README.md for the list and for what is "parity unpinned".
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the reference is built -O0).
 * Deliberately written as straightforward full-image, multi-pass code (one pass and one temporary
 * per OpenCV call of the reference) -- it is the checker, not the thing being optimised.
 */
#include "abub_oracle.h"

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================= */
/* helpers                                                                                 */
/* ======================================================================================= */

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    }
    return p;
}

static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ======================================================================================= */
/* Trainer                                                                                 */
/* ======================================================================================= */

/* Trainer::CalculateMeanSigmaImageVector, Trainer.cpp:144-216.
 * float32 Welford in frame order; divide by (processingFrame+1) converted to float (:186);
 * sigma = (int)sqrt(M2/(N-1)) (:191-195); mean stored by float->uchar truncation (:196).
 * N==1 gives 0/0 = NaN whose int cast is UB in the reference; x86 cvttss2si yields INT_MIN
 * whose low byte is 0, so the oracle defines that case as 0 (documented in DESIGN.md). */
void orc_welford(const uint8_t *slab, int N, int W, int H, uint8_t *mu, uint8_t *sigma)
{
    size_t P = (size_t)W * H;
    for (size_t px = 0; px < P; px++) {
        float mean = 0.f, m2 = 0.f;
        for (int k = 0; k < N; k++) {
            float x = (float)slab[(size_t)k * P + px];
            float delta = x - mean;
            float q = delta / (float)(size_t)(k + 1);
            mean = mean + q;
            float t = x - mean;
            float pr = delta * t;
            m2 = m2 + pr;
        }
        float var = m2 / (float)(size_t)(N - 1);
        float sd = sqrtf(var);
        int isd = (sd != sd) ? 0 : (int)sd;
        sigma[px] = (uint8_t)isd;
        float mf = mean;
        mu[px] = (uint8_t)(int)mf;
    }
}

/* cv::calcHist with nbins in {16,128,256} over [0,256): bin = v >> shift. */
static void hist_bins(const uint8_t *img, size_t P, int shift, uint32_t *h, int nb)
{
    memset(h, 0, sizeof(uint32_t) * nb);
    for (size_t i = 0; i < P; i++)
        h[img[i] >> shift]++;
}

/* Shannon entropy as in Trainer::calculateEntropyFrame (Trainer.cpp:365-373):
 * hist (float) / (rows*cols) is a cv::MatExpr scale by alpha = 1.0/(rows*cols) executed by
 * convertTo on a CV_32F matrix, i.e. p = h * (float)alpha in float32; then
 * ImgEntropy -= p * log2(p) with float operands (LBP/lbp.hpp:11 `using namespace std` selects the
 * float overload of log2) accumulated in float. */
static float entropy_from_hist(const uint32_t *h, int nb, size_t P)
{
    float scale = (float)(1.0 / (double)(int)P);
    float e = 0.f;
    for (int i = 0; i < nb; i++) {
        float p = (float)h[i] * scale;
        if (p != 0) {
            float t = p * log2f(p);
            e = e - t;
        }
    }
    return e;
}

float orc_entropy16(const uint8_t *img, int W, int H)
{
    uint32_t h[16];
    hist_bins(img, (size_t)W * H, 4, h, 16);
    return entropy_from_hist(h, 16, (size_t)W * H);
}

float orc_entropy128(const uint8_t *img, int W, int H)
{
    uint32_t h[128];
    hist_bins(img, (size_t)W * H, 1, h, 128);
    return entropy_from_hist(h, 128, (size_t)W * H);
}

/* Trainer.cpp:279 `TestingForEntropyArray[1]-TestingForEntropyArray[0]` (u8 saturating) then :280. */
float orc_pair_entropy16(const uint8_t *f1, const uint8_t *f0, int W, int H)
{
    size_t P = (size_t)W * H;
    uint8_t *d = (uint8_t *)malloc(P);
    for (size_t i = 0; i < P; i++)
        d[i] = sat_u8((int)f1[i] - (int)f0[i]);
    float e = orc_entropy16(d, W, H);
    free(d);
    return e;
}

/* ======================================================================================= */
/* ProcessFrame                                                                            */
/* ======================================================================================= */

/* cv::GaussianBlur(src, dst, Size(5,5), 0) on u8: fixed kernel [1 4 6 4 1]/16 per axis,
 * exact 8-bit fixed point: dst = (sum_ij w_i w_j src + 128) >> 8, BORDER_REFLECT_101. */
static void gauss5_u8(const uint8_t *src, int w, int h, uint8_t *dst)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = -2; j <= 2; j++)
                s += k[j + 2] * src[(size_t)y * w + reflect101(x + j, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -2; i <= 2; i++)
                s += k[i + 2] * tmp[(size_t)reflect101(y + i, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)((s + 128) >> 8);
        }
    free(tmp);
}

/* AnalyzerUnit::ProcessFrame ROI overload, AnalyzerUnit.cpp:346-377:
 *   diff_frame = zeros                                            (:349)
 *   pos = cur(ROI) - ref(ROI) - 6*sigma(ROI)   (saturating u8)    (:351)
 *   neg = ref(ROI) - cur(ROI) - 6*sigma(ROI)                      (:352)
 *   GaussianBlur 5x5 of each, in place                            (:359-360)
 *   absdiff(pos, neg) -> diff_frame(ROI)                          (:370)
 * The MatExpr a-b-6c evaluates (a-b) saturated, then subtracts 6c saturated: max(0, a-b-6c). */
void orc_process_frame_roi(const uint8_t *cur, const uint8_t *ref, const uint8_t *sigma, int W,
                           int H, int rx, int ry, int rw, int rh, uint8_t *D)
{
    memset(D, 0, (size_t)W * H);
    if (rw <= 0 || rh <= 0)
        return;
    size_t n = (size_t)rw * rh;
    uint8_t *pos = (uint8_t *)malloc(n), *neg = (uint8_t *)malloc(n);
    uint8_t *posb = (uint8_t *)malloc(n), *negb = (uint8_t *)malloc(n);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            size_t g = (size_t)(ry + y) * W + (rx + x);
            int c = cur[g], r = ref[g], s6 = 6 * (int)sigma[g];
            int t1 = sat_u8(c - r);
            int t2 = sat_u8(r - c);
            pos[(size_t)y * rw + x] = sat_u8(t1 - s6);
            neg[(size_t)y * rw + x] = sat_u8(t2 - s6);
        }
    gauss5_u8(pos, rw, rh, posb);
    gauss5_u8(neg, rw, rh, negb);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            int a = posb[(size_t)y * rw + x], b = negb[(size_t)y * rw + x];
            D[(size_t)(ry + y) * W + (rx + x)] = (uint8_t)(a > b ? a - b : b - a);
        }
    free(pos);
    free(neg);
    free(posb);
    free(negb);
}

/* Full-frame overload, AnalyzerUnit.cpp:341-344. */
void orc_process_frame(const uint8_t *cur, const uint8_t *ref, const uint8_t *sigma, int W, int H,
                       uint8_t *D)
{
    orc_process_frame_roi(cur, ref, sigma, W, H, 0, 0, W, H, D);
}

void orc_hist256(const uint8_t *img, size_t P, uint32_t hist[256])
{
    hist_bins(img, P, 0, hist, 256);
}

/* L3Localizer::CalculatePostTriggerFrameParams, L3Localizer.cpp:779-785:
 *   A = absdiff(frame, mu); O = A - 6*sigma (saturating); cv::blur(O, O, Size(3,3)).
 * cv::blur on u8 = 3x3 box sum S, normalised and rounded: round(S/9) = (S+4)/9 (9 is odd: no ties),
 * BORDER_REFLECT_101. */
void orc_posttrig_frame(const uint8_t *frame, const uint8_t *mu, const uint8_t *sigma, int W, int H,
                        uint8_t *O)
{
    size_t P = (size_t)W * H;
    uint8_t *o = (uint8_t *)malloc(P);
    for (size_t i = 0; i < P; i++) {
        int a = (int)frame[i] - (int)mu[i];
        if (a < 0)
            a = -a;
        o[i] = sat_u8(a - 6 * (int)sigma[i]);
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int s = 0;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++)
                    s += o[(size_t)reflect101(y + i, H) * W + reflect101(x + j, W)];
            O[(size_t)y * W + x] = (uint8_t)((s + 4) / 9);
        }
    free(o);
}

/* ======================================================================================= */
/* threshold / Otsu                                                                        */
/* ======================================================================================= */

/* OpenCV getThreshVal_Otsu_8u restated (imgproc/thresh.cpp): all in double. */
int orc_otsu(const uint32_t h[256], size_t P)
{
    double mu = 0, scale = 1. / (double)P;
    for (int i = 0; i < 256; i++)
        mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0;
    int max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        double q2 = 1. - q1;
        double mn = q1 < q2 ? q1 : q2, mx = q1 > q2 ? q1 : q2;
        if (mn < FLT_EPSILON || mx > 1. - FLT_EPSILON)
            continue;
        mu1 = (mu1 + i * p_i) / q1;
        double mu2 = (mu - q1 * mu1) / q2;
        double sg = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sg > max_sigma) {
            max_sigma = sg;
            max_val = i;
        }
    }
    return max_val;
}

/* cv::threshold(src, t, thr, 255, THRESH_TOZERO) then cv::threshold(t, t, 0, 255, BINARY|OTSU)
 * (L3Localizer.cpp:252-254, :365-367, :786-787). */
int orc_binarize(const uint8_t *img, int W, int H, int thr, uint8_t *mask)
{
    size_t P = (size_t)W * H;
    uint8_t *t = (uint8_t *)calloc(P, 1);
    for (size_t i = 0; i < P; i++)
        t[i] = img[i] > thr ? img[i] : 0;
    uint32_t h[256];
    orc_hist256(t, P, h);
    int T = orc_otsu(h, P);
    for (size_t i = 0; i < P; i++)
        mask[i] = t[i] > T ? 255 : 0;
    free(t);
    return T;
}

/* ======================================================================================= */
/* findContours(RETR_EXTERNAL, CHAIN_APPROX_TC89_L1)                                       */
/* ======================================================================================= */

/* Freeman code deltas used by OpenCV (y grows downwards): 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE */
static const int code_dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int code_dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

typedef struct {
    orc_point pt;
    int k; /* support region */
    int s; /* 1-curvature */
    int next; /* index of next surviving point, -1 = end */
} tc_pt;

/* OpenCV icvApproximateChainTC89 restated for method = CV_CHAIN_APPROX_TC89_L1 (imgproc/approx.cpp).
 * codes[i] is the Freeman step from point i to point i+1; origin is point 0.
 * Output: malloc'd vertex list. */
static orc_point *tc89_l1(orc_point origin, const signed char *codes, int count, int *npts_out)
{
    static const int abs_diff[15] = {1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1};
    if (count == 0) {
        orc_point *o = (orc_point *)malloc(sizeof(orc_point));
        o[0] = origin;
        *npts_out = 1;
        return o;
    }
    /* array has count+8 slots like the AutoBuffer of the original ("move to the end" case) */
    tc_pt *a = (tc_pt *)calloc((size_t)count + 8, sizeof(tc_pt));
    const int HEAD = -2; /* virtual list head ("temp") */
    int head_next = -1;
    int cur = HEAD;
#define SET_NEXT(idx, val)        \
    do {                          \
        if ((idx) == HEAD)        \
            head_next = (val);    \
        else                      \
            a[(idx)].next = (val); \
    } while (0)
#define GET_NEXT(idx) ((idx) == HEAD ? head_next : a[(idx)].next)

    /* Pass 0: restore points, drop zero 1-curvature points from the list. */
    orc_point pt = origin;
    int prev_code = codes[count - 1];
    for (int i = 0; i < count; i++) {
        int code = codes[i];
        int s = abs_diff[code - prev_code + 7];
        a[i].pt = pt;
        a[i].s = s;
        a[i].next = -1;
        if (s != 0) {
            SET_NEXT(cur, i);
            cur = i;
        }
        pt.x += code_dx[code];
        pt.y += code_dy[code];
        prev_code = code;
    }
    SET_NEXT(cur, -1);
    int len = count;
    int i1, i2;

    if (head_next < 0) {
        /* cannot happen for a closed 8-connected border of >= 1 step (direction must change);
         * the original asserts.  Return the origin. */
        free(a);
        orc_point *o = (orc_point *)malloc(sizeof(orc_point));
        o[0] = origin;
        *npts_out = 1;
        return o;
    }

    /* Pass 1: support region of every remaining point. */
    for (cur = head_next; cur >= 0; cur = a[cur].next) {
        int i = cur, k, l = 0, d_num = 0;
        orc_point pt0 = a[i].pt;
        for (k = 1;; k++) {
            i1 = i - k;
            i1 += i1 < 0 ? len : 0;
            i2 = i + k;
            i2 -= i2 >= len ? len : 0;
            int dx = a[i2].pt.x - a[i1].pt.x;
            int dy = a[i2].pt.y - a[i1].pt.y;
            int lk = dx * dx + dy * dy;
            int dk_num = (pt0.x - a[i1].pt.x) * dy - (pt0.y - a[i1].pt.y) * dx;
            /* original: d.f = (float)(((double)d_num)*lk - ((double)dk_num)*l); tests the sign
             * of the float through its integer bit pattern (d.i <= 0  <=>  d.f <= +0 or negative,
             * with -0.0f having i < 0). */
            float df = (float)(((double)d_num) * lk - ((double)dk_num) * l);
            int32_t di;
            memcpy(&di, &df, 4);
            if (k > 1 && (l >= lk || ((d_num > 0 && di <= 0) || (d_num < 0 && di >= 0))))
                break;
            d_num = dk_num;
            l = lk;
            if (k >= len) { /* safety: original asserts k <= len */
                k++;
                break;
            }
        }
        a[cur].k = --k;
    }

    /* Pass 2: non-maxima suppression. */
    {
        int prev = HEAD;
        cur = head_next;
        while (cur >= 0) {
            int k2 = a[cur].k >> 1;
            int s = a[cur].s, i = cur, j;
            for (j = 1; j <= k2; j++) {
                i2 = i - j;
                i2 += i2 < 0 ? len : 0;
                if (a[i2].s > s)
                    break;
                i2 = i + j;
                i2 -= i2 >= len ? len : 0;
                if (a[i2].s > s)
                    break;
            }
            int nxt = a[cur].next;
            if (j <= k2) {
                SET_NEXT(prev, nxt);
                a[cur].s = 0;
            } else
                prev = cur;
            cur = nxt;
        }
    }

    /* Pass 3: remove non-dominant points with 1-length support region. */
    {
        int prev = HEAD;
        cur = head_next;
        while (cur >= 0) {
            int nxt = a[cur].next;
            if (a[cur].k == 1) {
                int s = a[cur].s, i = cur;
                i1 = i - 1;
                i1 += i1 < 0 ? len : 0;
                i2 = i + 1;
                i2 -= i2 >= len ? len : 0;
                if (s <= a[i1].s || s <= a[i2].s) {
                    SET_NEXT(prev, nxt);
                    a[cur].s = 0;
                } else
                    prev = cur;
            } else
                prev = cur;
            cur = nxt;
        }
    }

    /* Pass 4: clean remaining couples of adjacent points (TC89_L1 only). */
    int all_survived = 0;
    if (head_next >= 0 && a[0].s != 0 && a[len - 1].s != 0) { /* wrap-around run */
        for (i1 = 1; i1 < len && a[i1].s != 0; i1++)
            a[i1 - 1].s = 0;
        if (i1 == len)
            all_survived = 1;
        else {
            i1--;
            for (i2 = len - 2; i2 > 0 && a[i2].s != 0; i2--) {
                a[i2].next = -1;
                a[i2 + 1].s = 0;
            }
            i2++;
            if (i1 == 0 && i2 == len - 1) { /* only two points */
                i1 = a[0].next;
                a[len] = a[0]; /* move to the end */
                a[len].next = -1;
                a[len - 1].next = len;
            }
            head_next = i1;
        }
    }
    if (!all_survived && head_next >= 0) {
        int first = HEAD, prev = HEAD;
        int cnt = 1;
        cur = head_next;
        while (cur >= 0) {
            int nxt = a[cur].next;
            if (nxt < 0 || nxt - cur != 1) {
                if (cnt >= 2) {
                    if (cnt == 2) {
                        int s1 = a[prev].s, s2 = a[cur].s;
                        if (s1 > s2 || (s1 == s2 && a[prev].k <= a[cur].k))
                            SET_NEXT(prev, nxt); /* remove second */
                        else
                            SET_NEXT(first, cur); /* remove first */
                    } else {
                        int fn = GET_NEXT(first);
                        a[fn].next = cur;
                    }
                }
                first = cur;
                cnt = 1;
            } else
                cnt++;
            prev = cur;
            cur = nxt;
        }
    }

    /* gather */
    int n = 0;
    for (cur = head_next; cur >= 0; cur = a[cur].next)
        n++;
    orc_point *out = (orc_point *)malloc(sizeof(orc_point) * (size_t)(n > 0 ? n : 1));
    n = 0;
    for (cur = head_next; cur >= 0; cur = a[cur].next)
        out[n++] = a[cur].pt;
    *npts_out = n;
    free(a);
    return out;
#undef SET_NEXT
#undef GET_NEXT
}

/* Suzuki-Abe outer-border following as OpenCV's icvFetchContour with CV_CHAIN_CODE
 * (imgproc/contours.cpp), on a signed-char image with a 1-pixel zero frame (step = padded width).
 * Marks visited border pixels with nbd=2 (left/inner side) or 2|-128 (right edge).
 * Returns the chain codes (malloc'd) of the border starting at p0. */
static signed char *follow_border(signed char *img, int step, size_t p0, int *ncodes)
{
    const signed char nbd = 2;
    int deltas[16];
    for (int i = 0; i < 8; i++)
        deltas[i] = deltas[i + 8] = code_dy[i] * step + code_dx[i];
    size_t cap = 64, n = 0;
    signed char *codes = (signed char *)malloc(cap);
    signed char *i0 = img + p0, *i1, *i3, *i4 = NULL;
    int s, s_end;
    s_end = s = 4; /* outer border: start looking from the west neighbour, clockwise */
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);
    if (s == s_end) { /* single pixel */
        *i0 = (signed char)(nbd | -128);
    } else {
        i3 = i0;
        for (;;) {
            s_end = s;
            while (s < 15) {
                i4 = i3 + deltas[++s];
                if (*i4 != 0)
                    break;
            }
            s &= 7;
            /* "right" bound: the east neighbour was examined and found empty */
            if ((unsigned)(s - 1) < (unsigned)s_end)
                *i3 = (signed char)(nbd | -128);
            else if (*i3 == 1)
                *i3 = nbd;
            if (n == cap) {
                cap *= 2;
                codes = (signed char *)realloc(codes, cap);
            }
            codes[n++] = (signed char)s;
            if (i4 == i0 && i3 == i1)
                break;
            i3 = i4;
            s = (s + 4) & 7;
        }
    }
    *ncodes = (int)n;
    return codes;
}

/* cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_TC89_L1) restated from OpenCV >= 3.2's
 * cvFindContours_Impl / cvFindNextContour (imgproc/contours.cpp):
 *  - non-zero -> 1 on a copy with a 1-pixel zero frame (image-border pixels take part);
 *  - raster scan; an outer-border start is a 0 -> 1 transition; in RETR_EXTERNAL mode it is kept
 *    only when the last marked pixel met on this row (lnbd) does not carry a positive mark
 *    (`img0[lnbd] > 0` => we are inside an already traced outer border => skip);
 *  - hole borders are never traced in this mode;
 *  - contours come out in reverse discovery order (each new contour is linked as the first child
 *    of the frame, cvInsertNodeIntoTree). */
orc_contours *orc_find_contours(const uint8_t *mask, int W, int H)
{
    int step = W + 2, ph = H + 2;
    signed char *img = (signed char *)calloc((size_t)step * ph, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            img[(size_t)(y + 1) * step + (x + 1)] = mask[(size_t)y * W + x] ? 1 : 0;

    size_t cap = 16;
    orc_contours *out = (orc_contours *)malloc(sizeof(orc_contours));
    out->n = 0;
    out->c = (orc_contour *)malloc(sizeof(orc_contour) * cap);

    for (int y = 1; y < ph - 1; y++) {
        signed char *row = img + (size_t)y * step;
        int lnbd_x = 0;
        int prev = 0;
        for (int x = 1; x < step - 1; x++) {
            int p = row[x];
            if (p == prev)
                continue;
            if (prev == 0 && p == 1) {
                /* candidate outer border */
                if (!(row[lnbd_x] > 0)) {
                    int ncodes;
                    signed char *codes = follow_border(img, step, (size_t)y * step + x, &ncodes);
                    orc_point origin = {x - 1, y - 1};
                    int npts;
                    orc_point *pts = tc89_l1(origin, codes, ncodes, &npts);
                    free(codes);
                    if ((size_t)out->n == cap) {
                        cap *= 2;
                        out->c = (orc_contour *)realloc(out->c, sizeof(orc_contour) * cap);
                    }
                    out->c[out->n].npts = npts;
                    out->c[out->n].pts = pts;
                    out->c[out->n].nchain = ncodes;
                    out->n++;
                    p = row[x];
                }
            } else {
                /* hole check: a hole border would start at p == 0 && prev >= 1 */
                if (!(p != 0 || prev < 1)) {
                    if (prev & -2)
                        lnbd_x = x - 1;
                    /* RETR_EXTERNAL: holes are skipped */
                }
            }
            prev = p;
            if (prev & -2)
                lnbd_x = x;
        }
    }
    free(img);
    /* reverse discovery order */
    for (int i = 0, j = out->n - 1; i < j; i++, j--) {
        orc_contour t = out->c[i];
        out->c[i] = out->c[j];
        out->c[j] = t;
    }
    return out;
}

void orc_contours_free(orc_contours *c)
{
    if (!c)
        return;
    for (int i = 0; i < c->n; i++)
        free(c->c[i].pts);
    free(c->c);
    free(c);
}
int orc_contours_count(const orc_contours *c) { return c->n; }
int orc_contour_npts(const orc_contours *c, int i) { return c->c[i].npts; }
int orc_contour_nchain(const orc_contours *c, int i) { return c->c[i].nchain; }
void orc_contour_points(const orc_contours *c, int i, int *xy)
{
    for (int k = 0; k < c->c[i].npts; k++) {
        xy[2 * k] = c->c[i].pts[k].x;
        xy[2 * k + 1] = c->c[i].pts[k].y;
    }
}

/* cv::boundingRect, cv::contourArea, cv::moments(contour) (Green's formula), restated from
 * OpenCV imgproc shapedescr.cpp / moments.cpp; then the BubbleImageFrame fields of
 * L3Localizer.cpp:400-419 (genesis, with the m00==0 fallback) / :817-823 (tracking, no fallback). */
void orc_blob_from_contour(const orc_point *pts, int n, int genesis_fallback, orc_blob *b)
{
    memset(b, 0, sizeof(*b));
    if (n == 0)
        return;
    int xmin = pts[0].x, xmax = pts[0].x, ymin = pts[0].y, ymax = pts[0].y;
    for (int i = 1; i < n; i++) {
        if (pts[i].x < xmin) xmin = pts[i].x;
        if (pts[i].x > xmax) xmax = pts[i].x;
        if (pts[i].y < ymin) ymin = pts[i].y;
        if (pts[i].y > ymax) ymax = pts[i].y;
    }
    b->x = xmin;
    b->y = ymin;
    b->w = xmax - xmin + 1;
    b->h = ymax - ymin + 1;

    /* contourArea: points converted to float, products in double */
    {
        double a00 = 0;
        float px = (float)pts[n - 1].x, py = (float)pts[n - 1].y;
        for (int i = 0; i < n; i++) {
            float qx = (float)pts[i].x, qy = (float)pts[i].y;
            a00 += (double)px * qy - (double)py * qx;
            px = qx;
            py = qy;
        }
        a00 *= 0.5;
        b->area = fabs(a00);
    }
    b->radius = sqrt(b->area / 3.14159);

    /* moments (spatial, order <= 1) */
    {
        double a00 = 0, a10 = 0, a01 = 0;
        double xi_1 = pts[n - 1].x, yi_1 = pts[n - 1].y;
        for (int i = 0; i < n; i++) {
            double xi = pts[i].x, yi = pts[i].y;
            double dxy = xi_1 * yi - xi * yi_1;
            double xii_1 = xi_1 + xi, yii_1 = yi_1 + yi;
            a00 += dxy;
            a10 += dxy * xii_1;
            a01 += dxy * yii_1;
            xi_1 = xi;
            yi_1 = yi;
        }
        if (fabs(a00) > FLT_EPSILON) {
            double db1_2, db1_6;
            if (a00 > 0) {
                db1_2 = 0.5;
                db1_6 = 0.16666666666666666666666666666667;
            } else {
                db1_2 = -0.5;
                db1_6 = -0.16666666666666666666666666666667;
            }
            b->m00 = a00 * db1_2;
            b->m10 = a10 * db1_6;
            b->m01 = a01 * db1_6;
        }
    }
    if (!genesis_fallback || b->m00 > 0) {
        /* cv::Point2f(m10/m00, m01/m00): double division narrowed to float (0/0 -> NaN) */
        b->cx = (float)(b->m10 / b->m00);
        b->cy = (float)(b->m01 / b->m00);
    } else {
        double x = 0, y = 0, nn = 0;
        for (int i = 0; i < n; i++) {
            x += pts[i].x;
            y += pts[i].y;
            nn++;
        }
        x /= nn;
        y /= nn;
        b->cx = (float)x;
        b->cy = (float)y;
    }
}

/* ======================================================================================= */
/* analyzer: significance, trigger, localizer, tracking                                    */
/* ======================================================================================= */

typedef struct {
    int n_desc;
    orc_blob desc[ORC_MAX_DESC + 4];
    float last_x, last_y;
    int lock;
    int n_dz;
    float dz[ORC_MAX_DESC + 4];
} orc_bubble;

typedef struct {
    int *v;
    int n, cap;
} ivec;

struct orc_analyzer {
    const uint8_t *frames;
    int F, W, H;
    const uint8_t *mu, *sigma;
    int training_set_size;
    const uint8_t *frame_ok;
    const uint8_t *fid_mask;
    int fmW, fmH;
    const uint8_t *bel_mask;
    int bmW, bmH;
    /* AnalyzerUnit public state */
    int MatTrigFrame;                      /* AnalyzerUnit.cpp:28 */
    int loc_thres;                         /* :29 */
    ivec pix_counts[256];                  /* :30 */
    int okToProceed;                       /* AnalyzerUnit.hpp:86 */
    int status;                            /* AnalyzerUnit.hpp:87 */
    orc_bubble *bubbles;
    int nbub, capbub;
    double *sig_trace;
    const uint8_t *bel_tmpl; /* bellows template or NULL */
    int tW, tH;
    int exception; /* a cv::Exception would have been thrown (ROI outside the image): status -6 */
};

void orc_analyzer_set_bellows_template(orc_analyzer *a, const uint8_t *tmpl, int tw, int th)
{
    a->bel_tmpl = tmpl;
    a->tW = tw;
    a->tH = th;
}

static void ivec_push(ivec *v, int x)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->v = (int *)realloc(v->v, sizeof(int) * v->cap);
    }
    v->v[v->n++] = x;
}

orc_analyzer *orc_analyzer_create(const uint8_t *frames, int F, int W, int H, const uint8_t *mu,
                                  const uint8_t *sigma, int training_set_size,
                                  const uint8_t *frame_ok, const uint8_t *fid_mask, int fmW,
                                  int fmH, const uint8_t *bel_mask, int bmW, int bmH)
{
    orc_analyzer *a = (orc_analyzer *)calloc(1, sizeof(*a));
    a->frames = frames;
    a->F = F;
    a->W = W;
    a->H = H;
    a->mu = mu;
    a->sigma = sigma;
    a->training_set_size = training_set_size;
    a->frame_ok = frame_ok;
    a->fid_mask = fid_mask;
    a->fmW = fmW;
    a->fmH = fmH;
    a->bel_mask = bel_mask;
    a->bmW = bmW;
    a->bmH = bmH;
    a->MatTrigFrame = 0;
    a->loc_thres = 3;
    a->okToProceed = 1;
    a->status = 0;
    a->sig_trace = (double *)malloc(sizeof(double) * (size_t)(F > 0 ? F : 1));
    for (int i = 0; i < F; i++)
        a->sig_trace[i] = NAN;
    return a;
}

void orc_analyzer_destroy(orc_analyzer *a)
{
    if (!a)
        return;
    for (int i = 0; i < 256; i++)
        free(a->pix_counts[i].v);
    free(a->bubbles);
    free(a->sig_trace);
    free(a);
}

/* CalcMean / CalcStdDev, AnalyzerUnit.cpp:514-532 on vector<int> with explicit size. */
static double calc_mean(const ivec *v, int size)
{
    double sum = 0;
    for (int i = 0; i < v->n; i++)
        sum += v->v[i];
    return sum / size;
}
static double calc_stddev(const ivec *v, double mean, int size)
{
    double sum = 0;
    for (int i = 0; i < v->n; i++) {
        /* `val*val` is int*int in the reference (:529); overflow is UB there, the -O0 x86 build wraps */
        int sq = (int)((unsigned)v->v[i] * (unsigned)v->v[i]);
        sum += sq;
    }
    return sqrt(sum / size - mean * mean);
}

/* AnalyzerUnit::calculateSignificanceFrame, AnalyzerUnit.cpp:435-504, fed with the 256-bin
 * histogram that cv::calcHist (:456) would produce (counts as float32). */
double orc_significance(orc_analyzer *a, const uint32_t hist[256], int store)
{
    double significance = 0;
    int pix_rem = a->W * a->H; /* ImageFrame.total() :458 */
    double mean, sigma;
    int first_over_3p5 = -1, max_adc = 0;
    const int loc_thres_max = 3; /* AnalyzerUnit.hpp:28 */
    for (int i = 0; i < 256; i++) {
        float binEntry = (float)hist[i];
        if (pix_rem > 0) {
            if (store)
                ivec_push(&a->pix_counts[i], (int)binEntry);
            if (i > 1) {
                int n0 = a->pix_counts[0].n;
                mean = calc_mean(&a->pix_counts[i], n0);
                sigma = calc_stddev(&a->pix_counts[i], mean, n0);
                if ((double)binEntry != mean || sigma > 0)
                    significance += ((double)binEntry - mean) / sigma;
                if (significance < 0)
                    significance = 0;
            }
            if (significance > 3.5 && first_over_3p5 < 0)
                first_over_3p5 = i;
            if (i > max_adc)
                max_adc = i;
            if (store) {
                int lt = first_over_3p5 - 1 > max_adc - 1 ? first_over_3p5 - 1 : max_adc - 1;
                if (lt < 2)
                    lt = 2;
                if (lt > loc_thres_max || a->training_set_size < 6)
                    lt = loc_thres_max;
                a->loc_thres = lt;
            }
            pix_rem = (int)((float)pix_rem - binEntry); /* `pix_rem -= binEntry` int -= float :493 */
        } else
            break;
    }
    return significance;
}

static const uint8_t *frame_ptr(const orc_analyzer *a, int i)
{
    return a->frames + (size_t)i * a->W * a->H;
}

static double sig_of_pair(orc_analyzer *a, int cur, int ref, int store, uint8_t *scratch)
{
    uint32_t h[256];
    orc_process_frame(frame_ptr(a, cur), frame_ptr(a, ref), a->sigma, a->W, a->H, scratch);
    orc_hist256(scratch, (size_t)a->W * a->H, h);
    return orc_significance(a, h, store);
}

/* AnalyzerUnit::FindTriggerFrame, AnalyzerUnit.cpp:119-324. */
void orc_find_trigger(orc_analyzer *a, int startframe)
{
    int n = a->F;
    if (n < 5) { /* :122-126 */
        a->okToProceed = 0;
        a->status = -9;
        return;
    }
    float entropyThreshold = 3.5f; /* :142 */
    float singleEntropy;
    if (startframe < 1)
        startframe = 1; /* :163 */
    if (startframe == 1) { /* :165-169 */
        for (int i = 0; i < 256; i++)
            a->pix_counts[i].n = 0;
    }
    /* :175-177 load prevFrame / prevPrevFrame (return codes ignored there) */
    int prev = startframe - 1;
    int prevprev = startframe < 2 ? prev : startframe - 2;
    a->status = -3; /* :180 */
    int twoFrameOffset = 1;
    if (a->training_set_size < 6) { /* :185-188 */
        twoFrameOffset = 0;
        entropyThreshold = (float)((double)entropyThreshold * (5 / 3.5));
    }
    for (int i = 0; i < n; i++)
        a->sig_trace[i] = NAN;
    uint8_t *scratch = (uint8_t *)malloc((size_t)a->W * a->H);
    for (int i = startframe; i < n; i++) { /* :191 */
        if (a->frame_ok && !a->frame_ok[i]) { /* :207-213 */
            a->okToProceed = 0;
            a->status = -9;
            free(scratch);
            return;
        }
        double s = sig_of_pair(a, i, twoFrameOffset ? prevprev : prev, 1, scratch); /* :218,228 */
        a->sig_trace[i] = s;
        singleEntropy = (float)s;
        if (singleEntropy > entropyThreshold && i >= 2 /* minEvalFrameNumber, hpp:26 */) { /* :254 */
            if (i != n - 1) { /* :261 */
                int numFramesCheck = 2;
                int tpp = prev, tp = i;
                double max_so_far = singleEntropy; /* :280 */
                for (int ii = 1; ii <= numFramesCheck && ii + i < n; ii++) { /* :282 */
                    int pk = i + ii;
                    singleEntropy = (float)sig_of_pair(a, pk, twoFrameOffset ? tpp : tp, 0, scratch);
                    if ((double)singleEntropy / ((double)entropyThreshold / 3.5 * 5) +
                            (double)singleEntropy / max_so_far <=
                        3)
                        break; /* :296 */
                    else if (ii == numFramesCheck) { /* :297-300 */
                        a->status = 0;
                        a->MatTrigFrame = i;
                    }
                    if ((double)singleEntropy > max_so_far)
                        max_so_far = singleEntropy; /* :302 */
                    tpp = tp;
                    tp = pk;
                }
                if (a->status == 0)
                    break; /* :307 */
            }
        }
        prevprev = prev; /* :312-313 */
        prev = i;
    }
    free(scratch);
    if (a->status == -3)
        a->okToProceed = 0; /* :317-321 */
}

/* L3Localizer::isInMask, L3Localizer.cpp:971-1012.  Out-of-range lookups (UB in the reference,
 * cv::Mat::at is unchecked) are defined as 0 here. */
static int is_in_mask(const orc_analyzer *a, const orc_blob *r, int bellows)
{
    int xpix = (int)(r->x + r->w / 2.);
    int ypix = r->y + r->h / 2;
    const uint8_t *m = bellows ? a->bel_mask : a->fid_mask;
    int mW = bellows ? a->bmW : a->fmW, mH = bellows ? a->bmH : a->fmH;
    if (!m)
        return !bellows;
    if (xpix < 0 || ypix < 0 || xpix >= mW || ypix >= mH)
        return 0;
    return m[(size_t)ypix * mW + xpix] > 0;
}

static orc_bubble *new_bubble(orc_analyzer *a, const orc_blob *g)
{
    if (a->nbub == a->capbub) {
        a->capbub = a->capbub ? a->capbub * 2 : 8;
        a->bubbles = (orc_bubble *)realloc(a->bubbles, sizeof(orc_bubble) * a->capbub);
    }
    orc_bubble *b = &a->bubbles[a->nbub++];
    memset(b, 0, sizeof(*b));
    /* bubble::bubble, bubble.cpp:34-48 */
    b->desc[0] = *g;
    b->n_desc = 1;
    b->last_x = g->cx;
    b->last_y = g->cy;
    b->lock = 1;
    return b;
}

/* cv::matchTemplate(img, templ, result, CV_TM_CCORR_NORMED) restated from OpenCV imgproc/templmatch.cpp
 * (crossCorr + common_matchTemplate).  The correlation itself is exact here (OpenCV evaluates it through a
 * DFT in float32: agreement only to float rounding, parity unpinned); the normalisation follows
 * common_matchTemplate: templNorm from meanStdDev, window energy from the double integral image,
 * the 1.125 guard band, result stored as float32. */
void orc_match_template_ccorr_normed(const uint8_t *img, int W, int H, const uint8_t *tmpl, int tw, int th,
                                     float *result)
{
    int rw = W - tw + 1, rh = H - th + 1;
    double N = (double)tw * th, sum = 0, sq = 0;
    for (int i = 0; i < tw * th; i++) {
        sum += tmpl[i];
        sq += (double)tmpl[i] * tmpl[i];
    }
    double invArea = 1. / N;
    double mean = sum * invArea;
    double var = sq * invArea - mean * mean;
    double sdv = sqrt(var > 0 ? var : 0);
    double templNorm = sdv * sdv + mean * mean; /* numType == 0: templNorm = templSum2 */
    templNorm = sqrt(templNorm);
    templNorm /= sqrt(invArea);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            uint64_t n = 0, w2 = 0;
            for (int r = 0; r < th; r++) {
                const uint8_t *ir = img + (size_t)(y + r) * W + x, *tr = tmpl + (size_t)r * tw;
                for (int c = 0; c < tw; c++) {
                    n += (uint32_t)tr[c] * ir[c];
                    w2 += (uint32_t)ir[c] * ir[c];
                }
            }
            double num = (double)(float)(double)n; /* crossCorr result is CV_32F */
            double wndSum2 = (double)w2, t;
            double diff2 = wndSum2 > 0 ? wndSum2 : 0;
            double lim = 10 * FLT_EPSILON * wndSum2;
            if (diff2 <= (0.5 < lim ? 0.5 : lim))
                t = 0;
            else
                t = sqrt(diff2) * templNorm;
            if (fabs(num) < t)
                num /= t;
            else if (fabs(num) < t * 1.125)
                num = num > 0 ? 1 : -1;
            else
                num = 0;
            result[(size_t)y * rw + x] = (float)num;
        }
}

/* L3Localizer::TrackAFeature, L3Localizer.cpp:473-543: matchTemplate, cv::normalize(0,1,NORM_MINMAX),
 * minMaxLoc, 3x3 sub-pixel centre of mass.  Neighbours outside the result matrix (unchecked upstream)
 * are skipped here. */
void orc_track_feature(const uint8_t *img, int W, int H, const uint8_t *tmpl, int tw, int th, float *bx, float *by)
{
    int rw = W - tw + 1, rh = H - th + 1;
    size_t n = (size_t)rw * rh;
    float *res = (float *)malloc(sizeof(float) * n);
    orc_match_template_ccorr_normed(img, W, H, tmpl, tw, th, res);
    /* cv::normalize NORM_MINMAX to [0,1]: scale/shift in double, applied by convertTo in float32 */
    double smin = res[0], smax = res[0];
    for (size_t i = 1; i < n; i++) {
        if (res[i] < smin) smin = res[i];
        if (res[i] > smax) smax = res[i];
    }
    double scale = (1.0 - 0.0) * (smax - smin > DBL_EPSILON ? 1. / (smax - smin) : 0);
    double shift = 0.0 - smin * scale;
    float a = (float)scale, b = (float)shift;
    for (size_t i = 0; i < n; i++) {
        float v = res[i] * a;
        res[i] = v + b;
    }
    /* minMaxLoc: first maximum in raster order */
    size_t best = 0;
    for (size_t i = 1; i < n; i++)
        if (res[i] > res[best])
            best = i;
    int mx = (int)(best % rw), my = (int)(best / rw);
    float sx = 0.f, sy = 0.f, total_mass = 0.f;
    for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++) {
            int xx = mx + i, yy = my + j;
            if (xx < 0 || yy < 0 || xx >= rw || yy >= rh)
                continue;
            float pv = res[(size_t)yy * rw + xx];
            float px = (float)xx * pv, py = (float)yy * pv;
            sx = sx + px;
            sy = sy + py;
            total_mass = total_mass + pv;
        }
    float inv_mass = (float)(1.0 / total_mass);
    *bx = sx * inv_mass;
    *by = sy * inv_mass;
    free(res);
}

/* Bellows-movement subtraction, L3Localizer.cpp:303-369: locate the bellows template in the trigger and
 * pre-trigger frames, paste it into two empty frames (nudged one pixel apart), ProcessFrame the pair on the
 * overlap ROI and subtract the result from D.  Returns 0, or -1 where OpenCV would throw (ROI outside). */
static int bellows_subtract(orc_analyzer *a, int trig, int pre, uint8_t *D)
{
    int W = a->W, H = a->H, tw = a->tW, th = a->tH;
    size_t P = (size_t)W * H;
    if (tw > W || th > H)
        return -1;
    float tx, ty, px, py;
    orc_track_feature(frame_ptr(a, trig), W, H, a->bel_tmpl, tw, th, &tx, &ty); /* :313 */
    orc_track_feature(frame_ptr(a, pre), W, H, a->bel_tmpl, tw, th, &px, &py);  /* :314 */
    if (tx >= px) { /* :318-325 */
        tx++;
        px--;
    } else {
        tx--;
        px++;
    }
    /* cv::Rect(float, float, int, int): implicit float -> int truncation (:345,:347) */
    int rtx = (int)tx, rty = (int)ty, rpx = (int)px, rpy = (int)py;
    if (rtx < 0 || rty < 0 || rtx + tw > W || rty + th > H || rpx < 0 || rpy < 0 || rpx + tw > W || rpy + th > H)
        return -1;
    uint8_t *tc = (uint8_t *)calloc(P, 1), *pc = (uint8_t *)calloc(P, 1), *df = (uint8_t *)malloc(P);
    for (int r = 0; r < th; r++) {
        memcpy(tc + (size_t)(rty + r) * W + rtx, a->bel_tmpl + (size_t)r * tw, (size_t)tw);
        memcpy(pc + (size_t)(rpy + r) * W + rpx, a->bel_tmpl + (size_t)r * tw, (size_t)tw);
    }
    /* GetDiffROI, :462-470 (float expressions assigned to int) */
    int start_x = (int)(tx > px ? tx : px), start_y = (int)(ty > py ? ty : py);
    int delta_x = (int)((tx < px ? tx : px) + tw - start_x), delta_y = (int)((ty < py ? ty : py) + th - start_y);
    int rc = 0;
    if (start_x < 0 || start_y < 0 || delta_x < 0 || delta_y < 0 || start_x + delta_x > W || start_y + delta_y > H)
        rc = -1;
    else {
        orc_process_frame_roi(tc, pc, a->sigma, W, H, start_x, start_y, delta_x, delta_y, df); /* :355 */
        for (size_t i = 0; i < P; i++)
            D[i] = sat_u8((int)D[i] - (int)df[i]); /* :362 */
    }
    free(tc);
    free(pc);
    free(df);
    return rc;
}

/* L3Localizer::CalculateInitialBubbleParams, L3Localizer.cpp:215-460. */
static void genesis(orc_analyzer *a, int trig, int pre)
{
    size_t P = (size_t)a->W * a->H;
    uint8_t *D = (uint8_t *)malloc(P), *mask = (uint8_t *)malloc(P);
    orc_process_frame(frame_ptr(a, trig), frame_ptr(a, pre), a->sigma, a->W, a->H, D); /* :232 */
    orc_binarize(D, a->W, a->H, a->loc_thres, mask);                                    /* :252-254 */
    orc_contours *cs = orc_find_contours(mask, a->W, a->H);                              /* :264 */
    int n = cs->n;
    orc_blob *rect = (orc_blob *)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_blob));
    int nk = 0;
    int largest = 0;
    int allInBellows = n > 0; /* :275 */
    for (int i = 0; i < n; i++) { /* :277-290; erasing keeps minRect aligned with contours */
        orc_blob r;
        orc_blob_from_contour(cs->c[i].pts, cs->c[i].npts, 1, &r);
        int area = r.w * r.h;
        if (!is_in_mask(a, &r, 1)) {
            allInBellows = 0;
            if (largest < area)
                largest = area;
            rect[nk++] = r;
        }
    }
    if (allInBellows) { /* :292-390: contours re-found, no mask filter */
        if (a->bel_tmpl) {
            if (bellows_subtract(a, trig, pre, D) != 0) {
                a->exception = 1;
                free(rect);
                orc_contours_free(cs);
                free(D);
                free(mask);
                return;
            }
            orc_binarize(D, a->W, a->H, a->loc_thres, mask); /* :365-367 */
            orc_contours_free(cs);
            cs = orc_find_contours(mask, a->W, a->H); /* :374 */
            n = cs->n;
            free(rect);
            rect = (orc_blob *)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_blob));
        }
        nk = 0;
        largest = 0;
        for (int i = 0; i < n; i++) {
            orc_blob r;
            orc_blob_from_contour(cs->c[i].pts, cs->c[i].npts, 1, &r);
            int area = r.w * r.h;
            if (largest < area)
                largest = area;
            rect[nk++] = r;
        }
    }
    for (int i = 0; i < nk; i++) { /* :392-444 */
        int area = rect[i].w * rect[i].h;
        if (area > 10 || area >= largest) {
            if (!is_in_mask(a, &rect[i], 0))
                continue; /* :432-436 */
            new_bubble(a, &rect[i]);
        }
    }
    free(rect);
    orc_contours_free(cs);
    free(D);
    free(mask);
}

/* L3Localizer::CalculatePostTriggerFrameParams, L3Localizer.cpp:764-869. */
static void track(orc_analyzer *a, int frame)
{
    size_t P = (size_t)a->W * a->H;
    uint8_t *O = (uint8_t *)malloc(P), *mask = (uint8_t *)malloc(P);
    orc_posttrig_frame(frame_ptr(a, frame), a->mu, a->sigma, a->W, a->H, O); /* :779-785 */
    orc_binarize(O, a->W, a->H, 3, mask);                                     /* :786-787 */
    orc_contours *cs = orc_find_contours(mask, a->W, a->H);                   /* :793 */
    int n = cs->n, nd = 0;
    orc_blob *dets = (orc_blob *)calloc((size_t)(n > 0 ? n : 1), sizeof(orc_blob));
    for (int i = 0; i < n; i++) { /* :807-836 */
        orc_blob r;
        orc_blob_from_contour(cs->c[i].pts, cs->c[i].npts, 0, &r);
        if (r.w * r.h > 10) {
            if (!is_in_mask(a, &r, 0))
                continue;
            dets[nd++] = r;
        }
    }
    for (int k = 0; k < a->nbub; k++) /* :842-844 */
        a->bubbles[k].lock = 0;
    for (int j = 0; j < nd; j++) { /* :847-863 */
        float tx = dets[j].cx, ty = dets[j].cy;
        for (int k = 0; k < a->nbub; k++) {
            orc_bubble *b = &a->bubbles[k];
            float ex = b->last_x, ey = b->last_y;
            if ((ex - tx < 5) && (fabs(ey - ty) < 5)) {
                /* bubble::operator<<, bubble.cpp:54-72 */
                if (!b->lock) {
                    b->desc[b->n_desc++] = dets[j];
                    b->dz[b->n_dz++] = b->last_x - dets[j].x;
                    b->last_x = dets[j].cx;
                    b->last_y = dets[j].cy;
                    b->lock = 1;
                }
                break;
            }
        }
    }
    free(dets);
    orc_contours_free(cs);
    free(O);
    free(mask);
}

/* L3Localizer::LocalizeOMatic, L3Localizer.cpp:881-968. */
void orc_localize(orc_analyzer *a)
{
    if (a->F <= 5)
        a->okToProceed = 0; /* :889 */
    if (!a->okToProceed)
        return; /* :915 */
    int prev_offset = 2;
    if (a->training_set_size < 6)
        prev_offset = 1; /* :923-926 */
    int t = a->MatTrigFrame;
    int pre = t - prev_offset;
    if (pre < 0)
        pre = 0; /* :932-933 */
    genesis(a, t, pre); /* :942 */
    if (a->exception)
        return;
    if (t < 29) {       /* :944-954 */
        for (int k = 1; k <= ORC_MAX_TRACK; k++) {
            if (t + k >= a->F)
                break;
            track(a, t + k);
        }
    } else {
        for (int k = 1; k <= 39 - t; k++) {
            if (t + k >= a->F)
                break;
            track(a, t + k);
        }
    }
}

/* AnyCamAnalysis, AutoBubStart3.cpp:87-110. */
int orc_any_cam_analysis(orc_analyzer *a)
{
    int staged;
    do {
        orc_find_trigger(a, a->MatTrigFrame + 1);
        if (a->okToProceed) {
            orc_localize(a);
            if (a->exception) { /* cv::Exception -> catch -> -6 (AutoBubStart3.cpp:114-117) */
                staged = -6;
                break;
            }
            if (a->okToProceed)
                staged = a->nbub > 0 ? 0 : -1; /* stageCameraOutput: -1 when list empty (V4.cpp:102) */
            else {
                staged = -8;
                break;
            }
        } else {
            staged = a->status;
            break;
        }
    } while (a->nbub == 0);
    return staged;
}

int orc_get_trig_frame(const orc_analyzer *a) { return a->MatTrigFrame; }
int orc_get_status(const orc_analyzer *a) { return a->status; }
int orc_get_ok(const orc_analyzer *a) { return a->okToProceed; }
int orc_get_loc_thres(const orc_analyzer *a) { return a->loc_thres; }
int orc_get_nbubbles(const orc_analyzer *a) { return a->nbub; }
int orc_get_bubble_ndesc(const orc_analyzer *a, int b) { return a->bubbles[b].n_desc; }
void orc_get_bubble_desc(const orc_analyzer *a, int b, int d, orc_blob *out)
{
    *out = a->bubbles[b].desc[d];
}
int orc_get_bubble_ndz(const orc_analyzer *a, int b) { return a->bubbles[b].n_dz; }
float orc_get_bubble_dz(const orc_analyzer *a, int b, int i) { return a->bubbles[b].dz[i]; }

/* bubble::dZdT, bubble.cpp:101-108 */
float orc_get_bubble_dzdt(const orc_analyzer *a, int bi)
{
    const orc_bubble *b = &a->bubbles[bi];
    int numFrames = b->n_desc;
    float total_z = (float)(b->desc[0].x - b->desc[numFrames - 1].x);
    return (float)(total_z / ((float)numFrames - 1.0));
}
/* bubble::dRdT, bubble.cpp:110-118 */
float orc_get_bubble_drdt(const orc_analyzer *a, int bi)
{
    const orc_bubble *b = &a->bubbles[bi];
    int numFrames = b->n_desc;
    float dx = (float)(b->desc[0].w - b->desc[numFrames - 1].w);
    float dy = (float)(b->desc[0].h - b->desc[numFrames - 1].h);
    float dr = (float)sqrt(dx * dx + dy * dy);
    return (float)(dr / ((float)numFrames - 1.0));
}

int orc_get_sig_trace(const orc_analyzer *a, double *sig, int cap)
{
    int n = a->F < cap ? a->F : cap;
    for (int i = 0; i < n; i++)
        sig[i] = a->sig_trace[i];
    return n;
}
int orc_get_pixcount_len(const orc_analyzer *a, int bin) { return a->pix_counts[bin].n; }

uint64_t orc_bench_trigger_pass(const uint8_t *frames, int F, int W, int H, const uint8_t *sigma,
                                int ref_offset, int first, int count, uint32_t *hists)
{
    size_t P = (size_t)W * H;
    uint8_t *D = (uint8_t *)malloc(P);
    uint64_t ck = 0;
    for (int n = 0; n < count; n++) {
        int i = first + n;
        if (i >= F)
            break;
        int r = i - ref_offset;
        if (r < 0)
            r = 0;
        uint32_t h[256];
        orc_process_frame(frames + (size_t)i * P, frames + (size_t)r * P, sigma, W, H, D);
        orc_hist256(D, P, h);
        for (int b = 0; b < 256; b++) {
            ck = ck * 1099511628211ULL + h[b];
            if (hists)
                hists[(size_t)n * 256 + b] = h[b];
        }
    }
    free(D);
    return ck;
}

/* ======================================================================================= */
/* PICO recon format (PICOFormatWriterV4.cpp)                                              */
/* ======================================================================================= */

typedef struct {
    char *p;
    int cap, n, overflow;
} sbuf;

static void sb_printf(sbuf *b, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    int room = b->cap - b->n;
    int w = vsnprintf(b->p + b->n, room > 0 ? (size_t)room : 0, fmt, ap);
    va_end(ap);
    if (w < 0 || w >= room)
        b->overflow = 1;
    else
        b->n += w;
}

/* writeHeader, :54-88 */
int orc_format_header(char *out, int cap)
{
    sbuf b = {out, cap, 0, 0};
    const int n = ORC_MAX_TRACK;
    sb_printf(&b, "Output of AutoBub v3 - the automatic unified bubble finder code by Pitam, using OpenCV.\n");
    sb_printf(&b, "run  ev  ibubimage  TotalBub4CamImg  camera  frame0  hori  vert  GenesisW  GenesisH  dZdt  dRdt  ");
    sb_printf(&b, "TrkFrame(%d)  TrkHori(%d)  TrkVert(%d)  ", n, n, n);
    sb_printf(&b, "TrkBubW(%d)  TrkBubH(%d)  TrkBubRadius(%d)  FakeValue\n", n, n, n);
    sb_printf(&b, "%%12s  %%5d  %%d  %%d  %%d  %%d  %%.02f  %%.02f  %%d  %%d  %%.02f  %%.02f  ");
    for (int j = 1; j <= n; j++)
        sb_printf(&b, "%%d  ");
    for (int k = 0; k < 5; k++)
        for (int j = 1; j <= n; j++)
            sb_printf(&b, "%%.02f  ");
    sb_printf(&b, "%%d");
    sb_printf(&b, "\n8\n\n\n");
    return b.overflow ? -1 : b.n;
}

/* writeCameraOutput :287-302 + formEachBubbleOutput :132-283; std::fixed, precision 2, "  " separators */
int orc_format_event(orc_analyzer *const *cams, const int *staged, int ncams, const char *run_number,
                     int event, int frameOffset, char *out, int cap)
{
    sbuf b = {out, cap, 0, 0};
    const int n = ORC_MAX_TRACK;
    int ibub = 1, nBubTotal = 0;
    for (int c = 0; c < ncams; c++)
        nBubTotal += staged[c] != 0 ? 0 : cams[c]->nbub; /* :291-294 */
    for (int c = 0; c < ncams; c++) {
        const orc_analyzer *a = cams[c];
        if (staged[c] != 0) { /* :146-174 */
            sb_printf(&b, "%s  %d  %d  %d  %d  %d  %.2f  %.2f  %d  %d", run_number, event, 0, 0, c, staged[c], 0.0, 0.0, 0, 0);
            sb_printf(&b, "  %.2f  %.2f  ", 0.0, 0.0);
            for (int j = 1; j <= n; j++)
                sb_printf(&b, "%d  ", 0);
            for (int j = 1; j <= 5 * n; j++)
                sb_printf(&b, "%.2f  ", 0.0);
            sb_printf(&b, "1  \n");
            continue;
        }
        for (int i = 0; i < a->nbub; i++) { /* :180-276 */
            const orc_bubble *bb = &a->bubbles[i];
            int frame0 = a->MatTrigFrame + frameOffset; /* :184 */
            float width = (float)bb->desc[0].w, height = (float)bb->desc[0].h;
            float x = bb->desc[0].cx, y = bb->desc[0].cy;
            float dzdt = orc_get_bubble_dzdt(a, i), drdt = orc_get_bubble_drdt(a, i);
            int tracked = bb->n_desc - 1;
            int excess = n > tracked ? n - tracked : 0;
            sb_printf(&b, "%s  %d  %d  %d  %d  ", run_number, event, ibub + i, nBubTotal, c);
            sb_printf(&b, "%d  ", frame0);
            sb_printf(&b, "%.2f  %.2f  %d  %d  %.2f  %.2f  ", (double)x, (double)y, (int)width, (int)height, (double)dzdt,
                      (double)drdt);
            for (int j = 1; j <= tracked; j++)
                sb_printf(&b, "%d  ", frame0 + j);
            for (int j = 0; j < excess; j++)
                sb_printf(&b, "%d  ", frame0 + tracked + j);
            for (int j = 1; j <= tracked; j++)
                sb_printf(&b, "%.2f  ", (double)bb->desc[j].cx);
            for (int j = 0; j < excess; j++)
                sb_printf(&b, "%d  ", -1);
            for (int j = 1; j <= tracked; j++)
                sb_printf(&b, "%.2f  ", (double)bb->desc[j].cy);
            for (int j = 0; j < excess; j++)
                sb_printf(&b, "%d  ", -1);
            for (int j = 1; j <= tracked; j++)
                sb_printf(&b, "%d  ", bb->desc[j].w);
            for (int j = 0; j < excess; j++)
                sb_printf(&b, "%d  ", -1);
            for (int j = 1; j <= tracked; j++)
                sb_printf(&b, "%d  ", bb->desc[j].h);
            for (int j = 0; j < excess; j++)
                sb_printf(&b, "%d  ", -1);
            for (int j = 1; j <= tracked; j++)
                sb_printf(&b, "%.2f  ", bb->desc[j].radius);
            for (int j = 0; j < excess; j++)
                sb_printf(&b, "%d  ", -1);
            sb_printf(&b, "1  \n");
        }
        ibub += a->nbub;
    }
    return b.overflow ? -1 : b.n;
}
