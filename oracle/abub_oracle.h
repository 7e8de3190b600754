/*
 * abub_oracle.h -- CPU ORACLE for the AutoBub3hs bubble-detection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (autobub3hs_amd/) never does.
 *
 * It is a plain-C restatement of the reference algorithm (picoexperiment/AutoBub3hs @ 2024-11-25).
 * Every function cites the reference file:line it follows.  The per-pixel arithmetic of the
 * reference lives in OpenCV (unpinned, CMakeLists.txt:4, absent from this image): those
 * primitives are restated from OpenCV's published algorithms (see oracle/README.md) and are
 * PARITY UNPINNED against OpenCV itself -- the reference ships no tests, golden vectors or
 * fixtures for this path.  What pins the oracle instead: analytic
 * known-answer tests and independent numpy/scipy cross-checks in tests/test_oracle_*.py.
 */
#ifndef ABUB_ORACLE_H
#define ABUB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_TRACK 10 /* common/CommonParameters.h:5 NumFramesBubbleTrack */
#define ORC_MAX_DESC (ORC_MAX_TRACK + 1)

/* ---- per-pixel primitives ------------------------------------------------------------- */

/* Trainer::CalculateMeanSigmaImageVector, Trainer.cpp:144-216.  slab = [N][H][W] u8. */
void orc_welford(const uint8_t *slab, int N, int W, int H, uint8_t *mu, uint8_t *sigma);

/* Trainer.cpp:279-280 (Mat - Mat, saturating) + Trainer::calculateEntropyFrame :341-376. */
float orc_pair_entropy16(const uint8_t *f1, const uint8_t *f0, int W, int H);
/* 16-bin Shannon entropy of an image (Trainer.cpp:341-376; ImageEntropyMethods.cpp:32-57). */
float orc_entropy16(const uint8_t *img, int W, int H);
/* AnalyzerUnit::calculateEntropyFrame, AnalyzerUnit.cpp:386-423 (128 bins; compiled, unused). */
float orc_entropy128(const uint8_t *img, int W, int H);

/* AnalyzerUnit::ProcessFrame full-frame overload, AnalyzerUnit.cpp:341-377. */
void orc_process_frame(const uint8_t *cur, const uint8_t *ref, const uint8_t *sigma, int W, int H,
                       uint8_t *D);
/* ROI overload, AnalyzerUnit.cpp:346-377: result pasted into a zeroed full frame. */
void orc_process_frame_roi(const uint8_t *cur, const uint8_t *ref, const uint8_t *sigma, int W,
                           int H, int rx, int ry, int rw, int rh, uint8_t *D);

/* cv::calcHist 256 bins, AnalyzerUnit.cpp:456. */
void orc_hist256(const uint8_t *img, size_t P, uint32_t hist[256]);

/* L3Localizer.cpp:779-785: absdiff(frame, mu) - 6*sigma, then cv::blur 3x3. */
void orc_posttrig_frame(const uint8_t *frame, const uint8_t *mu, const uint8_t *sigma, int W, int H,
                        uint8_t *O);

/* cv::threshold(THRESH_BINARY|THRESH_OTSU) threshold value from a 256-bin histogram. */
int orc_otsu(const uint32_t hist[256], size_t P);
/* TOZERO(thr) followed by BINARY|OTSU (L3Localizer.cpp:252-254, :786-787): mask 0/255.
 * Returns the Otsu threshold computed on the TOZERO'd image. */
int orc_binarize(const uint8_t *img, int W, int H, int thr, uint8_t *mask);

/* ---- contours ------------------------------------------------------------------------- */

typedef struct {
    int x, y;
} orc_point;

typedef struct {
    int npts;
    orc_point *pts; /* TC89_L1 polygon vertices */
    int nchain;     /* number of chain codes of the traced border (0 for a 1-pixel blob) */
} orc_contour;

typedef struct {
    int n;
    orc_contour *c; /* in cv::findContours output order (reverse discovery order) */
} orc_contours;

/* cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_TC89_L1), L3Localizer.cpp:264,793. */
orc_contours *orc_find_contours(const uint8_t *mask, int W, int H);
void orc_contours_free(orc_contours *c);
/* accessors for ctypes */
int orc_contours_count(const orc_contours *c);
int orc_contour_npts(const orc_contours *c, int i);
int orc_contour_nchain(const orc_contours *c, int i);
void orc_contour_points(const orc_contours *c, int i, int *xy /* [npts][2] */);

/* BubbleImageFrame, bubble/bubble.hpp:22-31. */
typedef struct {
    int x, y, w, h;     /* newPosition = boundingRect */
    double area;        /* ContArea  = cv::contourArea */
    double radius;      /* ContRadius = sqrt(area/3.14159) */
    double m00, m10, m01;
    float cx, cy;       /* MassCentres */
} orc_blob;

/* boundingRect / contourArea / moments / radius / centroid (genesis_fallback: L3Localizer.cpp:405-419). */
void orc_blob_from_contour(const orc_point *pts, int npts, int genesis_fallback, orc_blob *out);

/* ---- significance / trigger state machine -------------------------------------------- */

typedef struct orc_analyzer orc_analyzer;

/* An in-memory (event, camera): frames = [F][H][W] u8 in frame order (lexicographic file order,
 * SURVEY A9).  frame_ok may be NULL (all decodable) or F flags (0 = Parser::GetImage returned -1).
 * Masks may be NULL (== MaskDir "" or unloadable: isInMask returns !bellows, L3Localizer.cpp:988-994). */
orc_analyzer *orc_analyzer_create(const uint8_t *frames, int F, int W, int H, const uint8_t *mu,
                                  const uint8_t *sigma, int training_set_size,
                                  const uint8_t *frame_ok, const uint8_t *fid_mask, int fmW,
                                  int fmH, const uint8_t *bel_mask, int bmW, int bmH);
void orc_analyzer_destroy(orc_analyzer *a);

/* Bellows template (MaskDir/cam<N>_bellows_template.png, L3Localizer.cpp:294-295); NULL = not loadable. */
void orc_analyzer_set_bellows_template(orc_analyzer *a, const uint8_t *tmpl, int tw, int th);
/* cv::matchTemplate(CV_TM_CCORR_NORMED) restated (imgproc/templmatch.cpp): result is (H-th+1) x (W-tw+1). */
void orc_match_template_ccorr_normed(const uint8_t *img, int W, int H, const uint8_t *tmpl, int tw, int th,
                                     float *result);
/* L3Localizer::TrackAFeature, L3Localizer.cpp:473-543: best match with 3x3 sub-pixel centre of mass. */
void orc_track_feature(const uint8_t *img, int W, int H, const uint8_t *tmpl, int tw, int th, float *bx, float *by);

/* AnalyzerUnit::calculateSignificanceFrame, AnalyzerUnit.cpp:435-504 on a histogram. */
double orc_significance(orc_analyzer *a, const uint32_t hist[256], int store);

/* AnalyzerUnit::FindTriggerFrame, AnalyzerUnit.cpp:119-324. */
void orc_find_trigger(orc_analyzer *a, int startframe);
/* L3Localizer::LocalizeOMatic, L3Localizer.cpp:881-968. */
void orc_localize(orc_analyzer *a);
/* AnyCamAnalysis retry loop, AutoBubStart3.cpp:87-110.  Returns the staged status:
 * 0 = bubbles staged, or -3/-9/-8 error row.  (-1 "no bubble" is transient) */
int orc_any_cam_analysis(orc_analyzer *a);

/* field access */
int orc_get_trig_frame(const orc_analyzer *a);   /* MatTrigFrame */
int orc_get_status(const orc_analyzer *a);       /* TriggerFrameIdentificationStatus */
int orc_get_ok(const orc_analyzer *a);           /* okToProceed */
int orc_get_loc_thres(const orc_analyzer *a);    /* loc_thres */
int orc_get_nbubbles(const orc_analyzer *a);     /* BubbleList.size() */
int orc_get_bubble_ndesc(const orc_analyzer *a, int b);
void orc_get_bubble_desc(const orc_analyzer *a, int b, int d, orc_blob *out);
int orc_get_bubble_ndz(const orc_analyzer *a, int b);
float orc_get_bubble_dz(const orc_analyzer *a, int b, int i);
float orc_get_bubble_dzdt(const orc_analyzer *a, int b); /* bubble::dZdT bubble.cpp:101-108 */
float orc_get_bubble_drdt(const orc_analyzer *a, int b); /* bubble::dRdT bubble.cpp:110-118 */
/* significance of every main-loop evaluation of the last orc_find_trigger call (debug aid):
 * returns count, fills sig[i] for frame index i (NaN where not evaluated). */
int orc_get_sig_trace(const orc_analyzer *a, double *sig, int cap);
int orc_get_pixcount_len(const orc_analyzer *a, int bin);

/* PICO recon format, PICOFormatWriter/PICOFormatWriterV4.cpp: header :54-88; one event block :132-302
 * (cams[c] already analysed with orc_any_cam_analysis, staged[c] its return value).  Returns the number
 * of bytes written (excluding the terminating NUL), or -1 if cap is too small. */
int orc_format_header(char *out, int cap);
int orc_format_event(orc_analyzer *const *cams, const int *staged, int ncams, const char *run_number,
                     int event, int frameOffset, char *out, int cap);

/* Timed CPU-baseline helper for bench.py: ProcessFrame + hist256 over frames first..first+count-1
 * of a [F][H][W] stack with ref = frame[max(i-ref_offset,0)]; returns a checksum of the histograms. */
uint64_t orc_bench_trigger_pass(const uint8_t *frames, int F, int W, int H, const uint8_t *sigma,
                                int ref_offset, int first, int count, uint32_t *hists);

#ifdef __cplusplus
}
#endif
#endif
