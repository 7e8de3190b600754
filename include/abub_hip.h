/*
 * abub_hip.h -- C ABI of the MI355X (gfx950) bubble-detection hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, no exceptions.
 * The reference (picoexperiment/AutoBub3hs) is a single C++ process with no FFI of its own; these
 * entry points are what its AnalyzerUnit / L3Localizer / Trainer methods would bind for the
 * per-pixel work they delegate to OpenCV today.  Each entry cites the reference code it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the call sites a maintainer edits.
 *
 * Two layers:
 *   (A) abub_*_dev  : stateless launchers on DEVICE pointers + a hipStream_t (passed as void*).
 *                     Used by bench.py / tests with torch-owned HBM and by layer (B).
 *   (B) abub_ctx_*  : a per-host-thread context that owns HBM slabs (frame stack, model, scratch)
 *                     and moves HOST buffers in and out.  One ctx per host thread (the reference runs
 *                     one analyzer per OpenMP thread, AutoBubStart3.cpp:342); contexts are independent
 *                     and re-entrant.
 *
 * All functions return 0 on success or a negative code (ABUB_E_*); abub_last_error() gives text.
 * Images are single-channel u8, row-major, pitch == W.
 */
#ifndef ABUB_HIP_H
#define ABUB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABUB_OK 0
#define ABUB_E_INVALID (-1)   /* bad argument (shape, null pointer, range) */
#define ABUB_E_HIP (-2)       /* HIP runtime error (see abub_last_error) */
#define ABUB_E_NODEVICE (-3)  /* no usable gfx950 device */
#define ABUB_E_OVERFLOW (-4)  /* caller-provided capacity too small (count still reported) */

/* A unit of K2/K3 work: indices (in frames) into a frame slab and a model slab. */
typedef struct {
    uint32_t cur;   /* frame index of the current frame in `frames`            */
    uint32_t ref;   /* frame index of the reference frame in `frames` (K2)      */
    uint32_t model; /* index into the [nmodels][H][W] sigma6 / mu slabs         */
    uint32_t out;   /* output slot: hist[out][256] and, if stored, img[out][H][W] */
} abub_job;

const char *abub_last_error(void);
int abub_device_count(void);
/* name/CU count/HBM bytes of a device, e.g. for bench.py's config block */
int abub_device_info(int device, char *name, int name_cap, int *cus, uint64_t *hbm_bytes);

/* ------------------------------------------------------------------------------------------- */
/* (A) stateless device-pointer launchers                                                      */
/* ------------------------------------------------------------------------------------------- */

/* sigma6[i] = min(6*sigma[i], 255): the saturated `6*TrainedSigmaImage` operand of
 * AnalyzerUnit.cpp:351-352 and L3Localizer.cpp:782, computed once per model. n = bytes. */
int abub_sigma6_dev(const uint8_t *sigma, uint8_t *sigma6, size_t n, void *stream);

/* Fill jobs for regular stacks: `nstacks` stacks of F frames laid out [nstacks][F][H][W]; for stack s
 * and i in [first, first+count): cur = s*F+i, ref = s*F+max(i-ref_offset,0), model = s % nmodels,
 * out = s*count + (i-first).  (FindTriggerFrame's frame pairing, AnalyzerUnit.cpp:175-177,218,312-313.) */
int abub_fill_stack_jobs_dev(abub_job *jobs, int nstacks, int F, int first, int count, int ref_offset,
                             int nmodels, void *stream);

/* K2: fused AnalyzerUnit::ProcessFrame (AnalyzerUnit.cpp:346-377: two saturating differences
 * minus 6 sigma, 5x5 Gaussian of each, absdiff) + cv::calcHist 256 bins (AnalyzerUnit.cpp:456).
 *   frames  : [..][H][W] u8 slab            sigma6 : [nmodels][H][W] from abub_sigma6_dev
 *   jobs    : njobs entries (device)         hist   : [nslots][256] u32 (device), fully overwritten
 *   diff    : NULL (trigger-only mode, 3*W*H algorithmic bytes/job) or [nslots][H][W] u8
 *             (store mode, 4*W*H bytes/job)
 *   rows_per_chunk: 0 = auto.  Fast register-rolling kernel when W%4==0 and W<=2048, generic
 *   LDS-tile kernel otherwise (same results). */
int abub_diff_hist_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs,
                       int W, int H, uint32_t *hist, uint8_t *diff, int rows_per_chunk, void *stream);

/* Same on a sub-rectangle (ProcessFrame ROI overload, AnalyzerUnit.cpp:346: borders reflect at the
 * ROI edge, zeros outside).  One job; always the generic kernel. */
int abub_diff_roi_dev(const uint8_t *cur, const uint8_t *ref, const uint8_t *sigma6, int W, int H,
                      int rx, int ry, int rw, int rh, uint8_t *diff, uint32_t *hist, void *stream);

/* K1: Trainer::CalculateMeanSigmaImageVector (Trainer.cpp:144-216): float32 Welford over the
 * N frames `idx[0..N)` (device array of frame indices, NULL = 0..N-1) of `frames`. */
int abub_train_dev(const uint8_t *frames, const uint32_t *idx, int N, int W, int H, uint8_t *mu,
                   uint8_t *sigma, void *stream);

/* K1b: histogram (256 bins) of the saturating difference f1 - f0 for npairs frame pairs
 * (Trainer.cpp:279 + the calcHist of :365); pairs[k] = {cur=f1, ref=f0, -, out}. */
int abub_pair_hist_dev(const uint8_t *frames, const abub_job *pairs, int npairs, int W, int H,
                       uint32_t *hist, void *stream);

/* K3: fused L3Localizer::CalculatePostTriggerFrameParams pixel stage (L3Localizer.cpp:779-785:
 * absdiff against mu, minus 6 sigma, 3x3 box blur) + 256-bin histogram (for the Otsu of :787).
 * jobs[k] = {cur, -, model, out}; img = [nslots][H][W] (may be NULL: histogram only). */
int abub_posttrig_dev(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6,
                      const abub_job *jobs, int njobs, int W, int H, uint32_t *hist, uint8_t *img,
                      void *stream);

/* K4: binarize (v > thr[k]) + stream-compaction of foreground pixel indices, for nimg images
 * img[k] (the TOZERO + BINARY|OTSU mask of L3Localizer.cpp:252-254,786-787 with
 * thr = max(loc_thres, otsu)).  idx: [nimg][cap] u32 raster indices (unordered); count: [nimg]
 * (true counts, may exceed cap -> caller falls back to abub_fetch). */
int abub_fg_compact_dev(const uint8_t *img, int nimg, int W, int H, const int32_t *thr,
                        uint32_t *idx, int cap, uint32_t *count, void *stream);

/* Fused forms for the batched run pipeline: K2 / K3 as above, and every output pixel with value
 * > cthr[slot] is appended to ONE shared list, pairs[2k] = slot | value << 24, pairs[2k+1] = y*W+x
 * with slot = slot_base + job.out; cthr is indexed by job.out
 * (unordered; *count = true total, may exceed cap; count must be zeroed by the caller).  With the
 * list the images need not be materialised (diff / img may be NULL): cthr = the TOZERO threshold is
 * known before the launch, the final Otsu cut is applied to the listed values on the host.
 * Fast path only (abub_fast_path(W) != 0). */
int abub_fast_path(int W);

/* abub_diff_hist_dev without the stored image (trigger search, AnalyzerUnit.cpp:119-324) plus a hint about the job
 * list: it consists of blocks of `chain_len` consecutive jobs in which job q takes the cur frame of job q - chain_stride
 * as its ref and all share a model -- what abub_fill_stack_jobs_dev(first = 1, count = F - 1, ref_offset) produces
 * with chain_len = F - 1, chain_stride = ref_offset.  The kernels then load every frame row once for the two jobs
 * that use it.  The hint is verified on the device per chain; a list that does not have the structure gives the same
 * histograms, only slower.  chain_len = 0: no hint. */
int abub_diff_hist_chained_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs, int W,
                               int H, uint32_t *hist, int chain_len, int chain_stride, void *stream);

/* Store-mode form of the same: D is written too ([nslots][H][W]); the scan writes the rows it proves zero, the listed
 * groups and the handed-over rows overwrite their pixels.  BASELINE configs[2] (10k-frame slab, i = 2..9999, ref = i-2)
 * is this call with chain_len = njobs, chain_stride = 2. */
int abub_diff_hist_chained_store_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs, int W,
                                     int H, uint32_t *hist, uint8_t *diff, int chain_len, int chain_stride, void *stream);

/* Deferred pieces.  The trigger search stops at a stack's trigger frame (AnalyzerUnit.cpp:191, break at :307), but a batched
 * launch covers whole blocks of frames before the host knows where that is.  abub_diff_hist_chained_deferred_dev() is
 * abub_diff_hist_chained_dev() with the row machine left out: row ranges the bound scan hands over (dense frames) go to the
 * caller's `pieces` list (abub_k2_pieces_cap(njobs, W, H) entries of 8 bytes, `*npieces` used) and incomplete[job] (njobs
 * bytes) is set to 1 for every job that has some -- the histograms of exactly those jobs are not final.
 * abub_diff_hist_pieces_dev() then runs the row machine on the pieces of the jobs with want[job] != 0 (njobs bytes) and
 * finalises their histograms; same frames / sigma6 / jobs / njobs / hist as the deferred call.  A job must be completed at
 * most once.  Needs abub_fast_path(W) and the "bound" option on. */
size_t abub_k2_pieces_cap(int njobs, int W, int H);
int abub_diff_hist_chained_deferred_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs, int W,
                                        int H, uint32_t *hist, int chain_len, int chain_stride, void *pieces,
                                        uint32_t pieces_cap, uint32_t *npieces, uint8_t *incomplete, void *stream);
int abub_diff_hist_pieces_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs, int W, int H,
                              uint32_t *hist, const void *pieces, const uint32_t *npieces, const uint8_t *want, void *stream);

/* Run-time tuning knobs of the K2 launchers (defaults from ABUB_K2_BOUND / _CHAIN / _BUDGET / _PF / _SPLIT / _LIST / _WG /
 * _SYNC / _SCANPF in the environment):
 *   "bound"  1 = bound-and-verify pass (default), 0 = the plain row machine for every row (the dense-regime worst case)
 *   "chain"  jobs per wave of the chained scan: 2 or 4 (>= 3 means 4); 0 = never chain; -1 (default) = 4
 *   "split"  1 (default) = the chained scan's whole-piece lane mapping wherever the row width allows it, 0 = never
 *   "budget" suspect groups a (job, chunk) may list in LDS before it hands its remaining rows to the row machine
 *   "list"   1 = suspect groups go to a global list that a second kernel evaluates exactly with the whole chip;
 *            0 (default) = every scanning wave evaluates its own suspects at its end
 *   "pf"     software-prefetch depth of the row machine (1 or 2)
 *   "scanpf" rows the chained scan fetches ahead: 1, 2, -1 (default) = 2 where instantiated (trigger-only, W = 1280 class)
 *   "wg"     waves per workgroup of the chained scan = consecutive segments of one chain (1 .. 8; -1 default = 1)
 *   "sync"   row steps a wave of such a workgroup may run ahead of its slowest wave (0 = never waits; -1 default = 0)
 * Results never depend on them. */
int abub_k2_set_option(const char *name, int value);

/* The bound-and-verify form of abub_diff_hist_dev keeps a work list in device scratch memory that the
 * library owns, one buffer per (device, stream), grown on demand.  Call this before destroying a stream that was
 * used for such launches (or at any quiet moment) to give its buffer back; it waits for the stream to drain. */
int abub_scratch_release(void *stream);

/* Diagnostics of the LAST bound-and-verify launch (K2 or K3) on `stream`, read back from that scratch buffer after the
 * stream has drained: counts[0] = row pieces handed over to the row machine (32 rows each at most), counts[1] = entries
 * of the global suspect list (0 when the scanning waves evaluate their own suspects).  Measurement only. */
int abub_bound_counts_dev(void *stream, uint32_t counts[2]);
int abub_diff_hist_compact_dev(const uint8_t *frames, const uint8_t *sigma6, const abub_job *jobs, int njobs,
                               int W, int H, uint32_t *hist, uint8_t *diff, const int32_t *cthr,
                               uint32_t *pairs, uint32_t cap, uint32_t *count, uint32_t slot_base,
                               void *stream);
int abub_posttrig_compact_dev(const uint8_t *frames, const uint8_t *mu, const uint8_t *sigma6,
                              const abub_job *jobs, int njobs, int W, int H, uint32_t *hist, uint8_t *img,
                              const int32_t *cthr, uint32_t *pairs, uint32_t cap, uint32_t *count,
                              uint32_t slot_base, void *stream);

/* Group such a list by slot on the device (counting sort): offsets[s] .. offsets[s+1] delimit slot s
 * in idx_out / val_out (raster index, value); offsets[nslots] = total (<= cap).  `count` is read on the
 * device, no host synchronisation needed between the producing launches and this call. */
int abub_pairs_group_dev(const uint32_t *pairs, const uint32_t *count, uint32_t cap, int nslots,
                         uint32_t *scratch /* [2*nslots] */, uint32_t *offsets /* [nslots+1] */,
                         uint32_t *idx_out /* [cap] */, uint8_t *val_out /* [cap] */, void *stream);

/* Same, with the per-slot counts taken from the producers' histograms (entries of slot s = pixels of hist[s]
 * with value > cthr[s]) instead of a counting pass over the list; valid when the list did not overflow. */
int abub_pairs_group_hist_dev(const uint32_t *pairs, const uint32_t *count, uint32_t cap, int nslots,
                              uint32_t *scratch, uint32_t *offsets, uint32_t *idx_out, uint8_t *val_out,
                              const uint32_t *hist /* [nslots][256] */, const int32_t *cthr /* [nslots] */,
                              void *stream);

/* K4, batched form: one shared output list for all nimg images, pairs[2*k] = image | value << 24,
 * pairs[2*k+1] = y*W+x (unordered); *count = true total (may exceed cap). */
int abub_fg_compact_pairs_dev(const uint8_t *img, int nimg, int W, int H, const int32_t *thr,
                              uint32_t *pairs, uint32_t cap, uint32_t *count, void *stream);

/* Raw terms of cv::matchTemplate(CV_TM_CCORR_NORMED) for the bellows veto (L3Localizer::TrackAFeature,
 * L3Localizer.cpp:499-500): for each of the (W-tw+1) x (H-th+1) placements the exact integer sums
 * num = sum(T*I) and wsum2 = sum(I*I) over the window.  Normalisation is host work (double). */
int abub_match_ccorr_dev(const uint8_t *img, int W, int H, const uint8_t *tmpl, int tw, int th,
                         unsigned long long *num, unsigned long long *wsum2, void *stream);
/* img = saturate(img - sub) in place + its 256-bin histogram (`overTheSigma -= diff_frame`, L3Localizer.cpp:362). */
int abub_subsat_hist_dev(uint8_t *img, const uint8_t *sub, int W, int H, uint32_t *hist, void *stream);

/* ------------------------------------------------------------------------------------------- */
/* (B) context API (host buffers in/out)                                                       */
/* ------------------------------------------------------------------------------------------- */

typedef struct abub_ctx abub_ctx;

/* One context per host thread; owns a frame slab for up to max_frames frames of W x H. */
int abub_ctx_create(abub_ctx **out, int device, int W, int H, int max_frames);
void abub_ctx_destroy(abub_ctx *ctx);

/* Trainer::CalculateMeanSigmaImageVector on N host frames (Trainer.cpp:316); fills host mu/sigma.
 * N is not limited by max_frames: a training set larger than the ctx slab gets a temporary device slab of
 * N frames for the call (the Welford recurrence needs every frame of a pixel in order). */
int abub_ctx_train(abub_ctx *ctx, const uint8_t *const *frames, int N, uint8_t *mu_out,
                   uint8_t *sigma_out);
/* 256-bin histogram of sat(f1 - f0) (training entropy veto, Trainer.cpp:279-280). */
int abub_ctx_pair_hist(abub_ctx *ctx, const uint8_t *f0, const uint8_t *f1, uint32_t hist[256]);

/* Make (mu, sigma) the context's current model (AnalyzerUnit.cpp:27 deep-copies the Trainer per
 * analyzer; here the model lives once in HBM). */
int abub_ctx_set_model(abub_ctx *ctx, const uint8_t *mu, const uint8_t *sigma);

/* Upload the frame stack of one (event, camera): F host frame pointers (Parser::GetImage results). */
int abub_ctx_upload_stack(abub_ctx *ctx, const uint8_t *const *frames, int F);

/* Histograms of D(frame[i]; frame[max(i-ref_offset,0)]) for i in [first, first+count):
 * everything FindTriggerFrame needs (AnalyzerUnit.cpp:191-314).  hist_out: [count][256] host. */
int abub_ctx_diff_hist_batch(abub_ctx *ctx, int ref_offset, int first, int count, uint32_t *hist_out);

/* D(frame[i]; frame[ref]) materialised (L3Localizer.cpp:232); D_out/hist_out may be NULL.
 * The image stays resident as the context's "current image" for abub_ctx_foreground. */
int abub_ctx_diff_frame(abub_ctx *ctx, int i, int ref, uint8_t *D_out, uint32_t *hist_out);

/* ROI overload of ProcessFrame on resident frames (AnalyzerUnit.cpp:346, bellows path L3Localizer.cpp:355). */
int abub_ctx_diff_frame_roi(abub_ctx *ctx, int i, int ref, int rx, int ry, int rw, int rh, uint8_t *D_out,
                            uint32_t *hist_out);

/* Post-trigger image of frame i (L3Localizer.cpp:779-785) + histogram; becomes the current image. */
int abub_ctx_posttrig(abub_ctx *ctx, int i, uint8_t *O_out, uint32_t *hist_out);

/* Foreground pixels (v > thr) of the current image as raster indices; *n = true count.
 * Returns ABUB_E_OVERFLOW if *n > cap (idx_out then holds the first cap found). */
int abub_ctx_foreground(abub_ctx *ctx, int thr, uint32_t *idx_out, int cap, int *n);

/* Bellows veto: correlation terms of resident frame i against a host template (see abub_match_ccorr_dev);
 * num_out / wsum2_out: [(H-th+1)][(W-tw+1)] host arrays. */
int abub_ctx_match_template(abub_ctx *ctx, int i, const uint8_t *tmpl, int tw, int th, unsigned long long *num_out,
                            unsigned long long *wsum2_out);
/* current image = saturate(current image - sub) with sub a host image; returns the new histogram. */
int abub_ctx_subtract_image(abub_ctx *ctx, const uint8_t *sub, uint32_t *hist_out);
/* Replace the current image by a host image (re-install D after a ROI ProcessFrame used the slot). */
int abub_ctx_set_image(abub_ctx *ctx, const uint8_t *img);

/* Copy the current image to the host (debug write-out / overflow fallback). */
int abub_ctx_fetch_image(abub_ctx *ctx, uint8_t *out);

/* ---- PNG frames decoded on the GPU (abub_png.hip) -------------------------------------------------------------
 * Replaces, for batches of frames, the host decode behind Parser::GetImage (ZipParser.cpp:186-239, RawParser.cpp:30-47:
 * cv::imdecode / cv::imread = libpng + zlib).  The caller uploads the FILES as they are on disk and describes each
 * frame: where its IDAT chunks lie, where its zlib stream may be assembled, where the decoded frame goes.  Handles
 * 8-bit grey and 8-bit palette images without interlace, any filter type, any deflate block type; a frame the kernels
 * refuse gets a positive status (below) and is left as it is -- the caller decodes such a frame on the host.
 * All pointers are device pointers; nothing here reads host memory. */
typedef struct abub_png_seg {
    uint32_t off; /* byte offset of an IDAT chunk's DATA in `files` */
    uint32_t len; /* its length */
} abub_png_seg;
typedef struct abub_png_frame {
    uint32_t seg_begin; /* first entry of this frame in segs[] */
    uint32_t seg_count; /* its IDAT chunks, in file order */
    uint32_t zoff;      /* where the frame's zlib stream is assembled in `zbuf`: a multiple of 16, with room for zlen
                           rounded up to 16, plus 16 */
    uint32_t zlen;      /* sum of the segment lengths */
    uint32_t lut;       /* palette images: index of the frame's 256-byte palette-index -> grey table in `luts`;
                           0xffffffff for grey images */
    uint32_t reserved;
    uint64_t dst;       /* byte offset of the decoded W*H frame from `out` (a multiple of 4) */
} abub_png_frame;
/* status[frame] after abub_png_decode_dev: 0 = decoded */
#define ABUB_PNG_E_DESC 1       /* descriptor out of the stated buffer sizes */
#define ABUB_PNG_E_HEADER 2     /* zlib header (RFC 1950) */
#define ABUB_PNG_E_TRUNCATED 3  /* stream ends early */
#define ABUB_PNG_E_BLOCKTYPE 4  /* deflate block type 3 */
#define ABUB_PNG_E_STORED 5     /* stored block: LEN / NLEN mismatch */
#define ABUB_PNG_E_SYMBOLS 6    /* more than 286 literal/length or 30 distance codes */
#define ABUB_PNG_E_CODES 7      /* over-subscribed / incomplete code, bad repeat */
#define ABUB_PNG_E_NOEOB 8      /* no end-of-block code */
#define ABUB_PNG_E_CODE 9       /* invalid literal/length or distance code in the data */
#define ABUB_PNG_E_DISTANCE 10  /* distance reaches before the start of the output */
#define ABUB_PNG_E_TOOMUCH 11   /* more than H*(W+1) bytes */
#define ABUB_PNG_E_TOOLITTLE 12 /* fewer */
#define ABUB_PNG_E_ADLER 13     /* Adler-32 of the output differs from the trailer */
#define ABUB_PNG_E_FILTER 14    /* filter type above 4 */
#define ABUB_PNG_E_INTERNAL 15  /* the two waves of a stream lost each other (never expected) */
/* bytes of `rawbuf` one frame takes (its inflated, still filtered scanlines) */
size_t abub_png_raw_stride(int W, int H);
/* files: the uploaded file bytes (4-byte aligned, files_bytes of them); frames[nframes], segs[nsegs], luts[nluts][256];
 * zbuf / rawbuf: scratch (16-byte aligned; rawbuf >= nframes * abub_png_raw_stride); out: where frames go (out_bytes);
 * status[nframes] (written for every frame).  W: a multiple of 4 in [4, 2048]. */
int abub_png_decode_dev(const uint8_t *files, size_t files_bytes, const abub_png_frame *frames, int nframes,
                        const abub_png_seg *segs, int nsegs, const uint8_t *luts, int nluts, int W, int H,
                        uint8_t *zbuf, size_t zbuf_bytes, uint8_t *rawbuf, size_t rawbuf_bytes, uint8_t *out,
                        size_t out_bytes, int32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif
