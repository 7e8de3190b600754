// abub_ctx.hip -- layer (B) of include/abub_hip.h: a per-host-thread context that owns the HBM slabs
// of one (event, camera) at a time and moves host buffers in and out.  Everything here is plumbing
// around the launchers of abub_k2.hip / abub_k3.hip / abub_misc.hip; no pixel arithmetic happens on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/abub_hip.h"

struct abub_ctx {
    int device, W, H, maxF;
    size_t P;
    hipStream_t stream;
    uint8_t *d_frames;  // [maxF][H][W]   frame stack of the current (event, camera)
    uint8_t *d_mu, *d_sigma, *d_sigma6; // [H][W] current model
    uint8_t *d_img;     // [H][W]         current image (last D or post-trigger image)
    uint32_t *d_hist;   // [maxF][256]
    abub_job *d_jobs;   // [maxF]
    uint32_t *d_idx;    // [idx_cap]
    uint32_t *d_count;
    int32_t *d_thr;
    int idx_cap;
    uint8_t *h_stage;   // pinned [maxF][H][W]
    uint32_t *h_hist;   // pinned [maxF][256]
    abub_job *h_job;    // pinned [1]
    uint32_t *h_small;  // pinned [4]
    uint32_t *h_idx;    // pinned [idx_cap]
    int F;
    int have_model;
};

int abub_set_err_(int code, const char *what, hipError_t e); // abub_misc.hip: text for abub_last_error()
static int cfail(int code, const char *what, hipError_t e = hipSuccess) { return abub_set_err_(code, what, e); }
#define CCHK(x)                                  \
    do {                                         \
        hipError_t e_ = (x);                     \
        if (e_ != hipSuccess)                    \
            return cfail(ABUB_E_HIP, #x, e_);    \
    } while (0)
#define CALL(x)                                  \
    do {                                         \
        int r_ = (x);                            \
        if (r_ != ABUB_OK)                       \
            return r_;                           \
    } while (0)

extern "C" int abub_ctx_create(abub_ctx **out, int device, int W, int H, int max_frames)
{
    if (!out || W <= 0 || H <= 0 || max_frames <= 0)
        return cfail(ABUB_E_INVALID, "abub_ctx_create: bad arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return cfail(ABUB_E_NODEVICE, "abub_ctx_create: no such HIP device");
    CCHK(hipSetDevice(device));
    abub_ctx *c = (abub_ctx *)calloc(1, sizeof(abub_ctx));
    c->device = device;
    c->W = W;
    c->H = H;
    c->maxF = max_frames;
    c->P = (size_t)W * H;
    c->idx_cap = 1 << 16;
    *out = nullptr;
    // any failure below gives the partly built context back (destroy tolerates null members)
#define CTRY(x)                                           \
    do {                                                  \
        hipError_t e_ = (x);                              \
        if (e_ != hipSuccess) {                           \
            const int rc_ = cfail(ABUB_E_HIP, #x, e_);    \
            abub_ctx_destroy(c);                          \
            return rc_;                                   \
        }                                                 \
    } while (0)
    CTRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CTRY(hipMalloc((void **)&c->d_frames, c->P * max_frames));
    CTRY(hipMalloc((void **)&c->d_mu, c->P));
    CTRY(hipMalloc((void **)&c->d_sigma, c->P));
    CTRY(hipMalloc((void **)&c->d_sigma6, c->P));
    CTRY(hipMalloc((void **)&c->d_img, c->P));
    CTRY(hipMalloc((void **)&c->d_hist, (size_t)max_frames * 256 * sizeof(uint32_t)));
    CTRY(hipMalloc((void **)&c->d_jobs, (size_t)max_frames * sizeof(abub_job)));
    CTRY(hipMalloc((void **)&c->d_idx, (size_t)c->idx_cap * sizeof(uint32_t)));
    CTRY(hipMalloc((void **)&c->d_count, sizeof(uint32_t)));
    CTRY(hipMalloc((void **)&c->d_thr, sizeof(int32_t)));
    CTRY(hipHostMalloc((void **)&c->h_stage, c->P * max_frames, hipHostMallocDefault));
    CTRY(hipHostMalloc((void **)&c->h_hist, (size_t)max_frames * 256 * sizeof(uint32_t), hipHostMallocDefault));
    CTRY(hipHostMalloc((void **)&c->h_job, sizeof(abub_job), hipHostMallocDefault));
    CTRY(hipHostMalloc((void **)&c->h_small, 4 * sizeof(uint32_t), hipHostMallocDefault));
    CTRY(hipHostMalloc((void **)&c->h_idx, (size_t)c->idx_cap * sizeof(uint32_t), hipHostMallocDefault));
#undef CTRY
    *out = c;
    return ABUB_OK;
}

extern "C" void abub_ctx_destroy(abub_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_frames);
    (void)hipFree(c->d_mu);
    (void)hipFree(c->d_sigma);
    (void)hipFree(c->d_sigma6);
    (void)hipFree(c->d_img);
    (void)hipFree(c->d_hist);
    (void)hipFree(c->d_jobs);
    (void)hipFree(c->d_idx);
    (void)hipFree(c->d_count);
    (void)hipFree(c->d_thr);
    (void)hipHostFree(c->h_stage);
    (void)hipHostFree(c->h_hist);
    (void)hipHostFree(c->h_job);
    (void)hipHostFree(c->h_small);
    (void)hipHostFree(c->h_idx);
    if (c->stream) {
        (void)abub_scratch_release(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    free(c);
}

// stage host frames through pinned memory, one async H2D per frame so that the next host memcpy
// overlaps the previous transfer
static int upload_frames(abub_ctx *c, const uint8_t *const *frames, int n, uint8_t *d_dst, uint8_t *h_stage)
{
    for (int k = 0; k < n; k++) {
        if (!frames[k])
            return cfail(ABUB_E_INVALID, "upload_frames: null frame pointer");
        memcpy(h_stage + (size_t)k * c->P, frames[k], c->P);
        CCHK(hipMemcpyAsync(d_dst + (size_t)k * c->P, h_stage + (size_t)k * c->P, c->P,
                            hipMemcpyHostToDevice, c->stream));
    }
    return ABUB_OK;
}

extern "C" int abub_ctx_train(abub_ctx *c, const uint8_t *const *frames, int N, uint8_t *mu_out,
                              uint8_t *sigma_out)
{
    if (!c || !frames || N <= 0 || !mu_out || !sigma_out)
        return cfail(ABUB_E_INVALID, "abub_ctx_train: bad arguments");
    CCHK(hipSetDevice(c->device));
    uint8_t *d_train = nullptr, *h_st = nullptr;
    bool own = N > c->maxF;
    if (own) {
        CCHK(hipMalloc((void **)&d_train, c->P * (size_t)N));
        CCHK(hipHostMalloc((void **)&h_st, c->P * (size_t)N, hipHostMallocDefault));
    } else {
        d_train = c->d_frames;
        h_st = c->h_stage;
        c->F = 0; // the resident stack is overwritten
    }
    int rc = upload_frames(c, frames, N, d_train, h_st);
    if (rc == ABUB_OK)
        rc = abub_train_dev(d_train, nullptr, N, c->W, c->H, c->d_mu, c->d_sigma, c->stream);
    if (rc == ABUB_OK)
        rc = abub_sigma6_dev(c->d_sigma, c->d_sigma6, c->P, c->stream);
    if (rc == ABUB_OK) {
        hipError_t e = hipMemcpyAsync(mu_out, c->d_mu, c->P, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(sigma_out, c->d_sigma, c->P, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess)
            rc = cfail(ABUB_E_HIP, "abub_ctx_train: copy back", e);
    }
    if (own) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(d_train);
        (void)hipHostFree(h_st);
    }
    if (rc == ABUB_OK)
        c->have_model = 1;
    return rc;
}

extern "C" int abub_ctx_pair_hist(abub_ctx *c, const uint8_t *f0, const uint8_t *f1, uint32_t hist[256])
{
    if (!c || !f0 || !f1 || !hist)
        return cfail(ABUB_E_INVALID, "abub_ctx_pair_hist: bad arguments");
    if (c->maxF < 2)
        return cfail(ABUB_E_INVALID, "abub_ctx_pair_hist: context needs max_frames >= 2");
    CCHK(hipSetDevice(c->device));
    const uint8_t *fr[2] = {f0, f1};
    c->F = 0;
    CALL(upload_frames(c, fr, 2, c->d_frames, c->h_stage));
    *c->h_job = abub_job{1, 0, 0, 0};
    CCHK(hipMemcpyAsync(c->d_jobs, c->h_job, sizeof(abub_job), hipMemcpyHostToDevice, c->stream));
    CALL(abub_pair_hist_dev(c->d_frames, c->d_jobs, 1, c->W, c->H, c->d_hist, c->stream));
    CCHK(hipMemcpyAsync(c->h_hist, c->d_hist, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    memcpy(hist, c->h_hist, 256 * sizeof(uint32_t));
    return ABUB_OK;
}

extern "C" int abub_ctx_set_model(abub_ctx *c, const uint8_t *mu, const uint8_t *sigma)
{
    if (!c || !mu || !sigma)
        return cfail(ABUB_E_INVALID, "abub_ctx_set_model: bad arguments");
    CCHK(hipSetDevice(c->device));
    CCHK(hipMemcpyAsync(c->d_mu, mu, c->P, hipMemcpyHostToDevice, c->stream));
    CCHK(hipMemcpyAsync(c->d_sigma, sigma, c->P, hipMemcpyHostToDevice, c->stream));
    CALL(abub_sigma6_dev(c->d_sigma, c->d_sigma6, c->P, c->stream));
    CCHK(hipStreamSynchronize(c->stream)); // mu/sigma may be pageable and reused by the caller
    c->have_model = 1;
    return ABUB_OK;
}

extern "C" int abub_ctx_upload_stack(abub_ctx *c, const uint8_t *const *frames, int F)
{
    if (!c || !frames || F <= 0 || F > c->maxF)
        return cfail(ABUB_E_INVALID, "abub_ctx_upload_stack: bad arguments (F > max_frames?)");
    CCHK(hipSetDevice(c->device));
    CALL(upload_frames(c, frames, F, c->d_frames, c->h_stage));
    c->F = F;
    return ABUB_OK;
}

extern "C" int abub_ctx_diff_hist_batch(abub_ctx *c, int ref_offset, int first, int count,
                                        uint32_t *hist_out)
{
    if (!c || !hist_out || first < 0 || count < 0 || first + count > c->F || ref_offset < 0)
        return cfail(ABUB_E_INVALID, "abub_ctx_diff_hist_batch: bad arguments");
    if (!c->have_model)
        return cfail(ABUB_E_INVALID, "abub_ctx_diff_hist_batch: no model set");
    if (count == 0)
        return ABUB_OK;
    CCHK(hipSetDevice(c->device));
    CALL(abub_fill_stack_jobs_dev(c->d_jobs, 1, c->F, first, count, ref_offset, 1, c->stream));
    // one stack: the whole job list is one block of chains (job q refs the cur frame of job q - ref_offset)
    CALL(abub_diff_hist_chained_dev(c->d_frames, c->d_sigma6, c->d_jobs, count, c->W, c->H, c->d_hist,
                                    ref_offset > 0 ? count : 0, ref_offset, c->stream));
    CCHK(hipMemcpyAsync(c->h_hist, c->d_hist, (size_t)count * 256 * sizeof(uint32_t),
                        hipMemcpyDeviceToHost, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    memcpy(hist_out, c->h_hist, (size_t)count * 256 * sizeof(uint32_t));
    return ABUB_OK;
}

static int one_job(abub_ctx *c, uint32_t cur, uint32_t ref)
{
    *c->h_job = abub_job{cur, ref, 0, 0};
    CCHK(hipMemcpyAsync(c->d_jobs, c->h_job, sizeof(abub_job), hipMemcpyHostToDevice, c->stream));
    return ABUB_OK;
}

static int finish_image(abub_ctx *c, uint8_t *img_out, uint32_t *hist_out)
{
    if (hist_out)
        CCHK(hipMemcpyAsync(c->h_hist, c->d_hist, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (img_out)
        CCHK(hipMemcpyAsync(img_out, c->d_img, c->P, hipMemcpyDeviceToHost, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    if (hist_out)
        memcpy(hist_out, c->h_hist, 256 * sizeof(uint32_t));
    return ABUB_OK;
}

extern "C" int abub_ctx_diff_frame(abub_ctx *c, int i, int ref, uint8_t *D_out, uint32_t *hist_out)
{
    if (!c || i < 0 || ref < 0 || i >= c->F || ref >= c->F)
        return cfail(ABUB_E_INVALID, "abub_ctx_diff_frame: frame index out of range");
    if (!c->have_model)
        return cfail(ABUB_E_INVALID, "abub_ctx_diff_frame: no model set");
    CCHK(hipSetDevice(c->device));
    CALL(one_job(c, (uint32_t)i, (uint32_t)ref));
    CALL(abub_diff_hist_dev(c->d_frames, c->d_sigma6, c->d_jobs, 1, c->W, c->H, c->d_hist, c->d_img, 0,
                            c->stream));
    return finish_image(c, D_out, hist_out);
}

extern "C" int abub_ctx_diff_frame_roi(abub_ctx *c, int i, int ref, int rx, int ry, int rw, int rh,
                                       uint8_t *D_out, uint32_t *hist_out)
{
    if (!c || i < 0 || ref < 0 || i >= c->F || ref >= c->F)
        return cfail(ABUB_E_INVALID, "abub_ctx_diff_frame_roi: frame index out of range");
    if (!c->have_model)
        return cfail(ABUB_E_INVALID, "abub_ctx_diff_frame_roi: no model set");
    CCHK(hipSetDevice(c->device));
    // the generic kernel addresses ref relative to cur; order the pair so that the offset is >= 0
    const uint8_t *pc = c->d_frames + (size_t)i * c->P, *pr = c->d_frames + (size_t)ref * c->P;
    if (ref >= i) {
        CALL(abub_diff_roi_dev(pc, pr, c->d_sigma6, c->W, c->H, rx, ry, rw, rh, c->d_img, c->d_hist, c->stream));
    } else {
        // |G(pos)-G(neg)| is symmetric in (cur, ref): swapping the frames swaps the two planes only
        CALL(abub_diff_roi_dev(pr, pc, c->d_sigma6, c->W, c->H, rx, ry, rw, rh, c->d_img, c->d_hist, c->stream));
    }
    return finish_image(c, D_out, hist_out);
}

extern "C" int abub_ctx_posttrig(abub_ctx *c, int i, uint8_t *O_out, uint32_t *hist_out)
{
    if (!c || i < 0 || i >= c->F)
        return cfail(ABUB_E_INVALID, "abub_ctx_posttrig: frame index out of range");
    if (!c->have_model)
        return cfail(ABUB_E_INVALID, "abub_ctx_posttrig: no model set");
    CCHK(hipSetDevice(c->device));
    CALL(one_job(c, (uint32_t)i, 0));
    CALL(abub_posttrig_dev(c->d_frames, c->d_mu, c->d_sigma6, c->d_jobs, 1, c->W, c->H, c->d_hist,
                           c->d_img, c->stream));
    return finish_image(c, O_out, hist_out);
}

extern "C" int abub_ctx_foreground(abub_ctx *c, int thr, uint32_t *idx_out, int cap, int *n)
{
    if (!c || !idx_out || !n || cap <= 0)
        return cfail(ABUB_E_INVALID, "abub_ctx_foreground: bad arguments");
    CCHK(hipSetDevice(c->device));
    int dcap = cap < c->idx_cap ? cap : c->idx_cap;
    int32_t *h_thr = reinterpret_cast<int32_t *>(c->h_small);
    *h_thr = thr;
    CCHK(hipMemcpyAsync(c->d_thr, h_thr, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CALL(abub_fg_compact_dev(c->d_img, 1, c->W, c->H, c->d_thr, c->d_idx, dcap, c->d_count, c->stream));
    CCHK(hipMemcpyAsync(c->h_small + 1, c->d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    uint32_t cnt = c->h_small[1];
    *n = (int)cnt;
    uint32_t got = cnt < (uint32_t)dcap ? cnt : (uint32_t)dcap;
    if (got) {
        CCHK(hipMemcpyAsync(c->h_idx, c->d_idx, (size_t)got * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        CCHK(hipStreamSynchronize(c->stream));
        memcpy(idx_out, c->h_idx, (size_t)got * sizeof(uint32_t));
    }
    return cnt > (uint32_t)dcap ? ABUB_E_OVERFLOW : ABUB_OK;
}

extern "C" int abub_ctx_match_template(abub_ctx *c, int i, const uint8_t *tmpl, int tw, int th,
                                       unsigned long long *num_out, unsigned long long *wsum2_out)
{
    if (!c || !tmpl || !num_out || !wsum2_out || i < 0 || i >= c->F || tw <= 0 || th <= 0 || tw > c->W || th > c->H)
        return cfail(ABUB_E_INVALID, "abub_ctx_match_template: bad arguments");
    CCHK(hipSetDevice(c->device));
    const size_t n = (size_t)(c->W - tw + 1) * (c->H - th + 1);
    uint8_t *d_t = nullptr;
    unsigned long long *d_num = nullptr, *d_w = nullptr;
    // (the temporaries are freed on every exit path: a failed allocation must not leak the earlier ones)
    hipError_t e = hipMalloc((void **)&d_t, (size_t)tw * th);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_num, n * 8);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_w, n * 8);
    int rc = ABUB_OK;
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_t, tmpl, (size_t)tw * th, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        rc = abub_match_ccorr_dev(c->d_frames + (size_t)i * c->P, c->W, c->H, d_t, tw, th, d_num, d_w, c->stream);
    if (e == hipSuccess && rc == ABUB_OK)
        e = hipMemcpyAsync(num_out, d_num, n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == ABUB_OK)
        e = hipMemcpyAsync(wsum2_out, d_w, n * 8, hipMemcpyDeviceToHost, c->stream);
    const hipError_t es = hipStreamSynchronize(c->stream); // also before freeing buffers a queued kernel may use
    if (e == hipSuccess)
        e = es;
    (void)hipFree(d_t);
    (void)hipFree(d_num);
    (void)hipFree(d_w);
    if (e != hipSuccess)
        return cfail(ABUB_E_HIP, "abub_ctx_match_template", e);
    return rc;
}

extern "C" int abub_ctx_subtract_image(abub_ctx *c, const uint8_t *sub, uint32_t *hist_out)
{
    if (!c || !sub)
        return cfail(ABUB_E_INVALID, "abub_ctx_subtract_image: bad arguments");
    CCHK(hipSetDevice(c->device));
    uint8_t *d_sub = nullptr;
    CCHK(hipMalloc((void **)&d_sub, c->P));
    hipError_t e = hipMemcpyAsync(d_sub, sub, c->P, hipMemcpyHostToDevice, c->stream);
    int rc = ABUB_OK;
    if (e == hipSuccess)
        rc = abub_subsat_hist_dev(c->d_img, d_sub, c->W, c->H, c->d_hist, c->stream);
    if (e == hipSuccess && rc == ABUB_OK)
        rc = finish_image(c, nullptr, hist_out);
    else
        (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_sub);
    if (e != hipSuccess)
        return cfail(ABUB_E_HIP, "abub_ctx_subtract_image", e);
    return rc;
}

extern "C" int abub_ctx_set_image(abub_ctx *c, const uint8_t *img)
{
    if (!c || !img)
        return cfail(ABUB_E_INVALID, "abub_ctx_set_image: bad arguments");
    CCHK(hipSetDevice(c->device));
    CCHK(hipMemcpyAsync(c->d_img, img, c->P, hipMemcpyHostToDevice, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    return ABUB_OK;
}

extern "C" int abub_ctx_fetch_image(abub_ctx *c, uint8_t *out)
{
    if (!c || !out)
        return cfail(ABUB_E_INVALID, "abub_ctx_fetch_image: bad arguments");
    CCHK(hipSetDevice(c->device));
    CCHK(hipMemcpyAsync(out, c->d_img, c->P, hipMemcpyDeviceToHost, c->stream));
    CCHK(hipStreamSynchronize(c->stream));
    return ABUB_OK;
}
