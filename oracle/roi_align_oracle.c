/* TEST INFRASTRUCTURE — CPU restatement (oracle) of RoIAlign "v2" (aligned=True, adaptive
 * sampling), the op that feeds AR-RFF.  Not part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Restates the algorithm of the reference's CPU implementation
 *   /root/reference/mmdet/ops/roi_align/src/cpu/roi_align_v2.cpp
 *     forward  : :107-191  (sample grid :143-147, count :151, bilinear taps :22-104)
 *     backward : :247-332  (tap weights :198-241, scatter :317-327)
 * in plain C, written from the published description of the op:
 *
 *   box edge e -> e*scale - 0.5 ; size = end - start (must be >= 0)
 *   bin = size / P ; grid = ceil(size / P) per axis (sample_num == 0) or sample_num
 *   sample (iy,ix) of bin (ph,pw): y = y0 + ph*bin_h + (iy+0.5)*bin_h/grid_h, x likewise
 *   a sample with y < -1 or y > H or x < -1 or x > W contributes 0
 *   otherwise clamp to >= 0, lo = (int)coord, if lo >= size-1 then lo = hi = size-1, coord = lo
 *   value = hy*hx*v(lo,lo) + hy*lx*v(lo,hi) + ly*hx*v(hi,lo) + ly*lx*v(hi,hi)
 *   out = sum(samples) / max(grid_h*grid_w, 1)
 *
 * Pinned against the compiled reference (oracle/_ref, see oracle/build_ref.py) by
 * tests/test_oracle_golden.py and against tests/golden/roialign_*.npz.
 *
 * Threads (oracle_set_threads, default 1): the forward is split over RoIs (disjoint outputs), the backward over
 * CHANNELS (disjoint feature planes; every element still receives its contributions in RoI order), so results are
 * bit-identical for every thread count.  pthreads, no OpenMP runtime beside torch's.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -pthread -o oracle/_build/liboracle.so oracle/roi_align_oracle.c -lm
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }

typedef void (*range_fn)(void *ctx, int lo, int hi);
typedef struct { range_fn fn; void *ctx; int lo, hi; } job_t;
static void *job_main(void *p) { job_t *j = (job_t *)p; j->fn(j->ctx, j->lo, j->hi); return 0; }

/* fn(ctx, lo, hi) over [0, n) in contiguous slices, one per thread */
static void parallel_ranges(range_fn fn, void *ctx, int n) {
    int t = g_threads < n ? g_threads : n;
    if (t <= 1) { fn(ctx, 0, n); return; }
    pthread_t th[256]; job_t jobs[256];
    for (int i = 0; i < t; ++i) {
        jobs[i].fn = fn; jobs[i].ctx = ctx;
        jobs[i].lo = (int)((int64_t)n * i / t); jobs[i].hi = (int)((int64_t)n * (i + 1) / t);
        if (i + 1 < t) pthread_create(&th[i], 0, job_main, &jobs[i]);
    }
    job_main(&jobs[t - 1]);
    for (int i = 0; i + 1 < t; ++i) pthread_join(th[i], 0);
}

typedef struct {
    int y0, y1, x0, x1;   /* tap rows/cols, -1 when the sample is outside */
    float wy0, wy1, wx0, wx1;
} tap_t;

/* one-axis bilinear tap: returns 0 when the coordinate is out of range */
static int axis_tap(float c, int n, int *lo, int *hi, float *wlo, float *whi) {
    if (c < -1.0f || c > (float)n) return 0;
    if (c <= 0.0f) c = 0.0f;
    int l = (int)c;
    int h;
    if (l >= n - 1) { l = h = n - 1; c = (float)l; }
    else h = l + 1;
    float frac = c - (float)l;
    *lo = l; *hi = h; *whi = frac; *wlo = 1.0f - frac;
    return 1;
}

typedef struct {
    int batch;
    float y_start, x_start, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
    int bad;             /* negative size */
} roi_geom_t;

static roi_geom_t roi_geometry(const float *roi, float scale, int P_h, int P_w, int sample_num) {
    roi_geom_t g;
    g.batch = (int)roi[0];
    const float off = 0.5f;
    float xs = roi[1] * scale - off, ys = roi[2] * scale - off;
    float xe = roi[3] * scale - off, ye = roi[4] * scale - off;
    float w = xe - xs, h = ye - ys;
    g.bad = !(w >= 0.0f && h >= 0.0f);
    g.x_start = xs; g.y_start = ys;
    g.bin_h = h / (float)P_h; g.bin_w = w / (float)P_w;
    g.grid_h = sample_num > 0 ? sample_num : (int)ceilf(h / (float)P_h);
    g.grid_w = sample_num > 0 ? sample_num : (int)ceilf(w / (float)P_w);
    int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

/* feat [B,C,H,W], rois [R,5] (b,x1,y1,x2,y2), out [R,C,P_h,P_w] with explicit roi/chan strides
 * (so the caller can write straight into a channel slice of a wider tensor).
 * returns 0, or -(1+index of first roi with negative size). */
typedef struct {
    const float *feat; int C, H, W; const float *rois; float scale; int P_h, P_w, sample_num;
    float *out; int64_t out_roi_stride, out_chan_stride;
} fwd_ctx_t;

static void fwd_rois(void *p, int r_lo, int r_hi) {
    const fwd_ctx_t *k = (const fwd_ctx_t *)p;
    const int C = k->C, H = k->H, W = k->W, P_h = k->P_h, P_w = k->P_w;
    for (int r = r_lo; r < r_hi; ++r) {
        roi_geom_t g = roi_geometry(k->rois + 5 * (int64_t)r, k->scale, P_h, P_w, k->sample_num);
        const float *fb = k->feat + (int64_t)g.batch * C * H * W;
        for (int ph = 0; ph < P_h; ++ph)
            for (int pw = 0; pw < P_w; ++pw) {
                float *o = k->out + (int64_t)r * k->out_roi_stride + ph * P_w + pw;
                for (int c = 0; c < C; ++c) o[(int64_t)c * k->out_chan_stride] = 0.0f;
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    float y = g.y_start + ph * g.bin_h +
                              ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
                    int y0, y1; float wy0, wy1;
                    int oky = axis_tap(y, H, &y0, &y1, &wy0, &wy1);
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        float x = g.x_start + pw * g.bin_w +
                                  ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
                        int x0, x1; float wx0, wx1;
                        int okx = axis_tap(x, W, &x0, &x1, &wx0, &wx1);
                        if (!(oky && okx)) continue;
                        float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
                        for (int c = 0; c < C; ++c) {
                            const float *f = fb + (int64_t)c * H * W;
                            o[(int64_t)c * k->out_chan_stride] +=
                                w00 * f[y0 * W + x0] + w01 * f[y0 * W + x1] +
                                w10 * f[y1 * W + x0] + w11 * f[y1 * W + x1];
                        }
                    }
                }
                for (int c = 0; c < C; ++c) o[(int64_t)c * k->out_chan_stride] /= g.count;
            }
    }
}

/* index of the first roi with negative size, or -1 */
static int first_bad_roi(const float *rois, int R, float scale, int P_h, int P_w, int sample_num) {
    for (int r = 0; r < R; ++r)
        if (roi_geometry(rois + 5 * (int64_t)r, scale, P_h, P_w, sample_num).bad) return r;
    return -1;
}

int32_t oracle_roi_align_fwd(const float *feat, int B, int C, int H, int W,
                             const float *rois, int R, float scale, int P_h, int P_w,
                             int sample_num, float *out, int64_t out_roi_stride,
                             int64_t out_chan_stride) {
    (void)B;
    int bad = first_bad_roi(rois, R, scale, P_h, P_w, sample_num);
    fwd_ctx_t k = {feat, C, H, W, rois, scale, P_h, P_w, sample_num, out, out_roi_stride, out_chan_stride};
    parallel_ranges(fwd_rois, &k, bad >= 0 ? bad : R);      /* the RoIs in front of a bad one are still written */
    return bad >= 0 ? -(1 + bad) : 0;
}

/* dfeat [B,C,H,W] is ACCUMULATED into (caller zeroes); gout [R,C,P_h,P_w] with strides. */
typedef struct {
    const float *gout; int64_t g_roi_stride, g_chan_stride; const float *rois; int R; float scale;
    int P_h, P_w, sample_num; float *dfeat; int C, H, W;
} bwd_ctx_t;

static void bwd_channels(void *p, int c_lo, int c_hi) {
    const bwd_ctx_t *k = (const bwd_ctx_t *)p;
    const int C = k->C, H = k->H, W = k->W, P_h = k->P_h, P_w = k->P_w;
    for (int r = 0; r < k->R; ++r) {
        roi_geom_t g = roi_geometry(k->rois + 5 * (int64_t)r, k->scale, P_h, P_w, k->sample_num);
        float *db = k->dfeat + (int64_t)g.batch * C * H * W;
        float cnt = (float)(g.grid_h * g.grid_w);
        for (int ph = 0; ph < P_h; ++ph)
            for (int pw = 0; pw < P_w; ++pw) {
                const float *go = k->gout + (int64_t)r * k->g_roi_stride + ph * P_w + pw;
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    float y = g.y_start + ph * g.bin_h +
                              ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
                    int y0, y1; float wy0, wy1;
                    int oky = axis_tap(y, H, &y0, &y1, &wy0, &wy1);
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        float x = g.x_start + pw * g.bin_w +
                                  ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
                        int x0, x1; float wx0, wx1;
                        int okx = axis_tap(x, W, &x0, &x1, &wx0, &wx1);
                        if (!(oky && okx)) continue;
                        float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
                        for (int c = c_lo; c < c_hi; ++c) {
                            float gv = go[(int64_t)c * k->g_chan_stride];
                            float *d = db + (int64_t)c * H * W;
                            d[y0 * W + x0] += gv * w00 / cnt;
                            d[y0 * W + x1] += gv * w01 / cnt;
                            d[y1 * W + x0] += gv * w10 / cnt;
                            d[y1 * W + x1] += gv * w11 / cnt;
                        }
                    }
                }
            }
    }
}

int32_t oracle_roi_align_bwd(const float *gout, int64_t g_roi_stride, int64_t g_chan_stride,
                             const float *rois, int R, float scale, int P_h, int P_w,
                             int sample_num, float *dfeat, int B, int C, int H, int W) {
    (void)B;
    int bad = first_bad_roi(rois, R, scale, P_h, P_w, sample_num);
    bwd_ctx_t k = {gout, g_roi_stride, g_chan_stride, rois, bad >= 0 ? bad : R, scale, P_h, P_w, sample_num, dfeat, C, H, W};
    parallel_ranges(bwd_channels, &k, C);
    return bad >= 0 ? -(1 + bad) : 0;
}
