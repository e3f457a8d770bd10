// AR-RFF RoI-side kernels: enlarged-RoI generation, RoI -> level assignment (bit-exact), RoIAlign v2
// forward/backward fused with the per-level dispatch, and the elementwise fusion z = ori + ori*(lw+lh).
#include "common.hpp"

namespace arfe {

// ---- get_adaptive_scale_rois(rois, 1) — same fp32 operation order as additional.py:38-71 -----------------
// contraction is switched off so no mul+add pair is fused into an fma (the reference rounds every op).
__global__ void rff_make_rois_kernel(int R, const float* __restrict__ rois, float* __restrict__ h_rois,
                                     float* __restrict__ w_rois) {
#pragma clang fp contract(off)
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    const float b = rois[5 * k], x1 = rois[5 * k + 1], y1 = rois[5 * k + 2], x2 = rois[5 * k + 3], y2 = rois[5 * k + 4];
    const float cx = (x1 + x2) * 0.5f, cy = (y1 + y2) * 0.5f;
    const float rw = x2 - x1 + 1.0f, rh = y2 - y1 + 1.0f;
    const float facs = 1.0f;
    const float h_rate = (rw / rh) * facs + 1.0f;
    const float w_rate = (rh / rw) * facs + 1.0f;
    const float large_h = rh * h_rate, large_w = rw * w_rate;
    const float lo = 0.1f;   // torch.ones_like(rw) * 0.1 == fp32(0.1)
    float* h = h_rois + 5 * k;
    float* w = w_rois + 5 * k;
    h[0] = b;
    h[1] = fmaxf(cx - rw * 0.5f, lo);
    h[2] = fmaxf(cy - large_h * 0.5f, lo);
    h[3] = cx + rw * 0.5f;
    h[4] = cy + large_h * 0.5f;
    w[0] = b;
    w[1] = fmaxf(cx - large_w * 0.5f, lo);
    w[2] = fmaxf(cy - large_h * 0.5f, lo);
    w[3] = cx + large_w * 0.5f;
    w[4] = cy + large_h * 0.5f;
}

// ---- map_roi_levels: floor(log2(v)) evaluated as "how many of torch's fp32 step points has v reached" -----
// v = sqrt(w*h)/finest + 1e-6 with IEEE sqrt/div/add in fp32 (hipcc's defaults are correctly rounded).
// Step points of torch-CPU floor(log2(.)): 2, 4, pred(8), pred(16)  (oracle/arfe_oracle.py LEVEL_STEPS_F32).
__global__ void map_roi_levels_kernel(int R, const float* __restrict__ rois, int num_levels, float finest,
                                      int64_t* __restrict__ levels) {
#pragma clang fp contract(off)
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    const float w = rois[5 * k + 3] - rois[5 * k + 1];
    const float h = rois[5 * k + 4] - rois[5 * k + 2];
    const float scale = __fsqrt_rn(w * h);
    const float v = __fdiv_rn(scale, finest) + 1e-6f;
    const float s3 = __uint_as_float(0x40FFFFFFu);   // largest fp32 below 8
    const float s4 = __uint_as_float(0x417FFFFFu);   // largest fp32 below 16
    int lvl = (v >= 2.0f) + (v >= 4.0f) + (v >= s3) + (v >= s4);   // NaN (negative area) compares false -> 0
    if (lvl > num_levels - 1) lvl = num_levels - 1;
    levels[k] = (int64_t)lvl;
}

// ---- RoIAlign v2 -----------------------------------------------------------------------------------------
struct RoiLevels {
    const float* feat[ARFE_MAX_LEVELS];
    float* dfeat[ARFE_MAX_LEVELS];
    int H[ARFE_MAX_LEVELS], W[ARFE_MAX_LEVELS];
    float scale[ARFE_MAX_LEVELS];
    int n, C, B;
};

struct RoiGeom {
    float y_start, x_start, bin_h, bin_w;
    int grid_h, grid_w, batch;
    float count;
};
__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int P, int sample_num) {
    RoiGeom g;
    g.batch = (int)roi[0];
    const float xs = roi[1] * scale - 0.5f, ys = roi[2] * scale - 0.5f;
    const float xe = roi[3] * scale - 0.5f, ye = roi[4] * scale - 0.5f;
    const float rw = xe - xs, rh = ye - ys;
    g.x_start = xs; g.y_start = ys;
    g.bin_h = rh / (float)P; g.bin_w = rw / (float)P;
    g.grid_h = sample_num > 0 ? sample_num : (int)ceilf(rh / (float)P);
    g.grid_w = sample_num > 0 ? sample_num : (int)ceilf(rw / (float)P);
    if (g.grid_h < 0) g.grid_h = 0;      // negative-size RoI (the reference asserts): contributes nothing
    if (g.grid_w < 0) g.grid_w = 0;
    const int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}
__device__ __forceinline__ bool axis_tap(float c, int n, int& lo, int& hi, float& wlo, float& whi) {
    if (c < -1.0f || c > (float)n) return false;
    if (c <= 0.f) c = 0.f;
    lo = (int)c;
    if (lo >= n - 1) { lo = hi = n - 1; c = (float)lo; }
    else hi = lo + 1;
    whi = c - (float)lo;
    wlo = 1.0f - whi;
    return true;
}

// one thread per (roi, channel, bin), bins fastest -> coalesced output
__global__ __launch_bounds__(256) void roialign_fwd_kernel(RoiLevels L, int R, const float* __restrict__ rois,
                                                           const int64_t* __restrict__ levels, int P, int sample_num,
                                                           float* __restrict__ out, int64_t out_roi_stride) {
    const int64_t total = (int64_t)R * L.C * P * P;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int pw = (int)(idx % P), ph = (int)((idx / P) % P);
    const int c = (int)((idx / (P * P)) % L.C);
    const int k = (int)(idx / ((int64_t)P * P * L.C));
    int lvl = (int)levels[k];
    lvl = lvl < 0 ? 0 : (lvl >= L.n ? L.n - 1 : lvl);
    int H = L.H[0], W = L.W[0];
    float scale = L.scale[0];
    const float* feat = L.feat[0];
#pragma unroll
    for (int i = 1; i < ARFE_MAX_LEVELS; ++i)
        if (i == lvl) { H = L.H[i]; W = L.W[i]; scale = L.scale[i]; feat = L.feat[i]; }
    const RoiGeom g = roi_geom(rois + 5 * (int64_t)k, scale, P, sample_num);
    const float* f = feat + ((int64_t)g.batch * L.C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
        const float y = g.y_start + ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
        int y0, y1; float wy0, wy1;
        const bool oky = axis_tap(y, H, y0, y1, wy0, wy1);
        for (int ix = 0; ix < g.grid_w; ++ix) {
            const float x = g.x_start + pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
            int x0, x1; float wx0, wx1;
            const bool okx = axis_tap(x, W, x0, x1, wx0, wx1);
            if (!(oky && okx)) continue;
            acc += wy0 * wx0 * f[y0 * W + x0] + wy0 * wx1 * f[y0 * W + x1] + wy1 * wx0 * f[y1 * W + x0] +
                   wy1 * wx1 * f[y1 * W + x1];
        }
    }
    out[(int64_t)k * out_roi_stride + (int64_t)c * P * P + ph * P + pw] = acc / g.count;
}

__global__ __launch_bounds__(256) void roialign_bwd_kernel(RoiLevels L, int R, const float* __restrict__ rois,
                                                           const int64_t* __restrict__ levels, int P, int sample_num,
                                                           const float* __restrict__ gout, int64_t g_roi_stride) {
    const int64_t total = (int64_t)R * L.C * P * P;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int pw = (int)(idx % P), ph = (int)((idx / P) % P);
    const int c = (int)((idx / (P * P)) % L.C);
    const int k = (int)(idx / ((int64_t)P * P * L.C));
    int lvl = (int)levels[k];
    lvl = lvl < 0 ? 0 : (lvl >= L.n ? L.n - 1 : lvl);
    int H = L.H[0], W = L.W[0];
    float scale = L.scale[0];
    float* dfeat = L.dfeat[0];
#pragma unroll
    for (int i = 1; i < ARFE_MAX_LEVELS; ++i)
        if (i == lvl) { H = L.H[i]; W = L.W[i]; scale = L.scale[i]; dfeat = L.dfeat[i]; }
    const RoiGeom g = roi_geom(rois + 5 * (int64_t)k, scale, P, sample_num);
    float* d = dfeat + ((int64_t)g.batch * L.C + c) * H * W;
    const float gv = gout[(int64_t)k * g_roi_stride + (int64_t)c * P * P + ph * P + pw];
    const float cnt = (float)(g.grid_h * g.grid_w);
    for (int iy = 0; iy < g.grid_h; ++iy) {
        const float y = g.y_start + ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
        int y0, y1; float wy0, wy1;
        const bool oky = axis_tap(y, H, y0, y1, wy0, wy1);
        for (int ix = 0; ix < g.grid_w; ++ix) {
            const float x = g.x_start + pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
            int x0, x1; float wx0, wx1;
            const bool okx = axis_tap(x, W, x0, x1, wx0, wx1);
            if (!(oky && okx)) continue;
            unsafeAtomicAdd(d + y0 * W + x0, gv * (wy0 * wx0) / cnt);
            unsafeAtomicAdd(d + y0 * W + x1, gv * (wy0 * wx1) / cnt);
            unsafeAtomicAdd(d + y1 * W + x0, gv * (wy1 * wx0) / cnt);
            unsafeAtomicAdd(d + y1 * W + x1, gv * (wy1 * wx1) / cnt);
        }
    }
}

// ---- fusion: z = ori + ori * (lw + lh) -----------------------------------------------------------------------
__global__ void rff_fuse_fwd_kernel(int64_t total, int per_roi, const float* __restrict__ ori, int64_t ori_roi_stride,
                                    const float* __restrict__ lw, const float* __restrict__ lh, float* __restrict__ z) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / per_roi;
    const int q = (int)(i - r * per_roi);
    const float o = ori[r * ori_roi_stride + q];
    z[i] = o + o * (lw[i] + lh[i]);
}
__global__ void rff_fuse_bwd_kernel(int64_t total, int per_roi, const float* __restrict__ ori, int64_t ori_roi_stride,
                                    const float* __restrict__ lw, const float* __restrict__ lh, const float* __restrict__ dz,
                                    float* __restrict__ d_ori, int64_t d_ori_roi_stride, float* __restrict__ dl) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t r = i / per_roi;
    const int q = (int)(i - r * per_roi);
    const float o = ori[r * ori_roi_stride + q];
    const float g = dz[i];
    d_ori[r * d_ori_roi_stride + q] = g + g * (lw[i] + lh[i]);
    dl[i] = g * o;
}

static int32_t fill_roi_levels(RoiLevels& L, const arfe_pyramid_t* pyr, const float* const* feats, float* const* dfeats,
                               const int32_t* strides) {
    L.n = pyr->num_levels; L.C = pyr->channels; L.B = pyr->batch;
    for (int i = 0; i < ARFE_MAX_LEVELS; ++i) {
        L.feat[i] = nullptr; L.dfeat[i] = nullptr; L.H[i] = 1; L.W[i] = 1; L.scale[i] = 1.f;
    }
    for (int i = 0; i < L.n; ++i) {
        if (strides[i] <= 0) return ARFE_ERR_ARG;
        if (feats) { if (!feats[i]) return ARFE_ERR_ARG; L.feat[i] = feats[i]; }
        if (dfeats) { if (!dfeats[i]) return ARFE_ERR_ARG; L.dfeat[i] = dfeats[i]; }
        L.H[i] = pyr->height[i]; L.W[i] = pyr->width[i];
        L.scale[i] = 1.0f / (float)strides[i];     // spatial_scale = 1 / s (single_level.py:50)
    }
    return ARFE_OK;
}

}  // namespace arfe

using namespace arfe;

extern "C" int32_t arfe_rff_make_rois(int32_t R, const float* rois, float* h_rois, float* w_rois, void* stream) {
    ARFE_CHECK_ARG(R >= 0, ARFE_ERR_ARG, "rff_make_rois: R < 0");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(rois && h_rois && w_rois, ARFE_ERR_ARG, "rff_make_rois: null pointer");
    hipLaunchKernelGGL(rff_make_rois_kernel, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), R, rois, h_rois, w_rois);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_map_roi_levels(int32_t R, const float* rois, int32_t num_levels, float finest_scale,
                                       int64_t* levels, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && num_levels >= 1 && num_levels <= 5 && finest_scale > 0.f, ARFE_ERR_ARG,
                   "map_roi_levels: bad arguments (num_levels=%d)", num_levels);
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(rois && levels, ARFE_ERR_ARG, "map_roi_levels: null pointer");
    hipLaunchKernelGGL(map_roi_levels_kernel, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), R, rois, num_levels,
                       finest_scale, levels);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_roialign_fwd(const arfe_pyramid_t* pyr, const float* const* feats, const int32_t* strides,
                                     int32_t R, const float* rois, const int64_t* levels, int32_t pooled,
                                     int32_t sample_num, float* out, int64_t out_roi_stride, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    ARFE_CHECK_ARG(R >= 0 && pooled > 0 && pooled <= 16 && sample_num >= 0, ARFE_ERR_ARG, "roialign_fwd: bad arguments");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(feats && strides && rois && levels && out, ARFE_ERR_ARG, "roialign_fwd: null pointer");
    RoiLevels L;
    ARFE_CHECK_ARG(fill_roi_levels(L, pyr, feats, nullptr, strides) == ARFE_OK, ARFE_ERR_ARG, "roialign_fwd: bad level table");
    const int64_t total = (int64_t)R * L.C * pooled * pooled;
    hipLaunchKernelGGL(roialign_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), L, R, rois,
                       levels, pooled, sample_num, out, out_roi_stride);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_roialign_bwd(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides, int32_t R,
                                     const float* rois, const int64_t* levels, int32_t pooled, int32_t sample_num,
                                     const float* gout, int64_t gout_roi_stride, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    ARFE_CHECK_ARG(R >= 0 && pooled > 0 && pooled <= 16 && sample_num >= 0, ARFE_ERR_ARG, "roialign_bwd: bad arguments");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(dfeats && strides && rois && levels && gout, ARFE_ERR_ARG, "roialign_bwd: null pointer");
    RoiLevels L;
    ARFE_CHECK_ARG(fill_roi_levels(L, pyr, nullptr, dfeats, strides) == ARFE_OK, ARFE_ERR_ARG, "roialign_bwd: bad level table");
    const int64_t total = (int64_t)R * L.C * pooled * pooled;
    hipLaunchKernelGGL(roialign_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), L, R, rois,
                       levels, pooled, sample_num, gout, gout_roi_stride);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_rff_fuse_fwd(int32_t R, int32_t C, const float* ori, int64_t ori_roi_stride, const float* lw,
                                     const float* lh, float* z, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0, ARFE_ERR_ARG, "rff_fuse_fwd: bad shape");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(ori && lw && lh && z, ARFE_ERR_ARG, "rff_fuse_fwd: null pointer");
    const int64_t total = (int64_t)R * C * 49;
    hipLaunchKernelGGL(rff_fuse_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), total,
                       C * 49, ori, ori_roi_stride, lw, lh, z);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_rff_fuse_bwd(int32_t R, int32_t C, const float* ori, int64_t ori_roi_stride, const float* lw,
                                     const float* lh, const float* dz, float* d_ori, int64_t d_ori_roi_stride, float* dl,
                                     void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0, ARFE_ERR_ARG, "rff_fuse_bwd: bad shape");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(ori && lw && lh && dz && d_ori && dl, ARFE_ERR_ARG, "rff_fuse_bwd: null pointer");
    const int64_t total = (int64_t)R * C * 49;
    hipLaunchKernelGGL(rff_fuse_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), total,
                       C * 49, ori, ori_roi_stride, lw, lh, dz, d_ori, d_ori_roi_stride, dl);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
