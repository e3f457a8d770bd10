// AR-RFF RoI-side kernels: enlarged-RoI generation, RoI -> level assignment (bit-exact), RoIAlign v2
// forward/backward fused with the per-level dispatch, and the elementwise fusion z = ori + ori*(lw+lh).
#include "common.hpp"
#include <type_traits>

namespace arfe {

// ---- get_adaptive_scale_rois(rois, 1) — same fp32 operation order as additional.py:38-71 -----------------
// contraction is switched off so no mul+add pair is fused into an fma (the reference rounds every op).
__device__ __forceinline__ void make_rois_one(const float* __restrict__ roi, float* __restrict__ h, float* __restrict__ w) {
#pragma clang fp contract(off)
    const float b = roi[0], x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float cx = (x1 + x2) * 0.5f, cy = (y1 + y2) * 0.5f;
    const float rw = x2 - x1 + 1.0f, rh = y2 - y1 + 1.0f;
    const float facs = 1.0f;
    const float h_rate = (rw / rh) * facs + 1.0f;
    const float w_rate = (rh / rw) * facs + 1.0f;
    const float large_h = rh * h_rate, large_w = rw * w_rate;
    const float lo = 0.1f;   // torch.ones_like(rw) * 0.1 == fp32(0.1)
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
__global__ void rff_make_rois_kernel(int R, const float* __restrict__ rois, float* __restrict__ h_rois,
                                     float* __restrict__ w_rois) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    make_rois_one(rois + 5 * k, h_rois + 5 * k, w_rois + 5 * k);
}

// ---- map_roi_levels: floor(log2(v)) evaluated as "how many of torch's fp32 step points has v reached" -----
// v = sqrt(w*h)/finest + 1e-6 with IEEE sqrt/div/add in fp32 (hipcc's defaults are correctly rounded).
// Step points of torch-CPU floor(log2(.)): 2, 4, pred(8), pred(16)  (oracle/arfe_oracle.py LEVEL_STEPS_F32).
__device__ __forceinline__ int64_t map_level_one(const float* __restrict__ roi, int num_levels, float finest) {
#pragma clang fp contract(off)
    const float w = roi[3] - roi[1];
    const float h = roi[4] - roi[2];
    const float scale = __fsqrt_rn(w * h);
    const float v = __fdiv_rn(scale, finest) + 1e-6f;
    const float s3 = __uint_as_float(0x40FFFFFFu);   // largest fp32 below 8
    const float s4 = __uint_as_float(0x417FFFFFu);   // largest fp32 below 16
    int lvl = (v >= 2.0f) + (v >= 4.0f) + (v >= s3) + (v >= s4);   // NaN (negative area) compares false -> 0
    if (lvl > num_levels - 1) lvl = num_levels - 1;
    return (int64_t)lvl;
}
__global__ void map_roi_levels_kernel(int R, const float* __restrict__ rois, int num_levels, float finest,
                                      int64_t* __restrict__ levels) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    levels[k] = map_level_one(rois + 5 * k, num_levels, finest);
}
// the AR-RFF glue in one launch: both enlarged RoI sets and the level of all three sets (the enlarged RoIs are read
// back from the values just stored, i.e. exactly the fp32 numbers the two-call form sees)
__global__ void rff_roi_sets_kernel(int R, const float* __restrict__ rois, int num_levels, float finest,
                                    float* __restrict__ h_rois, float* __restrict__ w_rois, int64_t* __restrict__ lv_ori,
                                    int64_t* __restrict__ lv_w, int64_t* __restrict__ lv_h) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    float h[5], w[5];
    make_rois_one(rois + 5 * k, h, w);
#pragma unroll
    for (int e = 0; e < 5; ++e) { h_rois[5 * k + e] = h[e]; w_rois[5 * k + e] = w[e]; }
    lv_ori[k] = map_level_one(rois + 5 * k, num_levels, finest);
    lv_w[k] = map_level_one(w, num_levels, finest);
    lv_h[k] = map_level_one(h, num_levels, finest);
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

// ---- RoIAlign backward without atomics ------------------------------------------------------------------------
// The scatter form (one thread per output bin, 4 atomics per sample) moves ~5 GB of dword atomics per step and
// ran at 9 ms per RoI set.  The bilinear sample weights are separable (w = wy(row) * wx(col)), so the gradient
// of one RoI over a feature tile is   dF[c][y][x] = (1/count) * sum_ph wy[ph][y] * sum_pw wx[pw][x] * G[c][ph][pw].
// RoIs (of all sets) are binned per 16x16 feature tile; one workgroup owns (tile, 32 channels), walks its RoI
// list and WRITES every element of dfeats exactly once (no memset, no atomics on the feature maps).
constexpr int RT = 16;          // tile edge
#ifndef ARFE_RB_NB
#define ARFE_RB_NB 2
#endif
#ifndef ARFE_RB_CC
#define ARFE_RB_CC 32
#endif
constexpr int RB_CC = ARFE_RB_CC;   // channels per workgroup (round 3: 32, a wave owns 8; the per-RoI bookkeeping - list entries, header,
                                    // weight staging, two barriers - is paid once per workgroup and was 2/3 of the kernel's time at 16)
constexpr int RB_CPW = RB_CC / 4;   // channels per wave
constexpr int RB_MAXSETS = 4;

struct RoiSets {
    const float* rois[RB_MAXSETS];
    const int64_t* levels[RB_MAXSETS];
    int nsets, R;
};
struct RoiTileTable {
    int first[ARFE_MAX_LEVELS + 1];     // first tile id of each level (all images of a level are consecutive)
    int tiles_x[ARFE_MAX_LEVELS], tiles_y[ARFE_MAX_LEVELS];
};

// conservative feature-row/column range touched by a RoI axis: samples lie strictly inside (start, start+size)
__device__ __forceinline__ bool axis_range(float start, float size, int n, int& lo, int& hi) {
    const float end = start + size;
    if (!(size >= 0.f) || end < -1.0f || start > (float)n) return false;
    int a = (int)floorf(start), b = (int)floorf(end) + 1;
    lo = a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
    hi = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
    return true;
}

struct RoiWeightHdr {
    int id, lvl, img;               // RoI (set * R + k) of this slot, its level and image
    int y0, x0, big; float inv;
    int hext, wext;                 // rows / columns of the window that carry weight
    int ylo[7], yhi[7];             // row support [ylo, yhi) of bin ph (offsets into the table)
    int xlo[7], xhi[7];             // column support of bin pw
};
template <bool FILL>
__global__ void roi_bin_kernel(RoiLevels L, RoiSets S, RoiTileTable T, int P, int* __restrict__ counts,
                               const int* __restrict__ offsets, int* __restrict__ cursors, int* __restrict__ list,
                               const RoiWeightHdr* __restrict__ hdr, int* __restrict__ lmeta) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= S.nsets * S.R) return;
    const int s = id / S.R, k = id - s * S.R;
    const float* roi = S.rois[s] + 5 * (int64_t)k;
    int lvl = (int)S.levels[s][k];
    lvl = lvl < 0 ? 0 : (lvl >= L.n ? L.n - 1 : lvl);
    const float scale = L.scale[lvl];
    const int H = L.H[lvl], W = L.W[lvl];
    const int b = (int)roi[0];
    if (b < 0 || b >= L.B) return;
    const float xs = roi[1] * scale - 0.5f, ys = roi[2] * scale - 0.5f;
    const float xe = roi[3] * scale - 0.5f, ye = roi[4] * scale - 0.5f;
    int y0, y1, x0, x1;
    if (!axis_range(ys, ye - ys, H, y0, y1) || !axis_range(xs, xe - xs, W, x0, x1)) return;
    // the window origin and form of the weight table travel with the list entry, so that the tile kernel's reads of
    // table and gradients depend on ONE earlier read (the entry), not on entry -> header -> table
    int meta = 0;
    if (FILL) meta = (hdr[id].big << 28) | ((hdr[id].y0 & 0x3fff) << 14) | (hdr[id].x0 & 0x3fff);
    for (int ty = y0 / RT; ty <= y1 / RT; ++ty)
        for (int tx = x0 / RT; tx <= x1 / RT; ++tx) {
            const int tile = T.first[lvl] + (b * T.tiles_y[lvl] + ty) * T.tiles_x[lvl] + tx;
            if (FILL) {
                const int pos = atomicAdd(cursors + tile, 1);
                list[offsets[tile] + pos] = id;
                lmeta[offsets[tile] + pos] = meta;
            } else {
                atomicAdd(counts + tile, 1);
            }
        }
}

// exclusive scan of counts -> offsets (single workgroup), cursors zeroed
__global__ __launch_bounds__(1024) void roi_bin_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets,
                                                            int* __restrict__ cursors, int ntiles) {
    __shared__ int wave_tot[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < ntiles; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < ntiles ? counts[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int j = 0; j < wv; ++j) wave_off += wave_tot[j];
        const int c = carry;
        if (i < ntiles) { offsets[i] = c + wave_off + incl - v; cursors[i] = 0; }
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[ntiles] = carry;
}

// The fill above appends through atomic cursors, so a tile's list comes out in a different order every run - and with it the
// order in which the backward adds the RoIs' contributions.  One wave per tile puts its list in RoI order (rank by counting
// in LDS; lists longer than RL_SORT_MAX keep their arrival order), which makes the backward bit-reproducible.
constexpr int RL_SORT_MAX = 2048;
__global__ __launch_bounds__(64) void roi_list_sort_kernel(const int* __restrict__ offsets, int* __restrict__ list,
                                                          int* __restrict__ lmeta) {
    __shared__ int id_s[RL_SORT_MAX], meta_s[RL_SORT_MAX];
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int beg = offsets[tile], n = offsets[tile + 1] - beg;
    if (n < 2 || n > RL_SORT_MAX) return;
    for (int i = lane; i < n; i += 64) { id_s[i] = list[beg + i]; meta_s[i] = lmeta[beg + i]; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {
        const int mine = id_s[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += id_s[j] < mine ? 1 : 0;      // ids are unique within a list
        list[beg + rank] = mine;
        lmeta[beg + rank] = meta_s[i];
    }
}

// 1-D bilinear weight of feature index `target` for output bin `pb`: sum over the bin's grid samples
__device__ __forceinline__ float axis_weight(float start, float bin, int grid, int pb, int n, int target) {
    float w = 0.f;
    for (int i = 0; i < grid; ++i) {
        const float c = start + pb * bin + ((float)i + 0.5f) * bin / (float)grid;
        int lo, hi; float wlo, whi;
        if (!axis_tap(c, n, lo, hi, wlo, whi)) continue;
        if (lo == target) w += wlo;
        if (hi == target) w += whi;
    }
    return w;
}

// Per-RoI 1-D bilinear weight tables (one small workgroup per RoI, all sets): for the RoI's first RW_EXT feature
// rows / columns, w[axis][bin][i] = sum over the bin's grid samples of the tap weight on feature index origin+i.
// The tile kernel then only copies its 16-row / 16-column slice instead of re-deriving the geometry per tile.
constexpr int RW_EXT = 160;    // table entries per (axis, bin): windows up to 159 rows / columns (was 64: 5 % of the bench's RoIs -
                               // thin long boxes on the finest level, windows up to 135 - fell to the per-sample forms below and
                               // made the tail of both kernels: 349 -> 281 us forward, 487 -> 426 us backward without them)
constexpr int RW_SLACK = 16;   // entries past a window's extent that are still written (zeros): a 16-row tile or an 8-row read
                               // group that overlaps the window's end reads that far
// big: 0 table valid, 1 window larger than the table (consumers compute on the fly), 2 no sample inside the map

__global__ __launch_bounds__(256) void roi_weight_table_kernel(RoiLevels L, RoiSets S, int P, int sample_num,
                                                               RoiWeightHdr* __restrict__ hdr, float* __restrict__ wtab,
                                                               const int32_t* __restrict__ order, int prescale_y) {
    __shared__ float w_s[2][7][RW_EXT];
    // slot blockIdx.x of the tables describes RoI order[slot] (forward: spatial processing order) or RoI slot itself
    const int slot = blockIdx.x;
    const int id = order ? order[slot] : slot;
    const int s = id / S.R, k = id - s * S.R;
    int lvl = (int)S.levels[s][k];
    lvl = lvl < 0 ? 0 : (lvl >= L.n ? L.n - 1 : lvl);
    const int H = L.H[lvl], W = L.W[lvl];
    const RoiGeom g = roi_geom(S.rois[s] + 5 * (int64_t)k, L.scale[lvl], P, sample_num);
    int y0 = 0, y1 = -1, x0 = 0, x1 = -1;
    const bool oky = axis_range(g.y_start, g.bin_h * (float)P, H, y0, y1);
    const bool okx = axis_range(g.x_start, g.bin_w * (float)P, W, x0, x1);
    const int big = !(oky && okx) ? 2 : ((y1 - y0 >= RW_EXT - 1 || x1 - x0 >= RW_EXT - 1) ? 1 : 0);
    RoiWeightHdr* h = hdr + slot;
    if (threadIdx.x == 0) {
        h->id = id; h->lvl = lvl; h->img = (int)S.rois[s][5 * (int64_t)k];
        h->y0 = y0; h->x0 = x0; h->big = big;
        h->inv = 1.0f / (float)(g.grid_h * g.grid_w > 1 ? g.grid_h * g.grid_w : 1);
    }
    if (big) return;
    float* wt = wtab + (int64_t)slot * 2 * 7 * RW_EXT;
    // only the window's extent (+ slack) is evaluated and written: nothing reads past it
    const int ny = y1 - y0 + 1 + RW_SLACK < RW_EXT ? y1 - y0 + 1 + RW_SLACK : RW_EXT;
    const int nxw = x1 - x0 + 1 + RW_SLACK < RW_EXT ? x1 - x0 + 1 + RW_SLACK : RW_EXT;
    for (int i = threadIdx.x; i < 2 * 7 * RW_EXT; i += 256) {
        const int axis = i / (7 * RW_EXT), rem = i - axis * 7 * RW_EXT;
        const int pb = rem / RW_EXT, off = rem - pb * RW_EXT;
        if (off >= (axis == 0 ? ny : nxw)) continue;
        float w = axis == 0 ? axis_weight(g.y_start, g.bin_h, g.grid_h, pb, H, y0 + off)
                            : axis_weight(g.x_start, g.bin_w, g.grid_w, pb, W, x0 + off);
        if ((axis == 0 ? y0 : x0) + off >= (axis == 0 ? H : W)) w = 0.f;
        if (prescale_y && axis == 0) w *= 1.0f / (float)(g.grid_h * g.grid_w > 1 ? g.grid_h * g.grid_w : 1);   // = h->inv
        wt[i] = w;
        (&w_s[0][0][0])[i] = w;
    }
    __syncthreads();
    if (threadIdx.x < 14) {            // support of each bin along each axis
        const int axis = threadIdx.x / 7, pb = threadIdx.x - axis * 7;
        int lo = 0, hi = 0;
        bool any = false;
        for (int i = 0; i < (axis == 0 ? ny : nxw); ++i)
            if (w_s[axis][pb][i] != 0.f) { if (!any) { lo = i; any = true; } hi = i + 1; }
        if (axis == 0) { h->ylo[pb] = lo; h->yhi[pb] = hi; } else { h->xlo[pb] = lo; h->xhi[pb] = hi; }
        w_s[axis][pb][0] = (float)hi;  // reuse as scratch for the extents (column 0 is consumed above)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int he = 0, we = 0;
        for (int pb = 0; pb < 7; ++pb) {
            if ((int)w_s[0][pb][0] > he) he = (int)w_s[0][pb][0];
            if ((int)w_s[1][pb][0] > we) we = (int)w_s[1][pb][0];
        }
        h->hext = he; h->wext = we;
    }
}

// ---- RoIAlign forward, separable form ---------------------------------------------------------------------------
// out[c][ph][pw] = inv * sum_y wy[ph][y] * sum_x wx[pw][x] * F[c][y][x] over the RoI's feature window, using the
// per-RoI weight tables above.  One workgroup = (RoI of any set, RF_CH channels); a wave owns one channel at a
// time with lanes along the window's columns, so every feature row is ONE coalesced read (the sampling form
// issues 4 gathers per sample point), accumulates U[ph] per column while walking down the rows, and finishes
// with a 49-lane contraction over columns through LDS.  Windows wider/taller than RW_EXT (extreme aspect
// ratios) take the sampling form inside the same kernel.
// Processing order of the forward pass: level, image, then 8-row bands swept left to right, so that the RoIs in flight
// at any time read one neighbourhood of one feature map (L2 reuse between overlapping RoIs).
__global__ __launch_bounds__(256) void roi_key_kernel(RoiLevels L, RoiSets S, int n, int P, int sample_num,
                                                      uint32_t* __restrict__ keys) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int s = id / S.R, k = id - s * S.R;
    int lvl = (int)S.levels[s][k];
    lvl = lvl < 0 ? 0 : (lvl >= L.n ? L.n - 1 : lvl);
    const float* roi = S.rois[s] + 5 * (int64_t)k;
    const RoiGeom g = roi_geom(roi, L.scale[lvl], P, sample_num);
    int y0 = 0, y1 = -1, x0 = 0, x1 = -1;
    axis_range(g.y_start, g.bin_h * (float)P, L.H[lvl], y0, y1);
    axis_range(g.x_start, g.bin_w * (float)P, L.W[lvl], x0, x1);
    const int img = (int)roi[0];
    keys[id] = ((uint32_t)lvl << 29) | (((uint32_t)img & 0x1ffu) << 20) | ((((uint32_t)y0 >> 3) & 0x3ffu) << 10) |
               (((uint32_t)x0 >> 2) & 0x3ffu);
}

// order[rank] = id, rank = number of RoIs with a smaller (key, id): a counting rank is enough for a few thousand RoIs.
// One wave per RoI, the lanes share the comparisons (a thread-per-RoI loop kept 12 CUs busy for 80 us at 3072 RoIs).
// Round 3: the order is GROUP-major - all RoIs of one set first, in spatial order, then the next set - so that a caller can
// launch the sets one after the other (arfe_roialign_fwd_group) and start a consumer of one set's slice while the next set is
// still being extracted.  Processing group g holds set (g + 1) % nsets: with the AR-RFF sets (ori, w, h) the two enlarged-RoI
// sets, which feed the gating convs, come first and the original RoIs, needed only at the fusion, last.
__device__ __forceinline__ int roi_group_of_set(int s, int nsets) { return s == 0 ? nsets - 1 : s - 1; }
__global__ __launch_bounds__(256) void roi_rank_kernel(int n, const uint32_t* __restrict__ keys, int32_t* __restrict__ order,
                                                       int R, int nsets) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const uint32_t mine = keys[i];
    const int gi = roi_group_of_set(i / R, nsets);
    int cnt = 0;
    for (int j = lane; j < n; j += 64) {
        const uint32_t o = keys[j];
        const int gj = roi_group_of_set(j / R, nsets);
        cnt += (gj < gi || (gj == gi && (o < mine || (o == mine && j < i)))) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) order[cnt] = i;
}

constexpr int RF_ROWS = 8;  // window rows in flight per lane
constexpr int RF_CH = 32;   // channels per block: C/32 = 8 slices of a 256-channel map, one per XCD (see the block map)
// FT / OT: storage type of the feature maps / of the output (float, or _Float16 in the reference's fp16 mode, where the
// extractor reads the half maps a neck handed on and hands half RoI features to the head: the values it computes with are the
// same as after a conversion pass, the arithmetic is fp32 either way - single_level.py:109 @force_fp32(out_fp16=True))
template <class FT, class OT>
__global__ __launch_bounds__(256) void roialign_fwd_sep_kernel(RoiLevels L, RoiSets S, int P, int sample_num,
                                                               const RoiWeightHdr* __restrict__ hdr,
                                                               const float* __restrict__ wtab, int nslices,
                                                               OT* __restrict__ out, int64_t out_roi_stride, int slot0) {
    __shared__ __attribute__((aligned(16))) float wyT_s[RW_EXT][8];   // [window row][bin], pre-scaled by 1/samples
    __shared__ float wx_s[7][RW_EXT];
    __shared__ float U_s[4][7][64];                                    // one 64-column chunk of a wave's row sums
    __shared__ RoiWeightHdr h_s;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): the channel slice is the fast index, so
    // with 8 slices an XCD only ever sees its own 32 channels of the pyramid, and the RoIs come in spatial order.
    // Slot r of the header / weight tables was written for the r-th RoI of that order, so everything this workgroup
    // needs hangs off blockIdx alone (one global round trip before the window reads, not a chain of four).
    const int slot = slot0 + blockIdx.x / nslices;
    const RoiWeightHdr* hp = hdr + slot;
    const int id = hp->id;
    const int s = id / S.R, k = id - s * S.R;
    const int c0 = (blockIdx.x % nslices) * RF_CH;
    const int ncc = L.C - c0 < RF_CH ? L.C - c0 : RF_CH;
    const int lvl = hp->lvl;
    int H = L.H[0], W = L.W[0];
    const FT* feat = (const FT*)L.feat[0];
#pragma unroll
    for (int i = 1; i < ARFE_MAX_LEVELS; ++i)
        if (i == lvl) { H = L.H[i]; W = L.W[i]; feat = (const FT*)L.feat[i]; }
    const int bimg = hp->img;
    OT* obase = out + (int64_t)k * out_roi_stride + ((int64_t)s * L.C + c0) * 49;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int big = hp->big;

    if (big == 2) {     // every sample falls outside the feature map (or negative size): zeros
        for (int o = tid; o < ncc * 49; o += 256) obase[o] = (OT)0.f;
        return;
    }
    if (big) {          // sampling form (reference order), one thread per output element
        const RoiGeom g = roi_geom(S.rois[s] + 5 * (int64_t)k, L.scale[lvl], P, sample_num);
        for (int o = tid; o < ncc * 49; o += 256) {
            const int c = o / 49, q = o - c * 49, ph = q / 7, pw = q - ph * 7;
            const FT* f = feat + ((int64_t)bimg * L.C + c0 + c) * H * W;
            float acc = 0.f;
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const float yy = g.y_start + ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
                int y0, y1; float wy0, wy1;
                const bool oky = axis_tap(yy, H, y0, y1, wy0, wy1);
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    const float xx = g.x_start + pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
                    int x0, x1; float wx0, wx1;
                    const bool okx = axis_tap(xx, W, x0, x1, wx0, wx1);
                    if (!(oky && okx)) continue;
                    acc += wy0 * wx0 * (float)f[y0 * W + x0] + wy0 * wx1 * (float)f[y0 * W + x1] + wy1 * wx0 * (float)f[y1 * W + x0] +
                           wy1 * wx1 * (float)f[y1 * W + x1];
                }
            }
            obase[o] = (OT)(acc / g.count);
        }
        return;
    }
    // ---- header + weight tables -> LDS ----
    if (tid < (int)(sizeof(RoiWeightHdr) / sizeof(int))) ((int*)&h_s)[tid] = ((const int*)hp)[tid];
    const float* wt = wtab + (int64_t)slot * 2 * 7 * RW_EXT;
    const float inv = hp->inv;
    {
        const int hn = hp->hext + 8 < RW_EXT ? hp->hext + 8 : RW_EXT, wn = hp->wext < RW_EXT ? hp->wext : RW_EXT;
        for (int i = tid; i < 7 * RW_EXT; i += 256) {
            const int pb = i / RW_EXT, off = i - pb * RW_EXT;
            if (off < hn) wyT_s[off][pb] = wt[i] * inv;
            if (off < wn) (&wx_s[0][0])[i] = wt[7 * RW_EXT + i];
        }
        if (tid < hn) wyT_s[tid][7] = 0.f;
    }
    __syncthreads();
#if defined(ARFE_PROBES) && defined(ARFE_RF_PROBE) && (ARFE_RF_PROBE & 8)
    if (wyT_s[1][1] != 123.456f) return;
#endif
    const int wext = h_s.wext, hext = h_s.hext;
    // A lane is one window column; narrow windows put 4 (<= 16 columns) or 2 (<= 32) channels on one wave, so that a
    // row read (one dword per lane: it costs the address unit as much as a 16-byte one) carries several channels.
    // Every window row is read ONCE, eight rows in flight, and folded into all seven bins with the row's weight
    // vector (zero outside a bin's support) - no per-bin loops whose reads would queue behind each other.
    const int psh = wext <= 16 ? 2 : (wext <= 32 ? 1 : 0), pack = 1 << psh, pw_cols = 64 >> psh;
    const int half = lane >> (6 - psh);
    const int col = lane & (pw_cols - 1);
    const int colc = col < wext ? col : wext - 1;          // lanes past the window re-read its last column (unused)
    const int q = lane < 49 ? lane : 48, qph = q / 7, qpw = q - qph * 7;
    const int xlo = h_s.xlo[qpw], xhi = h_s.xhi[qpw];
    if (wext > 64) {
        // windows wider than a wave (5 % of the bench's RoIs: thin long boxes): 64 columns at a time, one channel per wave,
        // the column contraction adds up over the chunks.  Kept apart from the loop below so that its registers stay few.
        const int nchunk = (wext + 63) >> 6;
        for (int cb = wave; cb < ncc; cb += 4) {
            float wide_sum = 0.f;
            for (int ch = 0; ch < nchunk; ++ch) {
                const int colk = ch * 64 + lane < wext ? ch * 64 + lane : wext - 1;
                const FT* f = feat + (((int64_t)bimg * L.C + c0 + cb) * H + h_s.y0) * W + h_s.x0 + colk;
                float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int y = 0; y < hext; y += 4) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (float)f[(int64_t)(y + j < hext ? y + j : hext - 1) * W];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4_u wa = *reinterpret_cast<const float4_u*>(&wyT_s[y + j][0]);
                        const float4_u wb = *reinterpret_cast<const float4_u*>(&wyT_s[y + j][4]);
                        acc[0] += wa[0] * v[j]; acc[1] += wa[1] * v[j]; acc[2] += wa[2] * v[j]; acc[3] += wa[3] * v[j];
                        acc[4] += wb[0] * v[j]; acc[5] += wb[1] * v[j]; acc[6] += wb[2] * v[j];
                    }
                }
#pragma unroll
                for (int ph = 0; ph < 7; ++ph) U_s[wave][ph][lane] = acc[ph];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if (lane < 49) {
                    const int xa = xlo > ch * 64 ? xlo : ch * 64, xb = xhi < ch * 64 + 64 ? xhi : ch * 64 + 64;
                    for (int x = xa; x < xb; ++x) wide_sum += U_s[wave][qph][x - ch * 64] * wx_s[qpw][x];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);     // the row sums are read before the next chunk overwrites them
                __builtin_amdgcn_wave_barrier();
            }
            if (lane < 49) obase[cb * 49 + lane] = (OT)wide_sum;
        }
        return;
    }
    for (int cb = wave * pack; cb < ncc; cb += 4 * pack) {
        const int c = cb + half;
        const FT* f = feat + (((int64_t)bimg * L.C + c0 + (c < ncc ? c : cb)) * H + h_s.y0) * W + h_s.x0 + colc;
        // (the compiler already pairs the bins into v_pk_fma_f32; a hand-written 8-bin form with a scalar row pointer measured the
        // same 300 us, and so did the probes that removed the loads (245), the FMAs (222) or the column pass (258) one at a time:
        // 43 us of the kernel are its prologue, ~65 us the loop skeleton - no single bound)
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < hext; y += RF_ROWS) {
            float v[RF_ROWS];
#pragma unroll
#if defined(ARFE_PROBES) && defined(ARFE_RF_PROBE) && (ARFE_RF_PROBE & 1)
            for (int j = 0; j < RF_ROWS; ++j) v[j] = (float)(lane + y + j);
#else
            for (int j = 0; j < RF_ROWS; ++j) v[j] = (float)f[(int64_t)(y + j < hext ? y + j : hext - 1) * W];
#endif
            // rows past the window (y + j >= hext) carry zero weight in the table: no branch per row
#pragma unroll
            for (int j = 0; j < RF_ROWS; ++j) {
#if defined(ARFE_PROBES) && defined(ARFE_RF_PROBE) && (ARFE_RF_PROBE & 2)
                acc[j % 7] += v[j];
#else
                const float4_u wa = *reinterpret_cast<const float4_u*>(&wyT_s[y + j][0]);
                const float4_u wb = *reinterpret_cast<const float4_u*>(&wyT_s[y + j][4]);
                acc[0] += wa[0] * v[j]; acc[1] += wa[1] * v[j]; acc[2] += wa[2] * v[j]; acc[3] += wa[3] * v[j];
                acc[4] += wb[0] * v[j]; acc[5] += wb[1] * v[j]; acc[6] += wb[2] * v[j];
#endif
            }
        }
#pragma unroll
        for (int ph = 0; ph < 7; ++ph) U_s[wave][ph][lane] = acc[ph];
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): this wave's U_s writes have landed
        __builtin_amdgcn_wave_barrier();
        for (int hh = 0; hh < pack; ++hh) {
            if (lane < 49 && cb + hh < ncc) {
                float acc2 = 0.f;
#if defined(ARFE_PROBES) && defined(ARFE_RF_PROBE) && (ARFE_RF_PROBE & 4)
                acc2 = U_s[wave][qph][hh * pw_cols + xlo];
#else
                for (int x = xlo; x < xhi; ++x) acc2 += U_s[wave][qph][hh * pw_cols + x] * wx_s[qpw][x];
#endif
                obase[(cb + hh) * 49 + lane] = (OT)acc2;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// One workgroup = (16x16 feature tile, RB_CC channels).  RoIs of the tile's list are taken RB_NB at a time: their
// gradient blocks and 1-D weights (read one batch ahead, the list entries two) are staged in LDS, then each wave
// contracts its 4 channels on the matrix pipe:
//   C:  T[(c,ph)][x] = sum_pw G[c][ph][pw] wx[pw][x]      D:  dX_c[y][x] += sum_ph wy[y][ph] T_c[ph][x]
// (separable form of the bilinear scatter; every feature element is written once, no atomics unless a long list is cut).
constexpr int RB_NB = ARFE_RB_NB;
// Slicing a long RoI list over several workgroups (grid.z; the slices add atomically into pre-zeroed tiles) is OFF by default:
// the lists of the bench (up to 201 RoIs on a coarse tile, dispatched first) finish inside the kernel's run anyway, while every
// extra grid.z layer is 12 k workgroups that start only to exit - measured 434 us (32, 4) / 467 us (150, 2) / 391 us (off)
// for the backward, the same ranking at 1000 RoIs per image.  ARFE_RB_SEG=<min list length>,<slices> switches it on.
constexpr int RB_SEGS_DEF = 1;    // slices of a long RoI list
constexpr int RB_SEG_MIN_DEF = 1 << 30;   // lists up to this length stay in one workgroup (plain stores)
struct RbSeg { int segs, seg_min; };
static RbSeg rb_seg() {
    static const RbSeg v = []() {
        RbSeg r{RB_SEGS_DEF, RB_SEG_MIN_DEF};
        const char* e = getenv("ARFE_RB_SEG");
        if (e && sscanf(e, "%d,%d", &r.seg_min, &r.segs) == 2) {
            r.segs = r.segs < 1 ? 1 : (r.segs > 16 ? 16 : r.segs);      // grid.z of the tile kernel: never 0, never huge
            r.seg_min = r.seg_min < 1 ? 1 : r.seg_min;
        } else {
            r = RbSeg{RB_SEGS_DEF, RB_SEG_MIN_DEF};
        }
        return r;
    }();
    return v;
}
constexpr int RB_GS = 60, RB_TS = 20;
__device__ __forceinline__ int rb_wrow(int r) { return r * 8 + ((r >> 3) << 2); }   // start of weight row r (see the LDS note)
// Only the tiles whose RoI list is cut into slices are summed with atomics and need a zeroed output: every other tile
// (the empty ones included) is stored exactly once by the tile kernel.  Zeroing just those replaces a fill of the whole
// pyramid gradient (183 MB per launch at the bench size).
__global__ __launch_bounds__(256) void roi_zero_sliced_tiles_kernel(RoiLevels L, RoiTileTable T, const int* __restrict__ offsets,
                                                                    int ntiles, int nchunks, int RB_SEG_MIN) {
    const int tile = (int)(blockIdx.x / (unsigned)nchunks);
    if (offsets[tile + 1] - offsets[tile] <= RB_SEG_MIN) return;
    int lvl = 0;
    while (lvl + 1 < L.n && tile >= T.first[lvl + 1]) ++lvl;
    const int local = tile - T.first[lvl];
    const int tx = local % T.tiles_x[lvl];
    const int ty = (local / T.tiles_x[lvl]) % T.tiles_y[lvl];
    const int b = local / (T.tiles_x[lvl] * T.tiles_y[lvl]);
    const int H = L.H[lvl], W = L.W[lvl];
    const int c0 = (int)(blockIdx.x % (unsigned)nchunks) * RB_CC;
    float* d = L.dfeat[0];
#pragma unroll
    for (int i = 1; i < ARFE_MAX_LEVELS; ++i) if (i == lvl) d = L.dfeat[i];
    const int ox = tx * RT + (threadIdx.x & 15), oy = ty * RT + (threadIdx.x >> 4);
    if (ox >= W || oy >= H) return;
    for (int c = c0; c < c0 + RB_CC && c < L.C; ++c) d[(((int64_t)b * L.C + c) * H + oy) * W + ox] = 0.f;
    (void)ntiles;
}

static_assert(RB_CC % 8 == 0 && RT == 16, "the MFMA tiles are 16x16: a wave owns RB_CC / 4 channels (pairs in the C phase), a tile is 16x16 pixels");
// GT / DT: storage type of the incoming RoI-feature gradient / of the feature-map gradients (float or _Float16, see the forward)
template <class GT, class DT>
__global__ __launch_bounds__(256) void roialign_bwd_tiles_kernel(RoiLevels L, RoiSets S, RoiTileTable T, int P, int sample_num,
                                                                 const GT* __restrict__ gout, int64_t g_roi_stride,
                                                                 const int* __restrict__ offsets, const int* __restrict__ list,
                                                                 const int* __restrict__ lmeta,
                                                                 const float* __restrict__ wtab, int ntiles, int nchunks, int RB_SEGS, int RB_SEG_MIN) {
    // LDS strides are chosen for the MFMA fragment accesses (64 banks x 4 B; PMC on the unpadded arrays: 56 % of the LDS
    // cycles were bank conflicts): a channel of G takes 60 floats (the two channels of an A fragment then sit 60, not 56,
    // floats apart), a T row 20 floats (the four 16-lane groups of a result store land 80 / 160 floats apart),
    // weight rows 8..15 are shifted by 4 floats against rows 0..7.
    __shared__ __attribute__((aligned(16))) float G_s[RB_NB][RB_CC][RB_GS];      // [roi][c][ph*8 + pw], pw = 7 zero pad
    // [c][x][ph] with the two 4-bin halves of a column swapped for x >= 8: a lane's four results are ONE 16-byte store, and the
    // D-phase reads (lane -> column l15, bins lq and 4 + lq) of columns x and x + 8 fall on different banks
    __shared__ __attribute__((aligned(16))) float T_s[RB_CC][RT][8];
    __shared__ __attribute__((aligned(16))) float wy_s[RB_NB][RT * 8 + 4], wx_s[RB_NB][RT * 8 + 4];
    // coarse levels hold few tiles with long RoI lists: dispatch them first (reverse tile order) and cut a long
    // list into RB_SEGS slices handled by different workgroups (those add atomically into the zeroed output)
    // The channel chunk is the fast index of blockIdx.x: workgroups go round-robin to the 8 XCDs, so (16 chunks) an XCD
    // always works on the same two chunks, and the neighbouring tiles that re-read a RoI's gradient block for that
    // chunk follow each other on the same L2 (tile-fast order sent them to eight different L2s: 15 % hit rate).
    int tile = ntiles - 1 - (int)(blockIdx.x / (unsigned)nchunks), lvl = 0;
    const int lbeg = offsets[tile], lend = offsets[tile + 1];
    const int nseg = (lend - lbeg) > RB_SEG_MIN ? RB_SEGS : 1;
    const int seg = blockIdx.z;
    if (seg >= nseg) return;
    while (lvl + 1 < L.n && tile >= T.first[lvl + 1]) ++lvl;
    const int local = tile - T.first[lvl];
    const int tx = local % T.tiles_x[lvl];
    const int ty = (local / T.tiles_x[lvl]) % T.tiles_y[lvl];
    const int b = local / (T.tiles_x[lvl] * T.tiles_y[lvl]);
    const int H = L.H[lvl], W = L.W[lvl];
    const float scale = L.scale[lvl];
    const int c0 = (int)(blockIdx.x % (unsigned)nchunks) * RB_CC;
    const int ncc = L.C - c0 < RB_CC ? L.C - c0 : RB_CC;
    const int tid = threadIdx.x;
    // Both contractions run on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).  Wave v owns channels 4v..4v+3 of the chunk:
    //   C: T[(c,ph)][x] = sum_pw G[(c,ph)][pw] wx[pw][x]   (two channels = 14 of the 16 rows of one M tile, K = 8)
    //   D: dX_c[y][x] += sum_ph wy[y][ph] T_c[ph][x]        (accumulated over the RoIs of the list in the MFMA D tile)
    // D tile layout: lane l holds rows 4*(l/16)+r, r = 0..3, of column l%16.
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    f32x4_t acc[RB_CPW];
#pragma unroll
    for (int c = 0; c < RB_CPW; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // pad lanes of the k = 8 contractions (bin 7) are never written by the staging below: zero them once
    for (int i = tid; i < RB_NB * RT; i += 256) { wy_s[i / RT][rb_wrow(i % RT) + 7] = 0.f; wx_s[i / RT][rb_wrow(i % RT) + 7] = 0.f; }
    for (int i = tid; i < RB_NB * RB_CC * 7; i += 256) G_s[i / (RB_CC * 7)][(i / 7) % RB_CC][(i % 7) * 8 + 7] = 0.f;

    const int per_seg = (lend - lbeg + nseg - 1) / nseg;
    const int beg = lbeg + seg * per_seg;
    const int end = beg + per_seg < lend ? beg + per_seg : lend;
    // phase A/B roles: thread t < 112 -> weight of (tile row t/7, bin t%7); 128 <= t < 240 -> (tile column, bin);
    // every thread copies up to RB_GPT floats of each RoI's 16x49 gradient block
    const bool a_role = tid < RT * 7 || (tid >= 128 && tid < 128 + RT * 7);
    const int a_axis = tid >> 7, a_u = tid & 127, a_r = a_u / 7, a_pb = a_u - a_r * 7;
    // float gradients: one element per load; half gradients: PAIRS as 32-bit words (two-byte loads write half a register
    // and serialise: 562 us against 366 us for the float kernel; as words 2 x fewer loads than the float kernel issues)
    constexpr bool GH = sizeof(GT) == 2;
    constexpr int RB_GPT = GH ? (RB_CC * 49 / 2 + 255) / 256 : (RB_CC * 49 + 255) / 256;      // 4 | 7
    // register prefetch of one batch: the dependent global reads (list -> header -> table, gradients) of batch n+1
    // are in flight while batch n is contracted
    float pw[RB_NB];
    typename std::conditional<GH, unsigned, float>::type pg[RB_NB][RB_GPT];
    // software pipeline of the dependent reads: RoI ids two batches ahead, header -> table / gradients one batch ahead
    int ids[RB_NB], metas[RB_NB];
    auto fetch_ids = [&](int e0) {
#pragma unroll
        for (int j = 0; j < RB_NB; ++j) {
            ids[j] = e0 + j < end ? list[e0 + j] : 0;
            metas[j] = e0 + j < end ? lmeta[e0 + j] : 0;
        }
    };
    auto prefetch = [&](int e0) {
        const int nb = end - e0 < RB_NB ? end - e0 : RB_NB;
#pragma unroll
        for (int j = 0; j < RB_NB; ++j) {
            pw[j] = 0.f;
#pragma unroll
            for (int i = 0; i < RB_GPT; ++i) pg[j][i] = 0;
            if (j >= nb) continue;
            const int id = ids[j];
            // set index without a division (RB_MAXSETS = 4; an integer division is ~20 vector instructions, per RoI and lane)
            const int s = (id >= S.R ? 1 : 0) + (id >= 2 * S.R ? 1 : 0) + (id >= 3 * S.R ? 1 : 0), k = id - s * S.R;
            if (a_role) {
                const int big = metas[j] >> 28, wy0 = (metas[j] >> 14) & 0x3fff, wx0 = metas[j] & 0x3fff;
                float w;
                if (big == 0) {          // table form (axis 0 already carries 1/samples)
                    const int off = (a_axis ? tx * RT - wx0 : ty * RT - wy0) + a_r;
                    w = (off >= 0 && off < RW_EXT) ? wtab[((int64_t)id * 2 + a_axis) * 7 * RW_EXT + a_pb * RW_EXT + off] : 0.f;
                } else {
                    const RoiGeom g = roi_geom(S.rois[s] + 5 * (int64_t)k, scale, P, sample_num);
                    w = a_axis ? axis_weight(g.x_start, g.bin_w, g.grid_w, a_pb, W, tx * RT + a_r)
                               : axis_weight(g.y_start, g.bin_h, g.grid_h, a_pb, H, ty * RT + a_r) *
                                     (1.0f / (float)(g.grid_h * g.grid_w > 1 ? g.grid_h * g.grid_w : 1));
                }
                pw[j] = w;
            }
            const GT* gp = gout + (int64_t)k * g_roi_stride + ((int64_t)s * L.C + c0) * 49;
            if constexpr (GH) {
                const int n = ncc * 49;
                const bool aligned = (reinterpret_cast<uintptr_t>(gp) & 3) == 0;      // block-uniform
#pragma unroll
                for (int i = 0; i < RB_GPT; ++i) {
                    const int q = 2 * (tid + 256 * i);
                    if (aligned && q + 1 < n) pg[j][i] = *reinterpret_cast<const unsigned*>(gp + q);
                    else {
                        const unsigned lo = q < n ? (unsigned)reinterpret_cast<const unsigned short*>(gp)[q] : 0u;
                        const unsigned hi = q + 1 < n ? (unsigned)reinterpret_cast<const unsigned short*>(gp)[q + 1] : 0u;
                        pg[j][i] = lo | (hi << 16);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < RB_GPT; ++i) {
                    const int q = tid + 256 * i;
#if defined(ARFE_PROBES) && defined(ARFE_RB_PROBE) && (ARFE_RB_PROBE & 2)
                    if (q < ncc * 49) pg[j][i] = (float)(q + id);
#else
                    if (q < ncc * 49) pg[j][i] = (float)gp[q];
#endif
                }
            }
        }
    };
    if (beg < end) { fetch_ids(beg); prefetch(beg); fetch_ids(beg + RB_NB); }
    for (int e0 = beg; e0 < end; e0 += RB_NB) {
        const int nb = end - e0 < RB_NB ? end - e0 : RB_NB;
        __syncthreads();                       // previous batch's LDS reads are done
        // ---- A/B: registers -> LDS ----
#pragma unroll
        for (int j = 0; j < RB_NB; ++j) {
            if (a_role) { if (a_axis) wx_s[j][rb_wrow(a_r) + a_pb] = pw[j]; else wy_s[j][rb_wrow(a_r) + a_pb] = pw[j]; }
#pragma unroll
            for (int i = 0; i < RB_GPT; ++i) {
                if constexpr (GH) {
                    const int q = 2 * (tid + 256 * i);
                    if (q < RB_CC * 49) {          // RB_CC * 49 is even: both halves of the word are inside the chunk
                        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                        const h2_t h = __builtin_bit_cast(h2_t, pg[j][i]);
                        const int c = q / 49, r2 = q - c * 49, ph = r2 / 7;
                        G_s[j][c][ph * 8 + r2 - ph * 7] = (float)h[0];
                        const int q1 = q + 1, c1 = q1 / 49, r21 = q1 - c1 * 49, ph1 = r21 / 7;
                        G_s[j][c1][ph1 * 8 + r21 - ph1 * 7] = (float)h[1];
                    }
                } else {
                    const int q = tid + 256 * i;
                    if (q < RB_CC * 49) {
                        const int c = q / 49, r2 = q - c * 49, ph = r2 / 7;
                        G_s[j][c][ph * 8 + r2 - ph * 7] = pg[j][i];
                    }
                }
            }
        }
        __syncthreads();
        if (e0 + RB_NB < end) { prefetch(e0 + RB_NB); fetch_ids(e0 + 2 * RB_NB); }
        // ---- C then D per RoI (per wave, its own 4 channels: no workgroup barrier, T_s rows are private to the wave) ----
        for (int j = 0; j < nb; ++j) {
#pragma unroll
            for (int pr = 0; pr < RB_CPW / 2; ++pr) {
                const int cl = wv * RB_CPW + pr * 2 + (l15 >> 3), ph = l15 & 7;      // A row l15 -> (channel, bin)
                f32x4_t t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const float a = ph < 7 ? G_s[j][cl][(ph < 7 ? ph : 6) * 8 + 4 * kk + lq] : 0.f;
                    const float b = wx_s[j][rb_wrow(l15) + 4 * kk + lq];
                    t = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, t, 0, 0, 0);
                }
                // rows 4*lq + r of the M tile = (channel 4v + 2pr + (lq >> 1), bin 4*(lq & 1) + r), column l15
                *reinterpret_cast<f32x4_t*>(&T_s[wv * RB_CPW + pr * 2 + (lq >> 1)][l15][(4 * (lq & 1)) ^ ((l15 & 8) >> 1)]) = t;
            }
            const float a0 = wy_s[j][rb_wrow(l15) + lq], a1 = wy_s[j][rb_wrow(l15) + 4 + lq];
            __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): this wave's T rows have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < RB_CPW; ++c) {
                const float b0 = T_s[wv * RB_CPW + c][l15][lq ^ ((l15 & 8) >> 1)], b1 = T_s[wv * RB_CPW + c][l15][(4 + lq) ^ ((l15 & 8) >> 1)];
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);     // T rows are read before the next RoI overwrites them
            __builtin_amdgcn_wave_barrier();
        }
    }
    {
        DT* d = (DT*)L.dfeat[0];
#pragma unroll
        for (int i = 1; i < ARFE_MAX_LEVELS; ++i) if (i == lvl) d = (DT*)L.dfeat[i];
        const int ox = tx * RT + l15;
#pragma unroll
        for (int c = 0; c < RB_CPW; ++c) {
            const int cc = wv * RB_CPW + c;
            if (cc >= ncc || ox >= W) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = ty * RT + 4 * lq + r;
                if (oy >= H) continue;
                DT* dst = d + (((int64_t)b * L.C + c0 + cc) * H + oy) * W + ox;
                if constexpr (sizeof(DT) == 4) {
                    if (nseg == 1) *dst = acc[c][r];
                    else unsafeAtomicAdd(dst, acc[c][r]);
                } else {
#if defined(ARFE_PROBES) && defined(ARFE_RB_PROBE) && (ARFE_RB_PROBE & 1)
                    if (acc[c][r] == 123.456f)
#endif
                    *dst = (DT)acc[c][r];          // (half maps: the host never slices RoI lists, every tile is stored once)
                }
            }
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
// grid = (chunks of a RoI, RoIs): no division per element; VEC = 4 when every row is 16-byte aligned
template <int VEC>
__global__ __launch_bounds__(256) void rff_fuse_bwd_kernel(int per_roi, const float* __restrict__ ori, int64_t ori_roi_stride,
                                                           const float* __restrict__ lw, const float* __restrict__ lh,
                                                           const float* __restrict__ dz, float* __restrict__ d_ori,
                                                           int64_t d_ori_roi_stride, float* __restrict__ dlw,
                                                           float* __restrict__ dlh) {
    const int q = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (q >= per_roi) return;
    const int64_t r = blockIdx.y, i = r * per_roi + q;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const vec_t o = *reinterpret_cast<const vec_t*>(ori + r * ori_roi_stride + q);
    const vec_t g = *reinterpret_cast<const vec_t*>(dz + i);
    const vec_t a = *reinterpret_cast<const vec_t*>(lw + i), b = *reinterpret_cast<const vec_t*>(lh + i);
    vec_t dxo, da, db;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float ge = g[e], oe = o[e], ae = a[e], be = b[e];
        const float go = ge * oe;
        const float x = ge + ge * (ae + be);
        const float u = ae > 0.f ? go : 0.f;       // ReLU of wh_conv / hh_conv folded in (lw, lh are post-ReLU)
        const float w = be > 0.f ? go : 0.f;
        dxo[e] = x; da[e] = u; db[e] = w;
    }
    *reinterpret_cast<vec_t*>(d_ori + r * d_ori_roi_stride + q) = dxo;
    *reinterpret_cast<vec_t*>(dlw + i) = da;
    *reinterpret_cast<vec_t*>(dlh + i) = db;
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

extern "C" int32_t arfe_rff_roi_sets(int32_t R, const float* rois, int32_t num_levels, float finest_scale, float* h_rois,
                                     float* w_rois, int64_t* levels_ori, int64_t* levels_w, int64_t* levels_h, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && num_levels >= 1 && num_levels <= 5 && finest_scale > 0.f, ARFE_ERR_ARG,
                   "rff_roi_sets: bad arguments (num_levels=%d)", num_levels);
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(rois && h_rois && w_rois && levels_ori && levels_w && levels_h, ARFE_ERR_ARG, "rff_roi_sets: null pointer");
    hipLaunchKernelGGL(rff_roi_sets_kernel, dim3(cdiv(R, 256)), dim3(256), 0, as_stream(stream), R, rois, num_levels,
                       finest_scale, h_rois, w_rois, levels_ori, levels_w, levels_h);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int64_t arfe_roialign_fwd_workspace_bytes(int32_t nsets, int32_t R) {
    if (nsets < 1 || nsets > RB_MAXSETS || R < 0) return -1;
    return (int64_t)nsets * R * ((int64_t)sizeof(RoiWeightHdr) + (int64_t)2 * 7 * RW_EXT * sizeof(float) + 8) + 64;
}

namespace {
struct RoiFwdPlan {
    RoiLevels L; RoiSets S;
    int nroi, nslices;
    RoiWeightHdr* hdr; float* wtab; uint32_t* keys; int32_t* order;
};
int32_t roialign_fwd_plan(RoiFwdPlan& p, const arfe_pyramid_t* pyr, const void* const* feats_any, const int32_t* strides,
                          int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                          int32_t sample_num, void* workspace) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    ARFE_CHECK_ARG(R >= 0 && nsets >= 1 && nsets <= RB_MAXSETS && sample_num >= 0, ARFE_ERR_ARG, "roialign_fwd: bad arguments");
    ARFE_CHECK_ARG(pooled == 7, ARFE_ERR_ARG, "roialign_fwd: only 7x7 pooling has a kernel (got %d)", pooled);
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(strides && rois && levels && workspace, ARFE_ERR_ARG, "roialign_fwd: null pointer");
    ARFE_CHECK_ARG(fill_roi_levels(p.L, pyr, (const float* const*)feats_any, nullptr, strides) == ARFE_OK, ARFE_ERR_ARG,
                   "roialign_fwd: bad level table");
    p.S.nsets = nsets; p.S.R = R;
    for (int i = 0; i < RB_MAXSETS; ++i) { p.S.rois[i] = nullptr; p.S.levels[i] = nullptr; }
    for (int i = 0; i < nsets; ++i) {
        ARFE_CHECK_ARG(rois[i] && levels[i], ARFE_ERR_ARG, "roialign_fwd: null RoI set %d", i);
        p.S.rois[i] = rois[i]; p.S.levels[i] = levels[i];
    }
    p.nroi = nsets * R;
    p.hdr = (RoiWeightHdr*)workspace;
    p.wtab = (float*)(p.hdr + p.nroi);
    p.keys = (uint32_t*)(p.wtab + (int64_t)p.nroi * 2 * 7 * RW_EXT);
    p.order = (int32_t*)(p.keys + p.nroi);
    p.nslices = cdiv(p.L.C, RF_CH);
    ARFE_CHECK_ARG((int64_t)p.nroi * p.nslices < (int64_t)1 << 31, ARFE_ERR_ARG, "roialign_fwd: too many RoIs");
    return ARFE_OK;
}
// the RoI-only half: processing order (group-major, spatial inside a group) and the per-RoI weight tables
int32_t roialign_fwd_prep(const RoiFwdPlan& p, int pooled, int sample_num, hipStream_t st) {
    hipLaunchKernelGGL(roi_key_kernel, dim3(cdiv(p.nroi, 256)), dim3(256), 0, st, p.L, p.S, p.nroi, pooled, sample_num, p.keys);
    ARFE_CHECK_LAUNCH();
    hipLaunchKernelGGL(roi_rank_kernel, dim3(cdiv(p.nroi, 4)), dim3(256), 0, st, p.nroi, (const uint32_t*)p.keys, p.order, p.S.R,
                       p.S.nsets);
    ARFE_CHECK_LAUNCH();
    hipLaunchKernelGGL(roi_weight_table_kernel, dim3(p.nroi), dim3(256), 0, st, p.L, p.S, pooled, sample_num, p.hdr, p.wtab,
                       (const int32_t*)p.order, 0);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
// the extraction of processing groups [g0, g0 + gcount)
int32_t roialign_fwd_run(const RoiFwdPlan& p, int pooled, int sample_num, void* out, int64_t out_roi_stride, int g0, int gcount,
                         bool half_io, hipStream_t st) {
    const int slot0 = g0 * p.S.R, nslot = gcount * p.S.R;
    if (nslot <= 0) return ARFE_OK;
    if (half_io)
        hipLaunchKernelGGL((roialign_fwd_sep_kernel<_Float16, _Float16>), dim3(nslot * p.nslices), dim3(256), 0, st, p.L, p.S,
                           pooled, sample_num, (const RoiWeightHdr*)p.hdr, (const float*)p.wtab, p.nslices, (_Float16*)out,
                           out_roi_stride, slot0);
    else
        hipLaunchKernelGGL((roialign_fwd_sep_kernel<float, float>), dim3(nslot * p.nslices), dim3(256), 0, st, p.L, p.S, pooled,
                           sample_num, (const RoiWeightHdr*)p.hdr, (const float*)p.wtab, p.nslices, (float*)out, out_roi_stride,
                           slot0);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
}  // namespace

static int32_t roialign_fwd_any(const arfe_pyramid_t* pyr, const void* const* feats_any, const int32_t* strides,
                                int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                int32_t pooled, int32_t sample_num, void* out, int64_t out_roi_stride,
                                void* workspace, void* stream, bool half_io) {
    RoiFwdPlan p;
    int32_t rc = roialign_fwd_plan(p, pyr, feats_any, strides, nsets, R, rois, levels, pooled, sample_num, workspace);
    if (rc || R == 0) return rc;
    ARFE_CHECK_ARG(feats_any && out, ARFE_ERR_ARG, "roialign_fwd: null pointer");
    rc = roialign_fwd_prep(p, pooled, sample_num, as_stream(stream));
    if (rc) return rc;
    return roialign_fwd_run(p, pooled, sample_num, out, out_roi_stride, 0, nsets, half_io, as_stream(stream));
}

// The forward in two parts (round 3).  _prepare is the RoI-only part (processing order, weight tables): it needs no feature
// map.  _group extracts the RoIs of processing groups [group_first, group_first + group_count) - group g is RoI set
// (g + 1) % nsets, see roi_rank_kernel - on a prepared workspace: a caller launches the sets one after the other and starts the
// consumer of a set's channel slice (the AR-RFF gating convs) while the next set is extracted.  arfe_roialign_fwd = both.
extern "C" int32_t arfe_roialign_fwd_prepare(const arfe_pyramid_t* pyr, const int32_t* strides, int32_t nsets, int32_t R,
                                             const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                             int32_t sample_num, void* workspace, void* stream) {
    RoiFwdPlan p;
    int32_t rc = roialign_fwd_plan(p, pyr, nullptr, strides, nsets, R, rois, levels, pooled, sample_num, workspace);
    if (rc || R == 0) return rc;
    return roialign_fwd_prep(p, pooled, sample_num, as_stream(stream));
}
extern "C" int32_t arfe_roialign_fwd_group(const arfe_pyramid_t* pyr, const float* const* feats, const int32_t* strides,
                                           int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                           int32_t pooled, int32_t sample_num, float* out, int64_t out_roi_stride,
                                           const void* workspace, int32_t group_first, int32_t group_count, void* stream) {
    RoiFwdPlan p;
    int32_t rc = roialign_fwd_plan(p, pyr, (const void* const*)feats, strides, nsets, R, rois, levels, pooled, sample_num,
                                   const_cast<void*>(workspace));
    if (rc || R == 0) return rc;
    ARFE_CHECK_ARG(feats && out, ARFE_ERR_ARG, "roialign_fwd_group: null pointer");
    ARFE_CHECK_ARG(group_first >= 0 && group_count >= 0 && group_first + group_count <= nsets, ARFE_ERR_ARG,
                   "roialign_fwd_group: groups [%d, %d) outside [0, %d)", group_first, group_first + group_count, nsets);
    return roialign_fwd_run(p, pooled, sample_num, out, out_roi_stride, group_first, group_count, false, as_stream(stream));
}

extern "C" int32_t arfe_roialign_fwd(const arfe_pyramid_t* pyr, const float* const* feats, const int32_t* strides,
                                     int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                     int32_t pooled, int32_t sample_num, float* out, int64_t out_roi_stride,
                                     void* workspace, void* stream) {
    return roialign_fwd_any(pyr, (const void* const*)feats, strides, nsets, R, rois, levels, pooled, sample_num, out,
                            out_roi_stride, workspace, stream, false);
}
// half feature maps in, half RoI features out (the extractor of a model in fp16 mode); out_roi_stride in ELEMENTS
extern "C" int32_t arfe_roialign_fwd_h(const arfe_pyramid_t* pyr, const void* const* feats, const int32_t* strides,
                                       int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                       int32_t pooled, int32_t sample_num, void* out, int64_t out_roi_stride,
                                       void* workspace, void* stream) {
    return roialign_fwd_any(pyr, feats, strides, nsets, R, rois, levels, pooled, sample_num, out, out_roi_stride, workspace,
                            stream, true);
}

static RoiTileTable make_roi_tile_table(const arfe_pyramid_t* pyr, int* ntiles, int* max_tiles_per_img) {
    RoiTileTable T;
    int acc = 0, mx = 1;
    for (int i = 0; i < ARFE_MAX_LEVELS; ++i) {
        T.first[i] = acc;
        if (i < pyr->num_levels) {
            T.tiles_x[i] = cdiv(pyr->width[i], RT);
            T.tiles_y[i] = cdiv(pyr->height[i], RT);
            acc += T.tiles_x[i] * T.tiles_y[i] * pyr->batch;
            if (T.tiles_x[i] * T.tiles_y[i] > mx) mx = T.tiles_x[i] * T.tiles_y[i];
        } else {
            T.tiles_x[i] = T.tiles_y[i] = 0;
        }
    }
    T.first[ARFE_MAX_LEVELS] = acc;
    *ntiles = acc;
    *max_tiles_per_img = mx;
    return T;
}

extern "C" int64_t arfe_roialign_bwd_workspace_bytes(const arfe_pyramid_t* pyr, int32_t nsets, int32_t R) {
    if (check_pyramid(pyr) || nsets < 1 || nsets > RB_MAXSETS || R < 0) return -1;
    int ntiles, mx;
    make_roi_tile_table(pyr, &ntiles, &mx);
    // counts[ntiles] | offsets[ntiles+1] | cursors[ntiles] | list, lmeta[nsets*R*mx] | hdr[nsets*R] | wtab[nsets*R][2][7][RW_EXT]
    return ((int64_t)3 * ntiles + 1 + (int64_t)2 * nsets * R * mx + 16) * (int64_t)sizeof(int) +
           (int64_t)nsets * R * ((int64_t)sizeof(RoiWeightHdr) + (int64_t)2 * 7 * RW_EXT * sizeof(float)) + 64;
}

namespace {
struct RoiBwdPlan {
    RoiTileTable T;
    int ntiles, mx, nroi;
    int *counts, *offsets, *cursors, *list, *lmeta;
    RoiWeightHdr* hdr;
    float* wtab;
};
RoiBwdPlan plan_roi_bwd(const arfe_pyramid_t* pyr, int nsets, int R, void* workspace) {
    RoiBwdPlan p;
    p.T = make_roi_tile_table(pyr, &p.ntiles, &p.mx);
    p.nroi = nsets * R;
    p.counts = (int*)workspace;
    p.offsets = p.counts + p.ntiles;
    p.cursors = p.offsets + p.ntiles + 1;
    p.list = p.cursors + p.ntiles;
    p.lmeta = p.list + (size_t)p.nroi * p.mx;
    uintptr_t after = (uintptr_t)(p.lmeta + (size_t)p.nroi * p.mx);
    after = (after + 15) & ~(uintptr_t)15;
    p.hdr = (RoiWeightHdr*)after;
    p.wtab = (float*)(p.hdr + p.nroi);
    return p;
}
int32_t fill_roi_sets(RoiSets& S, int nsets, int R, const float* const* rois, const int64_t* const* levels, const char* who) {
    S.nsets = nsets; S.R = R;
    for (int i = 0; i < RB_MAXSETS; ++i) { S.rois[i] = nullptr; S.levels[i] = nullptr; }
    for (int i = 0; i < nsets && R > 0; ++i) {
        ARFE_CHECK_ARG(rois[i] && levels[i], ARFE_ERR_ARG, "%s: null RoI set %d", who, i);
        S.rois[i] = rois[i]; S.levels[i] = levels[i];
    }
    return ARFE_OK;
}
}  // namespace

// Everything of the backward that depends on the RoIs alone (tile lists, per-RoI weight tables): a caller can run it
// beside the forward RoIAlign, on another stream, and hand the filled workspace to arfe_roialign_bwd_prepared later.
extern "C" int32_t arfe_roialign_bwd_prepare(const arfe_pyramid_t* pyr, const int32_t* strides, int32_t nsets, int32_t R,
                                             const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                             int32_t sample_num, void* workspace, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    ARFE_CHECK_ARG(R >= 0 && nsets >= 1 && nsets <= RB_MAXSETS && sample_num >= 0, ARFE_ERR_ARG, "roialign_bwd: bad arguments");
    ARFE_CHECK_ARG(pooled == 7, ARFE_ERR_ARG, "roialign_bwd: only 7x7 pooling has a kernel (got %d)", pooled);
    ARFE_CHECK_ARG(strides && workspace && (R == 0 || (rois && levels)), ARFE_ERR_ARG, "roialign_bwd: null pointer");
    RoiLevels L;
    ARFE_CHECK_ARG(fill_roi_levels(L, pyr, nullptr, nullptr, strides) == ARFE_OK, ARFE_ERR_ARG, "roialign_bwd: bad level table");
    RoiSets S;
    rc = fill_roi_sets(S, nsets, R, rois, levels, "roialign_bwd");
    if (rc) return rc;
    RoiBwdPlan p = plan_roi_bwd(pyr, nsets, R, workspace);
    hipStream_t st = as_stream(stream);
    {   // the tile counters
        ZeroList z;
        z.add(p.counts, p.ntiles);
        rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    if (p.nroi > 0) {
        hipLaunchKernelGGL((roi_bin_kernel<false>), dim3(cdiv(p.nroi, 256)), dim3(256), 0, st, L, S, p.T, pooled, p.counts,
                           (const int*)p.offsets, p.cursors, p.list, (const RoiWeightHdr*)nullptr, (int*)nullptr);
        ARFE_CHECK_LAUNCH();
        hipLaunchKernelGGL(roi_weight_table_kernel, dim3(p.nroi), dim3(256), 0, st, L, S, pooled, sample_num, p.hdr, p.wtab,
                           (const int32_t*)nullptr, 1);
        ARFE_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(roi_bin_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)p.counts, p.offsets, p.cursors, p.ntiles);
    ARFE_CHECK_LAUNCH();
    if (p.nroi > 0) {
        hipLaunchKernelGGL((roi_bin_kernel<true>), dim3(cdiv(p.nroi, 256)), dim3(256), 0, st, L, S, p.T, pooled, p.counts,
                           (const int*)p.offsets, p.cursors, p.list, (const RoiWeightHdr*)p.hdr, p.lmeta);
        ARFE_CHECK_LAUNCH();
        hipLaunchKernelGGL(roi_list_sort_kernel, dim3(p.ntiles), dim3(64), 0, st, (const int*)p.offsets, p.list, p.lmeta);
        ARFE_CHECK_LAUNCH();
    }
    return ARFE_OK;
}

namespace {
// shared argument checks / tables of the two halves of the prepared backward
int32_t roi_bwd_args(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides, int nsets, int R,
                     const float* const* rois, const int64_t* const* levels, int pooled, int sample_num, const void* workspace,
                     RoiLevels& L, RoiSets& S) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    ARFE_CHECK_ARG(R >= 0 && nsets >= 1 && nsets <= RB_MAXSETS && sample_num >= 0, ARFE_ERR_ARG, "roialign_bwd: bad arguments");
    ARFE_CHECK_ARG(pooled == 7, ARFE_ERR_ARG, "roialign_bwd: only 7x7 pooling has a kernel (got %d)", pooled);
    ARFE_CHECK_ARG(dfeats && strides && workspace && (R == 0 || (rois && levels)), ARFE_ERR_ARG, "roialign_bwd: null pointer");
    ARFE_CHECK_ARG(fill_roi_levels(L, pyr, nullptr, dfeats, strides) == ARFE_OK, ARFE_ERR_ARG, "roialign_bwd: bad level table");
    return fill_roi_sets(S, nsets, R, rois, levels, "roialign_bwd");
}
}  // namespace

// First half of the prepared backward: zero the tiles of dfeats that the second half sums with atomics (those whose RoI list
// is cut into slices).  Needs only the prepared workspace, so a caller that allocates dfeats early runs it off the critical
// path; arfe_roialign_bwd_prepared = _prezero + _apply.
extern "C" int32_t arfe_roialign_bwd_prezero(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides,
                                             int32_t nsets, int32_t R, const void* workspace, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    ARFE_CHECK_ARG(R >= 0 && nsets >= 1 && nsets <= RB_MAXSETS, ARFE_ERR_ARG, "roialign_bwd: bad arguments");
    ARFE_CHECK_ARG(dfeats && strides && workspace, ARFE_ERR_ARG, "roialign_bwd: null pointer");
    RoiLevels L;
    ARFE_CHECK_ARG(fill_roi_levels(L, pyr, nullptr, dfeats, strides) == ARFE_OK, ARFE_ERR_ARG, "roialign_bwd: bad level table");
    if (rb_seg().segs <= 1) return ARFE_OK;          // no list is sliced: every tile is stored exactly once, nothing to zero
    RoiBwdPlan p = plan_roi_bwd(pyr, nsets, R, const_cast<void*>(workspace));
    const int nchunks = cdiv(L.C, RB_CC);
    ARFE_CHECK_ARG((int64_t)p.ntiles * nchunks < (int64_t)1 << 31, ARFE_ERR_ARG, "roialign_bwd: pyramid too large");
    hipLaunchKernelGGL(roi_zero_sliced_tiles_kernel, dim3(p.ntiles * nchunks), dim3(256), 0, as_stream(stream), L, p.T,
                       (const int*)p.offsets, p.ntiles, nchunks, rb_seg().seg_min);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

static int32_t roialign_bwd_apply_any(const arfe_pyramid_t* pyr, void* const* dfeats_any, const int32_t* strides,
                                      int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                      int32_t pooled, int32_t sample_num, const void* gout, int64_t gout_roi_stride,
                                      const void* workspace, void* stream, bool half_io) {
    RoiLevels L;
    RoiSets S;
    int32_t rc = roi_bwd_args(pyr, (float* const*)dfeats_any, strides, nsets, R, rois, levels, pooled, sample_num, workspace, L, S);
    if (rc) return rc;
    ARFE_CHECK_ARG(R == 0 || gout, ARFE_ERR_ARG, "roialign_bwd: null pointer");
    RoiBwdPlan p = plan_roi_bwd(pyr, nsets, R, const_cast<void*>(workspace));
    const int nchunks = cdiv(L.C, RB_CC);
    ARFE_CHECK_ARG((int64_t)p.ntiles * nchunks < (int64_t)1 << 31, ARFE_ERR_ARG, "roialign_bwd: pyramid too large");
    if (half_io)      // one workgroup per tile and chunk, plain stores (no sliced lists: there is no half atomic path)
        hipLaunchKernelGGL((roialign_bwd_tiles_kernel<_Float16, _Float16>), dim3(p.ntiles * nchunks, 1, 1), dim3(256), 0,
                           as_stream(stream), L, S, p.T, pooled, sample_num, (const _Float16*)gout, gout_roi_stride,
                           (const int*)p.offsets, (const int*)p.list, (const int*)p.lmeta, (const float*)p.wtab, p.ntiles, nchunks,
                           1, 1 << 30);
    else
        hipLaunchKernelGGL((roialign_bwd_tiles_kernel<float, float>), dim3(p.ntiles * nchunks, 1, rb_seg().segs), dim3(256), 0,
                           as_stream(stream), L, S, p.T, pooled, sample_num, (const float*)gout, gout_roi_stride,
                           (const int*)p.offsets, (const int*)p.list, (const int*)p.lmeta, (const float*)p.wtab, p.ntiles, nchunks,
                           rb_seg().segs, rb_seg().seg_min);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_roialign_bwd_apply(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides,
                                           int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                           int32_t pooled, int32_t sample_num, const float* gout, int64_t gout_roi_stride,
                                           const void* workspace, void* stream) {
    return roialign_bwd_apply_any(pyr, (void* const*)dfeats, strides, nsets, R, rois, levels, pooled, sample_num, gout,
                                  gout_roi_stride, workspace, stream, false);
}
// half gradients in, half feature-map gradients out, on a workspace prepared by arfe_roialign_bwd_prepare (every element of
// dfeats is written; gout_roi_stride in ELEMENTS)
extern "C" int32_t arfe_roialign_bwd_prepared_h(const arfe_pyramid_t* pyr, void* const* dfeats, const int32_t* strides,
                                                int32_t nsets, int32_t R, const float* const* rois,
                                                const int64_t* const* levels, int32_t pooled, int32_t sample_num,
                                                const void* gout, int64_t gout_roi_stride, const void* workspace,
                                                void* stream) {
    return roialign_bwd_apply_any(pyr, dfeats, strides, nsets, R, rois, levels, pooled, sample_num, gout, gout_roi_stride,
                                  workspace, stream, true);
}

extern "C" int32_t arfe_roialign_bwd_prepared(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides,
                                              int32_t nsets, int32_t R, const float* const* rois,
                                              const int64_t* const* levels, int32_t pooled, int32_t sample_num,
                                              const float* gout, int64_t gout_roi_stride, const void* workspace,
                                              void* stream) {
    ARFE_CHECK_ARG(R == 0 || gout, ARFE_ERR_ARG, "roialign_bwd: null pointer");
    int32_t rc = arfe_roialign_bwd_prezero(pyr, dfeats, strides, nsets, R, workspace, stream);
    if (rc) return rc;
    return arfe_roialign_bwd_apply(pyr, dfeats, strides, nsets, R, rois, levels, pooled, sample_num, gout, gout_roi_stride,
                                   workspace, stream);
}

extern "C" int32_t arfe_roialign_bwd(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides,
                                     int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                                     int32_t pooled, int32_t sample_num, const float* gout, int64_t gout_roi_stride,
                                     void* workspace, void* stream) {
    int32_t rc = arfe_roialign_bwd_prepare(pyr, strides, nsets, R, rois, levels, pooled, sample_num, workspace, stream);
    if (rc) return rc;
    return arfe_roialign_bwd_prepared(pyr, dfeats, strides, nsets, R, rois, levels, pooled, sample_num, gout, gout_roi_stride,
                                      workspace, stream);
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
                                     const float* lh, const float* dz, float* d_ori, int64_t d_ori_roi_stride, float* dlw,
                                     float* dlh, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0, ARFE_ERR_ARG, "rff_fuse_bwd: bad shape");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(ori && lw && lh && dz && d_ori && dlw && dlh, ARFE_ERR_ARG, "rff_fuse_bwd: null pointer");
    const int per_roi = C * 49;
    const bool vec4 = per_roi % 4 == 0 && ori_roi_stride % 4 == 0 && d_ori_roi_stride % 4 == 0 &&
                      (((uintptr_t)ori | (uintptr_t)lw | (uintptr_t)lh | (uintptr_t)dz | (uintptr_t)d_ori | (uintptr_t)dlw |
                        (uintptr_t)dlh) & 15) == 0;
    if (vec4)
        hipLaunchKernelGGL((rff_fuse_bwd_kernel<4>), dim3(cdiv(per_roi / 4, 256), R), dim3(256), 0, as_stream(stream), per_roi,
                           ori, ori_roi_stride, lw, lh, dz, d_ori, d_ori_roi_stride, dlw, dlh);
    else
        hipLaunchKernelGGL((rff_fuse_bwd_kernel<1>), dim3(cdiv(per_roi, 256), R), dim3(256), 0, as_stream(stream), per_roi,
                           ori, ori_roi_stride, lw, lh, dz, d_ori, d_ori_roi_stride, dlw, dlh);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
