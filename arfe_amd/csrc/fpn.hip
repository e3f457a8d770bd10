// AR-FPN (WFPNDualSpatial) streaming stages on gfx950: gather, gate maps, scatter+gate+residual and their
// backward passes.  All of these are HBM-bound: one pyramid copy is 91.7 MB / image.
//   layout: every level NCHW fp32, contiguous; lanes run along W so that a wave touches whole 128-B lines.
#include "common.hpp"

namespace arfe {

constexpr int NL_ = ARFE_MAX_LEVELS;

struct LevelPtrs { const float* p[NL_]; };
struct LevelMutPtrs { float* p[NL_]; };

// activation storage (arfe_pyramid_t::storage): fp32 or IEEE half, converted on load
typedef _Float16 half4_v __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ float4_u ld4(const float* p) { return *reinterpret_cast<const float4_u*>(p); }
__device__ __forceinline__ float4_u ld4(const _Float16* p) {
    const half4_v h = *reinterpret_cast<const half4_v*>(p);
    return float4_u{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// =========================================================================================================
// gather:  m[b,c,py,px] = (sum_i G_i) / n ; one thread per refine-level element, lanes along px
// =========================================================================================================
__global__ __launch_bounds__(256) void fpn_gather_fwd_kernel(Levels lv, LevelPtrs x, float* __restrict__ m,
                                                             int32_t* __restrict__ argmax) {
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const int64_t total = (int64_t)lv.B * lv.C * hr * wr;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % wr);
    const int py = (int)((idx / wr) % hr);
    const int64_t bc = idx / ((int64_t)wr * hr);
    float acc = 0.f;
    for (int i = 0; i < lv.n; ++i) {
        const int h = lv.H[i], w = lv.W[i];
        const float* plane = x.p[i] + bc * (int64_t)h * w;
        float g;
        if (i < lv.r) {
            const int ys = pool_start(py, h, hr), ye = pool_end(py, h, hr);
            const int xs = pool_start(px, w, wr), xe = pool_end(px, w, wr);
            float mx = -INFINITY;
            int am = ys * w + xs;
            if (xe - xs == 4) {
                for (int yy = ys; yy < ye; ++yy) {
                    float4_u v = *reinterpret_cast<const float4_u*>(plane + (int64_t)yy * w + xs);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float e = v[t];
                        if (e > mx || e != e) { mx = e; am = yy * w + xs + t; }
                    }
                }
            } else {
                for (int yy = ys; yy < ye; ++yy)
                    for (int xx = xs; xx < xe; ++xx) {
                        float e = plane[(int64_t)yy * w + xx];
                        if (e > mx || e != e) { mx = e; am = yy * w + xx; }
                    }
            }
            g = mx;
            if (argmax) argmax[(int64_t)i * total + idx] = am;
        } else {
            const int sy = nearest_src(py, h, nearest_scale(h, hr));
            const int sx = nearest_src(px, w, nearest_scale(w, wr));
            g = plane[(int64_t)sy * w + sx];
        }
        acc = acc + g;
    }
    m[idx] = acc / (float)lv.n;
}

// =========================================================================================================
// gate maps: partial 3x3 conv sums over a channel group, LDS-tiled.
//   tile = 16 x 64 output pixels, 256 threads x 4 px, channel chunk of 8 staged through LDS with its halo
//   (interior columns at float index 4..67 so the interior read is one aligned ds_read_b128).
//   grid.x = spatial tiles of all levels, grid.y = channel group; partial pre-activations go to
//   pre[group][2][pixel] and are finished (bias, ReLU, tanh) by gate_finish_kernel.
// =========================================================================================================
// =========================================================================================================
// Gate maps  a = tanh(relu(conv3x3(x; w1) + b1)),  b = ... (w2, b2)   (wfpn_dual_spatial.py: reduce_convs/2)
// A 3x3 conv with ONE output channel is, per tap, a [pixels x C] . [C] product: all 18 (gate, tap) products of a pixel
// are one row of   T[m=(gate,tap)][px] = sum_c W[m][c] * x[c][px]   -- an exact-fp32 MFMA GEMM with M = 18 (of 32),
// N = pixels, K = C whose B operand (lane = pixel, 2 channels per instruction) is read straight from the NCHW map with
// coalesced 128-B rows: x crosses the chip exactly once and no lane does per-tap index arithmetic.  The 9-tap
// shift-add over T (18 one-channel maps) is the second, tiny kernel.
// =========================================================================================================
typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int GM_TILES = 4;                  // 32-pixel MFMA tiles per wave
constexpr int GM_WAVES = 2;                  // waves per workgroup (small workgroups: 700 of them balance over 256 CUs)
constexpr int GM_PX = GM_WAVES * GM_TILES * 32;   // pixels per workgroup
constexpr int GM_M = 18;                     // rows of T per image: 2 gates x 9 taps
constexpr int GM_KC = 256;                   // channels whose weights are staged at a time (32 KB of LDS)

struct TileTable {             // name kept: the per-level pixel-chunk table of the gate kernels
    int first_tile[NL_ + 1];   // prefix sum of workgroups per level
    int tiles_x[NL_], tiles_y[NL_];   // tiles_x = chunks per image, tiles_y unused
    int64_t px_off[NL_ + 1];   // prefix sum of B*h*w (offset of a level inside a flattened per-pixel buffer)
};

// Wp[level][channel][32]: row m = gate * 9 + tap of the A operand, zero for m >= 18 (so that staging it is a plain copy)
__global__ void fpn_gate_pack_kernel(Levels lv, LevelPtrs w1, LevelPtrs w2, float* __restrict__ Wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.n * lv.C * 32) return;
    const int lvl = i / (lv.C * 32), rem = i - lvl * lv.C * 32, c = rem >> 5, m = rem & 31;
    Wp[i] = m < 9 ? w1.p[lvl][c * 9 + m] : (m < GM_M ? w2.p[lvl][c * 9 + m - 9] : 0.f);
}

__global__ __launch_bounds__(64 * GM_WAVES) void fpn_gate_mfma_kernel(Levels lv, TileTable tt, LevelPtrs x,
                                                                       const float* __restrict__ Wp, float* __restrict__ T) {
    __shared__ __attribute__((aligned(16))) float Ws[GM_KC][32];            // [channel][m], m >= 18 zero
    int t = blockIdx.x, lvl = 0;
    while (lvl + 1 < lv.n && t >= tt.first_tile[lvl + 1]) ++lvl;
    t -= tt.first_tile[lvl];
    const int hw = lv.H[lvl] * lv.W[lvl];
    const int bi = t / tt.tiles_x[lvl];
    const int chunk = t - bi * tt.tiles_x[lvl];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const float* xb = x.p[lvl] + (int64_t)bi * lv.C * hw;
    f32x16_t acc[GM_TILES];
#pragma unroll
    for (int i = 0; i < GM_TILES; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // The wave's four MFMA tiles interleave: tile i holds pixels p4 + i of every lane's quad p4 = ... + 4 * l31, so one
    // 16-byte read per (channel pair, lane) feeds all four tiles (a dword read costs the address unit the same) and the
    // four results of a row m go back as one 16-byte store.
    const int p4 = chunk * GM_PX + wave * (GM_TILES * 32) + 4 * l31;
    const bool quad_in = p4 + 3 < hw;                         // whole quad inside the map
    const int pq = p4 < hw ? p4 : (hw - 1);                   // clamped start for the element-wise tail form
    for (int c0 = 0; c0 < lv.C; c0 += GM_KC) {
        if (c0) __syncthreads();
        {
            const int kc_ = lv.C - c0 < GM_KC ? lv.C - c0 : GM_KC;
            const float4_u* src = reinterpret_cast<const float4_u*>(Wp + ((int64_t)lvl * lv.C + c0) * 32);
            for (int i = tid; i < GM_KC * 8; i += 64 * GM_WAVES) {
                float4_u v = {0.f, 0.f, 0.f, 0.f};
                if (i < kc_ * 8) v = src[i];
                *reinterpret_cast<float4_u*>(&Ws[0][0] + 4 * i) = v;
            }
        }
        __syncthreads();
        if (chunk * GM_PX + wave * (GM_TILES * 32) >= hw) continue;          // wave-uniform: nothing of this wave is inside
        const int kc = lv.C - c0 < GM_KC ? lv.C - c0 : GM_KC;
        // three register sets of 4 channel pairs: the reads of groups g+1 and g+2 are out before the 16 MFMAs of group g
        constexpr int GK = 4;
        float a[3][GK];
        float4_u b[3][GK];
        auto issue = [&](int g, int set) {
#pragma unroll
            for (int j = 0; j < GK; ++j) {
                const int k = 2 * (g * GK + j) + hh;
                const int kw = k < GM_KC ? k : GM_KC - 1;
                const int c = c0 + k < lv.C ? c0 + k : lv.C - 1;             // odd / short tails: the weight row is 0
                a[set][j] = k < kc ? Ws[kw][l31] : 0.f;
                const float* src = xb + (int64_t)c * hw;
                if (quad_in) {
                    b[set][j] = *reinterpret_cast<const float4_u*>(src + p4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[set][j][e] = src[pq + e < hw ? pq + e : hw - 1];
                }
            }
        };
        auto mma = [&](int set) {
#pragma unroll
            for (int j = 0; j < GK; ++j)
#pragma unroll
                for (int i = 0; i < GM_TILES; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][j], b[set][j][i], acc[i], 0, 0, 0);
        };
        const int ngroups = ((kc + 1) / 2 + GK - 1) / GK;
        issue(0, 0);
        if (ngroups > 1) issue(1, 1);
        for (int g = 0; g < ngroups; g += 3) {
            if (g + 2 < ngroups) issue(g + 2, 2);
            mma(0);
            if (g + 1 < ngroups) {
                if (g + 3 < ngroups) issue(g + 3, 0);
                mma(1);
            }
            if (g + 2 < ngroups) {
                if (g + 4 < ngroups) issue(g + 4, 1);
                mma(2);
            }
        }
    }
    // D layout: lane -> column l31 (here: quad p4); register r -> row m = (r & 3) + 8 * (r >> 2) + 4 * hh
    float* Tb = T + tt.px_off[lvl] * GM_M + (int64_t)bi * GM_M * hw;
    if (p4 < hw) {
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m >= GM_M) continue;
            float* dst = Tb + (int64_t)m * hw + p4;
            if (quad_in) {
                float4_u v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                *reinterpret_cast<float4_u*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (p4 + e < hw) dst[e] = acc[e][r];
            }
        }
    }
}

// a[y][x] = tanh(relu(b1 + sum_tap T[tap][y+dy][x+dx])) with zero padding (tap = 3*(dy+1) + (dx+1)); same for b
__global__ void fpn_gate_finish_kernel(Levels lv, TileTable tt, const float* __restrict__ T, int64_t pre_total,
                                       LevelPtrs b1, LevelPtrs b2, LevelMutPtrs a, LevelMutPtrs b) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pre_total) return;
    int lvl = 0;
    while (lvl + 1 < lv.n && i >= tt.px_off[lvl + 1]) ++lvl;
    const int h = lv.H[lvl], w = lv.W[lvl], hw = h * w;
    const int64_t o = i - tt.px_off[lvl];
    const int bi = (int)(o / hw), p = (int)(o - (int64_t)bi * hw), y = p / w, xx = p - y * w;
    const float* Tb = T + tt.px_off[lvl] * GM_M + (int64_t)bi * GM_M * hw;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xs = xx + tap % 3 - 1;
        const bool ok = yy >= 0 && yy < h && xs >= 0 && xs < w;
        const int q = ok ? yy * w + xs : p;
        const float v1 = Tb[(int64_t)tap * hw + q], v2 = Tb[(int64_t)(9 + tap) * hw + q];
        s1 += ok ? v1 : 0.f;
        s2 += ok ? v2 : 0.f;
    }
    s1 += b1.p[lvl][0];
    s2 += b2.p[lvl][0];
    a.p[lvl][o] = tanhf(fmaxf(s1, 0.f));
    b.p[lvl][o] = tanhf(fmaxf(s2, 0.f));
}

// =========================================================================================================
// scatter + gate + residual: out = x + nearest(bsf)[c] * (a + b); one thread per 4 consecutive pixels
// =========================================================================================================
struct QuadTable {
    int64_t first[NL_ + 1];   // prefix sum over levels of B*C*h*ceil(w/4)
    int wq[NL_];
};

struct BlockQuadTable {       // pixel quads (B*h*ceil(w/4)) of each level, in whole 64-quad blocks
    int first_blk[NL_ + 1];
    int64_t quads[NL_];
    int wq[NL_];
    int nwork[NL_];           // XCD-ordered kernels: 64-quad blocks of the level with work (first_blk advances in multiples of 8)
};

// out = x + nearest(bsf) * (a + b).  Workgroup = 64 pixel quads (block-aligned per level, so the level is a function
// of blockIdx) x 4 waves; a wave walks SC_NB batches of SC_CG channels of its quad column: the gate quads and the
// nearest indices are computed once per thread, and the reads of a batch are issued before its stores.
constexpr int SC_CG = 4, SC_NB = 4;
__global__ __launch_bounds__(256) void fpn_scatter_fwd_kernel(Levels lv, BlockQuadTable bt, LevelPtrs x, const float* __restrict__ bsf,
                                                              LevelPtrs a, LevelPtrs b, LevelMutPtrs out) {
    const int lq = threadIdx.x & 63, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lvl = 0;
    while (lvl + 1 < lv.n && (int)blockIdx.x >= bt.first_blk[lvl + 1]) ++lvl;
    // XCD-aware order (round 3): a level's workgroups start at a multiple of 8 and gridDim.x is one, so workgroup j of the
    // level runs on XCD j % 8 whatever blockIdx.y is; it takes quad block (j % 8) * ceil(n / 8) + j / 8, i.e. each XCD walks
    // one contiguous range of the level and the bsf rows a range shares (a level-0 row block re-reads its bsf row 4 times)
    // are fetched by ONE L2 instead of eight (PMC: 268 MB fetched for 192).
    const int jb = (int)blockIdx.x - bt.first_blk[lvl];
    const int nwork = bt.nwork[lvl];
    const int wb = (jb & 7) * ((nwork + 7) >> 3) + (jb >> 3);
    if (wb >= nwork) return;
    const unsigned q_blk = (unsigned)wb * 64u;
    unsigned q = q_blk + lq;
    const bool active = q < (unsigned)bt.quads[lvl];
    if (!active) q = q_blk;
    const int h = lv.H[lvl], w = lv.W[lvl], wq = bt.wq[lvl];
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const unsigned qrow = q / (unsigned)wq;
    const int x0 = (int)(q - qrow * (unsigned)wq) * 4;
    const int bi = (int)(qrow / (unsigned)h);
    const int y = (int)(qrow - (unsigned)bi * (unsigned)h);
    const bool full = x0 + 3 < w;
    const int e1 = x0 + 1 < w ? 1 : 0, e2 = x0 + 2 < w ? 2 : e1, e3 = x0 + 3 < w ? 3 : e2;
    const float sy_scale = nearest_scale(hr, h), sx_scale = nearest_scale(wr, w);
    const int sy = nearest_src(y, hr, sy_scale);
    const int sx[4] = {nearest_src(x0, wr, sx_scale), nearest_src(x0 + e1, wr, sx_scale), nearest_src(x0 + e2, wr, sx_scale),
                       nearest_src(x0 + e3, wr, sx_scale)};
    float s4[4];
    {
        const int64_t moff = ((int64_t)bi * h + y) * w + x0;
        const float* ap = a.p[lvl] + moff;
        const float* bp = b.p[lvl] + moff;
        s4[0] = ap[0] + bp[0]; s4[1] = ap[e1] + bp[e1]; s4[2] = ap[e2] + bp[e2]; s4[3] = ap[e3] + bp[e3];
    }
    const int c_wave = (blockIdx.y * 4 + cw) * (SC_CG * SC_NB);
    const int c_last = lv.C - 1;
#pragma unroll 1
    for (int nb = 0; nb < SC_NB; ++nb) {
        const int c_first = c_wave + nb * SC_CG;
        if (c_first >= lv.C) break;
        float xv[SC_CG][4], u[SC_CG][4];
#pragma unroll
        for (int cj = 0; cj < SC_CG; ++cj) {
            const int c = c_first + cj < lv.C ? c_first + cj : c_last;
            const int64_t bc = (int64_t)bi * lv.C + c;
            const float* xp = x.p[lvl] + (bc * h + y) * (int64_t)w + x0;
            const float* brow = bsf + (bc * hr + sy) * (int64_t)wr;
            if (full) {
                const float4_u v = *reinterpret_cast<const float4_u*>(xp);
                xv[cj][0] = v[0]; xv[cj][1] = v[1]; xv[cj][2] = v[2]; xv[cj][3] = v[3];
            } else {
                xv[cj][0] = xp[0]; xv[cj][1] = xp[e1]; xv[cj][2] = xp[e2]; xv[cj][3] = xp[e3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) u[cj][e] = brow[sx[e]];
        }
#pragma unroll
        for (int cj = 0; cj < SC_CG; ++cj) {
            if (!active || c_first + cj >= lv.C) continue;
            float* op = out.p[lvl] + (((int64_t)bi * lv.C + c_first + cj) * h + y) * (int64_t)w + x0;
            if (full) {
                float4_u o = {xv[cj][0] + u[cj][0] * s4[0], xv[cj][1] + u[cj][1] * s4[1], xv[cj][2] + u[cj][2] * s4[2],
                              xv[cj][3] + u[cj][3] * s4[3]};
                *reinterpret_cast<float4_u*>(op) = o;
            } else {
                op[0] = xv[cj][0] + u[cj][0] * s4[0];
                if (x0 + 1 < w) op[1] = xv[cj][1] + u[cj][1] * s4[1];
                if (x0 + 2 < w) op[2] = xv[cj][2] + u[cj][2] * s4[2];
            }
        }
    }
}

// =========================================================================================================
// backward A: dm_i = sum_c g_i[c] * U(bsf)[c]  -> dpre_a = dm (1 - a^2) [a > 0], dpre_b likewise.
//   block = 64 pixel quads x 4 channel quarters; the quarters are reduced through LDS.
// =========================================================================================================
struct PixQuadTable {
    int64_t first[NL_ + 1];   // prefix sum over levels of B*h*ceil(w/4)
    int wq[NL_];
};

__global__ __launch_bounds__(256) void fpn_scatter_bwd_maps_kernel(Levels lv, PixQuadTable pt, LevelPtrs gout,
                                                                   const float* __restrict__ bsf, LevelPtrs a, LevelPtrs b,
                                                                   LevelMutPtrs dpa, LevelMutPtrs dpb) {
    __shared__ float red[3][64][4];
    const int lq = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    int64_t q = (int64_t)blockIdx.x * 64 + lq;
    const bool active = q < pt.first[lv.n];
    float dm[4] = {0.f, 0.f, 0.f, 0.f};
    int lvl = 0, h = 1, w = 1, x0 = 0, y = 0, bi = 0;
    if (active) {
        while (lvl + 1 < lv.n && q >= pt.first[lvl + 1]) ++lvl;
        q -= pt.first[lvl];
        h = lv.H[lvl]; w = lv.W[lvl];
        const int wq = pt.wq[lvl];
        x0 = (int)(q % wq) * 4;
        y = (int)((q / wq) % h);
        bi = (int)(q / ((int64_t)wq * h));
        const int hr = lv.H[lv.r], wr = lv.W[lv.r];
        const float sy_scale = nearest_scale(hr, h), sx_scale = nearest_scale(wr, w);
        const int sy = nearest_src(y, hr, sy_scale);
        int sx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sx[e] = nearest_src(x0 + e < w ? x0 + e : w - 1, wr, sx_scale);
        // channels: gridDim.y chunks (more workgroups in flight than the 700 pixel blocks alone), a quarter of a chunk per wave
        const int cchunk = (lv.C + (int)gridDim.y - 1) / (int)gridDim.y;
        const int ch0 = blockIdx.y * cchunk, ch1 = ch0 + cchunk < lv.C ? ch0 + cchunk : lv.C;
        const int cq = (ch1 - ch0 + 3) / 4;
        const int c_begin = ch0 + quarter * cq < ch1 ? ch0 + quarter * cq : ch1;
        const int c_end = c_begin + cq < ch1 ? c_begin + cq : ch1;
        const bool full = x0 + 3 < w;
        const float* gp = gout.p[lvl] + (((int64_t)bi * lv.C + c_begin) * h + y) * w + x0;
        const float* bp = bsf + (((int64_t)bi * lv.C + c_begin) * hr + sy) * wr;
        const int64_t gstep = (int64_t)h * w, bstep = (int64_t)hr * wr;
        // clamped element offsets for the row-end quad (its lanes past the row are never stored): no branch in the
        // channel loop, eight channels' reads in flight
        const int e1 = x0 + 1 < w ? 1 : 0, e2 = x0 + 2 < w ? 2 : e1, e3 = x0 + 3 < w ? 3 : e2;
#pragma unroll 8
        for (int c = c_begin; c < c_end; ++c) {
            float4_u gv;
            if (full) gv = *reinterpret_cast<const float4_u*>(gp);
            else { gv[0] = gp[0]; gv[1] = gp[e1]; gv[2] = gp[e2]; gv[3] = gp[e3]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) dm[e] += gv[e] * bp[sx[e]];
            gp += gstep; bp += bstep;
        }
    }
    if (quarter > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[quarter - 1][lq][e] = dm[e];
    }
    __syncthreads();
    if (quarter == 0 && active) {
        const int64_t moff = ((int64_t)bi * h + y) * w + x0;
        for (int e = 0; e < 4 && x0 + e < w; ++e) {
            const float d = ((dm[e] + red[0][lq][e]) + red[1][lq][e]) + red[2][lq][e];
            if (gridDim.y > 1) {                 // partial sum over this chunk's channels: dpa collects dm (zeroed before)
                unsafeAtomicAdd(dpa.p[lvl] + moff + e, d);
            } else {
                const float av = a.p[lvl][moff + e], bv = b.p[lvl][moff + e];
                dpa.p[lvl][moff + e] = av > 0.f ? d * (1.f - av * av) : 0.f;
                dpb.p[lvl][moff + e] = bv > 0.f ? d * (1.f - bv * bv) : 0.f;
            }
        }
    }
}
// dpa holds dm = sum_c g * nearest(bsf): turn it into the two pre-activation gradients (tanh' * relu')
__global__ void fpn_scatter_bwd_maps_finish_kernel(Levels lv, TileTable tt, LevelPtrs a, LevelPtrs b, LevelMutPtrs dpa,
                                                   LevelMutPtrs dpb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tt.px_off[lv.n]) return;
    int lvl = 0;
    while (lvl + 1 < lv.n && i >= tt.px_off[lvl + 1]) ++lvl;
    const int64_t o = i - tt.px_off[lvl];
    const float d = dpa.p[lvl][o], av = a.p[lvl][o], bv = b.p[lvl][o];
    dpa.p[lvl][o] = av > 0.f ? d * (1.f - av * av) : 0.f;
    dpb.p[lvl][o] = bv > 0.f ? d * (1.f - bv * bv) : 0.f;
}

constexpr int DB_CG = 4;       // channels per thread: the gate quads (a + b) of a window are read once for all of them
__global__ __launch_bounds__(256) void fpn_scatter_bwd_dbsf_kernel(Levels lv, LevelPtrs gout, LevelPtrs a, LevelPtrs b,
                                                                   float* __restrict__ dbsf) {
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const int cgs = (lv.C + DB_CG - 1) / DB_CG;
    const int64_t total = (int64_t)lv.B * cgs * hr * wr;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int sx = (int)(idx % wr);
    const int sy = (int)((idx / wr) % hr);
    const int64_t bg = idx / ((int64_t)wr * hr);
    const int bi = (int)(bg / cgs), c0 = (int)(bg - (int64_t)bi * cgs) * DB_CG;
    float acc[DB_CG];
#pragma unroll
    for (int j = 0; j < DB_CG; ++j) acc[j] = 0.f;
    int cc[DB_CG];
#pragma unroll
    for (int j = 0; j < DB_CG; ++j) cc[j] = c0 + j < lv.C ? c0 + j : lv.C - 1;
    for (int i = 0; i < lv.n; ++i) {
        const int h = lv.H[i], w = lv.W[i];
        int y0, y1, x0, x1;
        const int sh = lv.sh[i];
        if (sh != SH_GENERAL && sh >= 0) {            // finer or equal level: dst block [s<<sh, (s+1)<<sh)
            y0 = sy << sh; y1 = (sy + 1) << sh; x0 = sx << sh; x1 = (sx + 1) << sh;
        } else if (sh != SH_GENERAL) {                // coarser level: dst d reads src d << -sh
            const int f = -sh, m = (1 << f) - 1;
            if ((sy & m) || (sx & m)) continue;
            y0 = sy >> f; y1 = y0 + 1; x0 = sx >> f; x1 = x0 + 1;
        } else {
            const float ys = nearest_scale(hr, h), xs = nearest_scale(wr, w);
            y0 = nearest_dst_begin(sy, hr, h, ys); y1 = nearest_dst_begin(sy + 1, hr, h, ys);
            x0 = nearest_dst_begin(sx, wr, w, xs); x1 = nearest_dst_begin(sx + 1, wr, w, xs);
        }
        const int64_t plane = (int64_t)h * w;
        const float* gp = gout.p[i] + (int64_t)bi * lv.C * plane;
        const float* ap = a.p[i] + (int64_t)bi * plane;
        const float* bp = b.p[i] + (int64_t)bi * plane;
        if (x1 - x0 == 4) {
            for (int yy = y0; yy < y1; ++yy) {
                const int64_t o = (int64_t)yy * w + x0;
                const float4_u av = *reinterpret_cast<const float4_u*>(ap + o);
                const float4_u bv = *reinterpret_cast<const float4_u*>(bp + o);
                float4_u gv[DB_CG];
#pragma unroll
                for (int j = 0; j < DB_CG; ++j) gv[j] = *reinterpret_cast<const float4_u*>(gp + cc[j] * plane + o);
#pragma unroll
                for (int j = 0; j < DB_CG; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j] += gv[j][e] * (av[e] + bv[e]);
            }
        } else {
            for (int yy = y0; yy < y1; ++yy)
                for (int xx = x0; xx < x1; ++xx) {
                    const int64_t o = (int64_t)yy * w + xx;
                    const float sv = ap[o] + bp[o];
#pragma unroll
                    for (int j = 0; j < DB_CG; ++j) acc[j] += gp[cc[j] * plane + o] * sv;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < DB_CG; ++j)
        if (c0 + j < lv.C) dbsf[(((int64_t)bi * lv.C + c0 + j) * hr + sy) * wr + sx] = acc[j];
}

// =========================================================================================================
// backward A+B fused (round 2): ONE pass over the upstream gradient g produces both
//     dm_i[dst]  = sum_c g_i[c][dst] * bsf[c][src(dst)]        (every dst pixel has exactly one nearest source)
//     dbsf[c][s] = sum_i sum_{dst: src(dst) = s} g_i[c][dst] * (a_i + b_i)[dst]
// instead of reading g twice (maps kernel + dbsf kernel: 2 x 183 MB at the bench size).  A thread owns one refine-level
// pixel s of one image and walks a range of channels; the destination windows of s (NF finer levels with windows of
// (2^sh)^2 pixels, the refine level itself, at most one pixel on every coarser level) and their gate sums live in
// registers, so per channel it reads the windows once (16 / 8 / 4-byte reads, lanes along sx: contiguous rows), adds
// g*bsf[c][s] into its dm registers and the gate-weighted sum into dbsf[c][s].  The 4 waves of a workgroup take the 4
// quarters of a channel chunk for the SAME 64 source pixels; dm is reduced across them through LDS and added atomically
// into the zeroed dpa maps (one add per chunk), which fpn_scatter_bwd_maps_finish_kernel turns into dpre_a / dpre_b.
// Instantiated for NF = 0, 1, 2 finer levels whose size ratio to the refine level is exactly 2^(r - i); other pyramids
// (ragged sizes, refine level 3 or 4) keep the two-kernel form.
// =========================================================================================================
template <int NF, class T>
__global__ __launch_bounds__(256) void fpn_scatter_bwd_fused_kernel(Levels lv, LevelPtrs gout, const float* __restrict__ bsf,
                                                                    LevelPtrs a, LevelPtrs b, LevelMutPtrs dpa,
                                                                    float* __restrict__ dbsf) {
    constexpr int W0 = NF >= 1 ? (1 << NF) : 1;              // window edge of level 0 (finest)
    constexpr int W1 = NF >= 2 ? (1 << (NF - 1)) : 1;        // window edge of level 1 when NF == 2
    constexpr int NC = ARFE_MAX_LEVELS;                       // slots for the refine level and the coarser ones
    __shared__ float red[3][64][W0 * W0 + W1 * W1 + NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hr = lv.H[lv.r], wr = lv.W[lv.r], npx = hr * wr;
    const int blocks_per_img = (npx + 63) / 64;
    const int bi = blockIdx.x / blocks_per_img;
    int s = (blockIdx.x - bi * blocks_per_img) * 64 + lane;
    const bool active = s < npx;
    if (!active) s = npx - 1;
    const int sy = s / wr, sx = s - sy * wr;
    // channel range of this wave
    const int cchunk = (lv.C + (int)gridDim.y - 1) / (int)gridDim.y;
    const int ch0 = blockIdx.y * cchunk, ch1 = ch0 + cchunk < lv.C ? ch0 + cchunk : lv.C;
    const int cq = (ch1 - ch0 + 3) / 4;
    const int c_begin = ch0 + wave * cq < ch1 ? ch0 + wave * cq : ch1;
    const int c_end = c_begin + cq < ch1 ? c_begin + cq : ch1;

    // ---- window geometry + gate sums (channel independent) ----
    float s0[W0 * W0], s1[W1 * W1], sc[NC];
    float d0[W0 * W0], d1[W1 * W1], dc[NC];
    int64_t o0 = 0, o1 = 0, oc[NC];            // element offset of the window's first pixel inside a channel plane
    bool vc[NC];
#pragma unroll
    for (int e = 0; e < W0 * W0; ++e) { s0[e] = 0.f; d0[e] = 0.f; }
#pragma unroll
    for (int e = 0; e < W1 * W1; ++e) { s1[e] = 0.f; d1[e] = 0.f; }
    if constexpr (NF >= 1) {
        const int w = lv.W[0];
        o0 = (int64_t)(sy * W0) * w + sx * W0;
        const float* ap = a.p[0] + (int64_t)bi * lv.H[0] * w + o0;
        const float* bp = b.p[0] + (int64_t)bi * lv.H[0] * w + o0;
#pragma unroll
        for (int yy = 0; yy < W0; ++yy)
#pragma unroll
            for (int xx = 0; xx < W0; ++xx) s0[yy * W0 + xx] = ap[yy * w + xx] + bp[yy * w + xx];
    }
    if constexpr (NF >= 2) {
        const int w = lv.W[1];
        o1 = (int64_t)(sy * W1) * w + sx * W1;
        const float* ap = a.p[1] + (int64_t)bi * lv.H[1] * w + o1;
        const float* bp = b.p[1] + (int64_t)bi * lv.H[1] * w + o1;
#pragma unroll
        for (int yy = 0; yy < W1; ++yy)
#pragma unroll
            for (int xx = 0; xx < W1; ++xx) s1[yy * W1 + xx] = ap[yy * w + xx] + bp[yy * w + xx];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int i = NF + k;                                  // refine level (k = 0) and the coarser ones
        sc[k] = 0.f; dc[k] = 0.f; oc[k] = 0; vc[k] = false;
        if (i >= lv.n) continue;
        const int h = lv.H[i], w = lv.W[i];
        int y0, y1, x0, x1;
        const int sh = lv.sh[i];
        if (sh == 0) { y0 = sy; y1 = sy + 1; x0 = sx; x1 = sx + 1; }
        else if (sh != SH_GENERAL && sh < 0) {
            const int f = -sh, m = (1 << f) - 1;
            const bool hit = !(sy & m) && !(sx & m);
            y0 = sy >> f; x0 = sx >> f; y1 = y0 + (hit ? 1 : 0); x1 = x0 + (hit ? 1 : 0);
        } else {
            const float ys = nearest_scale(hr, h), xs = nearest_scale(wr, w);
            y0 = nearest_dst_begin(sy, hr, h, ys); y1 = nearest_dst_begin(sy + 1, hr, h, ys);
            x0 = nearest_dst_begin(sx, wr, w, xs); x1 = nearest_dst_begin(sx + 1, wr, w, xs);
        }
        vc[k] = y1 > y0 && x1 > x0;       // (a coarser map has at most one pixel whose nearest source is s)
        if (vc[k]) {
            oc[k] = (int64_t)y0 * w + x0;
            sc[k] = a.p[i][(int64_t)bi * h * w + oc[k]] + b.p[i][(int64_t)bi * h * w + oc[k]];
        }
    }
    // ---- channel loop: two channels in flight ----
    const int64_t plane_r = (int64_t)npx;
    auto one = [&](int c, float (&g0)[W0 * W0], float (&g1)[W1 * W1], float (&gc)[NC], float& u) {
        const int64_t bc = (int64_t)bi * lv.C + c;
        u = bsf[bc * plane_r + s];
        if constexpr (NF >= 1) {
            const int w = lv.W[0];
            const T* gp = (const T*)gout.p[0] + bc * (int64_t)lv.H[0] * w + o0;
#pragma unroll
            for (int yy = 0; yy < W0; ++yy) {
                if constexpr (W0 == 4) {
                    const float4_u v = ld4(gp + yy * w);
                    g0[yy * 4 + 0] = v[0]; g0[yy * 4 + 1] = v[1]; g0[yy * 4 + 2] = v[2]; g0[yy * 4 + 3] = v[3];
                } else {
#pragma unroll
                    for (int xx = 0; xx < W0; ++xx) g0[yy * W0 + xx] = (float)gp[yy * w + xx];
                }
            }
        }
        if constexpr (NF >= 2) {
            const int w = lv.W[1];
            const T* gp = (const T*)gout.p[1] + bc * (int64_t)lv.H[1] * w + o1;
#pragma unroll
            for (int yy = 0; yy < W1; ++yy)
#pragma unroll
                for (int xx = 0; xx < W1; ++xx) g1[yy * W1 + xx] = (float)gp[yy * w + xx];
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int i = NF + k;
            gc[k] = 0.f;
            if (i < lv.n && vc[k]) gc[k] = (float)((const T*)gout.p[i])[bc * (int64_t)lv.H[i] * lv.W[i] + oc[k]];
        }
    };
    auto fold = [&](int c, const float (&g0)[W0 * W0], const float (&g1)[W1 * W1], const float (&gc)[NC], float u) {
        float acc = 0.f;
        if constexpr (NF >= 1) {
#pragma unroll
            for (int e = 0; e < W0 * W0; ++e) { d0[e] += g0[e] * u; acc += g0[e] * s0[e]; }
        }
        if constexpr (NF >= 2) {
#pragma unroll
            for (int e = 0; e < W1 * W1; ++e) { d1[e] += g1[e] * u; acc += g1[e] * s1[e]; }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) { dc[k] += gc[k] * u; acc += gc[k] * sc[k]; }
        if (active) dbsf[((int64_t)bi * lv.C + c) * plane_r + s] = acc;
    };
    int c = c_begin;
    for (; c + 1 < c_end; c += 2) {
        float ga0[W0 * W0], ga1[W1 * W1], gac[NC], ua;
        float gb0[W0 * W0], gb1[W1 * W1], gbc[NC], ub;
        one(c, ga0, ga1, gac, ua);
        one(c + 1, gb0, gb1, gbc, ub);
        fold(c, ga0, ga1, gac, ua);
        fold(c + 1, gb0, gb1, gbc, ub);
    }
    if (c < c_end) {
        float ga0[W0 * W0], ga1[W1 * W1], gac[NC], ua;
        one(c, ga0, ga1, gac, ua);
        fold(c, ga0, ga1, gac, ua);
    }
    // ---- dm: reduce the four channel quarters, one atomic add per destination pixel and chunk ----
    constexpr int ND = W0 * W0 + W1 * W1 + NC;
    if (wave > 0) {
        float* r = &red[wave - 1][lane][0];
#pragma unroll
        for (int e = 0; e < W0 * W0; ++e) r[e] = d0[e];
#pragma unroll
        for (int e = 0; e < W1 * W1; ++e) r[W0 * W0 + e] = d1[e];
#pragma unroll
        for (int k = 0; k < NC; ++k) r[W0 * W0 + W1 * W1 + k] = dc[k];
    }
    __syncthreads();
    if (wave == 0 && active) {
        auto total = [&](int e, float mine) { return ((mine + red[0][lane][e]) + red[1][lane][e]) + red[2][lane][e]; };
        if constexpr (NF >= 1) {
            const int w = lv.W[0];
            float* dp = dpa.p[0] + (int64_t)bi * lv.H[0] * w + o0;
#pragma unroll
            for (int yy = 0; yy < W0; ++yy)
#pragma unroll
                for (int xx = 0; xx < W0; ++xx) unsafeAtomicAdd(dp + yy * w + xx, total(yy * W0 + xx, d0[yy * W0 + xx]));
        }
        if constexpr (NF >= 2) {
            const int w = lv.W[1];
            float* dp = dpa.p[1] + (int64_t)bi * lv.H[1] * w + o1;
#pragma unroll
            for (int yy = 0; yy < W1; ++yy)
#pragma unroll
                for (int xx = 0; xx < W1; ++xx)
                    unsafeAtomicAdd(dp + yy * w + xx, total(W0 * W0 + yy * W1 + xx, d1[yy * W1 + xx]));
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int i = NF + k;
            if (i < lv.n && vc[k])
                unsafeAtomicAdd(dpa.p[i] + (int64_t)bi * lv.H[i] * lv.W[i] + oc[k], total(W0 * W0 + W1 * W1 + k, dc[k]));
        }
    }
    (void)ND;
}

// =========================================================================================================
// backward C: dx = g + conv3x3^T(dpre_a; w1) + conv3x3^T(dpre_b; w2) + gather^T(dm / n)
//   block = 64 pixel quads x 4 waves; a thread keeps the 3x6 neighbourhood of BOTH 1-channel gradient maps in
//   registers (they do not depend on the channel) and each wave handles IB_CG consecutive channels: per channel one 16-B
//   read of g, 18 scalar weight loads (wave-uniform), 72 FMAs, the pooling term and one 16-B store.
// =========================================================================================================
constexpr int IB_CG = 2;   // channels whose reads are in flight together
constexpr int IB_NB = 4;   // such batches per wave (amortises the neighbourhood reads of the two maps)
// Each index regime is its own instantiation so that only its paths are compiled in (all paths in one kernel made
// 167 KB of code at 256 VGPRs): levels with an exact power-of-two ratio to the refine level use shifts only.
// MODE 0 = POW2 (shift-only paths), 1 = generic level >= refine (nearest^T), 2 = generic level < refine (pool windows)
template <int MODE>
__global__ __launch_bounds__(256) void fpn_input_bwd_kernel(Levels lv, BlockQuadTable bt, LevelPtrs gout, LevelPtrs dpa,
                                                            LevelPtrs dpb, LevelPtrs w1, LevelPtrs w2,
                                                            const float* __restrict__ dm, const int32_t* __restrict__ argmax,
                                                            LevelMutPtrs dx) {
    // The level is a function of blockIdx only (block-aligned table) and the wave index goes through readfirstlane,
    // so level geometry, base pointers and the per-channel conv weights are all wave-uniform (scalar loads).
    const int lq = threadIdx.x & 63, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lvl = 0;
    while (lvl + 1 < lv.n && (int)blockIdx.x >= bt.first_blk[lvl + 1]) ++lvl;
    // quads of a level fit 31 bits (checked on the host): 32-bit divisions only
    const unsigned q_blk = (unsigned)((int)blockIdx.x - bt.first_blk[lvl]) * 64u;   // always a valid quad
    unsigned q = q_blk + lq;
    const bool active = q < (unsigned)bt.quads[lvl];
    if (!active) q = q_blk;                   // idle lanes follow along on the block's first quad and never store
    const int h = lv.H[lvl], w = lv.W[lvl], wq = bt.wq[lvl];
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const unsigned qrow = q / (unsigned)wq;
    const int x0 = (int)(q - qrow * (unsigned)wq) * 4;
    const int bi = (int)(qrow / (unsigned)h);
    const int y = (int)(qrow - (unsigned)bi * (unsigned)h);
    const bool full = x0 + 3 < w;

    // channel-independent neighbourhood of the two maps; addresses are clamped into the map and the value is
    // zeroed by a select, so the 36 loads are unconditional and issue back to back
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t vab[3][6];                 // (map a, map b) pairs: the conv-transpose below is one packed FMA per pair
    {
        const float* da = dpa.p[lvl] + (int64_t)bi * h * w;
        const float* db = dpb.p[lvl] + (int64_t)bi * h * w;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - (ky - 1);
            const bool rok = yy >= 0 && yy < h;
            const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int xx = x0 - 1 + t;
                const bool ok = rok && xx >= 0 && xx < w;
                const int xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                const float ta = da[yc * w + xc], tb = db[yc * w + xc];
                vab[ky][t] = f32x2_t{ok ? ta : 0.f, ok ? tb : 0.f};
            }
        }
    }
    const float nlev = (float)lv.n;
    const float inv_nlev = 1.0f / nlev;       // hot path: d(sum / n) as a multiplication (<= 1 ulp from the division; an IEEE
                                              // division is ~10 vector instructions, 4 per channel against 36 packed FMAs)
    const int64_t plane = (int64_t)h * w, plane_r = (int64_t)hr * wr;
    const int sh = lv.sh[lvl];
    // The conv weights are wave-uniform. Read through the constant address space they become scalar loads (the
    // compiler cannot prove the plain pointers are not aliased by the dx stores and would emit per-lane loads).
    constexpr int CG = MODE == 0 ? IB_CG : 1;   // generic-index levels are tiny: one channel per wave, more workgroups
    constexpr int NB = MODE == 0 ? IB_NB : 1;
    constexpr int CW = CG * NB;                 // channels per wave
    typedef const float __attribute__((address_space(4)))* const_fp;
    const const_fp k1g = (const_fp)(uintptr_t)w1.p[lvl], k2g = (const_fp)(uintptr_t)w2.p[lvl];
    // wave cw of block row blockIdx.y owns IB_CG consecutive channels; the loop is fully unrolled so that the
    // global reads of all of them are in flight together
    const int c_wave = (blockIdx.y * 4 + cw) * CW;
    constexpr bool POW2 = MODE == 0;
    if constexpr (MODE == 0) {
        // Shift-only levels carry almost all of the bytes. A wave walks NB batches of CG channels (amortising the
        // neighbourhood reads above); all global reads of a batch are issued together, ahead of any use: a store
        // between two channels would fence the next channel's reads (dx may alias the inputs as far as the compiler knows).
        const int c_last = lv.C - 1;
        const int e1 = x0 + 1 < w ? 1 : 0, e2 = x0 + 2 < w ? 2 : e1, e3 = x0 + 3 < w ? 3 : e2;
        const int f0 = y * w + x0;
        const int64_t oA = (int64_t)(y >> (sh > 0 ? sh : 0)) * wr + (x0 >> (sh > 0 ? sh : 0));
        const int64_t oB = (int64_t)(y >> (sh > 0 ? sh : 0)) * wr + ((x0 + e2) >> (sh > 0 ? sh : 0));
        float gq[CG][4], pq[CG][4];   // raw reads of the batch in flight: g and the pooling-term operands
        int ia[CG], ib[CG];
        // (the sh branches stay OUTSIDE the channel loops: one basic block holds all reads of a batch)
        auto issue = [&](int c_first) {
            int64_t bcs[CG];
#pragma unroll
            for (int cj = 0; cj < CG; ++cj) {
                const int c = c_first + cj < lv.C ? c_first + cj : c_last;
                bcs[cj] = (int64_t)bi * lv.C + c;
                const float* gp = gout.p[lvl] + (bcs[cj] * h + y) * (int64_t)w + x0;
                if (full) {
                    const float4_u gv = *reinterpret_cast<const float4_u*>(gp);
                    gq[cj][0] = gv[0]; gq[cj][1] = gv[1]; gq[cj][2] = gv[2]; gq[cj][3] = gv[3];
                } else {   // clamped addresses; the lanes past the row end are never stored
                    gq[cj][0] = gp[0]; gq[cj][1] = gp[e1]; gq[cj][2] = gp[e2]; gq[cj][3] = gp[e3];
                }
            }
            if (sh >= 1) {
                // a finer level below the refine level: windows are the disjoint (1<<sh)^2 blocks, so pixels (0,1)
                // and (2,3) of the quad share a window
#pragma unroll
                for (int cj = 0; cj < CG; ++cj) {
                    const int32_t* am = argmax + ((int64_t)lvl * lv.B * lv.C + bcs[cj]) * plane_r;
                    const float* dmp = dm + bcs[cj] * plane_r;
                    ia[cj] = am[oA]; ib[cj] = am[oB];
                    pq[cj][0] = dmp[oA]; pq[cj][1] = dmp[oB];
                }
            } else if (sh == 0) {
                // same size as the refine level: the gather is the identity
#pragma unroll
                for (int cj = 0; cj < CG; ++cj) {
                    const float* dp = dm + bcs[cj] * plane_r + (int64_t)y * wr + x0;
                    if (full) {
                        const float4_u dv = *reinterpret_cast<const float4_u*>(dp);
                        pq[cj][0] = dv[0]; pq[cj][1] = dv[1]; pq[cj][2] = dv[2]; pq[cj][3] = dv[3];
                    } else {
                        pq[cj][0] = dp[0]; pq[cj][1] = dp[e1]; pq[cj][2] = dp[e2]; pq[cj][3] = dp[e3];
                    }
                }
            } else {
                // coarser than the refine level (small maps): nearest^T sums the (1<<f)^2 block of dm
                const int f = -sh, n = 1 << f;
#pragma unroll
                for (int cj = 0; cj < CG; ++cj) {
                    const float* dmp = dm + bcs[cj] * plane_r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int xx = x0 + e < w ? x0 + e : w - 1;
                        float sum = 0.f;
                        for (int py = y << f; py < (y << f) + n; ++py)
                            for (int px = xx << f; px < (xx << f) + n; ++px) sum += dmp[(int64_t)py * wr + px] / nlev;
                        pq[cj][e] = sum;
                    }
                }
            }
        };
        issue(c_wave);
#pragma unroll 1
        for (int nb = 0; nb < NB; ++nb) {
            const int c_first = c_wave + nb * CG;
            if (c_first >= lv.C) break;
            float r[CG][4];
            if (sh >= 1) {
#pragma unroll
                for (int cj = 0; cj < CG; ++cj) {
                    const float da = pq[cj][0] * inv_nlev, db = pq[cj][1] * inv_nlev;
                    r[cj][0] = gq[cj][0] + (ia[cj] == f0 ? da : 0.f);
                    r[cj][1] = gq[cj][1] + (ia[cj] == f0 + 1 ? da : 0.f);
                    r[cj][2] = gq[cj][2] + (ib[cj] == f0 + 2 ? db : 0.f);
                    r[cj][3] = gq[cj][3] + (ib[cj] == f0 + 3 ? db : 0.f);
                }
            } else {
                const float dn = sh == 0 ? inv_nlev : 1.f;   // the coarser-level sums are already divided
#pragma unroll
                for (int cj = 0; cj < CG; ++cj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[cj][e] = gq[cj][e] + pq[cj][e] * dn;
            }
#pragma unroll
            for (int cj = 0; cj < CG; ++cj) {
                const int cc = c_first + cj < lv.C ? c_first + cj : c_last;
                const const_fp k1 = k1g + cc * 9, k2 = k2g + cc * 9;
                // the two maps' terms as ONE packed FMA per (tap, pixel): (acc_a, acc_b) += (u, s) * (d_a, d_b), summed at the end
                f32x2_t r2[4] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x2_t us = {k1[ky * 3 + kx], k2[ky * 3 + kx]};
                        // dx[y,x] += W[c,ky,kx] * d[y-(ky-1), x-(kx-1)] ; neighbourhood column index = e + 2 - kx
#pragma unroll
                        for (int e = 0; e < 4; ++e) r2[e] += us * vab[ky][e + 2 - kx];
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e) r[cj][e] += r2[e][0] + r2[e][1];
            }
#pragma unroll
            for (int cj = 0; cj < CG; ++cj) {
                if (!active || c_first + cj >= lv.C) continue;
                float* op = dx.p[lvl] + (((int64_t)bi * lv.C + c_first + cj) * h + y) * (int64_t)w + x0;
                if (full) {
                    float4_u o = {r[cj][0], r[cj][1], r[cj][2], r[cj][3]};
                    *reinterpret_cast<float4_u*>(op) = o;
                } else {
                    op[0] = r[cj][0];
                    if (x0 + 1 < w) op[1] = r[cj][1];
                    if (x0 + 2 < w) op[2] = r[cj][2];
                }
            }
            // (issuing these reads before the stores above was measured slower, not faster)
            if (nb + 1 < NB) issue(c_first + CG < lv.C ? c_first + CG : c_first);
        }
        return;
    }
    const int c_first = c_wave;
#pragma unroll
    for (int cj = 0; cj < CG; ++cj) {
        const int c = c_first + cj;
        if (c >= lv.C) break;
        const int64_t bc = (int64_t)bi * lv.C + c;
        const int64_t off = (bc * h + y) * (int64_t)w + x0;
        float r[4];
        {
            const float* gp = gout.p[lvl] + off;
            if (full) {
                const float4_u gv = *reinterpret_cast<const float4_u*>(gp);
                r[0] = gv[0]; r[1] = gv[1]; r[2] = gv[2]; r[3] = gv[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = x0 + e < w ? gp[e] : 0.f;
            }
        }
        const const_fp k1 = k1g + c * 9, k2 = k2g + c * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float u = k1[ky * 3 + kx], s = k2[ky * 3 + kx];
                // dx[y,x] += W[c,ky,kx] * d[y-(ky-1), x-(kx-1)] ; neighbourhood column index = e + 2 - kx
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] += u * vab[ky][e + 2 - kx][0] + s * vab[ky][e + 2 - kx][1];
            }
        // gather^T
        const float* dmp = dm + bc * plane_r;
        if (MODE != 1 && lvl < lv.r) {
            const int32_t* am = argmax + (int64_t)lvl * lv.B * lv.C * plane_r + bc * plane_r;
            if (POW2 || (sh != SH_GENERAL && sh >= 0)) {
                // windows are the disjoint (1<<sh)^2 blocks: one candidate per pixel; branch-free so that the index
                // and value reads of all pixels/channels are issued together instead of serialised behind compares
                const int oy = y >> sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xx = x0 + e < w ? x0 + e : w - 1;
                    const int64_t o = (int64_t)oy * wr + (xx >> sh);
                    const int idx = am[o];
                    const float dv = dmp[o] / nlev;
                    r[e] += (idx == y * w + xx && x0 + e < w) ? dv : 0.f;
                }
            } else if (MODE == 2) {
                const int oy0 = (int)(((int64_t)y * hr) / h) - 1;
                for (int e = 0; e < 4; ++e) {
                    const int xx = x0 + e;
                    if (xx >= w) break;
                    const int ox0 = (int)(((int64_t)xx * wr) / w) - 1;
                    const int flat = y * w + xx;
                    for (int oy = oy0; oy <= oy0 + 2; ++oy) {
                        if (oy < 0 || oy >= hr) continue;
                        if (y < pool_start(oy, h, hr) || y >= pool_end(oy, h, hr)) continue;
                        for (int ox = ox0; ox <= ox0 + 2; ++ox) {
                            if (ox < 0 || ox >= wr) continue;
                            if (xx < pool_start(ox, w, wr) || xx >= pool_end(ox, w, wr)) continue;
                            if (am[(int64_t)oy * wr + ox] == flat) r[e] += dmp[(int64_t)oy * wr + ox] / nlev;
                        }
                    }
                }
            }
        } else if (POW2 || (sh != SH_GENERAL && sh <= 0)) {
            const int f = -sh, n = 1 << f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xx = x0 + e < w ? x0 + e : w - 1;
                float s = 0.f;
                for (int py = y << f; py < (y << f) + n; ++py)
                    for (int px = xx << f; px < (xx << f) + n; ++px) s += dmp[(int64_t)py * wr + px] / nlev;
                r[e] += s;
            }
        } else if (MODE == 1) {
            const float ys = nearest_scale(h, hr), xs = nearest_scale(w, wr);
            const int py0 = nearest_dst_begin(y, h, hr, ys), py1 = nearest_dst_begin(y + 1, h, hr, ys);
            for (int e = 0; e < 4; ++e) {
                const int xx = x0 + e;
                if (xx >= w) break;
                const int px0 = nearest_dst_begin(xx, w, wr, xs), px1 = nearest_dst_begin(xx + 1, w, wr, xs);
                float s = 0.f;
                for (int py = py0; py < py1; ++py)
                    for (int px = px0; px < px1; ++px) s += dmp[(int64_t)py * wr + px] / nlev;
                r[e] += s;
            }
        }
        float* op = dx.p[lvl] + off;
        if (active) {
            if (full) {
                float4_u o = {r[0], r[1], r[2], r[3]};
                *reinterpret_cast<float4_u*>(op) = o;
            } else {
                for (int e = 0; e < 4 && x0 + e < w; ++e) op[e] = r[e];
            }
        }
    }
    (void)plane;
}

// =========================================================================================================
// gate-conv weight gradients: dw[g][c][tap] = sum_{b,y,x} x[c][y][x] * dpre_g[y - dy(tap)][x - dx(tap)]
//   = (D [18 x pixels]) . (X^T [pixels x C]) with D[m=(g,tap)][px] the shifted, zero-padded gate gradient: an
//   exact-fp32 MFMA GEMM whose K runs over pixels.  K is the contiguous axis of both operands, so a stage of 32 pixels
//   goes through LDS: X rows are read with coalesced 16-byte loads (x crosses the chip once), D is built per stage
//   (3 entries per thread), and the MFMA operands are then fetched k-major from the padded images.  One workgroup =
//   256 pixels of one image x 256 channels (4 waves x 2 channel tiles); partial sums are added atomically (18 per channel).
// =========================================================================================================
#ifndef ARFE_GW_STAGES
#define ARFE_GW_STAGES 8
#endif
constexpr int GW_PX = 32, GW_STAGES = ARFE_GW_STAGES, GW_LD = 36, GW_CH = 256;
__global__ void fpn_gate_wgrad_zero_kernel(int C, int n, LevelMutPtrs dw1, LevelMutPtrs db1, LevelMutPtrs dw2, LevelMutPtrs db2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 9 * C;
    if (i < n * per) { const int l = i / per, j = i - l * per; dw1.p[l][j] = 0.f; dw2.p[l][j] = 0.f; }
    if (i < n) { db1.p[i][0] = 0.f; db2.p[i][0] = 0.f; }
}
template <class T>
__global__ __launch_bounds__(256) void fpn_gate_wgrad_mfma_kernel(Levels lv, TileTable tt, LevelPtrs x, LevelPtrs dpa,
                                                                  LevelPtrs dpb, LevelMutPtrs dw1, LevelMutPtrs db1,
                                                                  LevelMutPtrs dw2, LevelMutPtrs db2) {
    __shared__ __attribute__((aligned(16))) float Xs[GW_CH][GW_LD];
    __shared__ float Ds[32][GW_LD];
    int t = blockIdx.x, lvl = 0;
    while (lvl + 1 < lv.n && t >= tt.first_tile[lvl + 1]) ++lvl;
    t -= tt.first_tile[lvl];
    const int h = lv.H[lvl], w = lv.W[lvl], hw = h * w;
    const int bi = t / tt.tiles_x[lvl];
    const int chunk = t - bi * tt.tiles_x[lvl];
    const int cg0 = blockIdx.y * GW_CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const T* xb = (const T*)x.p[lvl] + (int64_t)bi * lv.C * hw;
    const float* da = dpa.p[lvl] + (int64_t)bi * hw;
    const float* db = dpb.p[lvl] + (int64_t)bi * hw;
    const int px_begin = chunk * (GW_PX * GW_STAGES);
    int nst = (hw - px_begin + GW_PX - 1) / GW_PX;
    if (nst > GW_STAGES) nst = GW_STAGES;

    // rows GM_M..31 of the MFMA operand stay zero; rows below are rewritten by every stage (zeroing them here too would
    // race with the first stage's writes: there is no barrier in between)
    for (int i = tid; i < (32 - GM_M) * GW_LD; i += 256) (&Ds[GM_M][0])[i] = 0.f;
    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // X staging: pass p covers channels cg0 + 32 p .. + 31, a thread reads quad (tid & 7) of channel row (tid >> 3)
    const int xq = tid & 7, xc = tid >> 3;
    float4_u xr[8];
    auto load_x = [&](int stage) {
        const int p = px_begin + stage * GW_PX + 4 * xq;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int c = cg0 + ps * 32 + xc;
            const T* src = xb + (int64_t)(c < lv.C ? c : lv.C - 1) * hw;
            if (p + 3 < hw) {
                xr[ps] = ld4(src + p);
            } else {       // the map's last pixels: clamped element reads (their D entries are zero)
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[ps][e] = (float)src[p + e < hw ? p + e : hw - 1];
            }
        }
    };
    // D staging: thread -> pixel (tid & 31) of the stage, rows m = (tid >> 5) + 8 j
    const int dpx = tid & 31, dmg = tid >> 5;
    float sa = 0.f, sb = 0.f;
    load_x(0);
    for (int stage = 0; stage < nst; ++stage) {
        if (stage) __syncthreads();                 // the previous stage's operand reads are done
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<float4_u*>(&Xs[ps * 32 + xc][4 * xq]) = xr[ps];
        {
            const int p = px_begin + stage * GW_PX + dpx;
            const bool valid = p < hw;
            const int pc = valid ? p : hw - 1;
            const int y = pc / w, xx = pc - y * w;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int m = dmg + 8 * j;
                if (m >= GM_M) continue;
                const int g = m >= 9 ? 1 : 0, tap = m - 9 * g;
                const int yy = y - (tap / 3 - 1), xs = xx - (tap % 3 - 1);
                const bool ok = valid && yy >= 0 && yy < h && xs >= 0 && xs < w;
                const float v0 = (g ? db : da)[ok ? yy * w + xs : pc];
                const float v = ok ? v0 : 0.f;
                Ds[m][dpx] = v;
                if (m == 4) sa += v;
                if (m == 13) sb += v;
            }
        }
        __syncthreads();
        if (stage + 1 < nst) load_x(stage + 1);     // in flight during the MFMAs
#pragma unroll
        for (int s2 = 0; s2 < GW_PX / 2; ++s2) {
            const float a = Ds[l31][2 * s2 + hh];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float b = Xs[wave * 64 + nt * 32 + l31][2 * s2 + hh];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
            }
        }
    }
    // D layout: lane -> column l31 (channel); register r -> row m = (r & 3) + 8 * (r >> 2) + 4 * hh.
    // The sums go through LDS into the [channel][tap] order of dw, so that every atomic wave-instruction adds 256
    // contiguous bytes (lane stride 36 B, one value per channel row, runs an order of magnitude slower at the memory side).
    __syncthreads();                                  // all operand reads of the last stage are done: Xs is free
    float* Ts = &Xs[0][0] + wave * (2 * 64 * 9);      // [gate][64 channels of this wave][9]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m >= GM_M) continue;
            const int g = m >= 9 ? 1 : 0;
            Ts[(g * 64 + nt * 32 + l31) * 9 + (m - 9 * g)] = acc[nt][r];
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    {
        const int cw0 = cg0 + wave * 64;
        const int nch = lv.C - cw0 < 64 ? lv.C - cw0 : 64;
        for (int i = lane; i < nch * 9; i += 64) {
            unsafeAtomicAdd(dw1.p[lvl] + cw0 * 9 + i, Ts[i]);
            unsafeAtomicAdd(dw2.p[lvl] + cw0 * 9 + i, Ts[64 * 9 + i]);
        }
    }
    if (blockIdx.y == 0) {          // bias gradients = plain sums of the two maps (the centre taps of D)
        const float v1 = wave_sum(sa), v2 = wave_sum(sb);
        if (lane == 0) {
            if (v1 != 0.f) unsafeAtomicAdd(db1.p[lvl], v1);
            if (v2 != 0.f) unsafeAtomicAdd(db2.p[lvl], v2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
static TileTable make_tile_table(const Levels& lv, int chunk_px = GM_PX) {
    TileTable tt;
    int acc = 0;
    int64_t px = 0;
    for (int i = 0; i < NL_; ++i) {
        tt.tiles_x[i] = tt.tiles_y[i] = 0;
        tt.first_tile[i] = acc;
        tt.px_off[i] = px;
        if (i < lv.n) {
            tt.tiles_x[i] = cdiv(lv.H[i] * lv.W[i], chunk_px);
            tt.tiles_y[i] = 1;
            acc += tt.tiles_x[i] * lv.B;
            px += (int64_t)lv.B * lv.H[i] * lv.W[i];
        }
    }
    tt.first_tile[NL_] = acc;
    tt.px_off[NL_] = px;
    return tt;
}
static QuadTable make_quad_table(const Levels& lv) {
    QuadTable qt;
    int64_t acc = 0;
    for (int i = 0; i <= NL_; ++i) {
        qt.first[i] = acc;
        if (i < NL_) {
            qt.wq[i] = i < lv.n ? cdiv(lv.W[i], 4) : 0;
            if (i < lv.n) acc += (int64_t)lv.B * lv.C * lv.H[i] * qt.wq[i];
        }
    }
    return qt;
}
static PixQuadTable make_pixquad_table(const Levels& lv) {
    PixQuadTable pt;
    int64_t acc = 0;
    for (int i = 0; i <= NL_; ++i) {
        pt.first[i] = acc;
        if (i < NL_) {
            pt.wq[i] = i < lv.n ? cdiv(lv.W[i], 4) : 0;
            if (i < lv.n) acc += (int64_t)lv.B * lv.H[i] * pt.wq[i];
        }
    }
    return pt;
}
// second-generation kernels (fpn_stream.hip)
int32_t fpn_input_bwd_v2(const Levels& lv, const float* const* gout, const float* const* dpa, const float* const* dpb,
                         const float* const* w1, const float* const* w2, const float* dm, const int32_t* argmax,
                         float* const* dx, hipStream_t stream);

int32_t fpn_gather_gate_v2(const Levels& lv, const float* const* x, const float* const* w1, const float* const* w2, float* m,
                           int32_t* argmax, float* Gbuf, float* T, hipStream_t stream);
int32_t fpn_scatter_fwd_v2(const Levels& lv, const float* const* x, const float* bsf, const float* const* a,
                           const float* const* b, float* const* out, hipStream_t stream);

template <class T>
static bool fill_ptrs(T& dst, const float* const* src, int n) {
    for (int i = 0; i < NL_; ++i) dst.p[i] = nullptr;
    if (!src) return false;
    for (int i = 0; i < n; ++i) {
        if (!src[i]) return false;
        dst.p[i] = const_cast<float*>(src[i]);
    }
    return true;
}

}  // namespace arfe

using namespace arfe;

extern "C" int64_t arfe_fpn_gatemap_workspace_bytes(const arfe_pyramid_t* pyr) {
    if (check_pyramid(pyr)) return -1;
    int64_t px = 0;
    for (int i = 0; i < pyr->num_levels; ++i) px += (int64_t)pyr->batch * pyr->height[i] * pyr->width[i];
    return ((int64_t)GM_M * px + (int64_t)pyr->num_levels * pyr->channels * 32) * (int64_t)sizeof(float);
}

extern "C" int32_t arfe_fpn_gather_fwd(const arfe_pyramid_t* pyr, const float* const* x, float* m, int32_t* argmax,
                                       void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs xp;
    ARFE_CHECK_ARG(fill_ptrs(xp, x, lv.n) && m, ARFE_ERR_ARG, "fpn_gather_fwd: null pointer");
    ARFE_CHECK_ARG(!lv.half, ARFE_ERR_ARG, "fpn_gather_fwd: fp32 storage only (half pyramids: arfe_fpn_gather_gate_fwd)");
    const int64_t total = (int64_t)lv.B * lv.C * lv.H[lv.r] * lv.W[lv.r];
    hipLaunchKernelGGL(fpn_gather_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), lv, xp,
                       m, argmax);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_fpn_gatemap_fwd(const arfe_pyramid_t* pyr, const float* const* x, const float* const* w1,
                                        const float* const* b1, const float* const* w2, const float* const* b2,
                                        float* const* a, float* const* b, void* workspace, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs xp, w1p, w2p, b1p, b2p;
    LevelMutPtrs ap, bp;
    ARFE_CHECK_ARG(fill_ptrs(xp, x, lv.n) && fill_ptrs(w1p, w1, lv.n) && fill_ptrs(w2p, w2, lv.n) &&
                       fill_ptrs(b1p, b1, lv.n) && fill_ptrs(b2p, b2, lv.n) && fill_ptrs(ap, a, lv.n) &&
                       fill_ptrs(bp, b, lv.n) && workspace,
                   ARFE_ERR_ARG, "fpn_gatemap_fwd: null pointer");
    ARFE_CHECK_ARG(!lv.half, ARFE_ERR_ARG, "fpn_gatemap_fwd: fp32 storage only (half pyramids: arfe_fpn_gather_gate_fwd)");
    TileTable tt = make_tile_table(lv);
    const int64_t pre_total = tt.px_off[lv.n];
    float* pre = (float*)workspace;
    hipStream_t st = as_stream(stream);
    float* Wp = pre + GM_M * pre_total;
    hipLaunchKernelGGL(fpn_gate_pack_kernel, dim3(cdiv(lv.n * lv.C * 32, 256)), dim3(256), 0, st, lv, w1p, w2p, Wp);
    ARFE_CHECK_LAUNCH();
    hipLaunchKernelGGL(fpn_gate_mfma_kernel, dim3(tt.first_tile[lv.n]), dim3(64 * GM_WAVES), 0, st, lv, tt, xp,
                       (const float*)Wp, pre);
    ARFE_CHECK_LAUNCH();
    hipLaunchKernelGGL(fpn_gate_finish_kernel, dim3((unsigned)cdiv64(pre_total, 256)), dim3(256), 0, st, lv, tt, pre,
                       pre_total, b1p, b2p, ap, bp);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// gather + gate maps in one call (round 3): when the pyramid is eligible the fused kernel reads x ONCE for both
// (fpn_stream.hip); otherwise the two first-generation ops run one after the other.  Results are those of
// arfe_fpn_gather_fwd + arfe_fpn_gatemap_fwd (m and argmax bit-identical).
// which pyramids have half-storage instantiations of EVERY AR-FPN kernel (the conditions of the second-generation kernels and
// of the single-pass scatter backward)
extern "C" int32_t arfe_fpn_half_storage_supported(const arfe_pyramid_t* pyr) {
    if (check_pyramid(pyr)) return 0;
    const Levels lv = to_levels(pyr);
    static const bool v2_off = []() { const char* e = getenv("ARFE_FPN_V2"); return e && e[0] == '0'; }();
    static const char* sw = getenv("ARFE_FPN_BWD_FUSED");
    if (v2_off || (sw && sw[0] == '0')) return 0;
    if (lv.r > 2 || lv.C < 4) return 0;
    if ((int64_t)lv.B * lv.C * lv.H[lv.r] * lv.W[lv.r] * 4 >= ((int64_t)1 << 31)) return 0;
    for (int i = 0; i < lv.n; ++i) {
        if ((int64_t)lv.B * lv.C * lv.H[i] * lv.W[i] * 4 >= ((int64_t)1 << 31)) return 0;
        if (i < lv.r && (lv.sh[i] != lv.r - i || (lv.W[i] & 3) != 0)) return 0;    // pooled levels: ratio 4 / 2, whole quads
    }
    return 1;
}

extern "C" int64_t arfe_fpn_gather_gate_workspace_bytes(const arfe_pyramid_t* pyr) {
    const int64_t base = arfe_fpn_gatemap_workspace_bytes(pyr);
    if (base < 0) return base;
    const int64_t per = (int64_t)pyr->batch * pyr->channels * pyr->height[pyr->refine_level] * pyr->width[pyr->refine_level];
    return base + (int64_t)pyr->refine_level * per * (int64_t)sizeof(float);
}

extern "C" int32_t arfe_fpn_gather_gate_fwd(const arfe_pyramid_t* pyr, const float* const* x, const float* const* w1,
                                            const float* const* b1, const float* const* w2, const float* const* b2, float* m,
                                            int32_t* argmax, float* const* a, float* const* b, void* workspace,
                                            void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs xp, w1p, w2p, b1p, b2p;
    LevelMutPtrs ap, bp;
    ARFE_CHECK_ARG(fill_ptrs(xp, x, lv.n) && fill_ptrs(w1p, w1, lv.n) && fill_ptrs(w2p, w2, lv.n) &&
                       fill_ptrs(b1p, b1, lv.n) && fill_ptrs(b2p, b2, lv.n) && fill_ptrs(ap, a, lv.n) &&
                       fill_ptrs(bp, b, lv.n) && workspace && m,
                   ARFE_ERR_ARG, "fpn_gather_gate_fwd: null pointer");
    TileTable tt = make_tile_table(lv);
    const int64_t pre_total = tt.px_off[lv.n];
    float* pre = (float*)workspace;
    float* Gbuf = (float*)((char*)workspace + arfe_fpn_gatemap_workspace_bytes(pyr));
    hipStream_t st = as_stream(stream);
    const int32_t done = fpn_gather_gate_v2(lv, x, w1, w2, m, argmax, Gbuf, pre, st);
    if (done < 0) return done;
    if (!done) {
        ARFE_CHECK_ARG(!lv.half, ARFE_ERR_ARG, "fpn_gather_gate_fwd: this pyramid has no half-storage kernels (arfe_fpn_half_storage_supported)");
        rc = arfe_fpn_gather_fwd(pyr, x, m, argmax, stream);
        if (rc) return rc;
        return arfe_fpn_gatemap_fwd(pyr, x, w1, b1, w2, b2, a, b, workspace, stream);
    }
    hipLaunchKernelGGL(fpn_gate_finish_kernel, dim3((unsigned)cdiv64(pre_total, 256)), dim3(256), 0, st, lv, tt, pre,
                       pre_total, b1p, b2p, ap, bp);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_fpn_scatter_fwd(const arfe_pyramid_t* pyr, const float* const* x, const float* bsf,
                                        const float* const* a, const float* const* b, float* const* out, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs xp, ap, bp;
    LevelMutPtrs op;
    ARFE_CHECK_ARG(fill_ptrs(xp, x, lv.n) && fill_ptrs(ap, a, lv.n) && fill_ptrs(bp, b, lv.n) &&
                       fill_ptrs(op, out, lv.n) && bsf,
                   ARFE_ERR_ARG, "fpn_scatter_fwd: null pointer");
    {
        const int32_t done = fpn_scatter_fwd_v2(lv, x, bsf, a, b, out, as_stream(stream));      // round 3 (fpn_stream.hip)
        if (done) return done < 0 ? done : ARFE_OK;
    }
    ARFE_CHECK_ARG(!lv.half, ARFE_ERR_ARG, "fpn_scatter_fwd: this pyramid has no half-storage kernels (arfe_fpn_half_storage_supported)");
    BlockQuadTable bt;
    int nblk = 0;
    for (int i = 0; i <= NL_; ++i) {
        bt.first_blk[i] = nblk;
        if (i < NL_) {
            bt.wq[i] = i < lv.n ? cdiv(lv.W[i], 4) : 0;
            bt.quads[i] = i < lv.n ? (int64_t)lv.B * lv.H[i] * bt.wq[i] : 0;
            ARFE_CHECK_ARG(bt.quads[i] < (int64_t)1 << 31, ARFE_ERR_ARG, "fpn_scatter_fwd: level %d too large", i);
            bt.nwork[i] = (int)cdiv64(bt.quads[i], 64);
            nblk += 8 * cdiv(bt.nwork[i], 8);
        }
    }
    hipLaunchKernelGGL(fpn_scatter_fwd_kernel, dim3(nblk, cdiv(lv.C, 4 * SC_CG * SC_NB)), dim3(256), 0, as_stream(stream), lv,
                       bt, xp, bsf, ap, bp, op);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_fpn_scatter_bwd(const arfe_pyramid_t* pyr, const float* const* gout, const float* bsf,
                                        const float* const* a, const float* const* b, float* const* dpre_a,
                                        float* const* dpre_b, float* dbsf, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs gp, ap, bp;
    LevelMutPtrs dap, dbp;
    ARFE_CHECK_ARG(fill_ptrs(gp, gout, lv.n) && fill_ptrs(ap, a, lv.n) && fill_ptrs(bp, b, lv.n) &&
                       fill_ptrs(dap, dpre_a, lv.n) && fill_ptrs(dbp, dpre_b, lv.n) && bsf && dbsf,
                   ARFE_ERR_ARG, "fpn_scatter_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    {   // single-pass form: the r finer levels are exactly 2^(r-i) times the refine level, r <= 2 (see the kernel)
        bool fused = lv.r <= 2 && lv.n <= ARFE_MAX_LEVELS;
        for (int i = 0; i < lv.r; ++i) fused = fused && lv.sh[i] == lv.r - i;
        static const char* sw = getenv("ARFE_FPN_BWD_FUSED");
        if (sw && sw[0] == '0') fused = false;
        if (fused) {
            ZeroList z;
            for (int i = 0; i < lv.n; ++i) z.add(dap.p[i], (int64_t)lv.B * lv.H[i] * lv.W[i]);
            rc = zero_buffers(z, st);
            if (rc) return rc;
            const int npx = lv.H[lv.r] * lv.W[lv.r];
            const int nchunk = lv.C >= 128 ? 4 : (lv.C >= 32 ? 2 : 1);
            dim3 grid((unsigned)(cdiv(npx, 64) * lv.B), nchunk);
#define ARFE_SBF(NF)                                                                                                            \
    if (lv.half) hipLaunchKernelGGL((fpn_scatter_bwd_fused_kernel<NF, _Float16>), grid, dim3(256), 0, st, lv, gp, bsf, ap, bp, dap, dbsf); \
    else hipLaunchKernelGGL((fpn_scatter_bwd_fused_kernel<NF, float>), grid, dim3(256), 0, st, lv, gp, bsf, ap, bp, dap, dbsf);
            if (lv.r == 2) { ARFE_SBF(2) }
            else if (lv.r == 1) { ARFE_SBF(1) }
            else { ARFE_SBF(0) }
#undef ARFE_SBF
            ARFE_CHECK_LAUNCH();
            TileTable tt = make_tile_table(lv);
            hipLaunchKernelGGL(fpn_scatter_bwd_maps_finish_kernel, dim3((unsigned)cdiv64(tt.px_off[lv.n], 256)), dim3(256), 0, st,
                               lv, tt, ap, bp, dap, dbp);
            ARFE_CHECK_LAUNCH();
            return ARFE_OK;
        }
    }
    ARFE_CHECK_ARG(!lv.half, ARFE_ERR_ARG, "fpn_scatter_bwd: this pyramid has no half-storage kernels (arfe_fpn_half_storage_supported)");
    PixQuadTable pt = make_pixquad_table(lv);
    const int nchunk = lv.C >= 64 ? 4 : 1;
    if (nchunk > 1) {
        ZeroList z;
        for (int i = 0; i < lv.n; ++i) z.add(dap.p[i], (int64_t)lv.B * lv.H[i] * lv.W[i]);
        rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(fpn_scatter_bwd_maps_kernel, dim3((unsigned)cdiv64(pt.first[lv.n], 64), nchunk), dim3(256), 0, st, lv,
                       pt, gp, bsf, ap, bp, dap, dbp);
    ARFE_CHECK_LAUNCH();
    if (nchunk > 1) {
        TileTable tt = make_tile_table(lv);
        hipLaunchKernelGGL(fpn_scatter_bwd_maps_finish_kernel, dim3((unsigned)cdiv64(tt.px_off[lv.n], 256)), dim3(256), 0, st,
                           lv, tt, ap, bp, dap, dbp);
        ARFE_CHECK_LAUNCH();
    }
    const int64_t total = (int64_t)lv.B * cdiv(lv.C, DB_CG) * lv.H[lv.r] * lv.W[lv.r];
    hipLaunchKernelGGL(fpn_scatter_bwd_dbsf_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, lv, gp, ap, bp,
                       dbsf);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_fpn_input_bwd(const arfe_pyramid_t* pyr, const float* const* gout, const float* const* dpre_a,
                                      const float* const* dpre_b, const float* const* w1, const float* const* w2,
                                      const float* dm, const int32_t* argmax, float* const* dx, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs gp, dap, dbp, w1p, w2p;
    LevelMutPtrs dxp;
    ARFE_CHECK_ARG(fill_ptrs(gp, gout, lv.n) && fill_ptrs(dap, dpre_a, lv.n) && fill_ptrs(dbp, dpre_b, lv.n) &&
                       fill_ptrs(w1p, w1, lv.n) && fill_ptrs(w2p, w2, lv.n) && fill_ptrs(dxp, dx, lv.n) && dm &&
                       (argmax || lv.r == 0),
                   ARFE_ERR_ARG, "fpn_input_bwd: null pointer");
    // round 3: the item-map kernel (fpn_stream.hip) takes every level it supports in ONE launch; what is left (ragged finer
    // levels, ratios above 4) goes through the first-generation kernels below
    const int32_t done = fpn_input_bwd_v2(lv, gout, dpre_a, dpre_b, w1, w2, dm, argmax, dx, as_stream(stream));
    if (done < 0) return done;
    ARFE_CHECK_ARG(!lv.half || done == (1 << lv.n) - 1, ARFE_ERR_ARG,
                   "fpn_input_bwd: this pyramid has no half-storage kernels (arfe_fpn_half_storage_supported)");
    // levels with an exact power-of-two ratio to the refine level take the shift-only kernel, the rest (e.g. the
    // 13x21 top level of the 1333x800 pyramid, 25 -> 13 by a ceil) the generic one: two launches over disjoint levels
    for (int pass = 0; pass < 3; ++pass) {
        BlockQuadTable bt;
        int nblk = 0;
        for (int i = 0; i <= NL_; ++i) {
            bt.first_blk[i] = nblk;
            if (i < NL_) {
                const int shi = lv.sh[i];
                // shift-only kernel: finer-or-equal levels below the refine level, coarser-or-equal ones above it
                const bool pow2 = shi != SH_GENERAL && (i < lv.r ? shi >= 0 : shi <= 0);
                const int mode = (i >= lv.n || ((done >> i) & 1)) ? -1 : (pow2 ? 0 : (i >= lv.r ? 1 : 2));
                bt.wq[i] = i < lv.n ? cdiv(lv.W[i], 4) : 0;
                bt.quads[i] = mode == pass ? (int64_t)lv.B * lv.H[i] * bt.wq[i] : 0;
                ARFE_CHECK_ARG(bt.quads[i] < (int64_t)1 << 31, ARFE_ERR_ARG, "fpn_input_bwd: level %d too large", i);
                bt.nwork[i] = (int)cdiv64(bt.quads[i], 64);
                nblk += (int)cdiv64(bt.quads[i], 64);
            }
        }
        if (nblk == 0) continue;
        const dim3 grid(nblk, cdiv(lv.C, 4 * (pass == 0 ? IB_CG * IB_NB : 1)));
        if (pass == 0)
            hipLaunchKernelGGL((fpn_input_bwd_kernel<0>), grid, dim3(256), 0, as_stream(stream), lv, bt, gp, dap, dbp, w1p,
                               w2p, dm, argmax, dxp);
        else if (pass == 1)
            hipLaunchKernelGGL((fpn_input_bwd_kernel<1>), grid, dim3(256), 0, as_stream(stream), lv, bt, gp, dap, dbp, w1p,
                               w2p, dm, argmax, dxp);
        else
            hipLaunchKernelGGL((fpn_input_bwd_kernel<2>), grid, dim3(256), 0, as_stream(stream), lv, bt, gp, dap, dbp, w1p,
                               w2p, dm, argmax, dxp);
        ARFE_CHECK_LAUNCH();
    }
    return ARFE_OK;
}

extern "C" int32_t arfe_fpn_gate_wgrad(const arfe_pyramid_t* pyr, const float* const* x, const float* const* dpre_a,
                                       const float* const* dpre_b, float* const* dw1, float* const* db1,
                                       float* const* dw2, float* const* db2, void* stream) {
    int32_t rc = check_pyramid(pyr);
    if (rc) return rc;
    Levels lv = to_levels(pyr);
    LevelPtrs xp, dap, dbp;
    LevelMutPtrs w1p, b1p, w2p, b2p;
    ARFE_CHECK_ARG(fill_ptrs(xp, x, lv.n) && fill_ptrs(dap, dpre_a, lv.n) && fill_ptrs(dbp, dpre_b, lv.n) &&
                       fill_ptrs(w1p, dw1, lv.n) && fill_ptrs(b1p, db1, lv.n) && fill_ptrs(w2p, dw2, lv.n) &&
                       fill_ptrs(b2p, db2, lv.n),
                   ARFE_ERR_ARG, "fpn_gate_wgrad: null pointer");
    hipLaunchKernelGGL(fpn_gate_wgrad_zero_kernel, dim3(cdiv(lv.n * 9 * lv.C, 256)), dim3(256), 0, as_stream(stream), lv.C,
                       lv.n, w1p, b1p, w2p, b2p);
    ARFE_CHECK_LAUNCH();
    TileTable tt = make_tile_table(lv, GW_PX * GW_STAGES);
    if (lv.half)
        hipLaunchKernelGGL(fpn_gate_wgrad_mfma_kernel<_Float16>, dim3(tt.first_tile[lv.n], cdiv(lv.C, GW_CH)), dim3(256), 0,
                           as_stream(stream), lv, tt, xp, dap, dbp, w1p, b1p, w2p, b2p);
    else
        hipLaunchKernelGGL(fpn_gate_wgrad_mfma_kernel<float>, dim3(tt.first_tile[lv.n], cdiv(lv.C, GW_CH)), dim3(256), 0,
                           as_stream(stream), lv, tt, xp, dap, dbp, w1p, b1p, w2p, b2p);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
