// AR-FPN streaming stages, second generation (round 3): the kernels of fpn.hip that were latency- or instruction-bound,
// rebuilt around ONE item -> pixel map and a channel loop whose per-channel cost is a handful of instructions:
//
//   item map   a wave owns 64 pixel QUADS (4 consecutive pixels of a row) of one level, enumerated (window row, quad
//              column, row inside the window) with the row fastest, so that the R = 2^sh rows of a pooling window sit on
//              ADJACENT LANES: the window's argmax / dm operands are one broadcast read per lane group instead of R reads
//              from R workgroups on R different XCDs, and the gather's window maximum is a DPP exchange.  One wave-load of
//              a channel still touches whole 128-byte lines (R segments of 64/R quads).
//   addressing every tensor is addressed as (wave-uniform channel base in SGPRs) + (loop-invariant 32-bit lane offset):
//              advancing a channel is scalar arithmetic, a level must stay below 2^31 bytes (checked on the host).
//   in flight  IB_U channels' reads are issued together ahead of any use (dx may not alias the inputs: __restrict__ locals).
//
// reference: mmdet/models/necks/wfpn_dual_spatial.py:99-137 (forward) and its autograd.
#include <type_traits>

#include "common.hpp"

namespace arfe {

constexpr int NLS = ARFE_MAX_LEVELS;
struct SLevelPtrs { const float* p[NLS]; };
struct SLevelMutPtrs { float* p[NLS]; };

// ---- item map -------------------------------------------------------------------------------------------------
struct ItemTable {
    int first_blk[NLS + 1];    // prefix sum of workgroups per level (a workgroup never straddles levels)
    int items[NLS];            // B * h * ceil(w/4) quads of the level, 0 if the level is not handled by this launch
    int wq[NLS];               // quads per row
    int mode[NLS];             // IB_SH2 .. IB_GEN
    int lr[NLS];               // log2 of the rows that share a lane group (0: lanes simply run along the flattened quads)
    int cgroups[NLS];          // channel groups of the level: workgroup = (item block, channel group), 4 waves x cpw channels
    int cpw[NLS];              // channels per wave
    int ry[NLS], rx[NLS];      // IB_GEN: largest destination range of a source pixel (rows, columns)
    int nwork[NLS];            // workgroups of the level that have work (first_blk advances by this, rounded up to 8)
};
enum { IB_SH2 = 0, IB_SH1 = 1, IB_SH0 = 2, IB_SHM1 = 3, IB_GEN = 4, IB_NONE = -1 };

struct Item {
    int bi, y, x0;
    bool active, full;
    int e1, e2, e3;            // clamped element offsets of a row-end quad
};
__device__ __forceinline__ Item item_of(int LR, int blk_first_item, int lane, int items, int wq, int h, int w) {
    int i = blk_first_item + lane;
    Item it;
    it.active = i < items;
    if (!it.active) i = blk_first_item;                    // idle lanes follow along on a valid item and never store
    const int r = i & ((1 << LR) - 1);
    const unsigned qg = (unsigned)i >> LR;
    const unsigned wyg = qg / (unsigned)wq;
    it.x0 = (int)(qg - wyg * (unsigned)wq) * 4;
    const unsigned yg = (wyg << LR) + r;                   // row index over B * h (h is a multiple of the group height)
    it.bi = (int)(yg / (unsigned)h);
    it.y = (int)(yg - (unsigned)it.bi * (unsigned)h);
    it.full = it.x0 + 3 < w;
    it.e1 = it.x0 + 1 < w ? 1 : 0;
    it.e2 = it.x0 + 2 < w ? 2 : it.e1;
    it.e3 = it.x0 + 3 < w ? 3 : it.e2;
    return it;
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4)))* const_fp;    // wave-uniform data: scalar loads

// WIDE (level-uniform: w % 4 == 0, every quad whole): one 16-byte access, no branch inside the channel loop.  Otherwise
// every lane takes the element-wise form with clamped offsets (rare widths; the loop body stays one basic block).
template <bool WIDE>
__device__ __forceinline__ float4_u load_quad(const float* __restrict__ p, const Item& it) {
    if constexpr (WIDE) return *reinterpret_cast<const float4_u*>(p);
    float4_u v = {p[0], p[it.e1], p[it.e2], p[it.e3]};
    return v;
}
template <bool WIDE>
__device__ __forceinline__ void store_quad(float* __restrict__ p, const Item& it, int w, const float4_u v) {
    if constexpr (WIDE) { *reinterpret_cast<float4_u*>(p) = v; return; }
    p[0] = v[0];
    if (it.x0 + 1 < w) p[1] = v[1];
    if (it.x0 + 2 < w) p[2] = v[2];
    if (it.x0 + 3 < w) p[3] = v[3];
}
// half storage (arfe_pyramid_t::storage == ARFE_STORAGE_F16): 8-byte accesses, converted on the way (round to nearest even on
// the way out, as Tensor.half())
typedef _Float16 half4_u __attribute__((ext_vector_type(4), aligned(2)));
template <bool WIDE>
__device__ __forceinline__ float4_u load_quad(const _Float16* __restrict__ p, const Item& it) {
    if constexpr (WIDE) {
        const half4_u h = *reinterpret_cast<const half4_u*>(p);
        return float4_u{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
    float4_u v = {(float)p[0], (float)p[it.e1], (float)p[it.e2], (float)p[it.e3]};
    return v;
}
template <bool WIDE>
__device__ __forceinline__ void store_quad(_Float16* __restrict__ p, const Item& it, int w, const float4_u v) {
    if constexpr (WIDE) {
        const half4_u h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<half4_u*>(p) = h;
        return;
    }
    p[0] = (_Float16)v[0];
    if (it.x0 + 1 < w) p[1] = (_Float16)v[1];
    if (it.x0 + 2 < w) p[2] = (_Float16)v[2];
    if (it.x0 + 3 < w) p[3] = (_Float16)v[3];
}

// =========================================================================================================
// backward C, v2:  dx = g + conv3x3^T(dpre_a; w1) + conv3x3^T(dpre_b; w2) + gather^T(dm / n)
//   workgroup = 64 items x 4 waves; wave cw of block row blockIdx.y owns IB_CPW consecutive channels.
//   Per (item, channel): one 16-byte read of g, the pooling-term operands (one broadcast dword pair per window), 36 packed
//   FMAs with the two gates' weights as a scalar register pair, one 16-byte store.
// =========================================================================================================
#ifndef ARFE_IB_CPW
#define ARFE_IB_CPW 16
#endif
#ifndef ARFE_IB_U
#define ARFE_IB_U 4
#endif
#ifndef ARFE_IB_WAVES
#define ARFE_IB_WAVES 4
#endif
constexpr int IB_CPW = ARFE_IB_CPW;   // channels per wave on the large levels (the small coarse levels take IB_CPW_SMALL: their
#ifndef ARFE_IB_CPW_SMALL
#define ARFE_IB_CPW_SMALL 1
#endif
constexpr int IB_CPW_SMALL = ARFE_IB_CPW_SMALL;   // per-channel work is a latency chain, so they get many short waves, dispatched FIRST)
constexpr int IB_U = ARFE_IB_U;       // channels whose reads are in flight together
#ifndef ARFE_IB_ROWS_SH2
#define ARFE_IB_ROWS_SH2 4
#endif
#ifndef ARFE_IB_ROWS_SH1
#define ARFE_IB_ROWS_SH1 2
#endif
constexpr int IB_ROWS_SH2 = ARFE_IB_ROWS_SH2, IB_ROWS_SH1 = ARFE_IB_ROWS_SH1;   // rows per lane on WIDE levels (1, 2 or 4 / 1 or 2)

// ROWS: a lane owns the quad's column in ROWS consecutive rows - the rows of ONE pooling window (4 on a level 4x finer
// than the refine level, 2 on a 2x finer one).  The window's (argmax, dm) pair is then read once for ROWS quads and so are
// the 18 scalar weights: 2.5 / 4 vector-memory instructions per (quad, channel) instead of 4 / 6 (the streaming kernels run
// at ~35 cycles per vector-memory instruction and CU, see fpn_scatter_fwd2).  The items of such a level are (image, row
// group, quad column); the channel-independent neighbourhood is the (ROWS + 2) x 6 patch of the two gradient maps.
template <int MODE, bool WIDE, int ROWS, class T>
__device__ __forceinline__ void input_bwd2_level(const Levels& lv, const ItemTable& tb, int lvl, int blk, int lane, int cw,
                                                 const T* __restrict__ g, const float* __restrict__ da,
                                                 const float* __restrict__ db, const float* w1, const float* w2,
                                                 const float* __restrict__ dm, const int32_t* __restrict__ argmax,
                                                 T* __restrict__ dx) {
    static_assert(ROWS == 1 || MODE == IB_SH2 || MODE == IB_SH1, "row groups are pooling windows");
    constexpr int SH = MODE == IB_SH2 ? 2 : (MODE == IB_SH1 ? 1 : 0);
    const int h = lv.H[lvl], w = lv.W[lvl], wq = tb.wq[lvl];
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const int cgroups = tb.cgroups[lvl], cpw = tb.cpw[lvl];
    const int iblk = blk / cgroups, cgrp = blk - iblk * cgroups;
    // item rows are row GROUPS: h / ROWS of them per image (h is a multiple of ROWS on these levels)
    Item it = item_of(0, iblk * 64, lane, tb.items[lvl], wq, h / ROWS, w);
    it.y *= ROWS;
    const int C = lv.C;
    const unsigned plane = (unsigned)(h * w), plane_r = (unsigned)(hr * wr);

    // channel-independent (ROWS + 2) x 6 patch of the two gradient maps as (a, b) pairs: nb[j] = map row y - 1 + j;
    // clamped addresses + selects
    f32x2_t nb[ROWS + 2][6];
    {
        const float* pa = da + (size_t)it.bi * plane;
        const float* pb = db + (size_t)it.bi * plane;
#pragma unroll
        for (int j = 0; j < ROWS + 2; ++j) {
            const int yy = it.y - 1 + j;
            const bool rok = yy >= 0 && yy < h;
            const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int xx = it.x0 - 1 + t;
                const bool ok = rok && xx >= 0 && xx < w;
                const int xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                const float ta = pa[yc * w + xc], tb2 = pb[yc * w + xc];
                nb[j][t] = f32x2_t{ok ? ta : 0.f, ok ? tb2 : 0.f};
            }
        }
    }
    const float inv_n = 1.0f / (float)lv.n;
    const unsigned off = (unsigned)it.bi * (unsigned)C * plane + (unsigned)(it.y * w + it.x0);      // g / dx, + c * plane
    const int f0 = it.y * w + it.x0;
    // pooling-term operand offsets (+ c * plane_r)
    unsigned poff = (unsigned)it.bi * (unsigned)C * plane_r;
    if constexpr (MODE == IB_SH2 || MODE == IB_SH1) poff += (unsigned)((it.y >> SH) * wr + (it.x0 >> SH));
    else if constexpr (MODE == IB_SH0) poff += (unsigned)(it.y * wr + it.x0);
    const unsigned poffB = poff + (it.x0 + 2 < w ? 1u : 0u);       // second window of a quad (ratio 2); a row-end quad has one
    const int32_t* __restrict__ am = argmax ? argmax + (size_t)lvl * lv.B * C * plane_r : nullptr;
    const const_fp k1g = (const_fp)(uintptr_t)w1, k2g = (const_fp)(uintptr_t)w2;
    const int c_wave = (cgrp * 4 + cw) * cpw;

    // generic coarser level: destination ranges of the quad's pixels in the refine map (channel independent)
    int gy0 = 0, gy1 = 0, gx0[4] = {0, 0, 0, 0}, gx1[4] = {0, 0, 0, 0};
    if constexpr (MODE == IB_GEN) {
        const float ys = nearest_scale(h, hr), xs = nearest_scale(w, wr);
        gy0 = nearest_dst_begin(it.y, h, hr, ys); gy1 = nearest_dst_begin(it.y + 1, h, hr, ys);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = it.x0 + e < w ? it.x0 + e : w - 1;
            gx0[e] = nearest_dst_begin(xx, w, wr, xs); gx1[e] = nearest_dst_begin(xx + 1, w, wr, xs);
        }
    }

    // one output row: (acc_a, acc_b) += (w1, w2)[tap] * (d_a, d_b)[shifted]; the accumulator pair starts as (g, pooling term)
    auto conv_store = [&](int c, int r, const float4_u gq, const float4_u pool) {
        const const_fp k1 = k1g + c * 9, k2 = k2g + c * 9;
        f32x2_t r2[4] = {f32x2_t{gq[0], pool[0]}, f32x2_t{gq[1], pool[1]}, f32x2_t{gq[2], pool[2]}, f32x2_t{gq[3], pool[3]}};
#ifndef ARFE_IB_PROBE
#define ARFE_IB_PROBE 0
#endif
#pragma unroll
        for (int ky = 0; ky < (ARFE_IB_PROBE == 1 ? 0 : 3); ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#if ARFE_IB_PROBE == 2
                const f32x2_t us = {0.25f * (float)(ky + c), 0.5f * (float)(kx - c)};
#else
                const f32x2_t us = {k1[ky * 3 + kx], k2[ky * 3 + kx]};
#endif
                // dx[y, x] += W[c, ky, kx] * d[y - (ky - 1), x - (kx - 1)]: patch row r + 2 - ky, column e + 2 - kx
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] += us * nb[r + 2 - ky][e + 2 - kx];
            }
        const float4_u o = {r2[0][0] + r2[0][1], r2[1][0] + r2[1][1], r2[2][0] + r2[2][1], r2[3][0] + r2[3][1]};
        // (idle lanes mirror the block's first item and channels past C mirror channel C-1: they store the very same values
        // to the very same addresses, which keeps the loop body free of branches)
        store_quad<WIDE>(dx + (size_t)c * plane + off + (unsigned)(r * w), it, w, o);
    };

    if constexpr (MODE == IB_SH2 || MODE == IB_SH1 || MODE == IB_SH0) {
        constexpr int U = ROWS == 4 ? (IB_U >= 2 ? IB_U / 2 : 1) : IB_U;        // channels in flight: ~4 KB of g per wave
#pragma unroll 1
        for (int cb = 0; cb < cpw; cb += U) {
            const int c0 = c_wave + cb;
            if (c0 >= C) break;
            float4_u gq[U][ROWS];
            float4_u pq[U];                     // SH0: dm quad;  SH2: (dm, -, -, -);  SH1: (dmA, dmB, -, -)
            int ia[U], ib[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u < C ? c0 + u : C - 1;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) gq[u][r] = load_quad<WIDE>(g + (size_t)c * plane + off + (unsigned)(r * w), it);
#if ARFE_IB_PROBE == 3
                ia[u] = c; ib[u] = c + 1; pq[u] = float4_u{1.f, 2.f, 3.f, 4.f};
#else
                if constexpr (MODE == IB_SH2) {
                    ia[u] = am[(size_t)c * plane_r + poff];
                    pq[u][0] = dm[(size_t)c * plane_r + poff];
                } else if constexpr (MODE == IB_SH1) {
                    ia[u] = am[(size_t)c * plane_r + poff]; ib[u] = am[(size_t)c * plane_r + poffB];
                    pq[u][0] = dm[(size_t)c * plane_r + poff]; pq[u][1] = dm[(size_t)c * plane_r + poffB];
                } else {
                    pq[u] = load_quad<WIDE>(dm + (size_t)c * plane_r + poff, it);
                }
#endif
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u < C ? c0 + u : C - 1;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    float4_u pool;
                    if constexpr (MODE == IB_SH2) {
                        const float d = pq[u][0] * inv_n;
                        const int t = ia[u] - (f0 + r * w);
                        pool = float4_u{t == 0 ? d : 0.f, t == 1 ? d : 0.f, t == 2 ? d : 0.f, t == 3 ? d : 0.f};
                    } else if constexpr (MODE == IB_SH1) {
                        const float dA = pq[u][0] * inv_n, dB = pq[u][1] * inv_n;
                        const int tA = ia[u] - (f0 + r * w), tB = ib[u] - (f0 + r * w);
                        pool = float4_u{tA == 0 ? dA : 0.f, tA == 1 ? dA : 0.f, tB == 2 ? dB : 0.f, tB == 3 ? dB : 0.f};
                    } else {
                        pool = float4_u{pq[u][0] * inv_n, pq[u][1] * inv_n, pq[u][2] * inv_n, pq[u][3] * inv_n};
                    }
                    conv_store(c, r, gq[u][r], pool);
                }
            }
        }
    } else {
        // coarser levels (small maps): nearest^T sums a block of dm per pixel.  The block's extent differs from lane to
        // lane (3 or 4 rows / columns at 13 <- 50); the loops run the level's LARGEST extent with clamped addresses and
        // selects, so all reads of a channel are independent and issue together.
        const int ry = MODE == IB_SHM1 ? 2 : tb.ry[lvl], rx = MODE == IB_SHM1 ? 2 : tb.rx[lvl];
        if constexpr (MODE == IB_SHM1) {
            gy0 = it.y * 2; gy1 = gy0 + 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xx = it.x0 + e < w ? it.x0 + e : w - 1;
                gx0[e] = xx * 2; gx1[e] = gx0[e] + 2;
            }
        }
        // row / column offsets and validity of the (at most ry x rx) block, per pixel of the quad: channel independent
        auto sum_block = [&](const float* __restrict__ dmp, auto RY, auto RX) {
            float4_u pool = {0.f, 0.f, 0.f, 0.f};
            const int nry = RY.value > 0 ? RY.value : ry, nrx = RX.value > 0 ? RX.value : rx;
#pragma unroll
            for (int dy = 0; dy < nry; ++dy) {
                const int py = gy0 + dy;
                const bool yok = py < gy1;
                const float* row = dmp + (size_t)(yok ? py : gy0) * wr;
#pragma unroll
                for (int dxx = 0; dxx < nrx; ++dxx) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int px = gx0[e] + dxx;
                        const bool ok = yok && px < gx1[e];
                        const float v = row[ok ? px : gx0[e]];
                        pool[e] += ok ? v * inv_n : 0.f;
                    }
                }
            }
            return pool;
        };
        for (int cb = 0; cb < cpw; ++cb) {
            const int c = c_wave + cb;
            if (c >= C) break;
            const float4_u gq = load_quad<WIDE>(g + (size_t)c * plane + off, it);
            const float* __restrict__ dmp = dm + (size_t)c * plane_r + poff;
            float4_u pool;
            if constexpr (MODE == IB_SHM1) pool = sum_block(dmp, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
            else pool = sum_block(dmp, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            conv_store(c, 0, gq, pool);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256, ARFE_IB_WAVES) void fpn_input_bwd2_kernel(Levels lv, ItemTable tb, SLevelPtrs gout, SLevelPtrs dpa,
                                                             SLevelPtrs dpb, SLevelPtrs w1, SLevelPtrs w2,
                                                             const float* __restrict__ dm, const int32_t* __restrict__ argmax,
                                                             SLevelMutPtrs dx) {
    const int lane = threadIdx.x & 63, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the grid runs coarsest level first (the small levels' waves are latency chains and must not form the tail); every
    // level's workgroups start at a multiple of 8
    int lvl = lv.n - 1;
    while (lvl > 0 && (int)blockIdx.x >= tb.first_blk[lvl - 1]) --lvl;
    // XCD-aware order inside a level: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so local
    // workgroup j takes work item (j % 8) * ceil(n / 8) + j / 8: every XCD walks ONE contiguous range of the level.
    // Neighbouring item blocks share the argmax / dm lines of their pooling windows (4 rows of level 0 = 5 blocks apart)
    // and the gate maps' halo rows; dealt round-robin, every such line was fetched by 5 - 8 different L2s (PMC: 290 MB
    // fetched for 210).
    const int j = (int)blockIdx.x - tb.first_blk[lvl];
    const int nwork = tb.nwork[lvl];
    const int blk = (j & 7) * ((nwork + 7) >> 3) + (j >> 3);
    if (blk >= nwork) return;
    const T* g = (const T*)gout.p[lvl];
    const float* da = dpa.p[lvl];
    const float* db = dpb.p[lvl];
    T* d = (T*)dx.p[lvl];
    const bool wide = (lv.W[lvl] & 3) == 0;
#define ARFE_IB2(M, R)                                                                                                     \
    if (wide) input_bwd2_level<M, true, R, T>(lv, tb, lvl, blk, lane, cw, g, da, db, w1.p[lvl], w2.p[lvl], dm, argmax, d);  \
    else input_bwd2_level<M, false, 1, T>(lv, tb, lvl, blk, lane, cw, g, da, db, w1.p[lvl], w2.p[lvl], dm, argmax, d);
    switch (tb.mode[lvl]) {
        case IB_SH2: ARFE_IB2(IB_SH2, IB_ROWS_SH2) break;
        case IB_SH1: ARFE_IB2(IB_SH1, IB_ROWS_SH1) break;
        case IB_SH0: ARFE_IB2(IB_SH0, 1) break;
        case IB_SHM1: input_bwd2_level<IB_SHM1, false, 1, T>(lv, tb, lvl, blk, lane, cw, g, da, db, w1.p[lvl], w2.p[lvl], dm, argmax, d); break;
        default: input_bwd2_level<IB_GEN, false, 1, T>(lv, tb, lvl, blk, lane, cw, g, da, db, w1.p[lvl], w2.p[lvl], dm, argmax, d); break;
    }
#undef ARFE_IB2
}

// =========================================================================================================
// scatter + gate + residual, v2:  out = x + nearest(bsf) * (a + b)
//   Same item map as above.  What changes against the first-generation kernel is the number of vector-memory instructions
//   per (quad, channel): the four pixels of a quad map to ONE bsf pixel on a level 4x finer than the refine level, to TWO on
//   a 2x finer one and to four consecutive ones on a level of the same size - one dword, one 8-byte or one 16-byte read
//   instead of four dword reads (PMC: both streaming kernels ran at ~35 cycles per vector-memory instruction and CU,
//   whatever the occupancy or the number of reads in flight: bound by instruction count, not by bytes).
// =========================================================================================================
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));
constexpr int SC2_CPW = 16, SC2_U = 4;

template <int MODE, bool WIDE, class T>
__device__ __forceinline__ void scatter2_level(const Levels& lv, const ItemTable& tb, int lvl, int blk, int lane, int cw,
                                               const T* __restrict__ x, const float* __restrict__ bsf,
                                               const float* __restrict__ a, const float* __restrict__ b,
                                               T* __restrict__ out) {
    const int h = lv.H[lvl], w = lv.W[lvl], wq = tb.wq[lvl];
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const int cgroups = tb.cgroups[lvl], cpw = tb.cpw[lvl];
    const int iblk = blk / cgroups, cgrp = blk - iblk * cgroups;
    const Item it = item_of(0, iblk * 64, lane, tb.items[lvl], wq, h, w);
    const int C = lv.C;
    const unsigned plane = (unsigned)(h * w), plane_r = (unsigned)(hr * wr);
    float4_u s4;
    {
        const size_t moff = (size_t)it.bi * plane + (size_t)(it.y * w + it.x0);
        const float4_u av = load_quad<WIDE>(a + moff, it), bv = load_quad<WIDE>(b + moff, it);
        s4 = av + bv;
    }
    const unsigned off = (unsigned)it.bi * (unsigned)C * plane + (unsigned)(it.y * w + it.x0);
    unsigned boff = (unsigned)it.bi * (unsigned)C * plane_r;
    int sx[4] = {0, 0, 0, 0};
    if constexpr (MODE == IB_SH2) boff += (unsigned)((it.y >> 2) * wr + (it.x0 >> 2));
    else if constexpr (MODE == IB_SH1) boff += (unsigned)((it.y >> 1) * wr + (it.x0 >> 1));
    else if constexpr (MODE == IB_SH0) boff += (unsigned)(it.y * wr + it.x0);
    else {
        const float sy_scale = nearest_scale(hr, h), sx_scale = nearest_scale(wr, w);
        boff += (unsigned)(nearest_src(it.y, hr, sy_scale) * wr);
        sx[0] = nearest_src(it.x0, wr, sx_scale); sx[1] = nearest_src(it.x0 + it.e1, wr, sx_scale);
        sx[2] = nearest_src(it.x0 + it.e2, wr, sx_scale); sx[3] = nearest_src(it.x0 + it.e3, wr, sx_scale);
    }
    const bool two = it.x0 + 2 < w;                  // ratio 2: a row-end quad may cover one bsf pixel only
    const int c_wave = (cgrp * 4 + cw) * cpw;
#pragma unroll 1
    for (int cb = 0; cb < cpw; cb += SC2_U) {
        const int c0 = c_wave + cb;
        if (c0 >= C) break;
        float4_u xv[SC2_U], u[SC2_U];
#pragma unroll
        for (int j = 0; j < SC2_U; ++j) {
            const int c = c0 + j < C ? c0 + j : C - 1;
            xv[j] = load_quad<WIDE>(x + (size_t)c * plane + off, it);
            const float* bp = bsf + (size_t)c * plane_r + boff;
            if constexpr (MODE == IB_SH2) {
                const float v = bp[0];
                u[j] = float4_u{v, v, v, v};
            } else if constexpr (MODE == IB_SH1) {
                if constexpr (WIDE) {
                    const float2_u v = *reinterpret_cast<const float2_u*>(bp);
                    u[j] = float4_u{v[0], v[0], v[1], v[1]};
                } else {
                    const float v0 = bp[0], v1 = bp[two ? 1 : 0];
                    u[j] = float4_u{v0, v0, v1, v1};
                }
            } else if constexpr (MODE == IB_SH0) {
                u[j] = load_quad<WIDE>(bp, it);
            } else {
                u[j] = float4_u{bp[sx[0]], bp[sx[1]], bp[sx[2]], bp[sx[3]]};
            }
        }
#pragma unroll
        for (int j = 0; j < SC2_U; ++j) {
            const int c = c0 + j < C ? c0 + j : C - 1;       // (channels past C mirror C-1, idle lanes the first item: same stores)
            float4_u o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = xv[j][e] + u[j][e] * s4[e];
            store_quad<WIDE>(out + (size_t)c * plane + off, it, w, o);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void fpn_scatter_fwd2_kernel(Levels lv, ItemTable tb, SLevelPtrs x, const float* __restrict__ bsf,
                                                               SLevelPtrs a, SLevelPtrs b, SLevelMutPtrs out) {
    const int lane = threadIdx.x & 63, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lvl = lv.n - 1;
    while (lvl > 0 && (int)blockIdx.x >= tb.first_blk[lvl - 1]) --lvl;
    const int j = (int)blockIdx.x - tb.first_blk[lvl];
    const int nwork = tb.nwork[lvl];
    const int blk = (j & 7) * ((nwork + 7) >> 3) + (j >> 3);          // XCD-aware order, see fpn_input_bwd2_kernel
    if (blk >= nwork) return;
    const bool wide = (lv.W[lvl] & 3) == 0;
#define ARFE_SC2(M)                                                                                                 \
    if (wide) scatter2_level<M, true, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], bsf, a.p[lvl], b.p[lvl], (T*)out.p[lvl]);   \
    else scatter2_level<M, false, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], bsf, a.p[lvl], b.p[lvl], (T*)out.p[lvl]);
    switch (tb.mode[lvl]) {
        case IB_SH2: ARFE_SC2(IB_SH2) break;
        case IB_SH1: ARFE_SC2(IB_SH1) break;
        case IB_SH0: ARFE_SC2(IB_SH0) break;
        default: scatter2_level<IB_GEN, false, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], bsf, a.p[lvl], b.p[lvl], (T*)out.p[lvl]); break;
    }
#undef ARFE_SC2
}

// =========================================================================================================
// gather + gate maps, fused (round 3): ONE read of the pyramid feeds
//   (a) the pooled maps of the levels finer than the refine level (adaptive max over 4x4 / 2x2 windows + argmax), and
//   (b) the 18 (gate, tap) product maps T[m][px] = sum_c W[m][c] x[c][px] of BOTH 3x3 gate convs of every level
// (wfpn_dual_spatial.py:102-113 and :119-121).  The first-generation path read the pyramid twice (fpn_gather_fwd and
// fpn_gate_mfma: 2 x 183 MB at the bench size).
//   lanes     the item map with the rows of a pooling window on ADJACENT lanes (LR = log2 of the window height): the window
//             maximum is an in-lane fold over the quad plus LR quad-permute exchanges, in the scan order of
//             F.adaptive_max_pool2d (first maximum wins, the last NaN wins).
//   gate conv 36 packed FMAs per (quad, channel): (acc_gate1, acc_gate2)[tap][pixel] += (w1, w2)[c][tap] * x[c][pixel] with
//             the weight pair as a scalar operand; 72 accumulators per lane.  The workgroup's 4 waves split the channels and
//             add their accumulators through LDS (two halves of 36) in a fixed order; T is finished (9-tap shift-add, bias,
//             ReLU, tanh) by fpn_gate_finish_kernel as before.
//   mean      m = (((((0 + G0) + G1) + ...) / n is a separate small kernel over the refine-sized maps (fpn_gather_mean_kernel):
//             the levels at or above the refine level enter it straight from x through the nearest index.
// =========================================================================================================
#ifndef ARFE_GG_U
#define ARFE_GG_U 1
#endif
#ifndef ARFE_GG_WAVES
#define ARFE_GG_WAVES 3
#endif
#ifndef ARFE_GG_NW
#define ARFE_GG_NW 4
#endif
constexpr int GG_NW = ARFE_GG_NW;       // waves of a workgroup = channel slices of the same 64 quads
constexpr int GG_U = ARFE_GG_U;
constexpr int GG_M = 18;

__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ int dpp_xor2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }

// (value, index) of two window parts, `mine` and the partner lane's; `later` = this lane's part comes later in scan order
__device__ __forceinline__ void pool_combine(float& v, int& i, float ov, int oi, bool later) {
    // sequential rule: a later element replaces the running maximum iff it is greater or a NaN
    const bool mine_wins = later ? ((v > ov) || (v != v)) : !((ov > v) || (ov != ov));
    v = mine_wins ? v : ov;
    i = mine_wins ? i : oi;
}

struct GGTable {
    int first_blk[NLS + 1];
    int nwork[NLS];
    int items[NLS], wq[NLS];
    int lr[NLS];               // log2 rows per lane group: 2 (4x4 windows), 1 (2x2 windows), 0 (level not pooled)
    int64_t px_off[NLS + 1];   // prefix sum of B*h*w: offset of a level inside T
};

template <int LR, bool WIDE, class T>
__device__ __forceinline__ void gather_gate_level(const Levels& lv, const GGTable& tb, int lvl, int blk, int lane, int cw,
                                                  const T* __restrict__ x, const float* w1, const float* w2,
                                                  float* __restrict__ G, int32_t* __restrict__ argmax, float* __restrict__ Tm,
                                                  float* red) {
    const int h = lv.H[lvl], w = lv.W[lvl], wq = tb.wq[lvl];
    const int hr = lv.H[lv.r], wr = lv.W[lv.r];
    const Item it = item_of(LR, blk * 64, lane, tb.items[lvl], wq, h, w);
    const int C = lv.C;
    const unsigned plane = (unsigned)(h * w), plane_r = (unsigned)(hr * wr);
    const unsigned off = (unsigned)it.bi * (unsigned)C * plane + (unsigned)(it.y * w + it.x0);
    const int cpw = (C + GG_NW - 1) / GG_NW;
    const int c_begin = cw * cpw < C ? cw * cpw : C, c_end = c_begin + cpw < C ? c_begin + cpw : C;
    const const_fp k1g = (const_fp)(uintptr_t)w1, k2g = (const_fp)(uintptr_t)w2;
    const int r = lane & ((1 << LR) - 1);
    // pooled output offset (+ c * plane_r): window row y >> LR, first window column of the quad
    const unsigned goff = (unsigned)it.bi * (unsigned)C * plane_r + (unsigned)((it.y >> LR) * wr + (it.x0 >> LR));
    const int f0 = it.y * w + it.x0;

    f32x2_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = f32x2_t{0.f, 0.f};

    // software pipeline: the reads of batch k + 1 are issued before batch k is consumed (the workgroup's 4 waves are all the
    // waves these pixels get - 2.7 per SIMD at the bench size - so latency is hidden inside a wave, not across waves)
    auto issue = [&](float4_u (&xv)[GG_U], int c0) {
#pragma unroll
        for (int u = 0; u < GG_U; ++u) {
            const int c = c0 + u < c_end ? c0 + u : c_end - 1;
            xv[u] = load_quad<WIDE>(x + (size_t)(c < 0 ? 0 : c) * plane + off, it);
        }
    };
    auto consume = [&](const float4_u (&xv)[GG_U], int c0) {
#pragma unroll
        for (int u = 0; u < GG_U; ++u) {
            const bool live = c0 + u < c_end;                       // wave-uniform; a dead slot repeats the last channel
            const int c = live ? c0 + u : c_end - 1;
            const const_fp k1 = k1g + c * 9, k2 = k2g + c * 9;
            const float sc = live ? 1.f : 0.f;                      // ... and contributes nothing
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x2_t us = {k1[t] * sc, k2[t] * sc};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] += us * f32x2_t{xv[u][e], xv[u][e]};
            }
            if constexpr (LR == 2) {
                // 4x4 window = this quad in 4 rows (lanes 4k .. 4k+3)
                float v = xv[u][0];
                int i = f0;
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    const bool take = (xv[u][e] > v) || (xv[u][e] != xv[u][e]);
                    v = take ? xv[u][e] : v;
                    i = take ? f0 + e : i;
                }
                pool_combine(v, i, dpp_xor1(v), dpp_xor1(i), (r & 1) != 0);
                pool_combine(v, i, dpp_xor2(v), dpp_xor2(i), (r & 2) != 0);
                if (r == 0 && it.active && live) {
                    G[(size_t)c * plane_r + goff] = v;
                    if (argmax) argmax[(size_t)c * plane_r + goff] = i;
                }
            } else if constexpr (LR == 1) {
                // two 2x2 windows per quad (pixels 0,1 and 2,3) in 2 rows (lanes 2k, 2k+1)
                const bool tA = (xv[u][1] > xv[u][0]) || (xv[u][1] != xv[u][1]);
                const bool tB = (xv[u][3] > xv[u][2]) || (xv[u][3] != xv[u][3]);
                float vA = tA ? xv[u][1] : xv[u][0], vB = tB ? xv[u][3] : xv[u][2];
                int iA = tA ? f0 + 1 : f0, iB = tB ? f0 + 3 : f0 + 2;
                pool_combine(vA, iA, dpp_xor1(vA), dpp_xor1(iA), r != 0);
                pool_combine(vB, iB, dpp_xor1(vB), dpp_xor1(iB), r != 0);
                if (r == 0 && it.active && live) {
                    float* gp = G + (size_t)c * plane_r + goff;
                    gp[0] = vA; gp[1] = vB;
                    if (argmax) { int32_t* ap = argmax + (size_t)c * plane_r + goff; ap[0] = iA; ap[1] = iB; }
                }
            }
        }
    };
    float4_u xa[GG_U], xb[GG_U];
    if (c_begin < c_end) {
        issue(xa, c_begin);
#pragma unroll 1
        for (int c0 = c_begin; c0 < c_end; c0 += 2 * GG_U) {
            issue(xb, c0 + GG_U);                      // (past the end: clamped re-reads of the last channel, never consumed)
            consume(xa, c0);
            if (c0 + GG_U < c_end) {
                issue(xa, c0 + 2 * GG_U);
                consume(xb, c0 + GG_U);
            }
        }
    }
    // ---- channel quarters -> T: two halves (gate 1, gate 2) of 36 values per lane through LDS, summed wave 0..3 ----
    float* Tb = Tm + tb.px_off[lvl] * GG_M + (size_t)it.bi * GG_M * plane + (size_t)(it.y * w + it.x0);
#pragma unroll
    for (int gate = 0; gate < 2; ++gate) {
        if (gate) __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((cw * 9 + t) * 4 + e) * 64 + lane] = acc[t][e][gate];
        __syncthreads();
        // tap t is finished by wave t % GG_NW
#pragma unroll
        for (int tt = 0; tt < (9 + GG_NW - 1) / GG_NW; ++tt) {
            const int t = cw + GG_NW * tt;
            if (t < 9) {
                float4_u o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sum = red[((0 * 9 + t) * 4 + e) * 64 + lane];
#pragma unroll
                    for (int wv = 1; wv < GG_NW; ++wv) sum += red[((wv * 9 + t) * 4 + e) * 64 + lane];
                    o[e] = sum;
                }
                if (it.active) store_quad<WIDE>(Tb + (size_t)(gate * 9 + t) * plane, it, w, o);
            }
        }
    }
}

template <class T>
__global__ __launch_bounds__(64 * GG_NW, ARFE_GG_WAVES) void fpn_gather_gate_kernel(Levels lv, GGTable tb, SLevelPtrs x, SLevelPtrs w1, SLevelPtrs w2,
                                                              SLevelMutPtrs G, int32_t* __restrict__ argmax,
                                                              float* __restrict__ Tm) {
    __shared__ float red[GG_NW * 9 * 4 * 64];              // 36 KB at 4 waves: [wave][tap][pixel][lane]
    const int lane = threadIdx.x & 63, cw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int lvl = lv.n - 1;
    while (lvl > 0 && (int)blockIdx.x >= tb.first_blk[lvl - 1]) --lvl;
    const int j = (int)blockIdx.x - tb.first_blk[lvl];
    const int nwork = tb.nwork[lvl];
    const int blk = (j & 7) * ((nwork + 7) >> 3) + (j >> 3);          // XCD-aware order, see fpn_input_bwd2_kernel
    if (blk >= nwork) return;
    const size_t am_off = (size_t)lvl * lv.B * lv.C * lv.H[lv.r] * lv.W[lv.r];
    int32_t* am = argmax ? argmax + am_off : nullptr;
    switch (tb.lr[lvl]) {
        case 2: gather_gate_level<2, true, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], w1.p[lvl], w2.p[lvl], G.p[lvl], am, Tm, red); break;
        case 1: gather_gate_level<1, true, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], w1.p[lvl], w2.p[lvl], G.p[lvl], am, Tm, red); break;
        default:
            if ((lv.W[lvl] & 3) == 0)
                gather_gate_level<0, true, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], w1.p[lvl], w2.p[lvl], nullptr, nullptr, Tm, red);
            else
                gather_gate_level<0, false, T>(lv, tb, lvl, blk, lane, cw, (const T*)x.p[lvl], w1.p[lvl], w2.p[lvl], nullptr, nullptr, Tm, red);
            break;
    }
}

// m = (((((0 + G_0) + G_1) + ...) + G_{n-1}) / n over the refine-sized maps: pooled levels from G, the others from x through
// the nearest index (exactly the first-generation kernel's arithmetic: the result is bit-identical)
template <class T>
__global__ __launch_bounds__(256) void fpn_gather_mean_kernel(Levels lv, SLevelPtrs x, SLevelPtrs G, float* __restrict__ m) {
    // one thread per 4 consecutive pixels of a refine-level row (16-byte accesses where the row allows)
    const int hr = lv.H[lv.r], wr = lv.W[lv.r], wqr = (wr + 3) >> 2;
    const unsigned rows = (unsigned)(lv.B * lv.C * hr);
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows * (unsigned)wqr) return;
    const unsigned row = q / (unsigned)wqr;                  // (b, c, py)
    const int px0 = (int)(q - row * (unsigned)wqr) * 4;
    const unsigned bc = row / (unsigned)hr;
    const int py = (int)(row - bc * (unsigned)hr);
    const bool full = px0 + 3 < wr;
    const size_t o = (size_t)row * wr + px0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < lv.n; ++i) {
        float g[4];
        if (i < lv.r) {
            const float* src = G.p[i] + o;
            if (full) {
                const float4_u v = *reinterpret_cast<const float4_u*>(src);
                g[0] = v[0]; g[1] = v[1]; g[2] = v[2]; g[3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = src[px0 + e < wr ? e : 0];
            }
        } else if (lv.sh[i] == 0) {
            const T* src = (const T*)x.p[i] + o;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (float)src[px0 + e < wr ? e : 0];
        } else {
            const int h = lv.H[i], w = lv.W[i];
            const int sy = nearest_src(py, h, nearest_scale(h, hr));
            const float xs = nearest_scale(w, wr);
            const T* srow = (const T*)x.p[i] + ((size_t)bc * h + sy) * w;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (float)srow[nearest_src(px0 + e < wr ? px0 + e : wr - 1, w, xs)];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + g[e];
    }
    const float n = (float)lv.n;
    if (full) {
        const float4_u v = {acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n};
        *reinterpret_cast<float4_u*>(m + o) = v;
    } else {
        for (int e = 0; e < 4 && px0 + e < wr; ++e) m[o + e] = acc[e] / n;
    }
}

// -> 1 handled, 0 not eligible (the caller runs the two first-generation ops), < 0 error.
// Gbuf: refine_level maps [B,C,hr,wr] (workspace); T: [18 * sum(B h w)] floats (the gate maps' product buffer).
int32_t fpn_gather_gate_v2(const Levels& lv, const float* const* x, const float* const* w1, const float* const* w2, float* m,
                           int32_t* argmax, float* Gbuf, float* T, hipStream_t stream) {
    static const bool off = []() { const char* e = getenv("ARFE_FPN_V2"); return e && e[0] == '0'; }();
    if (off) return 0;
    if (lv.C < GG_NW) return 0;
    GGTable tb;
    SLevelPtrs xp, w1p, w2p, gp;
    SLevelMutPtrs gm;
    const int64_t per = (int64_t)lv.B * lv.C * lv.H[lv.r] * lv.W[lv.r];
    if (per * 4 >= ((int64_t)1 << 31)) return 0;
    int nblk = 0;
    int64_t px = 0;
    for (int i = 0; i < NLS; ++i) {
        tb.nwork[i] = 0; tb.items[i] = 0; tb.wq[i] = 1; tb.lr[i] = 0; tb.px_off[i] = px;
        xp.p[i] = w1p.p[i] = w2p.p[i] = gp.p[i] = nullptr; gm.p[i] = nullptr;
        if (i < lv.n) px += (int64_t)lv.B * lv.H[i] * lv.W[i];
    }
    tb.px_off[NLS] = px;
    for (int i = 0; i < lv.n; ++i) {
        if ((lv.W[i] & 3) != 0 && i < lv.r) return 0;    // pooled levels need whole quads
        if ((int64_t)lv.B * lv.C * lv.H[i] * lv.W[i] * 4 >= ((int64_t)1 << 31)) return 0;
        if (i < lv.r && !(lv.sh[i] == 1 || lv.sh[i] == 2)) return 0;
    }
    for (int i = lv.n - 1; i >= 0; --i) {
        tb.first_blk[i] = nblk;
        xp.p[i] = x[i]; w1p.p[i] = w1[i]; w2p.p[i] = w2[i];
        if (i < lv.r) { gm.p[i] = Gbuf + (size_t)i * per; gp.p[i] = gm.p[i]; tb.lr[i] = lv.sh[i]; }
        tb.wq[i] = cdiv(lv.W[i], 4);
        tb.items[i] = lv.B * lv.H[i] * tb.wq[i];
        tb.nwork[i] = cdiv(tb.items[i], 64);
        nblk += 8 * cdiv(tb.nwork[i], 8);
    }
    for (int i = lv.n; i <= NLS; ++i) tb.first_blk[i] = 0;
    tb.first_blk[NLS] = nblk;
    if (lv.half) hipLaunchKernelGGL(fpn_gather_gate_kernel<_Float16>, dim3(nblk), dim3(64 * GG_NW), 0, stream, lv, tb, xp, w1p, w2p, gm, argmax, T);
    else hipLaunchKernelGGL(fpn_gather_gate_kernel<float>, dim3(nblk), dim3(64 * GG_NW), 0, stream, lv, tb, xp, w1p, w2p, gm, argmax, T);
    if (hipGetLastError() != hipSuccess) { set_error("fpn_gather_gate launch failed"); return ARFE_ERR_LAUNCH; }
    const int64_t mquads = (int64_t)lv.B * lv.C * lv.H[lv.r] * cdiv(lv.W[lv.r], 4);
    if (lv.half) hipLaunchKernelGGL(fpn_gather_mean_kernel<_Float16>, dim3((unsigned)cdiv64(mquads, 256)), dim3(256), 0, stream, lv, xp, gp, m);
    else hipLaunchKernelGGL(fpn_gather_mean_kernel<float>, dim3((unsigned)cdiv64(mquads, 256)), dim3(256), 0, stream, lv, xp, gp, m);
    if (hipGetLastError() != hipSuccess) { set_error("fpn_gather_mean launch failed"); return ARFE_ERR_LAUNCH; }
    return 1;
}

// host mirror of nearest_src (the device form uses floorf on the same fp32 product)
static int longest_nearest_run(int in_size, int out_size) {
    const float scale = nearest_scale(in_size, out_size);
    int best = 1, run = 0, prev = -1;
    for (int d = 0; d < out_size; ++d) {
        int sidx = (int)floorf((float)d * scale);
        if (sidx > in_size - 1) sidx = in_size - 1;
        run = sidx == prev ? run + 1 : 1;
        prev = sidx;
        if (run > best) best = run;
    }
    return best;
}

// which levels the v2 kernel takes: finer levels below the refine level with ratio 4 or 2, the refine-sized level(s), coarser
// levels above it (ratio 1/2 by shifts, anything else by the generic nearest^T ranges); every level below 2^31 bytes
static int ib2_mode(const Levels& lv, int i) {
    const int sh = lv.sh[i];
    if ((int64_t)lv.B * lv.C * lv.H[i] * lv.W[i] * 4 >= ((int64_t)1 << 31)) return IB_NONE;
    if ((int64_t)lv.B * lv.C * lv.H[lv.r] * lv.W[lv.r] * 4 >= ((int64_t)1 << 31)) return IB_NONE;
    if (i < lv.r) return sh == 2 ? IB_SH2 : (sh == 1 ? IB_SH1 : IB_NONE);
    if (sh == 0) return IB_SH0;
    if (sh == -1) return IB_SHM1;
    return IB_GEN;                       // level >= refine, any size: nearest^T by destination ranges
}

// -> bit mask of the levels handled (the caller runs the first-generation kernels on the others), or < 0 on error
int32_t fpn_input_bwd_v2(const Levels& lv, const float* const* gout, const float* const* dpa, const float* const* dpb,
                         const float* const* w1, const float* const* w2, const float* dm, const int32_t* argmax,
                         float* const* dx, hipStream_t stream) {
    static const bool off = []() { const char* e = getenv("ARFE_FPN_V2"); return e && e[0] == '0'; }();
    if (off) return 0;
    ItemTable tb;
    SLevelPtrs gp, ap, bp, w1p, w2p;
    SLevelMutPtrs dxp;
    int nblk = 0, mask = 0;
    for (int i = 0; i < NLS; ++i) {
        tb.items[i] = 0; tb.wq[i] = 1; tb.mode[i] = IB_NONE; tb.lr[i] = 0; tb.cgroups[i] = 1; tb.cpw[i] = IB_CPW;
        tb.ry[i] = tb.rx[i] = 1; tb.nwork[i] = 0;
        gp.p[i] = ap.p[i] = bp.p[i] = w1p.p[i] = w2p.p[i] = nullptr; dxp.p[i] = nullptr;
    }
    // first_blk[i] = first workgroup of level i, levels in DESCENDING order (first_blk[n-1] = 0); first_blk[NLS] = total
    for (int i = lv.n - 1; i >= 0; --i) {
        tb.first_blk[i] = nblk;
        gp.p[i] = gout[i]; ap.p[i] = dpa[i]; bp.p[i] = dpb[i]; w1p.p[i] = w1[i]; w2p.p[i] = w2[i]; dxp.p[i] = dx[i];
        const int mode = ib2_mode(lv, i);
        if (mode == IB_NONE) continue;
        tb.mode[i] = mode;
        tb.wq[i] = cdiv(lv.W[i], 4);
        const bool wide = (lv.W[i] & 3) == 0;
        const int rows = !wide ? 1 : (mode == IB_SH2 ? IB_ROWS_SH2 : (mode == IB_SH1 ? IB_ROWS_SH1 : 1));
        tb.items[i] = lv.B * (lv.H[i] / rows) * tb.wq[i];
        tb.lr[i] = 0;
        tb.cpw[i] = (mode == IB_SHM1 || mode == IB_GEN) ? IB_CPW_SMALL : IB_CPW;
        tb.cgroups[i] = cdiv(lv.C, 4 * tb.cpw[i]);
        if (mode == IB_GEN) {       // the longest run of refine-level pixels with the same nearest source, per axis
            tb.ry[i] = longest_nearest_run(lv.H[i], lv.H[lv.r]);
            tb.rx[i] = longest_nearest_run(lv.W[i], lv.W[lv.r]);
        }
        tb.nwork[i] = cdiv(tb.items[i], 64) * tb.cgroups[i];
        nblk += 8 * cdiv(tb.nwork[i], 8);
        mask |= 1 << i;
    }
    for (int i = lv.n; i <= NLS; ++i) tb.first_blk[i] = 0;
    tb.first_blk[NLS] = nblk;
    if (nblk == 0) return 0;
    if (lv.half)
        hipLaunchKernelGGL(fpn_input_bwd2_kernel<_Float16>, dim3(nblk), dim3(256), 0, stream, lv, tb, gp, ap, bp, w1p, w2p, dm,
                           argmax, dxp);
    else
        hipLaunchKernelGGL(fpn_input_bwd2_kernel<float>, dim3(nblk), dim3(256), 0, stream, lv, tb, gp, ap, bp, w1p, w2p, dm,
                           argmax, dxp);
    if (hipGetLastError() != hipSuccess) { set_error("fpn_input_bwd2 launch failed"); return ARFE_ERR_LAUNCH; }
    return mask;
}

// the scatter takes every level: ratio 4 / 2 / 1 to the refine level by shifts (any position in the pyramid), all others
// through the generic nearest index (element-wise bsf reads)
int32_t fpn_scatter_fwd_v2(const Levels& lv, const float* const* x, const float* bsf, const float* const* a,
                           const float* const* b, float* const* out, hipStream_t stream) {
    static const bool off = []() { const char* e = getenv("ARFE_FPN_V2"); return e && e[0] == '0'; }();
    if (off) return 0;
    ItemTable tb;
    SLevelPtrs xp, ap, bp;
    SLevelMutPtrs op;
    int nblk = 0;
    for (int i = 0; i < NLS; ++i) {
        tb.items[i] = 0; tb.wq[i] = 1; tb.mode[i] = IB_NONE; tb.lr[i] = 0; tb.cgroups[i] = 1; tb.cpw[i] = SC2_CPW;
        tb.ry[i] = tb.rx[i] = 1; tb.nwork[i] = 0;
        xp.p[i] = ap.p[i] = bp.p[i] = nullptr; op.p[i] = nullptr;
    }
    if ((int64_t)lv.B * lv.C * lv.H[lv.r] * lv.W[lv.r] * 4 >= ((int64_t)1 << 31)) return 0;
    for (int i = 0; i < lv.n; ++i)
        if ((int64_t)lv.B * lv.C * lv.H[i] * lv.W[i] * 4 >= ((int64_t)1 << 31)) return 0;
    for (int i = lv.n - 1; i >= 0; --i) {
        tb.first_blk[i] = nblk;
        xp.p[i] = x[i]; ap.p[i] = a[i]; bp.p[i] = b[i]; op.p[i] = out[i];
        const int sh = lv.sh[i];
        tb.mode[i] = sh == 2 ? IB_SH2 : (sh == 1 ? IB_SH1 : (sh == 0 ? IB_SH0 : IB_GEN));
        tb.wq[i] = cdiv(lv.W[i], 4);
        tb.items[i] = lv.B * lv.H[i] * tb.wq[i];
        tb.cgroups[i] = cdiv(lv.C, 4 * tb.cpw[i]);
        tb.nwork[i] = cdiv(tb.items[i], 64) * tb.cgroups[i];
        nblk += 8 * cdiv(tb.nwork[i], 8);
    }
    for (int i = lv.n; i <= NLS; ++i) tb.first_blk[i] = 0;
    tb.first_blk[NLS] = nblk;
    if (lv.half) hipLaunchKernelGGL(fpn_scatter_fwd2_kernel<_Float16>, dim3(nblk), dim3(256), 0, stream, lv, tb, xp, bsf, ap, bp, op);
    else hipLaunchKernelGGL(fpn_scatter_fwd2_kernel<float>, dim3(nblk), dim3(256), 0, stream, lv, tb, xp, bsf, ap, bp, op);
    if (hipGetLastError() != hipSuccess) { set_error("fpn_scatter_fwd2 launch failed"); return ARFE_ERR_LAUNCH; }
    return 1;
}

}  // namespace arfe
