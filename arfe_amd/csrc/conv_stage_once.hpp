// 3x3 conv on [R, C, 7, 7] RoI tiles, "stage once" form of the implicit GEMM (forward and data gradient):
//   y[m][n] = sum_{tap, k} W[tap][m][k] * x[k][n + shift(tap)]      n = (RoI, pixel), shift = dy*7 + dx, 0 outside the 7x7
// The 256x224 GEMM kernel gathers, splits and stores the activation tile once per (tap, channel slice): nine times per
// slice, and that staging - not the MFMAs - bounds it (tools/probe/mma_probe.hip).  Here a 32-channel slice of x is staged
// ONCE per workgroup as a [240 pixels][32 ch] hi/lo image (the 224 pixels of the tile + 8 halo pixels each side, because a
// tile boundary may cut an RoI) and the nine taps read ROW-SHIFTED B fragments from it; a lane whose source pixel is
// outside the 7x7 map reads a zero row instead (address select, no arithmetic on the data).  Only the pre-split weights are
// staged per (tap, slice).  Same tile, wave layout, MFMA (v_mfma_f32_16x16x32_bf16) and epilogue as the 256x224 kernel.
// Round 2: the weights are stored by the pack kernel as two bf16 PLANES (hi, lo) [tap][m][k] and go global -> LDS by
// LDS-DMA (global_load_lds_dwordx4, 16 B per lane, 4 per wave and K-step): no staging registers, no unpack, no
// ds_write for the A operand (32 KB per K-step, nine times the bytes of the activation image).  The LDS image is
// lane-linear per wave-instruction (16 rows x 4 octets), so the XOR swizzle of the rows (xs_swz) is applied to the SOURCE
// octet each lane fetches.  The step's DMA is issued at the top of the iteration into the other buffer and has the whole
// MFMA phase to land; the barrier that ends the iteration (vmcnt(0) + s_barrier) retires it.
// Round 3: the tile's 224 pixels are 32 image rows of 7 (a tile starts on a row: 224 = 32*7), and a wave's 16-column MFMA block j
// is now pixel COLUMN j of its 16 rows (lane = row) instead of 16 consecutive pixels.  A tap that looks left (right) finds
// column 0 (6) outside every map: that whole block is skipped - 6 of 9 taps skip 1 of 7 blocks, 9.5 % of the MFMAs.  The
// row validity (taps that look up / down) is one address select per lane and tap; the seven blocks of a tap are immediate
// offsets from it.  In the epilogue a lane holds the 7 consecutive outputs of an image row per output channel: 28-byte
// stores, 16 lanes = 16 consecutive rows.
#pragma once
#include "gemm_bf16x3.hpp"
#include <type_traits>

namespace arfe {

constexpr int CS_BN = 224, CS_HALO = 8, CS_ROWS = CS_BN + 2 * CS_HALO;      // 240 staged pixel rows, row 240 = zeros
#ifndef ARFE_CS_BLD
#define ARFE_CS_BLD 40
#endif
constexpr int CS_BLD = ARFE_CS_BLD;                                         // B image: 80-byte rows (shifted reads: no swizzle)
constexpr int CS_ZROWS = 8;                                                 // zero rows behind the image: one per column block, + 1: a tap that
                                                                            // looks right reads (never uses) one row past block 6
constexpr int CS_A_IMG = XL_BM * XS_LD;                                     // A image (swizzled 64-byte rows), elements
constexpr int CS_B_IMG = (CS_ROWS + CS_ZROWS) * CS_BLD;
constexpr int CS_LDS_BYTES = (2 * 2 * CS_A_IMG + 2 * 2 * CS_B_IMG) * 2;     // 2 buffers x (hi, lo) of each operand

struct ConvStageDims { int M, N, C, tiles_n, dbg; };   // dbg: probe switches (ARFE_CONV_DBG), 0 in production

// PREC == 1 (half-operand mode, see gemm_bf16x3.hpp): wplanes is ONE plane of IEEE halves, the activations are converted to
// half on their way into LDS, one v_mfma_f32_16x16x32_f16 per block.
template <class EPI, int PREC = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_stage_once_kernel(ConvStageDims d, const unsigned short* __restrict__ wplanes,
                                                                    const float* __restrict__ x, int64_t roi_stride, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short cs_lds[];
    auto img_a = [&](int buf, int lo) -> unsigned short* { return cs_lds + (buf * 2 + lo) * CS_A_IMG; };
    auto img_b = [&](int buf, int lo) -> unsigned short* { return cs_lds + 4 * CS_A_IMG + (buf * 2 + lo) * CS_B_IMG; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    const int tn = blockIdx.x, tm = blockIdx.y;
    const int m0 = tm * XL_BM, n0 = tn * CS_BN;
    const int nslices = d.C / 32, nsteps = nslices * 9;

    f32x4 acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero rows of the four B images
    for (int i = tid; i < 4 * CS_ZROWS * CS_BLD; i += 512)
        (cs_lds + 4 * CS_A_IMG + (i / (CS_ZROWS * CS_BLD)) * CS_B_IMG + CS_ROWS * CS_BLD)[i % (CS_ZROWS * CS_BLD)] = 0;

    // ---- A staging by LDS-DMA: a wave-instruction fills 64 consecutive 16-byte chunks of an image = 16 rows x 4 octets;
    // wave w takes chunks [(2*i + ...)]: instruction i of wave w covers chunks (i*8 + w)*64 .. +63.  LDS chunk (row, pos)
    // holds source octet pos ^ xs_swz(row).
    const int64_t plane = (int64_t)9 * d.M * d.C;             // elements per plane (hi | lo)
    int a_src[2];                                              // element offset inside one tap's [M][C] matrix
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = (i * 8 + wave) * 64 + lane;
        const int row = chunk >> 2, oct = (chunk & 3) ^ xs_swz(row);
        const int m = m0 + row;
        a_src[i] = (m < d.M ? m : d.M - 1) * d.C + 8 * oct;    // + tap*M*C + k0 per step
    }
    // ---- B staging (activations, pixel-fast): row = staged pixel, two octets (8 channels each) per thread ----
    const int b_row = tid & 255;
    const bool b_on = b_row < CS_ROWS;
    int b_oct[2];
    const float* b_src;
    {
        int g = n0 - CS_HALO + (b_on ? b_row : 0);
        g = g < 0 ? 0 : (g >= d.N ? d.N - 1 : g);            // clamped pixels are only ever read through masked lanes
        const int r = g / 49, p = g - r * 49;
        b_src = x + (int64_t)r * roi_stride + p;
        b_oct[0] = tid >> 8; b_oct[1] = (tid >> 8) + 2;
    }
    float rb[2][8];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_a = [&](int step, int buf) {
        const int tap = step % 9, k0 = (step / 9) * 32;
        const unsigned short* base = wplanes + (int64_t)tap * d.M * d.C + k0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int dst = (i * 8 + wave) * 64 * 8;            // wave-uniform element offset of the 1-KB piece
            __builtin_amdgcn_global_load_lds(base + a_src[i], (lds_ptr_t)(img_a(buf, 0) + dst), 16, 0, 0);
            if constexpr (PREC == 0)
                __builtin_amdgcn_global_load_lds(base + plane + a_src[i], (lds_ptr_t)(img_a(buf, 1) + dst), 16, 0, 0);
        }
    };
    auto fetch_b = [&](int slice) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* q = b_src + (int64_t)(slice * 32 + 8 * b_oct[i]) * 49;
#pragma unroll
            for (int j = 0; j < 8; ++j) rb[i][j] = q[j * 49];
        }
    };
    auto stash_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            if constexpr (PREC == 1) { hi = cvt8_f16(rb[i]); lo = hi; }
            else split8(rb[i], hi, lo);
            if (b_on) {
                *reinterpret_cast<bf16x8*>(img_b(buf, 0) + b_row * CS_BLD + 8 * b_oct[i]) = hi;
                if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_b(buf, 1) + b_row * CS_BLD + 8 * b_oct[i]) = lo;
            }
        }
    };
    // ---- per-lane fragment geometry: this wave owns image rows wn*16 .. +15 of the tile, lane l15 = row, block j = column ----
    const int row_n = n0 + wn * 112 + l15 * 7;                                // first pixel of this lane's image row
    const int b_base = (wn * 112 + l15 * 7 + CS_HALO) * CS_BLD + 8 * lq;      // column 0, unshifted; column j: + j*CS_BLD
    const int zero_off = CS_ROWS * CS_BLD + 8 * lq;
    int b_off[3];           // per dy: the row to read from (the zero rows when the source row is outside the 7x7 map)
    {
        const int g = row_n < d.N ? row_n : d.N - 7;
        const int py = (g / 7) % 7;
        b_off[0] = py > 0 ? b_base - 7 * CS_BLD : zero_off;
        b_off[1] = b_base;
        b_off[2] = py < 6 ? b_base + 7 * CS_BLD : zero_off;
    }
    const int frag_oct = 8 * (lq ^ xs_swz(l15));
    // dxc = 0 / 1 / 2: the tap looks left / straight / right; block 0 (6) is not computed when it looks left (right): its
    // source column is outside every map.  One code path: the two conditional blocks are wave-uniform branches.
    //
    // Pipeline (round 3).  A step is one tap of one 32-channel slice: ONE MFMA K-step per (row block, column block), so a
    // barrier per step with the fragment reads behind it left the matrix pipe idle while every wave of the workgroup waited
    // for its first LDS reads (SQ_VALU_MFMA_BUSY = 50 % of the SIMD cycles).  Now the barrier sits in the MIDDLE of the step:
    //   blocks 0..3 of step s | vmcnt(0) + barrier | DMA of step s+2, A fragments of step s+1 -> second register set,
    //   blocks 4..6 of step s, first B fragment of step s+1 | blocks 0..3 of step s+1 | ...
    // so the reads that follow the barrier have three blocks of MFMAs to land behind.  Buffer s&1 is free at the barrier of
    // step s (its fragments were read before the barrier of step s-1 ... by every wave that reaches this one), and the DMA
    // of step s+1 went out a whole step ago.  Steps are unrolled in pairs (the two register sets swap roles).
    bf16x8 afh[2][4], afl[2][4];        // A fragments: [set][row block]
    bf16x8 bh[2], bl[2];                // B fragments: slot = (global block index) & 1
    auto boff_of = [&](int s) {
        const int tap = s % 9;
        const int dy = tap / 3, dxc = tap - dy * 3;
        return b_off[dy] + (dxc - 1) * CS_BLD;
    };
    auto load_a = [&](int abuf, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = (wm * 64 + t * 16 + l15) * XS_LD + frag_oct;
            afh[S][t] = *reinterpret_cast<const bf16x8*>(img_a(abuf, 0) + off);
            if constexpr (PREC == 0) afl[S][t] = *reinterpret_cast<const bf16x8*>(img_a(abuf, 1) + off);
        }
    };
    auto load_b = [&](int bbuf, int off, auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        bh[S] = *reinterpret_cast<const bf16x8*>(img_b(bbuf, 0) + off);
        if constexpr (PREC == 0) bl[S] = *reinterpret_cast<const bf16x8*>(img_b(bbuf, 1) + off);
    };
    auto block = [&](auto j_tag, auto set_tag, auto slot_tag) {
        constexpr int j = decltype(j_tag)::value, A = decltype(set_tag)::value, S = decltype(slot_tag)::value;
        if constexpr (PREC == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, afh[A][i]), __builtin_bit_cast(f16x8, bh[S]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[A][i], bh[S], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afh[A][i], bl[S], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afh[A][i], bh[S], acc[i][j], 0, 0, 0);
        }
    };
    // one step; P = s & 1: register set of its A fragments, and the parity of its first B slot (7 blocks per step)
    auto step = [&](int s, auto p_tag) {
        constexpr int P = decltype(p_tag)::value;
        const int tap = s % 9, slice = s / 9;
        const int dxc = tap - (tap / 3) * 3;
        const int bbuf = slice & 1, boff = boff_of(s);
        const bool next_b = tap == 8 && slice + 1 < nslices;
        const bool more = s + 1 < nsteps;
        // the activation registers (loaded nine steps ago) go to the other B image before the barrier of this step
        if (next_b) stash_b((slice + 1) & 1);
        // ---- blocks 0..3 ----
        load_b(bbuf, boff + 1 * CS_BLD, std::integral_constant<int, (P + 1) & 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (dxc != 0) block(std::integral_constant<int, 0>{}, p_tag, std::integral_constant<int, P>{});
        load_b(bbuf, boff + 2 * CS_BLD, std::integral_constant<int, P>{});
        __builtin_amdgcn_sched_barrier(0);
        block(std::integral_constant<int, 1>{}, p_tag, std::integral_constant<int, (P + 1) & 1>{});
        load_b(bbuf, boff + 3 * CS_BLD, std::integral_constant<int, (P + 1) & 1>{});
        __builtin_amdgcn_sched_barrier(0);
        block(std::integral_constant<int, 2>{}, p_tag, std::integral_constant<int, P>{});
        load_b(bbuf, boff + 4 * CS_BLD, std::integral_constant<int, P>{});
        __builtin_amdgcn_sched_barrier(0);
        block(std::integral_constant<int, 3>{}, p_tag, std::integral_constant<int, (P + 1) & 1>{});
        __syncthreads();        // vmcnt(0): the DMA of step s+1 (a step old) has landed; buffer s&1 is free
        if (s + 2 < nsteps) dma_a(s + 2, s & 1);
        if (next_b && slice + 2 < nslices) fetch_b(slice + 2);
        if (more) load_a((s + 1) & 1, std::integral_constant<int, (P + 1) & 1>{});
        // ---- blocks 4..6, and the first fragment of the next step ----
        load_b(bbuf, boff + 5 * CS_BLD, std::integral_constant<int, (P + 1) & 1>{});
        __builtin_amdgcn_sched_barrier(0);
        block(std::integral_constant<int, 4>{}, p_tag, std::integral_constant<int, P>{});
        load_b(bbuf, boff + 6 * CS_BLD, std::integral_constant<int, P>{});
        __builtin_amdgcn_sched_barrier(0);
        block(std::integral_constant<int, 5>{}, p_tag, std::integral_constant<int, (P + 1) & 1>{});
        if (more) load_b(((s + 1) / 9) & 1, boff_of(s + 1), std::integral_constant<int, (P + 1) & 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (dxc != 2) block(std::integral_constant<int, 6>{}, p_tag, std::integral_constant<int, P>{});
    };

#if defined(ARFE_PROBES) && defined(ARFE_CLOCK_PROBE)
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    dma_a(0, 0);                                        // the first weight tile and the first activation slice: one round trip, not two
    fetch_b(0);
    stash_b(0);
    if (nslices > 1) fetch_b(1);
    __syncthreads();                                    // vmcnt(0): step 0's weights have landed
    if (nsteps > 1) dma_a(1, 1);
    load_a(0, std::integral_constant<int, 0>{});
    load_b(0, boff_of(0), std::integral_constant<int, 0>{});
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
    }
    if (s < nsteps) step(s, std::integral_constant<int, 0>{});

#if defined(ARFE_PROBES) && defined(ARFE_CLOCK_PROBE)
    if (tid == 0 && (blockIdx.x % 37) == 0) {        // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md)
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
        printf("clk wg %d: %llu core cycles in %llu x 10 ns -> %.0f MHz\n", (int)blockIdx.x, dt, dr, dr ? 100.0 * (double)dt / (double)dr : 0.0);
    }
#endif
    if (d.dbg & 1) { if (acc[0][0][0] == 123.456f) epi.store_rb(0, 0, 0, acc[1][1][1], true, 0.f); return; }
    // ---- epilogue: D layout lane -> image row l15 (all 7 columns: one per block), output channels 4*lq + r ----
    float rbias[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + i * 16 + 4 * lq + r;
            rbias[i][r] = epi.row_bias(m < d.M ? m : d.M - 1);
        }
    if (row_n >= d.N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + i * 16 + 4 * lq + r;
            if (m >= d.M) continue;
            float v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = acc[i][j][r];
            epi.store_row7(m, row_n, v, rbias[i][r]);
        }
}

template <class EPI, int PREC>
static inline void launch_conv_stage_once_prec(const ConvStageDims& d, const unsigned short* wplanes, const float* x,
                                               int64_t roi_stride, EPI epi, hipStream_t stream, int32_t* rc) {
    static DeviceOnce attr;           // per instantiation and per device
    if (!attr.run([]() { return hipFuncSetAttribute((const void*)conv3x3_stage_once_kernel<EPI, PREC>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS_BYTES) == hipSuccess; })) {
        set_error("conv3x3: cannot reserve %d bytes of LDS", CS_LDS_BYTES);
        *rc = ARFE_ERR_LAUNCH;
        return;
    }
    hipLaunchKernelGGL((conv3x3_stage_once_kernel<EPI, PREC>), dim3(d.tiles_n, cdiv(d.C, XL_BM)), dim3(512), CS_LDS_BYTES, stream,
                       d, wplanes, x, roi_stride, epi);
    if (hipGetLastError() != hipSuccess) { set_error("conv3x3 stage-once launch failed"); *rc = ARFE_ERR_LAUNCH; }
}

// -> true when the stage-once kernel ran (C a multiple of 32, at least one full tile of pixels; ARFE_CONV_STAGE_ONCE=0
// switches it off), false to use the generic GEMM.  wplanes: bf16 (hi | lo) planes; wplane_f16: the same matrix as one plane
// of halves (half-operand mode)
template <class EPI>
static inline bool launch_conv_stage_once(int R, int C, const unsigned short* wplanes, const unsigned short* wplane_f16,
                                          const float* x, int64_t roi_stride, EPI epi, hipStream_t stream, int32_t* rc) {
    static const char* sw = getenv("ARFE_CONV_STAGE_ONCE");
    *rc = ARFE_OK;
    const int N = R * 49;
    if ((sw && sw[0] == '0') || gemm_use_f32() || N < CS_BN || C % 32 != 0 || C < 64) return false;
    // probe switches (skip the epilogue / MFMA phase / staging: wrong results on purpose) exist only in a build with
    // -DARFE_PROBES (tools/probe/build_variant.sh); the shipped library ignores ARFE_CONV_DBG
#ifdef ARFE_PROBES
    static const int dbg = []() { const char* e = getenv("ARFE_CONV_DBG"); return e ? atoi(e) : 0; }();
#else
    constexpr int dbg = 0;
#endif
    ConvStageDims d{C, N, C, cdiv(N, CS_BN), dbg};      // a ragged last tile: clamped loads, columns >= N are never stored
    if (gemm_half_operands()) launch_conv_stage_once_prec<EPI, 1>(d, wplane_f16, x, roi_stride, epi, stream, rc);
    else launch_conv_stage_once_prec<EPI, 0>(d, wplanes, x, roi_stride, epi, stream, rc);
    return true;
}

}  // namespace arfe
