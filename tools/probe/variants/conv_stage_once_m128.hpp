// 3x3 conv on [R, C, 7, 7] RoI tiles, stage-once form (see conv_stage_once.hpp), as TWO INDEPENDENT workgroups per CU:
// 256 threads own a 128 (output channels) x 224 (pixels) tile - the same 64 x 112 wave tile, column mapping, LDS-DMA weight
// staging and epilogue as the 512-thread kernel, half its rows.  Why: in the 512-thread kernel all eight waves of a CU meet
// at one barrier per K-step (ONE MFMA K-step of work each), so the two waves that share a SIMD wait for their LDS reads at
// the same time and then contend for the matrix pipe at the same time (SQ_VALU_MFMA_BUSY 53 % of the SIMD cycles, waves
// ready-but-not-issued 45 %).  Two workgroups have independent barriers: while one waits the other computes.
// LDS per workgroup: A 2 x (hi, lo) x 128 rows x 64 B = 32 KB, B ONE (hi, lo) image of 247 rows x 80 B = 39.5 KB (the slice
// change - every 9 steps - costs a second barrier instead of a second image): 71.5 KB, two fit the CU's 160 KB.
// Both workgroups of a pixel tile stage the same activations (the price of the split; the weights' traffic is unchanged).
#pragma once
#include "conv_stage_once.hpp"

namespace arfe {

constexpr int CH_BM = 128;
constexpr int CH_A_IMG = CH_BM * XS_LD;
constexpr int CH_LDS_BYTES = (2 * 2 * CH_A_IMG + 2 * CS_B_IMG) * 2;         // 2 A buffers x (hi, lo) + one B image (hi, lo)

// PREC == 1 (half-operand mode, see gemm_bf16x3.hpp): wplanes is ONE plane of IEEE halves, the activations are converted to
// half on their way into LDS, one v_mfma_f32_16x16x32_f16 per block.
template <class EPI, int PREC = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_stage_once_m128_kernel(ConvStageDims d, const unsigned short* __restrict__ wplanes,
                                                                    const float* __restrict__ x, int64_t roi_stride, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ch_lds[];
    unsigned short* const cs_lds = ch_lds;
    auto img_a = [&](int buf, int lo) -> unsigned short* { return cs_lds + (buf * 2 + lo) * CH_A_IMG; };
    auto img_b = [&](int lo) -> unsigned short* { return cs_lds + 4 * CH_A_IMG + lo * CS_B_IMG; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    const int tn = blockIdx.x, tm = blockIdx.y;
    const int m0 = tm * CH_BM, n0 = tn * CS_BN;
    const int nslices = d.C / 32, nsteps = nslices * 9;

    f32x4 acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero rows of the two B images
    for (int i = tid; i < 2 * CS_ZROWS * CS_BLD; i += 256)
        (img_b(i / (CS_ZROWS * CS_BLD)) + CS_ROWS * CS_BLD)[i % (CS_ZROWS * CS_BLD)] = 0;

    // ---- A staging by LDS-DMA: a wave-instruction fills 64 consecutive 16-byte chunks of an image = 16 rows x 4 octets;
    // wave w takes chunks [(2*i + ...)]: instruction i of wave w covers chunks (i*8 + w)*64 .. +63.  LDS chunk (row, pos)
    // holds source octet pos ^ xs_swz(row).
    const int64_t plane = (int64_t)9 * d.M * d.C;             // elements per plane (hi | lo)
    int a_src[2];                                              // element offset inside one tap's [M][C] matrix
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int chunk = (i * 4 + wave) * 64 + lane;
        const int row = chunk >> 2, oct = (chunk & 3) ^ xs_swz(row);
        const int m = m0 + row;
        a_src[i] = (m < d.M ? m : d.M - 1) * d.C + 8 * oct;    // + tap*M*C + k0 per step
    }
    // ---- B staging (activations, pixel-fast): row = staged pixel, two octets (8 channels each) per thread ----
    const int b_row = tid;                                   // one staged pixel row per thread, all four octets
    const bool b_on = b_row < CS_ROWS;
    const float* b_src;
    {
        int g = n0 - CS_HALO + (b_on ? b_row : 0);
        g = g < 0 ? 0 : (g >= d.N ? d.N - 1 : g);            // clamped pixels are only ever read through masked lanes
        const int r = g / 49, p = g - r * 49;
        b_src = x + (int64_t)r * roi_stride + p;
    }
    float rb[4][8];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_a = [&](int step, int buf) {
        const int tap = step % 9, k0 = (step / 9) * 32;
        const unsigned short* base = wplanes + (int64_t)tap * d.M * d.C + k0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int dst = (i * 4 + wave) * 64 * 8;            // wave-uniform element offset of the 1-KB piece
            __builtin_amdgcn_global_load_lds(base + a_src[i], (lds_ptr_t)(img_a(buf, 0) + dst), 16, 0, 0);
            if constexpr (PREC == 0)
                __builtin_amdgcn_global_load_lds(base + plane + a_src[i], (lds_ptr_t)(img_a(buf, 1) + dst), 16, 0, 0);
        }
    };
    auto fetch_b = [&](int slice) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* q = b_src + (int64_t)(slice * 32 + 8 * i) * 49;
#pragma unroll
            for (int j = 0; j < 8; ++j) rb[i][j] = q[j * 49];
        }
    };
    auto stash_b = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 hi, lo;
            if constexpr (PREC == 1) { hi = cvt8_f16(rb[i]); lo = hi; }
            else split8(rb[i], hi, lo);
            if (b_on) {
                *reinterpret_cast<bf16x8*>(img_b(0) + b_row * CS_BLD + 8 * i) = hi;
                if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_b(1) + b_row * CS_BLD + 8 * i) = lo;
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
    auto mma = [&](int abuf, int dy, int dxc) {
        const int boff = b_off[dy] + (dxc - 1) * CS_BLD;
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = (wm * 64 + t * 16 + l15) * XS_LD + frag_oct;
            ah[t] = *reinterpret_cast<const bf16x8*>(img_a(abuf, 0) + off);
            if constexpr (PREC == 0) al[t] = *reinterpret_cast<const bf16x8*>(img_a(abuf, 1) + off);
            else al[t] = ah[t];
        }
        bf16x8 bh[2], bl[2];
        {
            bh[0] = *reinterpret_cast<const bf16x8*>(img_b(0) + boff);
            if constexpr (PREC == 0) bl[0] = *reinterpret_cast<const bf16x8*>(img_b(1) + boff);
            else bl[0] = bh[0];
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j + 1 < 7) {
                bh[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(img_b(0) + boff + (j + 1) * CS_BLD);
                if constexpr (PREC == 0) bl[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(img_b(1) + boff + (j + 1) * CS_BLD);
                else bl[(j + 1) & 1] = bh[(j + 1) & 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((j == 0 && dxc == 0) || (j == 6 && dxc == 2)) continue;
            if constexpr (PREC == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bh[j & 1]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j & 1], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j & 1], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j & 1], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- pipeline: A by LDS-DMA, double-buffered per step (tap inner); B: ONE image per 9-step slice, replaced behind a second
    // barrier when the slice ends (its registers were fetched nine steps earlier) ----
    fetch_b(0);
    stash_b();
    dma_a(0, 0);
    if (nslices > 1) fetch_b(1);
    __syncthreads();                                    // vmcnt(0): step 0's weights have landed
    int abuf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int tap = s % 9, slice = s / 9;
        const bool next_b = tap == 8 && slice + 1 < nslices;
        if (s + 1 < nsteps) dma_a(s + 1, abuf ^ 1);     // the other buffer: its readers passed the last barrier
        __builtin_amdgcn_sched_barrier(0);
        mma(abuf, tap / 3, tap - (tap / 3) * 3);
        __syncthreads();                                // retires the DMA (vmcnt(0)), frees buffer abuf (and, at tap 8, the B image)
        if (next_b) {
            stash_b();
            if (slice + 2 < nslices) fetch_b(slice + 2);
            __syncthreads();
        }
        abuf ^= 1;
    }

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
static inline void launch_conv_stage_once_m128(const ConvStageDims& d, const unsigned short* wplanes, const float* x,
                                               int64_t roi_stride, EPI epi, hipStream_t stream, int32_t* rc) {
    static DeviceOnce attr;           // per instantiation and per device
    if (!attr.run([]() { return hipFuncSetAttribute((const void*)conv3x3_stage_once_m128_kernel<EPI, PREC>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES) == hipSuccess; })) {
        set_error("conv3x3: cannot reserve %d bytes of LDS", CH_LDS_BYTES);
        *rc = ARFE_ERR_LAUNCH;
        return;
    }
    hipLaunchKernelGGL((conv3x3_stage_once_m128_kernel<EPI, PREC>), dim3(d.tiles_n, cdiv(d.C, CH_BM)), dim3(256), CH_LDS_BYTES,
                       stream, d, wplanes, x, roi_stride, epi);
    if (hipGetLastError() != hipSuccess) { set_error("conv3x3 stage-once launch failed"); *rc = ARFE_ERR_LAUNCH; }
}

}  // namespace arfe
