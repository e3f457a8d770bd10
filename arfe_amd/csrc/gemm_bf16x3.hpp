// fp32 GEMM at bf16-MFMA speed by operand splitting ("bf16x3"):
//     a = a_hi + a_lo (+ <= 2^-18 |a|),  a_hi = bf16(a), a_lo = bf16(a - a_hi)
//     a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi          (the dropped a_lo*b_lo term is <= 2^-18 |a||b|)
// three v_mfma_f32_32x32x16_bf16 per 32x32x16 block with fp32 accumulation give ~1e-5 relative accuracy
// (vs 4e-3 for plain bf16) at 3/16 of the cost of the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), i.e. the
// 1e-3 parity budget of the path is kept with a 5.3x higher matrix-pipe ceiling.
//
// Same interface as gemm_f32_kernel (loader / epilogue functors, split-K, k-batches, XCD-aware tile map).
//   tile 128x128x32, 256 threads = 4 waves (2x2), wave = 64x64 = 2x2 MFMA blocks
//   LDS: four k-contiguous bf16 images (A_hi, A_lo, B_hi, B_lo), unpadded 64-byte rows with the 16-byte k-octets
//        XOR-swizzled by the row (xs_swz) so that the MFMA operand fetch (lane l: row l&31, k = 8*(l>>5)..+7) and
//        the staging stores are conflict-free ds_read_b128 / ds_write_b128
//   a thread stages two k-octets per operand (fp32 from global -> registers, in flight during the MFMAs of
//   the previous step), splits them and writes 16 B of hi and 16 B of lo.
#pragma once
#include "gemm_f32.hpp"

namespace arfe {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XB_BK = 32;
// Swizzled, unpadded operand image (64-byte rows): the 16-byte k-octet q of row r is stored at octet q ^ xs_swz(r).
// ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... :
// MI355X_MICROARCH.md, LDS) on 64 banks, ds_write_b128 in eight contiguous 8-lane groups on 32 banks.  With
// xs_swz(r) = F[(r >> 2) & 3] ^ ((r >> 1) & 1), F = {0, 2, 3, 1}, every group of the MFMA operand fetch (16x16x32: lane
// -> row l & 15, octet l >> 4; 32x32x16: row l & 31, octet l >> 5) and of both staging maps (k-fast: 2 rows x 4 octets
// per 8 lanes; row-fast: 8 rows x 1 octet) lands on distinct 16-byte slots.  The former 80-byte padded rows were
// 2-way conflicts on three slots of every read group (SQ_LDS_BANK_CONFLICT 15.5 M cycles per conv launch).
constexpr int XS_LD = 32;
__device__ __forceinline__ int xs_swz(int r) { return ((0x1320 >> (((r >> 2) & 3) * 4)) & 3) ^ ((r >> 1) & 1); }

// Pairs are converted with one v_cvt_pk_bf16_f32 each and kept as packed 32-bit words all the way to the 16-byte LDS
// store (element-wise __bf16 code compiled to 2.5x the conversions plus ~35 register moves per octet, all of them on
// the issue port the MFMAs share): 4 cvt + 8 unpack + 4 v_pk_add_f32 + 4 cvt = 20 vector instructions per octet.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 x = {v[2 * j], v[2 * j + 1]};
        const unsigned hw = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        const f32x2 hf = {__uint_as_float(hw << 16), __uint_as_float(hw & 0xffff0000u)};
        const f32x2 r = x - hf;
        h[j] = hw;
        l[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}

// Pre-split operands: a matrix that is re-laid anyway before it is used (the conv weights) is stored as one 32-bit word
// per element, bf16(x) in the upper half and bf16(x - bf16(x)) in the lower, i.e. exactly the pair split8 computes.
// The loaders move the words like floats; staging is then two v_perm_b32 per element pair instead of a split.
template <class L, class = void> struct is_presplit { static constexpr bool value = false; };
template <class L> struct is_presplit<L, decltype((void)L::presplit)> { static constexpr bool value = L::presplit; };
struct LoadSplitK : LoadRowMajorK {
    static constexpr bool presplit = true;
    typedef LoadRowMajorK Plain;
};
__device__ __forceinline__ unsigned presplit_word(float x) {
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | __builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ void unpack8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned w0 = __float_as_uint(v[2 * j]), w1 = __float_as_uint(v[2 * j + 1]);
        h[j] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
        l[j] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
    }
    hi = __builtin_bit_cast(bf16x8, h);
    lo = __builtin_bit_cast(bf16x8, l);
}
template <class L>
__device__ __forceinline__ void stage8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    if constexpr (is_presplit<L>::value) unpack8(v, hi, lo);
    else split8(v, hi, lo);
}

// ---- half-operand mode (PREC == 1; round 3) ---------------------------------------------------------------------------
// The reference's fp16 mode (mmdet/core/fp16/hooks.py:86-105) runs its contractions on half operands.  Here the operands are
// then converted to IEEE half on their way into LDS (exact for values that were stored as half, and the same rounding the
// reference applies to intermediates it keeps in half) and every product is ONE v_mfma_f32_*_f16 with fp32 accumulation
// instead of the three (six) bf16 MFMAs of the fp32-grade split: a third (sixth) of the matrix-pipe work, one LDS image per
// operand instead of two.  The kernels keep their layout; the "lo" images are simply not written or read.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 cvt8_f16(const float (&v)[8]) {
    u32x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 x = {v[2 * j], v[2 * j + 1]};
        h[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));      // v_cvt_pk_f16_f32... (RNE)
    }
    return __builtin_bit_cast(bf16x8, h);
}
// pre-split words (bf16 hi | bf16 lo, see presplit_word) back to the value, then to half
__device__ __forceinline__ bf16x8 cvt8_f16_presplit(const float (&v)[8]) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned w = __float_as_uint(v[j]);
        f[j] = __uint_as_float(w & 0xffff0000u) + __uint_as_float(w << 16);
    }
    return cvt8_f16(f);
}
template <int PREC, class L>
__device__ __forceinline__ void stage8p(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    if constexpr (PREC == 1) {
        if constexpr (is_presplit<L>::value) hi = cvt8_f16_presplit(v);
        else hi = cvt8_f16(v);
        lo = hi;
    } else {
        stage8<L>(v, hi, lo);
    }
}
template <int PREC>
__device__ __forceinline__ f32x16 mma32p(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
    if constexpr (PREC == 1) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), c, 0, 0, 0);
    } else {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    }
}
bool gemm_half_operands();           // per-thread mode, arfe_set_operand_precision (api.hip)

// Pipeline (round 2): two LDS buffers (2 x 32 KB, dynamic) and ONE barrier per K-step, as in the 256-wide kernels below:
// iteration s splits + stores step s+1 into the other buffer, re-issues the global loads of step s+2 into the registers
// just freed, then runs the 24 MFMAs of step s.  The first form (one buffer: loads -> MFMAs -> barrier -> split/store ->
// barrier) exposed the load latency and the whole staging pass every step: 1.8 us per K-step for 0.35 us of MFMAs.
constexpr int XM_BUF = 4 * GEMM_BM * XS_LD;                 // bf16 elements per buffer: A_hi A_lo B_hi B_lo
constexpr int XM_LDS_BYTES = 2 * XM_BUF * 2;                // 65 536

template <class LA, class LB, class EPI, int PREC = 0>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmDims d, LA la, LB lb, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short xm_lds[];
    auto img = [&](int buf, int which) -> unsigned short* { return xm_lds + (size_t)buf * XM_BUF + which * (GEMM_BM * XS_LD); };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;

    int ntiles = d.tiles_m * d.tiles_n;
    int bid = blockIdx.x;
    {
        int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / d.tiles_n, tn = bid % d.tiles_n;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const int ksteps_per_kb = (d.K + XB_BK - 1) / XB_BK;
    const int per = (d.ksteps_total + d.splitk - 1) / d.splitk;
    const int s_begin = blockIdx.y * per;
    int s_end = s_begin + per;
    if (s_end > d.ksteps_total) s_end = d.ksteps_total;
    const bool first_split = (blockIdx.y == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging coordinates: operand tile = 128 rows x 4 k-octets; two octets per thread and operand
    int a_row[2], a_oct[2], b_row[2], b_oct[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (LA::kfast) { a_oct[i] = tid & 3; a_row[i] = (tid >> 2) + 64 * i; }
        else           { a_row[i] = tid & 127; a_oct[i] = (tid >> 7) + 2 * i; }
        if (LB::kfast) { b_oct[i] = tid & 3; b_row[i] = (tid >> 2) + 64 * i; }
        else           { b_row[i] = tid & 127; b_oct[i] = (tid >> 7) + 2 * i; }
    }
    typename LA::Row arow[2];
    typename LB::Row brow[2];
    int a_off[2], b_off[2];          // swizzled element offsets of this thread's staging octets inside an image (xs_swz)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = la.row(z, m0 + a_row[i]);
        brow[i] = lb.row(z, n0 + b_row[i]);
        a_off[i] = a_row[i] * XS_LD + 8 * (a_oct[i] ^ xs_swz(a_row[i]));
        b_off[i] = b_row[i] * XS_LD + 8 * (b_oct[i] ^ xs_swz(b_row[i]));
    }
    float ra[2][8], rb[2][8];
    auto fetch = [&](int step) {
        // k-batch outermost (default) or innermost (kb_inner: the same K-slice of every k-batch back to back,
        // e.g. the 9 taps of a 3x3 conv re-use one staged channel slice while it is still in L1/L2)
        const int kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        const int k0 = (d.kb_inner ? step / d.kb_count : step - kb * ksteps_per_kb) * XB_BK;
        const bool full_k = k0 + XB_BK <= d.K;      // wave-uniform: every step but a K tail skips the k bound checks
        if (full_k) {
#pragma unroll
            for (int i = 0; i < 2; ++i) la.template load8<true>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lb.template load8<true>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) la.template load8<false>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lb.template load8<false>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        }
    };
    // branch-free steady-state loaders (K a multiple of the K-step): the (k-batch, inner step) pair of the step being
    // fetched is carried and advanced with selects, so the loop body is one scheduling region
    int f_kb = 0, f_ks = 0;
    auto fetch_seek = [&](int step) {
        f_kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        f_ks = d.kb_inner ? step / d.kb_count : step - f_kb * ksteps_per_kb;
    };
    auto fetch_full_a = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) la.template load8<true>(arow[i], z, f_kb, f_ks * XB_BK + 8 * a_oct[i], ra[i]);
    };
    auto fetch_full_b = [&]() {
        const int kb = f_kb, k0 = f_ks * XB_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) lb.template load8<true>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        const int inner = d.kb_inner;
        int kb2 = f_kb + inner, ks2 = f_ks + 1 - inner;
        const bool wrap_kb = kb2 == d.kb_count;               // only reachable with kb_inner
        ks2 += (inner & (wrap_kb ? 1 : 0));
        kb2 = wrap_kb ? 0 : kb2;
        const bool wrap_ks = (inner == 0) & (ks2 == ksteps_per_kb);
        kb2 += wrap_ks ? 1 : 0;
        ks2 = wrap_ks ? 0 : ks2;
        f_kb = kb2; f_ks = ks2;
    };
    auto stash_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            stage8p<PREC, LA>(ra[i], hi, lo);
            *reinterpret_cast<bf16x8*>(img(buf, 0) + a_off[i]) = hi;
            if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img(buf, 1) + a_off[i]) = lo;
        }
    };
    auto stash_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            stage8p<PREC, LB>(rb[i], hi, lo);
            *reinterpret_cast<bf16x8*>(img(buf, 2) + b_off[i]) = hi;
            if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img(buf, 3) + b_off[i]) = lo;
        }
    };
    const int kh = lane >> 5, l31 = lane & 31;
    const int fsw = xs_swz(l31);     // fragment row bases are multiples of 32
    auto mma = [&](int buf, int kk) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ar = wm * 64 + t * 32 + l31, br = wn * 64 + t * 32 + l31;
            const int ko = 8 * (((kk >> 3) + kh) ^ fsw);
            ah[t] = *reinterpret_cast<const bf16x8*>(img(buf, 0) + ar * XS_LD + ko);
            bh[t] = *reinterpret_cast<const bf16x8*>(img(buf, 2) + br * XS_LD + ko);
            if constexpr (PREC == 0) {
                al[t] = *reinterpret_cast<const bf16x8*>(img(buf, 1) + ar * XS_LD + ko);
                bl[t] = *reinterpret_cast<const bf16x8*>(img(buf, 3) + br * XS_LD + ko);
            } else { al[t] = ah[t]; bl[t] = bh[t]; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma32p<PREC>(ah[i], al[i], bh[j], bl[j], acc[i][j]);
    };

    if (s_begin < s_end) {
        fetch(s_begin);
        stash_a(0);
        stash_b(0);
        if (s_begin + 1 < s_end) fetch(s_begin + 1);
        __syncthreads();
        int buf = 0;
        int s = s_begin;
        if (d.K % XB_BK == 0) {
            fetch_seek(s + 2);
            for (; s + 2 < s_end; ++s) {
                stash_a(buf ^ 1);
                fetch_full_a();
                __builtin_amdgcn_sched_barrier(0);
                mma(buf, 0);
                stash_b(buf ^ 1);
                fetch_full_b();
                __builtin_amdgcn_sched_barrier(0);
                mma(buf, 16);
                __syncthreads();
                buf ^= 1;
            }
        } else {
            for (; s + 2 < s_end; ++s) {       // K tail somewhere: the bound-checked loaders (branches in the body)
                stash_a(buf ^ 1);
                mma(buf, 0);
                stash_b(buf ^ 1);
                fetch(s + 2);
                mma(buf, 16);
                __syncthreads();
                buf ^= 1;
            }
        }
        for (; s < s_end; ++s) {       // last two steps: nothing left to fetch
            const bool more = s + 1 < s_end;
            if (more) { stash_a(buf ^ 1); stash_b(buf ^ 1); }
            mma(buf, 0);
            mma(buf, 16);
            __syncthreads();
            buf ^= 1;
        }
    } else if (!first_split) {
        return;
    }
    epi_select_slice(epi, (int)blockIdx.y);

    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < d.M) epi(z, m, n, acc[i][j][r], first_split);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// fp32-grade contraction on the bf16 pipe ("bf16x6"): a = a1 + a2 + a3 with three bf16 terms (24 significand bits),
//   a*b ~= a3*b1 + a1*b3 + a2*b2 + a2*b1 + a1*b2 + a1*b1      (dropped terms <= 2^-24 |a||b|)
// six v_mfma_f32_32x32x16_bf16 per 32x32x16 block = 6/16 of the exact-fp32 MFMA's cycles at fp32-rounding-level
// accuracy per product.  Used for the contractions marked `exact` (the NonLocal logits feed exp()); ARFE_GEMM_F32=1 still
// selects the true fp32 MFMA everywhere.  Same 128x128x32 single-buffer pipeline as gemm_bf16x3_kernel, six LDS images.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8x3(const float (&v)[8], bf16x8& t1, bf16x8& t2, bf16x8& t3) {
    u32x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 x = {v[2 * j], v[2 * j + 1]};
        const unsigned hw = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        const f32x2 hf = {__uint_as_float(hw << 16), __uint_as_float(hw & 0xffff0000u)};
        const f32x2 r1 = x - hf;
        const unsigned mw = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
        const f32x2 mf = {__uint_as_float(mw << 16), __uint_as_float(mw & 0xffff0000u)};
        const f32x2 r2 = r1 - mf;
        h[j] = hw; m[j] = mw;
        l[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
    }
    t1 = __builtin_bit_cast(bf16x8, h);
    t2 = __builtin_bit_cast(bf16x8, m);
    t3 = __builtin_bit_cast(bf16x8, l);
}

template <class LA, class LB, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16x6_kernel(GemmDims d, LA la, LB lb, EPI epi) {
    // [operand image: 0..2 A terms, 3..5 B terms][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short lds[6][GEMM_BM][XS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;

    int ntiles = d.tiles_m * d.tiles_n;
    int bid = blockIdx.x;
    {
        int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / d.tiles_n, tn = bid % d.tiles_n;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const int ksteps_per_kb = (d.K + XB_BK - 1) / XB_BK;
    const int per = (d.ksteps_total + d.splitk - 1) / d.splitk;
    const int s_begin = blockIdx.y * per;
    int s_end = s_begin + per;
    if (s_end > d.ksteps_total) s_end = d.ksteps_total;
    const bool first_split = (blockIdx.y == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_row[2], a_oct[2], b_row[2], b_oct[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (LA::kfast) { a_oct[i] = tid & 3; a_row[i] = (tid >> 2) + 64 * i; }
        else           { a_row[i] = tid & 127; a_oct[i] = (tid >> 7) + 2 * i; }
        if (LB::kfast) { b_oct[i] = tid & 3; b_row[i] = (tid >> 2) + 64 * i; }
        else           { b_row[i] = tid & 127; b_oct[i] = (tid >> 7) + 2 * i; }
    }
    typename LA::Row arow[2];
    typename LB::Row brow[2];
    int a_so[2], b_so[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = la.row(z, m0 + a_row[i]);
        brow[i] = lb.row(z, n0 + b_row[i]);
        a_so[i] = 8 * (a_oct[i] ^ xs_swz(a_row[i]));      // swizzled octet inside the 64-byte LDS row (see xs_swz)
        b_so[i] = 8 * (b_oct[i] ^ xs_swz(b_row[i]));
    }
    float ra[2][8], rb[2][8];
    auto fetch = [&](int step) {
        const int kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        const int k0 = (d.kb_inner ? step / d.kb_count : step - kb * ksteps_per_kb) * XB_BK;
        const bool full_k = k0 + XB_BK <= d.K;
        if (full_k) {
#pragma unroll
            for (int i = 0; i < 2; ++i) la.template load8<true>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lb.template load8<true>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) la.template load8<false>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lb.template load8<false>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 t1, t2, t3;
            split8x3(ra[i], t1, t2, t3);
            *reinterpret_cast<bf16x8*>(&lds[0][a_row[i]][a_so[i]]) = t1;
            *reinterpret_cast<bf16x8*>(&lds[1][a_row[i]][a_so[i]]) = t2;
            *reinterpret_cast<bf16x8*>(&lds[2][a_row[i]][a_so[i]]) = t3;
            split8x3(rb[i], t1, t2, t3);
            *reinterpret_cast<bf16x8*>(&lds[3][b_row[i]][b_so[i]]) = t1;
            *reinterpret_cast<bf16x8*>(&lds[4][b_row[i]][b_so[i]]) = t2;
            *reinterpret_cast<bf16x8*>(&lds[5][b_row[i]][b_so[i]]) = t3;
        }
    };

    if (s_begin < s_end) {
        fetch(s_begin);
        stash();
        __syncthreads();
        const int kh = lane >> 5, l31 = lane & 31;
        const int fsw = xs_swz(l31);     // fragment row bases are multiples of 32
        for (int s = s_begin; s < s_end; ++s) {
            const bool more = (s + 1 < s_end);
#if defined(ARFE_PROBES) && defined(ARFE_X6_PROBE) && (ARFE_X6_PROBE & 4)
            (void)more;
#else
            if (more) fetch(s + 1);
#endif
#if defined(ARFE_PROBES) && defined(ARFE_X6_PROBE) && (ARFE_X6_PROBE & 2)
            if (s < 0)
#endif
#pragma unroll
            for (int kk = 0; kk < XB_BK; kk += 16) {
                bf16x8 a1[2], a2[2], a3[2], b1[2], b2[2], b3[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ar = wm * 64 + t * 32 + l31, br = wn * 64 + t * 32 + l31;
                    a1[t] = *reinterpret_cast<const bf16x8*>(&lds[0][ar][8 * (((kk >> 3) + kh) ^ fsw)]);
                    a2[t] = *reinterpret_cast<const bf16x8*>(&lds[1][ar][8 * (((kk >> 3) + kh) ^ fsw)]);
                    a3[t] = *reinterpret_cast<const bf16x8*>(&lds[2][ar][8 * (((kk >> 3) + kh) ^ fsw)]);
                    b1[t] = *reinterpret_cast<const bf16x8*>(&lds[3][br][8 * (((kk >> 3) + kh) ^ fsw)]);
                    b2[t] = *reinterpret_cast<const bf16x8*>(&lds[4][br][8 * (((kk >> 3) + kh) ^ fsw)]);
                    b3[t] = *reinterpret_cast<const bf16x8*>(&lds[5][br][8 * (((kk >> 3) + kh) ^ fsw)]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {       // smallest terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i], b1[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b3[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b2[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b1[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b2[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
                    }
            }
            __syncthreads();
#if !(defined(ARFE_PROBES) && defined(ARFE_X6_PROBE) && (ARFE_X6_PROBE & 4))
            if (more) {
                stash();
                __syncthreads();
            }
#endif
        }
    } else if (!first_split) {
        return;
    }
#if defined(ARFE_PROBES) && defined(ARFE_X6_PROBE) && (ARFE_X6_PROBE & 1)
    if (acc[0][0][0] != 123.456f) return;
#endif

    const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < d.M) epi(z, m, n, acc[i][j][r], first_split);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x6, second generation (round 3).  Probes on the kernel above showed its three parts adding up (staging 53 us + MFMAs 74 +
// epilogue 62 of 175 for the NonLocal logits): one LDS buffer and two barriers per K-step overlap nothing.  Here the K-step is
// ONE MFMA k-slice (16), which halves the six operand images (128 rows x 32 B each: 24.6 KB per buffer), so that TWO buffers fit
// beside a second workgroup (49 KB): the split + LDS stores of step s+1 and the global loads of step s+2 ride between the 24 MFMAs
// of step s, one barrier per step.  The epilogue is the large tile's: the two column blocks of a wave exchange halves
// (v_permlane32_swap) and every store covers 256 contiguous bytes.  LDS rows are 32 bytes; octet o of row r sits at
// o ^ ((r >> 3) & 1) (conflict-free b128 fragment reads for lane = (row, k half)).
constexpr int X6_BK = 16;
constexpr int X6_IMG = GEMM_BM * X6_BK;                 // elements per operand image
constexpr int X6_BUF = 6 * X6_IMG;
constexpr int X6_LDS_BYTES = 2 * X6_BUF * 2;            // 49 152

template <class LA, class LB, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16x6v2_kernel(GemmDims d, LA la, LB lb, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short x6_lds[];
    auto img = [&](int buf, int which) -> unsigned short* { return x6_lds + buf * X6_BUF + which * X6_IMG; };
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;

    int ntiles = d.tiles_m * d.tiles_n;
    int bid = blockIdx.x;
    {
        int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / d.tiles_n, tn = bid % d.tiles_n;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    // d.ksteps_total counts steps of X6_BK here (set by the launcher)
    const int ksteps_per_kb = (d.K + X6_BK - 1) / X6_BK;
    const int per = (d.ksteps_total + d.splitk - 1) / d.splitk;
    const int s_begin = blockIdx.y * per;
    int s_end = s_begin + per;
    if (s_end > d.ksteps_total) s_end = d.ksteps_total;
    const bool first_split = (blockIdx.y == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: operand tile = 128 rows x 2 k-octets = 256 (row, octet) pairs: one per thread and operand
    int a_row, a_oct, b_row, b_oct;
    if (LA::kfast) { a_oct = tid & 1; a_row = tid >> 1; } else { a_row = tid & 127; a_oct = tid >> 7; }
    if (LB::kfast) { b_oct = tid & 1; b_row = tid >> 1; } else { b_row = tid & 127; b_oct = tid >> 7; }
    const typename LA::Row arow = la.row(z, m0 + a_row);
    const typename LB::Row brow = lb.row(z, n0 + b_row);
    const int a_off = a_row * X6_BK + 8 * (a_oct ^ ((a_row >> 3) & 1));
    const int b_off = b_row * X6_BK + 8 * (b_oct ^ ((b_row >> 3) & 1));
    float ra[8], rb[8];
    auto fetch = [&](int step) {
        const int kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        const int k0 = (d.kb_inner ? step / d.kb_count : step - kb * ksteps_per_kb) * X6_BK;
        if (k0 + X6_BK <= d.K) {         // wave-uniform
            la.template load8<true>(arow, z, kb, k0 + 8 * a_oct, ra);
            lb.template load8<true>(brow, z, kb, k0 + 8 * b_oct, rb);
        } else {
            la.template load8<false>(arow, z, kb, k0 + 8 * a_oct, ra);
            lb.template load8<false>(brow, z, kb, k0 + 8 * b_oct, rb);
        }
    };
    auto stash = [&](int buf) {
        bf16x8 t1, t2, t3;
        split8x3(ra, t1, t2, t3);
        *reinterpret_cast<bf16x8*>(img(buf, 0) + a_off) = t1;
        *reinterpret_cast<bf16x8*>(img(buf, 1) + a_off) = t2;
        *reinterpret_cast<bf16x8*>(img(buf, 2) + a_off) = t3;
        split8x3(rb, t1, t2, t3);
        *reinterpret_cast<bf16x8*>(img(buf, 3) + b_off) = t1;
        *reinterpret_cast<bf16x8*>(img(buf, 4) + b_off) = t2;
        *reinterpret_cast<bf16x8*>(img(buf, 5) + b_off) = t3;
    };
    const int kh = lane >> 5, l31 = lane & 31;
    auto mma = [&](int buf) {
        bf16x8 a1[2], a2[2], a3[2], b1[2], b2[2], b3[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ar = wm * 64 + t * 32 + l31, br = wn * 64 + t * 32 + l31;      // (row >> 3) & 1 == (l31 >> 3) & 1
            const int ao = ar * X6_BK + 8 * (kh ^ ((l31 >> 3) & 1)), bo = br * X6_BK + 8 * (kh ^ ((l31 >> 3) & 1));
            a1[t] = *reinterpret_cast<const bf16x8*>(img(buf, 0) + ao);
            a2[t] = *reinterpret_cast<const bf16x8*>(img(buf, 1) + ao);
            a3[t] = *reinterpret_cast<const bf16x8*>(img(buf, 2) + ao);
            b1[t] = *reinterpret_cast<const bf16x8*>(img(buf, 3) + bo);
            b2[t] = *reinterpret_cast<const bf16x8*>(img(buf, 4) + bo);
            b3[t] = *reinterpret_cast<const bf16x8*>(img(buf, 5) + bo);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {       // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i], b1[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b3[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b2[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b1[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b2[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            }
    };

#if defined(ARFE_PROBES) && defined(ARFE_CLOCK_PROBE)
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (s_begin < s_end) {
        fetch(s_begin);
        stash(0);
        if (s_begin + 1 < s_end) fetch(s_begin + 1);
        __syncthreads();
        int buf = 0;
        for (int s = s_begin; s < s_end; ++s) {
            if (s + 1 < s_end) stash(buf ^ 1);          // step s + 1 (registers) -> the other buffer (its readers passed the last barrier)
            if (s + 2 < s_end) fetch(s + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(buf);
            __syncthreads();
            buf ^= 1;
        }
    } else if (!first_split) {
        return;
    }
    epi_select_slice(epi, (int)blockIdx.y);
#if defined(ARFE_PROBES) && defined(ARFE_CLOCK_PROBE)
    if (tid == 0 && (blockIdx.x % 331) == 0 && blockIdx.z == 0) {        // in-kernel clock and main-loop cycles (tools/probe/nl_clock.py)
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
        printf("x6 clk wg %d: %llu core cycles in %llu x 10 ns -> %.0f MHz\n", (int)blockIdx.x, dt, dr, dr ? 100.0 * (double)dt / (double)dr : 0.0);
    }
#endif

    // D layout of a 32x32 block: lane -> column l31, register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the two column blocks
    // exchange halves so that one register is ONE row across the wave's 64 columns (256-byte stores)
    const int n = n0 + wn * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][0][r]), __float_as_uint(acc[i][1][r]), false, false);
            const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
            if (n < d.N) {
                if (m < d.M) epi(z, m, n, __uint_as_float(sw[0]), first_split);
                if (m + 4 < d.M) epi(z, m + 4, n, __uint_as_float(sw[1]), first_split);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Large-tile variant for the big contractions (3x3 convs, FC1, theta^T phi): 256x256x32 tile, 512 threads =
// 8 waves as 2(M) x 4(N), wave tile 128x64 = 4x2 MFMA blocks (128 accumulator registers).  Against the 128x128
// kernel every staged operand element feeds twice the MFMAs (LDS write traffic per flop halves; ds_write_b128
// runs at only ~79 B/clk) and each fragment read feeds 1.5x the MFMAs.  The two LDS buffers (2 x 80 KB = the
// whole 160 KB of a CU) give one barrier per K-step: iteration s splits+stores step s+1 into the other buffer and
// issues the global loads of step s+2 around the 48 MFMAs of step s.
// Measured decomposition on the conv shape (M=256, N=50176, K=9x256; tools/mfma_calib.py): MFMA + fragment reads alone
// 173 us, staging alone (loads + split + LDS stores) 174 us, both 250 us; hipBLASLt needs 209 us for a plain bf16 GEMM
// with the same MFMA work.  The kernel sits at the 256-register limit of 2 waves/SIMD (128 accumulators + 48 fragment
// + 32 staging registers), which is what keeps the two halves from overlapping further.
// ---------------------------------------------------------------------------------------------------------
constexpr int XL_BM = 256, XL_BN = 256;
constexpr int XL_LDS_BYTES = 2 * 2 * (XL_BM + XL_BN) * XS_LD * 2;      // 131 072 (swizzled 64-byte rows, see xs_swz)

template <class LA, class LB, class EPI, int PREC = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_large_kernel(GemmDims d, LA la, LB lb, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short xl_lds[];
    // buffer b: A_hi [256][40] | A_lo | B_hi | B_lo
    auto img = [&](int buf, int which) -> unsigned short* {
        return xl_lds + ((size_t)buf * 4 + which) * (XL_BM * XS_LD);
    };
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // batch and K-slice of this workgroup.  Per-batch reduction lengths (zk_on): gridDim.y lists exactly the (batch, slice)
    // pairs - first the class-1 batches, then the class-0 ones - so that no workgroup is launched only to exit
    // (a grid with 24 idle workgroups beyond the 256 CUs ran the 246 working ones in two rounds).
    int z = blockIdx.z, slice = blockIdx.y;
    if (d.zk_on == 2) {
        const unsigned v = d.zk_pair[blockIdx.y];
        z = (int)(v >> 8);
        slice = (int)(v & 255u);
    } else if (d.zk_on) {
        int p = blockIdx.y, n1 = __builtin_popcount(d.zk_class);
        const bool cls = p < n1 * d.zk_splits1;
        if (!cls) p -= n1 * d.zk_splits1;
        const int sp = cls ? d.zk_splits1 : d.zk_splits0;
        int zi = p / sp;
        slice = p - zi * sp;
        z = 0;
        for (int b = 0; b < 32; ++b) {                  // zi-th batch whose class bit equals cls
            if ((((d.zk_class >> b) & 1u) != 0) == cls) { if (zi == 0) { z = b; break; } --zi; }
        }
    }

    int ntiles = d.tiles_m * d.tiles_n;
    int bid = blockIdx.x;
    {
        int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / d.tiles_n, tn = bid % d.tiles_n;
    const int m0 = tm * XL_BM, n0 = tn * XL_BN;

    const int ksteps_per_kb = (d.K + XB_BK - 1) / XB_BK;
    int kst = d.ksteps_total, spl = d.splitk;
    if (d.zk_on) {                       // per-batch reduction length / slice count
        const bool cls = (d.zk_class >> z) & 1u;
        kst = cls ? d.zk_steps1 : d.zk_steps0;
        spl = cls ? d.zk_splits1 : d.zk_splits0;
    }
    const int per = (kst + spl - 1) / spl;
    const int s_begin = slice * per;
    int s_end = s_begin + per;
    if (s_end > kst) s_end = kst;
    const bool first_split = (slice == 0);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: operand tile = 256 rows x 4 k-octets = 1024 octets; two per thread and operand
    int a_row[2], a_oct[2], b_row[2], b_oct[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (LA::kfast) { a_oct[i] = tid & 3; a_row[i] = (tid >> 2) + 128 * i; }
        else           { a_row[i] = tid & 255; a_oct[i] = (tid >> 8) + 2 * i; }
        if (LB::kfast) { b_oct[i] = tid & 3; b_row[i] = (tid >> 2) + 128 * i; }
        else           { b_row[i] = tid & 255; b_oct[i] = (tid >> 8) + 2 * i; }
    }
    typename LA::Row arow[2];
    typename LB::Row brow[2];
    int a_off[2], b_off[2];          // swizzled element offsets of this thread's staging octets inside an image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = la.row(z, m0 + a_row[i]);
        brow[i] = lb.row(z, n0 + b_row[i]);
        a_off[i] = a_row[i] * XS_LD + 8 * (a_oct[i] ^ xs_swz(a_row[i]));
        b_off[i] = b_row[i] * XS_LD + 8 * (b_oct[i] ^ xs_swz(b_row[i]));
    }
    float ra[2][8], rb[2][8];
    auto fetch = [&](int step) {
        const int kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        const int k0 = (d.kb_inner ? step / d.kb_count : step - kb * ksteps_per_kb) * XB_BK;
        const bool full_k = k0 + XB_BK <= d.K;
        if (full_k) {
#pragma unroll
            for (int i = 0; i < 2; ++i) la.template load8<true>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lb.template load8<true>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) la.template load8<false>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) lb.template load8<false>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        }
    };
    // steady-state form for K a multiple of the K-step: no branch at all, so that the whole iteration is ONE
    // scheduling region and the loads can be placed between the MFMAs
    // The (k-batch, inner step) pair of the step being fetched is carried and advanced with selects: the division
    // form above would put scalar branches into the loop body.
    int f_kb = 0, f_ks = 0;
    auto fetch_seek = [&](int step) {
        f_kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        f_ks = d.kb_inner ? step / d.kb_count : step - f_kb * ksteps_per_kb;
    };
    auto fetch_full_a = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) la.template load8<true>(arow[i], z, f_kb, f_ks * XB_BK + 8 * a_oct[i], ra[i]);
    };
    auto fetch_full_b = [&]() {
        const int kb = f_kb, k0 = f_ks * XB_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) lb.template load8<true>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        // advance to the next step
        const int inner = d.kb_inner;
        int kb2 = f_kb + inner, ks2 = f_ks + 1 - inner;
        const bool wrap_kb = kb2 == d.kb_count;               // only reachable with kb_inner
        ks2 += (inner & (wrap_kb ? 1 : 0));
        kb2 = wrap_kb ? 0 : kb2;
        const bool wrap_ks = (inner == 0) & (ks2 == ksteps_per_kb);
        kb2 += wrap_ks ? 1 : 0;
        ks2 = wrap_ks ? 0 : ks2;
        f_kb = kb2; f_ks = ks2;
    };
    auto stash_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            stage8p<PREC, LA>(ra[i], hi, lo);
            *reinterpret_cast<bf16x8*>(img(buf, 0) + a_off[i]) = hi;
            if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img(buf, 1) + a_off[i]) = lo;
        }
    };
    auto stash_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            stage8p<PREC, LB>(rb[i], hi, lo);
            *reinterpret_cast<bf16x8*>(img(buf, 2) + b_off[i]) = hi;
            if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img(buf, 3) + b_off[i]) = lo;
        }
    };
    const int kh = lane >> 5, l31 = lane & 31;
    const int fsw = xs_swz(l31);       // fragment row bases are multiples of 32: the row's swizzle is the lane's
    auto mma = [&](int buf, int kk) {
        bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ar = wm * 128 + t * 32 + l31;
            ah[t] = *reinterpret_cast<const bf16x8*>(img(buf, 0) + ar * XS_LD + 8 * (((kk >> 3) + kh) ^ fsw));
            if constexpr (PREC == 0) al[t] = *reinterpret_cast<const bf16x8*>(img(buf, 1) + ar * XS_LD + 8 * (((kk >> 3) + kh) ^ fsw));
            else al[t] = ah[t];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int br = wn * 64 + t * 32 + l31;
            bh[t] = *reinterpret_cast<const bf16x8*>(img(buf, 2) + br * XS_LD + 8 * (((kk >> 3) + kh) ^ fsw));
            if constexpr (PREC == 0) bl[t] = *reinterpret_cast<const bf16x8*>(img(buf, 3) + br * XS_LD + 8 * (((kk >> 3) + kh) ^ fsw));
            else bl[t] = bh[t];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma32p<PREC>(ah[i], al[i], bh[j], bl[j], acc[i][j]);
    };

    if (s_begin < s_end) {
        fetch(s_begin);
        stash_a(0);
        stash_b(0);
        if (s_begin + 1 < s_end) fetch(s_begin + 1);
        __syncthreads();
        int buf = 0;
        int s = s_begin;
        // steady state (branch-free body = one scheduling region): the split+store of step s+1 and the global loads of
        // step s+2 are interleaved with the 48 MFMAs of step s by the scheduler directives below
        if (d.K % XB_BK == 0) {
            // branch-free body = ONE scheduling region: the compiler places the split + LDS stores of step s+1 and the
            // global loads of step s+2 between the 48 MFMAs of step s (hand-written sched_group_barrier patterns, and a
            // ping-pong order in which SIMD partner waves alternate MFMA and staging halves, were both measured slower)
            fetch_seek(s + 2);
            for (; s + 2 < s_end; ++s) {
                // each operand's loads for step s + 2 go out as soon as its staging registers are free and are pinned
                // ahead of the MFMAs that follow (the scheduler otherwise sinks them behind the last MFMA, see n224)
                stash_a(buf ^ 1);
                fetch_full_a();
                __builtin_amdgcn_sched_barrier(0);
                mma(buf, 0);
                stash_b(buf ^ 1);
                fetch_full_b();
                __builtin_amdgcn_sched_barrier(0);
                mma(buf, 16);
                __syncthreads();
                buf ^= 1;
            }
        } else {
            for (; s + 2 < s_end; ++s) {       // K tail somewhere: the bound-checked loaders (branches in the body)
                stash_a(buf ^ 1);
                mma(buf, 0);
                stash_b(buf ^ 1);
                fetch(s + 2);
                mma(buf, 16);
                __syncthreads();
                buf ^= 1;
            }
        }
        for (; s < s_end; ++s) {       // last two steps: nothing left to fetch / stash
            const bool more = s + 1 < s_end;
            if (more) { stash_a(buf ^ 1); stash_b(buf ^ 1); }
            mma(buf, 0);
            mma(buf, 16);
            __syncthreads();
            buf ^= 1;
        }
    } else if (!first_split) {
        return;
    }
    epi_select_slice(epi, slice);

    // D layout of a 32x32 block: lane -> column l31, register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The two
    // column blocks of a wave exchange halves (v_permlane32_swap: upper lanes of the first with lower lanes of the
    // second), after which one register holds ONE row across the wave's 64 columns: every store / atomic wave-instruction
    // covers 256 contiguous bytes instead of two 128-byte runs (float atomics run at half rate on those).
    {
        const int n = n0 + wn * 64 + lane;
        if constexpr (epi_has_pre<EPI>::value) {
            if (epi.wants_pre()) {                 // wave-uniform: auxiliary operands of a 32-row block first, then the stores
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float ps[16][2], pd[16][2];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2);
                        const int nc = n < d.N ? n : d.N - 1;
                        epi.pre(z, m < d.M ? m : d.M - 1, nc, ps[r][0], pd[r][0]);
                        epi.pre(z, m + 4 < d.M ? m + 4 : d.M - 1, nc, ps[r][1], pd[r][1]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][0][r]),
                                                                         __float_as_uint(acc[i][1][r]), false, false);
                        const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (n < d.N) {
                            if (m < d.M) epi.store_pre(z, m, n, __uint_as_float(sw[0]), ps[r][0], pd[r][0]);
                            if (m + 4 < d.M) epi.store_pre(z, m + 4, n, __uint_as_float(sw[1]), ps[r][1], pd[r][1]);
                        }
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][0][r]), __float_as_uint(acc[i][1][r]),
                                                                 false, false);
                const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (n < d.N) {
                    if (m < d.M) epi(z, m, n, __uint_as_float(sw[0]), first_split);
                    if (m + 4 < d.M) epi(z, m + 4, n, __uint_as_float(sw[1]), first_split);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 224 variant of the large tile for outputs whose width is a multiple of 224 but gives the 256-wide tile a
// badly filled last round: the 3x3 convs at R = 1024 have N = 50 176 = 196 x 256 = 224 x 224, i.e. 196 workgroups on
// 256 CUs become 224 workgroups of 7/8 the work.  8 waves as 4(M) x 2(N), wave tile 64 x 112 = 4 x 7 blocks of
// v_mfma_f32_16x16x32_bf16 (one instruction spans the whole 32-wide K-step; 112 accumulator registers).  Full K-steps
// only (K a multiple of 32), otherwise the same pipeline as above.
// (Tried: the same tile as 8 x 1 waves of 32 x 224 on v_mfma_f32_32x32x16_bf16, whose 32-cycle gaps leave more vector
// issue for the staging code - 238 us against 220: every wave then reads all of B from LDS.)
// ---------------------------------------------------------------------------------------------------------
constexpr int XN_BN = 224;
constexpr int XN_BUF = 2 * (XL_BM + XN_BN) * XS_LD;                 // bf16 elements per buffer: A_hi A_lo B_hi B_lo
constexpr int XN_LDS_BYTES = 2 * XN_BUF * 2;                        // 122 880
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class LA, class LB, class EPI, int PREC = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_n224_kernel(GemmDims d, LA la, LB lb, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short xn_lds[];
    auto img_a = [&](int buf, int lo) -> unsigned short* { return xn_lds + (size_t)buf * XN_BUF + lo * (XL_BM * XS_LD); };
    auto img_b = [&](int buf, int lo) -> unsigned short* {
        return xn_lds + (size_t)buf * XN_BUF + 2 * XL_BM * XS_LD + lo * (XN_BN * XS_LD);
    };
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int z = blockIdx.z;

    int ntiles = d.tiles_m * d.tiles_n;
    int bid = blockIdx.x;
    {
        int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / d.tiles_n, tn = bid % d.tiles_n;
    const int m0 = tm * XL_BM, n0 = tn * XN_BN;

    const int ksteps_per_kb = d.K / XB_BK;
    const int per = (d.ksteps_total + d.splitk - 1) / d.splitk;
    const int s_begin = blockIdx.y * per;
    int s_end = s_begin + per;
    if (s_end > d.ksteps_total) s_end = d.ksteps_total;
    const bool first_split = (blockIdx.y == 0);

    f32x4 acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int a_row[2], a_oct[2], b_row[2], b_oct[2];
    bool b_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (LA::kfast) { a_oct[i] = tid & 3; a_row[i] = (tid >> 2) + 128 * i; }
        else           { a_row[i] = tid & 255; a_oct[i] = (tid >> 8) + 2 * i; }
        if (LB::kfast) { b_oct[i] = tid & 3; b_row[i] = (tid >> 2) + 128 * i; }
        else           { b_row[i] = tid & 255; b_oct[i] = (tid >> 8) + 2 * i; }
        b_ok[i] = b_row[i] < XN_BN;                  // rows 224..255 of the staging grid do not exist in this tile
        if (!b_ok[i]) b_row[i] = XN_BN - 1;
    }
    typename LA::Row arow[2];
    typename LB::Row brow[2];
    int a_off[2], b_off[2];          // swizzled element offsets of this thread's staging octets inside an image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = la.row(z, m0 + a_row[i]);
        brow[i] = lb.row(z, n0 + b_row[i]);
        a_off[i] = a_row[i] * XS_LD + 8 * (a_oct[i] ^ xs_swz(a_row[i]));
        b_off[i] = b_row[i] * XS_LD + 8 * (b_oct[i] ^ xs_swz(b_row[i]));
    }
    float ra[2][8], rb[2][8];
    int f_kb = 0, f_ks = 0;
    auto fetch_seek = [&](int step) {
        f_kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        f_ks = d.kb_inner ? step / d.kb_count : step - f_kb * ksteps_per_kb;
    };
    auto fetch = [&]() {
        const int kb = f_kb, k0 = f_ks * XB_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) la.template load8<true>(arow[i], z, kb, k0 + 8 * a_oct[i], ra[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) lb.template load8<true>(brow[i], z, kb, k0 + 8 * b_oct[i], rb[i]);
        const int inner = d.kb_inner;
        int kb2 = f_kb + inner, ks2 = f_ks + 1 - inner;
        const bool wrap_kb = kb2 == d.kb_count;
        ks2 += (inner & (wrap_kb ? 1 : 0));
        kb2 = wrap_kb ? 0 : kb2;
        const bool wrap_ks = (inner == 0) & (ks2 == ksteps_per_kb);
        kb2 += wrap_ks ? 1 : 0;
        ks2 = wrap_ks ? 0 : ks2;
        f_kb = kb2; f_ks = ks2;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            stage8p<PREC, LA>(ra[i], hi, lo);
            *reinterpret_cast<bf16x8*>(img_a(buf, 0) + a_off[i]) = hi;
            if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_a(buf, 1) + a_off[i]) = lo;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 hi, lo;
            stage8p<PREC, LB>(rb[i], hi, lo);
            if (b_ok[i]) {
                *reinterpret_cast<bf16x8*>(img_b(buf, 0) + b_off[i]) = hi;
                if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_b(buf, 1) + b_off[i]) = lo;
            }
        }
    };
    const int frag_oct = 8 * (lq ^ xs_swz(l15));    // all fragment row bases are multiples of 16: the swizzle is the lane's
    auto mma = [&](int buf) {
        // operand of v_mfma_f32_16x16x32_bf16: lane -> row l15, k = 8*lq .. 8*lq+7 (one 16-byte read per block and image)
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = (wm * 64 + t * 16 + l15) * XS_LD + frag_oct;
            ah[t] = *reinterpret_cast<const bf16x8*>(img_a(buf, 0) + off);
            if constexpr (PREC == 0) al[t] = *reinterpret_cast<const bf16x8*>(img_a(buf, 1) + off);
            else al[t] = ah[t];
        }
        // B fragments are double-buffered in registers: the two reads of column block j + 1 are issued before the twelve
        // MFMAs of block j, so the wave never sits at lgkmcnt between blocks (the compiler's own schedule read block
        // j + 1 only after issuing block j and drained the matrix pipe seven times per K-step)
        bf16x8 bh[2], bl[2];
        {
            const int off = (wn * 112 + l15) * XS_LD + frag_oct;
            bh[0] = *reinterpret_cast<const bf16x8*>(img_b(buf, 0) + off);
            if constexpr (PREC == 0) bl[0] = *reinterpret_cast<const bf16x8*>(img_b(buf, 1) + off);
            else bl[0] = bh[0];
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j + 1 < 7) {
                const int off = (wn * 112 + (j + 1) * 16 + l15) * XS_LD + frag_oct;
                bh[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(img_b(buf, 0) + off);
                if constexpr (PREC == 0) bl[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(img_b(buf, 1) + off);
                else bl[(j + 1) & 1] = bh[(j + 1) & 1];
            }
            __builtin_amdgcn_sched_barrier(0);
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

    if (s_begin < s_end) {
        fetch_seek(s_begin);
        fetch();
        stash(0);
        if (s_begin + 1 < s_end) fetch();
        __syncthreads();
        int buf = 0;
        int s = s_begin;
        // (taking stash/fetch and mma in opposite order on the two waves that share a SIMD - so that one converts while
        // the other multiplies - measured 231 us against 225: the waves already drift apart inside a barrier interval)
        for (; s + 2 < s_end; ++s) {
            stash(buf ^ 1);
            fetch();
            // the loads of the step after next must be IN FLIGHT during the MFMAs: left alone, the scheduler sinks them
            // behind the last MFMA (they share registers with nothing, it just shortens live ranges) and the next
            // iteration then waits out the whole memory latency at its first s_waitcnt vmcnt
            __builtin_amdgcn_sched_barrier(0);
            mma(buf);
            __syncthreads();
            buf ^= 1;
        }
        for (; s < s_end; ++s) {
            if (s + 1 < s_end) stash(buf ^ 1);
            mma(buf);
            __syncthreads();
            buf ^= 1;
        }
    } else if (!first_split) {
        return;
    }
    epi_select_slice(epi, (int)blockIdx.y);
    // D layout: lane -> column l15, rows 4*lq + r
    if constexpr (epi_has_rowbias<EPI>::value) {       // the 16 rows' biases first, then nothing but stores
        float rb[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + 4 * lq + r;
                rb[i][r] = epi.row_bias(m < d.M ? m : d.M - 1);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int n = n0 + wn * 112 + j * 16 + l15;
                if (n >= d.N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + 4 * lq + r;
                    if (m < d.M) epi.store_rb(z, m, n, acc[i][j][r], first_split, rb[i][r]);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int n = n0 + wn * 112 + j * 16 + l15;
            if (n >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + 4 * lq + r;
                if (m < d.M) epi(z, m, n, acc[i][j][r], first_split);
            }
        }
}

// precision mode of the dense contractions: bf16x3 by default, exact fp32 MFMA with ARFE_GEMM_F32=1
bool gemm_use_f32();

// The 256x256 kernel pays when the output holds enough large tiles to occupy the chip (one 8-wave workgroup per
// CU).  When the caller allows K-slicing (splitk > 1: atomic epilogue into a zeroed output) the slice count is
// re-derived for the large tile so that tiles x slices is about one workgroup per CU.
static inline bool plan_large_tile(int M, int N, int batch, int ksteps, int* splitk, bool caller_plans_k = false) {
    const int64_t tiles = (int64_t)cdiv(M, XL_BM) * cdiv(N, XL_BN) * batch;
    const double fill = (double)M * N / ((double)cdiv(M, XL_BM) * XL_BM * (double)cdiv(N, XL_BN) * XL_BN);
    if (fill < 0.85) return false;
    // Round 2 (profiles/r02_gemm_sweep.txt): with its two LDS buffers the 128x128 kernel keeps two workgroups per CU busy
    // at 1.23 tile-steps/us/CU against 1.08 for this one, and cuts K finer for less atomic traffic - FC1 (16 large tiles,
    // K = 12544) 149 -> 114 us, the NonLocal K = L contractions 137 -> 118 us.  The 256-wide tile only pays where the
    // output alone fills the chip AND the K loop is long enough to amortise its prologue (4096^3: 396 vs 528 us; the conv
    // GEMMs), where the caller has planned the K ranges for it (per-batch lengths), or where one whole dimension fits in a
    // single 256-tile: the NonLocal K = L contractions (M = 256 channels against the 70 MB attention matrix) read that
    // matrix once with the large tile and twice with two rows of 128-tiles - measured in the step 113 vs 141 us.
    const bool one_tile_dim = (M > GEMM_BM && M <= XL_BM) || (N > GEMM_BN && N <= XL_BN);
    if (!caller_plans_k && !one_tile_dim && !(tiles >= 512 || (tiles >= 192 && ksteps >= 32))) return false;
    int sk = 1;
    if (*splitk > 1) {
        sk = (int)(256 / (tiles > 0 ? tiles : 1));
        const int max_s = ksteps / 8 > 1 ? ksteps / 8 : 1;   // >= 8 K-steps per slice: below that the 2-step pipeline
                                                             // fill and the 256-KB atomic epilogue dominate (tools/gemm_sweep.py)
        if (sk > max_s) sk = max_s;
        if (sk < 1) sk = 1;
    }
    if (tiles * sk < 160) return false;
    *splitk = sk;
    return true;
}

template <int PREC, class LA, class LB, class EPI>
static inline int32_t launch_gemm_prec(int M, int N, int K, int kb_count, int batch, int splitk, LA la, LB lb, EPI epi,
                                       hipStream_t stream, bool exact, bool kb_inner, const GemmDims* zvar) {
    if constexpr (is_presplit<LA>::value) {
        // in fp32 mode the re-laid matrix holds plain floats (see conv_w_permute_kernel)
        if (gemm_use_f32())
            return launch_gemm(M, N, K, kb_count, batch, splitk, static_cast<const typename LA::Plain&>(la), lb, epi, stream,
                               kb_inner);
        if (exact && PREC == 0) {
            set_error("gemm: pre-split operands are not available on the bf16x6 path");
            return ARFE_ERR_ARG;
        }
    } else {
        if (gemm_use_f32()) return launch_gemm(M, N, K, kb_count, batch, splitk, la, lb, epi, stream, kb_inner);
    }
    if (M <= 0 || N <= 0 || batch <= 0) return ARFE_OK;
    if constexpr (!is_presplit<LA>::value && PREC == 0) if (exact) {        // fp32-grade products from six bf16 MFMAs (see gemm_bf16x6_kernel)
        GemmDims d6;
        d6.M = M; d6.N = N; d6.K = K; d6.kb_count = kb_count < 1 ? 1 : kb_count;
        d6.kb_inner = kb_inner ? 1 : 0;
        static const bool x6_v1 = []() { const char* e = getenv("ARFE_X6_V1"); return e && e[0] == '1'; }();
        const int bk6 = x6_v1 ? XB_BK : X6_BK;
        d6.ksteps_total = d6.kb_count * cdiv(K, bk6);
        if (splitk < 1) splitk = 1;
        if (splitk > d6.ksteps_total && d6.ksteps_total > 0) splitk = d6.ksteps_total;
        d6.splitk = splitk;
        d6.tiles_m = cdiv(M, GEMM_BM); d6.tiles_n = cdiv(N, GEMM_BN);
        dim3 grid6(d6.tiles_m * d6.tiles_n, splitk, batch);
        if (x6_v1) hipLaunchKernelGGL((gemm_bf16x6_kernel<LA, LB, EPI>), grid6, dim3(256), 0, stream, d6, la, lb, epi);
        else hipLaunchKernelGGL((gemm_bf16x6v2_kernel<LA, LB, EPI>), grid6, dim3(256), X6_LDS_BYTES, stream, d6, la, lb, epi);
        ARFE_CHECK_LAUNCH();
        return ARFE_OK;
    }
    GemmDims d;
    d.M = M; d.N = N; d.K = K; d.kb_count = kb_count < 1 ? 1 : kb_count;
    d.kb_inner = kb_inner ? 1 : 0;
    d.ksteps_total = d.kb_count * cdiv(K, XB_BK);
    if (splitk < 1) splitk = 1;
    if (splitk > d.ksteps_total && d.ksteps_total > 0) splitk = d.ksteps_total;
    int splitk_large = splitk;
    bool use_large = plan_large_tile(M, N, batch, d.ksteps_total, &splitk_large, zvar != nullptr);
    {   // tuning aid (tools/gemm_sweep.py): ARFE_GEMM_PLAN=L<k> / S<k> forces the tile and, for callers that already
        // take the atomic split-K path, the slice count
        static const char* plan = getenv("ARFE_GEMM_PLAN");
        if (plan && (plan[0] == 'L' || plan[0] == 'S')) {
            use_large = plan[0] == 'L';
            const int k = atoi(plan + 1);
            if (k >= 1 && splitk > 1) { splitk = k < d.ksteps_total ? k : d.ksteps_total; splitk_large = splitk; }
            else if (use_large) splitk_large = splitk;
        }
    }
    if (use_large && !zvar && N % XN_BN == 0 && d.K % XB_BK == 0) {
        // 224-wide tiles when they fill the chip's rounds better (cost in units of one 256x256 tile's time)
        const int64_t t256 = (int64_t)cdiv(M, XL_BM) * cdiv(N, XL_BN) * batch * splitk_large;
        const int64_t t224 = (int64_t)cdiv(M, XL_BM) * (N / XN_BN) * batch * splitk_large;
        const double c256 = (double)cdiv64(t256, 256), c224 = (double)cdiv64(t224, 256) * (double)XN_BN / XL_BN;
        static const char* plan = getenv("ARFE_GEMM_PLAN");
        const bool veto = plan && plan[0] == 'L';
        if (c224 < c256 && !veto) {
            splitk = splitk_large;
            d.splitk = splitk;
            d.tiles_m = cdiv(M, XL_BM); d.tiles_n = N / XN_BN;
            static DeviceOnce attr224;     // per instantiation and per device
            if (!attr224.run([]() { return hipFuncSetAttribute((const void*)gemm_bf16x3_n224_kernel<LA, LB, EPI, PREC>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, XN_LDS_BYTES) == hipSuccess; })) {
                set_error("gemm: cannot reserve %d bytes of LDS", XN_LDS_BYTES);
                return ARFE_ERR_LAUNCH;
            }
            dim3 grid(d.tiles_m * d.tiles_n, splitk, batch);
            hipLaunchKernelGGL((gemm_bf16x3_n224_kernel<LA, LB, EPI, PREC>), grid, dim3(512), XN_LDS_BYTES, stream, d, la, lb, epi);
            ARFE_CHECK_LAUNCH();
            return ARFE_OK;
        }
    }
    if (zvar && !use_large) {
        set_error("gemm: per-batch reduction lengths need the large-tile kernel (M=%d N=%d)", M, N);
        return ARFE_ERR_ARG;
    }
    if (use_large) {
        splitk = splitk_large;
        d.splitk = splitk;
        if (zvar) {                      // the caller planned steps and slices per batch (plan_large_tile said yes)
            d.zk_on = zvar->zk_on == 2 ? 2 : 1;
            if (d.zk_on == 2) memcpy(d.zk_pair, zvar->zk_pair, sizeof(d.zk_pair));
            d.zk_class = zvar->zk_class;
            d.zk_steps0 = zvar->zk_steps0; d.zk_steps1 = zvar->zk_steps1;
            d.zk_splits0 = zvar->zk_splits0; d.zk_splits1 = zvar->zk_splits1;
            const int n1 = __builtin_popcount(d.zk_class & ((1u << batch) - 1u));
            splitk = n1 * d.zk_splits1 + (batch - n1) * d.zk_splits0;      // gridDim.y = every (batch, slice) pair
            d.splitk = splitk;
            batch = 1;                                                     // gridDim.z: the batch is decoded from y
        }
        d.tiles_m = cdiv(M, XL_BM); d.tiles_n = cdiv(N, XL_BN);
        static DeviceOnce attr_set;        // per instantiation and per device: opt in to the full 160 KB of LDS once
        if (!attr_set.run([]() { return hipFuncSetAttribute((const void*)gemm_bf16x3_large_kernel<LA, LB, EPI, PREC>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, XL_LDS_BYTES) == hipSuccess; })) {
            set_error("gemm: cannot reserve %d bytes of LDS", XL_LDS_BYTES);
            return ARFE_ERR_LAUNCH;
        }
        dim3 grid(d.tiles_m * d.tiles_n, splitk, batch);
        hipLaunchKernelGGL((gemm_bf16x3_large_kernel<LA, LB, EPI, PREC>), grid, dim3(512), XL_LDS_BYTES, stream, d, la, lb, epi);
        ARFE_CHECK_LAUNCH();
        return ARFE_OK;
    }
    d.splitk = splitk;
    d.tiles_m = cdiv(M, GEMM_BM); d.tiles_n = cdiv(N, GEMM_BN);
    dim3 grid(d.tiles_m * d.tiles_n, splitk, batch);
    static DeviceOnce attr_mid;        // per instantiation and per device
    if (!attr_mid.run([]() { return hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<LA, LB, EPI, PREC>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, XM_LDS_BYTES) == hipSuccess; })) {
        set_error("gemm: cannot reserve %d bytes of LDS", XM_LDS_BYTES);
        return ARFE_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((gemm_bf16x3_kernel<LA, LB, EPI, PREC>), grid, dim3(256), XM_LDS_BYTES, stream, d, la, lb, epi);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// half-operand mode (arfe_set_operand_precision(1), the reference's fp16 mode): the same tiles and planner, one f16 MFMA per
// product; `exact` (the six-MFMA fp32-grade split of the softmax logits) has no meaning there - the reference's logits are
// half products too
template <class LA, class LB, class EPI>
static inline int32_t launch_gemm_auto(int M, int N, int K, int kb_count, int batch, int splitk, LA la, LB lb, EPI epi,
                                       hipStream_t stream, bool exact = false, bool kb_inner = false,
                                       const GemmDims* zvar = nullptr) {
    if (gemm_half_operands() && !gemm_use_f32())
        return launch_gemm_prec<1>(M, N, K, kb_count, batch, splitk, la, lb, epi, stream, exact, kb_inner, zvar);
    return launch_gemm_prec<0>(M, N, K, kb_count, batch, splitk, la, lb, epi, stream, exact, kb_inner, zvar);
}

static inline int gemm_bk() { return gemm_use_f32() ? GEMM_BK : XB_BK; }

}  // namespace arfe
