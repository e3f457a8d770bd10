// Dense contractions of the AR path on the exact-fp32 MFMA GEMM (gemm_f32.hpp):
//   NonLocal2D refine fwd/bwd, AR-RFF 3x3 convs (implicit GEMM) fwd/bwd, nn.Linear fwd/bwd.
#include <mutex>
#include <algorithm>
#include "gemm_bf16x3.hpp"

namespace arfe {

// ---------------------------------------------------------------------------------------------------------
// loaders with an optional ReLU mask (value is zeroed where mask <= 0; mask shares the value's indexing)
// ---------------------------------------------------------------------------------------------------------
// ---- 3x3 conv on [R,C,7,7] RoI tiles as 9 shifted GEMMs (k-batch = tap) ------------------------------------
struct ConvGatherB {   // B(n=(r,p), k=ci) for tap kb: x[r*roi_stride + ci*49 + shift(p, tap)], 0 outside the 7x7 tile
    const float* x; int64_t roi_stride; int N, Cin;
    const float* zeros;     // 8*49 zero floats in global memory: what a padding tap reads
    static constexpr bool kfast = false;
    __device__ __forceinline__ float operator()(int, int kb, int n, int k) const {
        if (n >= N || k >= Cin) return 0.f;
        int r = n / 49, p = n - r * 49;
        int py = p / 7, px = p - py * 7;
        int yy = py + kb / 3 - 1, xx = px + kb % 3 - 1;
        if ((unsigned)yy >= 7u || (unsigned)xx >= 7u) return 0.f;
        return x[(int64_t)r * roi_stride + k * 49 + yy * 7 + xx];
    }
    struct Row { const float* xb; int py, px; bool ok; };
    __device__ __forceinline__ Row row(int, int n) const {
        Row r;
        r.ok = n < N;
        const int nn = r.ok ? n : N - 1;
        const int ri = nn / 49, p = nn - ri * 49;
        r.py = p / 7; r.px = p - r.py * 7;
        r.xb = x + (int64_t)ri * roi_stride;
        return r;
    }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int, int kb, int k0, float (&v)[8]) const {
        const int yy = r.py + kb / 3 - 1, xx = r.px + kb % 3 - 1;
        const bool inside = (unsigned)yy < 7u && (unsigned)xx < 7u;
        if (FULLK) {
            // padding taps read a block of zeros instead (pointer select: 3 selects per octet, not 8); rows past N are
            // clamped copies that only reach columns the epilogue never stores
            // (the choice is spelled as two v_cndmask: a C ternary is turned back into a predicated block around the
            // address arithmetic, which cuts the K-step into several scheduling regions)
            const uint64_t qin = (uint64_t)(uintptr_t)(r.xb + (k0 * 49 + yy * 7 + xx)), qz = (uint64_t)(uintptr_t)zeros;
            const uint64_t lanes = __builtin_amdgcn_ballot_w64(inside);
            unsigned q_lo, q_hi;
            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(q_lo) : "v"((unsigned)qz), "v"((unsigned)qin), "s"(lanes));
            asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(q_hi) : "v"((unsigned)(qz >> 32)), "v"((unsigned)(qin >> 32)), "s"(lanes));
            typedef const float __attribute__((address_space(1)))* gptr;     // global (not generic): global_load, not flat_load
            const gptr q = (gptr)(((uint64_t)q_hi << 32) | q_lo);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = q[j * 49];
        } else {
            const float* q = r.xb + (inside ? yy * 7 + xx : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const float t = q[(k < Cin ? k : Cin - 1) * 49];
                v[j] = (r.ok && inside && k < Cin) ? t : 0.f;
            }
        }
    }
};
struct EpiConv {       // y[r*roi_stride + co*49 + p] = act(v + bias[co]) * [mask > 0]
    float* y; int64_t roi_stride; const float* bias; int relu; const float* mask; int64_t mask_roi_stride;
    __device__ __forceinline__ void operator()(int z, int m, int n, float v, bool first) const {
        store_rb(z, m, n, v, first, row_bias(m));
    }
    // the per-row bias can be fetched for all rows of a thread before its first store (a load between two stores of y
    // waits for the earlier store: possible alias) - see the 256x224 kernel's epilogue
    __device__ __forceinline__ float row_bias(int m) const { return bias ? epi_ld(bias, m) : 0.f; }
    __device__ __forceinline__ void store_rb(int, int m, int n, float v, bool, float b) const {
        int r = n / 49, p = n - r * 49;
        v += b;
        if (relu) v = fmaxf(v, 0.f);
        if (mask && !(epi_ld(mask, (int64_t)r * mask_roi_stride + m * 49 + p) > 0.f)) v = 0.f;
        y[(int64_t)r * roi_stride + m * 49 + p] = v;
    }
    // one image row: the 7 consecutive pixels n .. n+6 (n a multiple of 7) of output channel m
    __device__ __forceinline__ void store_row7(int m, int n, float (&v)[7], float b) const {
        typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;
        typedef float __attribute__((ext_vector_type(2), aligned(4))) f2u;
        const int r = n / 49, p = n - r * 49;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            v[j] += b;
            if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        if (mask) {
            // read-only operand: constant address space, so the loads may move ahead of earlier stores (see epi_ld)
            typedef const f4u __attribute__((address_space(4)))* c4;
            typedef const f2u __attribute__((address_space(4)))* c2;
            const float* q = mask + (int64_t)r * mask_roi_stride + m * 49 + p;
            const f4u m0 = *(c4)(uintptr_t)q;
            const f2u m1 = *(c2)(uintptr_t)(q + 4);
            const float m2 = epi_ld(q, 6);
            v[0] = m0[0] > 0.f ? v[0] : 0.f; v[1] = m0[1] > 0.f ? v[1] : 0.f; v[2] = m0[2] > 0.f ? v[2] : 0.f;
            v[3] = m0[3] > 0.f ? v[3] : 0.f; v[4] = m1[0] > 0.f ? v[4] : 0.f; v[5] = m1[1] > 0.f ? v[5] : 0.f;
            v[6] = m2 > 0.f ? v[6] : 0.f;
        }
        float* o = y + (int64_t)r * roi_stride + m * 49 + p;
        *reinterpret_cast<f4u*>(o) = f4u{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f2u*>(o + 4) = f2u{v[4], v[5]};
        o[6] = v[6];
    }
};
template <> struct epi_has_rowbias<EpiConv> { static constexpr bool value = true; };
// The second gating conv of the AR-RFF head with the fusion folded into its epilogue (multirois_bbox_head.py:172-181):
//   lh = ReLU(v + bias)      z = ori + ori * (lw + lh)
// lh and z share the layout [R][C][49]; ori is a channel slice of the [R, 3C, 7, 7] concat (its own RoI stride).
struct EpiConvFuseZ {
    float* lh; const float* bias; const float* ori; int64_t ori_roi_stride; const float* lw; float* z; int C;
    __device__ __forceinline__ float row_bias(int m) const { return bias ? epi_ld(bias, m) : 0.f; }
    __device__ __forceinline__ void store_rb(int, int m, int n, float v, bool, float b) const {
        const int r = n / 49, p = n - r * 49;
        const int64_t o = ((int64_t)r * C + m) * 49 + p;
        const float h = fmaxf(v + b, 0.f), x0 = epi_ld(ori, (int64_t)r * ori_roi_stride + m * 49 + p);
        lh[o] = h;
        z[o] = x0 + x0 * (epi_ld(lw, o) + h);
    }
    __device__ __forceinline__ void operator()(int zz, int m, int n, float v, bool first) const { store_rb(zz, m, n, v, first, row_bias(m)); }
    __device__ __forceinline__ void store_row7(int m, int n, float (&v)[7], float b) const {
        typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;
        typedef float __attribute__((ext_vector_type(2), aligned(4))) f2u;
        typedef const f4u __attribute__((address_space(4)))* c4;
        typedef const f2u __attribute__((address_space(4)))* c2;
        const int r = n / 49, p = n - r * 49;
        const int64_t o = ((int64_t)r * C + m) * 49 + p;
        const float* qo = ori + (int64_t)r * ori_roi_stride + m * 49 + p;
        const float* qa = lw + o;
        const f4u o0 = *(c4)(uintptr_t)qo, a0 = *(c4)(uintptr_t)qa;
        const f2u o1 = *(c2)(uintptr_t)(qo + 4), a1 = *(c2)(uintptr_t)(qa + 4);
        const float o2 = epi_ld(qo, 6), a2 = epi_ld(qa, 6);
        const float xo[7] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o2};
        const float xa[7] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a2};
        float h[7], zz[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            h[j] = fmaxf(v[j] + b, 0.f);
            zz[j] = xo[j] + xo[j] * (xa[j] + h[j]);
        }
        float* ph = lh + o;
        float* pz = z + o;
        *reinterpret_cast<f4u*>(ph) = f4u{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<f2u*>(ph + 4) = f2u{h[4], h[5]};
        ph[6] = h[6];
        *reinterpret_cast<f4u*>(pz) = f4u{zz[0], zz[1], zz[2], zz[3]};
        *reinterpret_cast<f2u*>(pz + 4) = f2u{zz[4], zz[5]};
        pz[6] = zz[6];
    }
};
template <> struct epi_has_rowbias<EpiConvFuseZ> { static constexpr bool value = true; };
}  // namespace arfe
#include "conv_stage_once.hpp"
#include "conv_wgrad_stage_once.hpp"
namespace arfe {

// Weight gradient of the 3x3 conv: reduction index k' = r*56 + p with the 49 pixels of an RoI padded to 7 octets
// (p >= 49 is zero), so a k-octet never straddles two RoIs: K' = R*56.
// Weight gradient of the 3x3 convs: dWp[tap][co][ci] = sum_{r, p valid under tap} dy[r][co][p] * x[r][ci][p + shift(tap)].
// The reduction index k' is row-structured: (RoI r, pixel row among the tap's VALID output rows, 8 slots), slot j < 7 =
// pixel column j, slot 7 = zero.  A tap with dy != 0 has 6 valid rows, so its K is 6/7 of the others' (per-batch K of the
// large-tile kernel; `uniform` = every tap walks 7 rows per RoI and masks the missing one, for the other kernels).
// An octet is one contiguous 8-float window in both operands; column validity (dx != 0) is one constant mask on B.
__device__ __forceinline__ void wg_tap_rows(int z, int& py0, int& nh) {
    const int dy = z / 3 - 1;
    py0 = dy < 0 ? 1 : 0;
    nh = dy == 0 ? 7 : 6;
}
struct WgradA {        // A(m=co, k') = dy[r*roi_stride + co*49 + py*7 + j]
    const float* dy; int64_t roi_stride; int Cout, R, uniform;
    static constexpr bool kfast = true;
    __device__ __forceinline__ float operator()(int z, int, int m, int k) const {
        int py0, nh; wg_tap_rows(z, py0, nh);
        const int per = uniform ? 7 : nh;
        const int o8 = k >> 3, j = k & 7, ri = o8 / per, rw = o8 - ri * per;
        if (m >= Cout || ri >= R || rw >= nh || j >= 7) return 0.f;
        return dy[(int64_t)ri * roi_stride + m * 49 + (py0 + rw) * 7 + j];
    }
    struct Row { const float* q; bool ok; };
    __device__ __forceinline__ Row row(int, int m) const { return Row{dy + (m < Cout ? m : Cout - 1) * 49, m < Cout}; }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int z, int, int k0, float (&v)[8]) const {
        int py0, nh; wg_tap_rows(z, py0, nh);
        const int o8 = k0 >> 3;
        const int ri = (uniform || nh == 7) ? o8 / 7 : o8 / 6;
        const int rw = o8 - ri * ((uniform || nh == 7) ? 7 : 6);
        const bool ok = r.ok && ri < R && rw < nh;
        const int py = py0 + (rw < nh ? rw : 0);
        const float* q = r.q + (int64_t)(ri < R ? ri : R - 1) * roi_stride;
        // rows 0..5: the 8-float window starts at the row (its 8th float belongs to the next row and is dropped);
        // row 6: read pixels 41..48 and shift by one, so the window never leaves the channel's 49 floats
        float t[8];
        ld8_contig(q + (py < 6 ? py * 7 : 41), t);
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = ok ? (py < 6 ? t[j] : t[j + 1]) : 0.f;
        v[7] = 0.f;
    }
};
struct WgradB {        // B(n=ci, k') for tap z: x[r*roi_stride + ci*49 + (py+dy)*7 + j + dx], 0 where j + dx leaves the row
    const float* x; int64_t roi_stride; int Cin, R, uniform;
    static constexpr bool kfast = true;
    __device__ __forceinline__ float operator()(int z, int, int n, int k) const {
        int py0, nh; wg_tap_rows(z, py0, nh);
        const int per = uniform ? 7 : nh;
        const int o8 = k >> 3, j = k & 7, ri = o8 / per, rw = o8 - ri * per;
        const int xx = j + z % 3 - 1;
        if (n >= Cin || ri >= R || rw >= nh || j >= 7 || (unsigned)xx >= 7u) return 0.f;
        return x[(int64_t)ri * roi_stride + n * 49 + (py0 + rw + z / 3 - 1) * 7 + xx];
    }
    struct Row { int n; bool ok; };
    __device__ __forceinline__ Row row(int, int n) const { return Row{n < Cin ? n : Cin - 1, n < Cin}; }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int z, int, int k0, float (&v)[8]) const {
        int py0, nh; wg_tap_rows(z, py0, nh);
        const int dyy = z / 3 - 1, dx = z % 3 - 1;
        const int o8 = k0 >> 3;
        const int ri0 = (uniform || nh == 7) ? o8 / 7 : o8 / 6;
        const int rw = o8 - ri0 * ((uniform || nh == 7) ? 7 : 6);
        const bool ok = r.ok && ri0 < R && rw < nh;
        const int ri = ri0 < R ? ri0 : R - 1;
        const int py = py0 + (rw < nh ? rw : 0) + dyy;                   // source row, 0..6
        // slots whose source column leaves the row are masked, so the 8-float window may start one float early
        // (dx = -1) or overhang into the neighbouring channel as long as it stays inside the tensor slice
        const int64_t start = (int64_t)ri * roi_stride + r.n * 49 + py * 7 + dx;
        const int64_t last = (int64_t)(R - 1) * roi_stride + (int64_t)Cin * 49 - 8;
        const unsigned mask = !ok ? 0u : (dx < 0 ? 0x7eu : (dx > 0 ? 0x3fu : 0x7fu));
        if (start >= 0 && start <= last) {
            ld8_contig(x + start, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (mask >> j) & 1u ? v[j] : 0.f;
        } else {                          // first / last few floats of the slice: element-wise, clamped
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t a = start + j;
                const float t = x[a < 0 ? 0 : (a > last + 7 ? last + 7 : a)];
                v[j] = (mask >> j) & 1u ? t : 0.f;
            }
        }
    }
};

// w [Cout,Cin,3,3] -> wp[tap][co][ci] (fwd) or wd[tap][ci][co] = w[co][ci][8-tap] (transposed + flipped, dgrad)
// presplit: the elements are stored as (bf16 hi << 16 | bf16 lo) words for LoadSplitK (bf16x3 mode), else as floats
// planes (bf16x3 mode): the same two matrices once more as bf16 PLANES - ph = [hi plane 9*C*C | lo plane 9*C*C] of wp,
// pd likewise of wd - the form the stage-once conv kernel copies to LDS by LDS-DMA
// fh / fd: the same two matrices as ONE plane of IEEE halves each (half-operand mode of the stage-once conv kernel)
__global__ void conv_w_permute_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wd, int C,
                                      int presplit, unsigned short* __restrict__ ph, unsigned short* __restrict__ pd,
                                      unsigned short* __restrict__ fh, unsigned short* __restrict__ fd) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int total = C * C * 9;
    if (i >= total) return;
    int tap = i / (C * C), rem = i - tap * C * C;
    int a = rem / C, b = rem - a * C;
    float f = w[((int64_t)a * C + b) * 9 + tap];           // wp[tap][co=a][ci=b]
    float g = w[((int64_t)b * C + a) * 9 + (8 - tap)];     // wd[tap][ci=a][co=b]
    fh[i] = __builtin_bit_cast(unsigned short, (_Float16)f);
    fd[i] = __builtin_bit_cast(unsigned short, (_Float16)g);
    if (presplit) {
        const unsigned fw = presplit_word(f), gw = presplit_word(g);
        ph[i] = (unsigned short)(fw >> 16); ph[total + i] = (unsigned short)(fw & 0xffffu);
        pd[i] = (unsigned short)(gw >> 16); pd[total + i] = (unsigned short)(gw & 0xffffu);
        f = __uint_as_float(fw);
        g = __uint_as_float(gw);
    }
    wp[i] = f;
    wd[i] = g;
}
// dwp[tap][co][ci] -> dw[co][ci][tap]
__global__ void conv_dw_unpermute_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int total = C * C * 9;
    if (i >= total) return;
    int pair = i / 9, tap = i - pair * 9;
    dw[i] = dwp[(int64_t)tap * C * C + pair];
}

// ---- reductions for bias gradients ---------------------------------------------------------------------
// out[mid] = sum_{o, i} x[o*outer_stride + mid*inner + i] * [mask > 0]; one workgroup per (mid, outer chunk).
__global__ void reduce_mid_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                                  int outer, int64_t outer_stride, int inner, int outer_chunk) {
    int mid = blockIdx.x;
    int o0 = blockIdx.y * outer_chunk;
    int o1 = o0 + outer_chunk < outer ? o0 + outer_chunk : outer;
    float s = 0.f;
    if (inner <= 64) {
        // short rows (the 49 pixels of an RoI channel): the (row, pixel) pairs of the chunk are dealt flat over the
        // workgroup, so every thread has several independent loads in flight (one wave per row ran at 1.8 TB/s)
        const int n = (o1 - o0) * inner;
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const int o = e / inner, i = e - o * inner;
            const int64_t off = (int64_t)(o0 + o) * outer_stride + (int64_t)mid * inner + i;
            float v = x[off];
            if (mask && !(mask[off] > 0.f)) v = 0.f;
            s += v;
        }
    } else {
        // long rows (NonLocal, inner = L): the whole workgroup along the row
        for (int o = o0; o < o1; ++o) {
            const float* row = x + (int64_t)o * outer_stride + (int64_t)mid * inner;
            const float* mrow = mask ? mask + (int64_t)o * outer_stride + (int64_t)mid * inner : nullptr;
            for (int i = threadIdx.x; i < inner; i += blockDim.x) {
                float v = row[i];
                if (mrow && !(mrow[i] > 0.f)) v = 0.f;
                s += v;
            }
        }
    }
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
        unsafeAtomicAdd(out + mid, t);
    }
}
// column sums of a row-major [rows, cols] matrix (inner == 1 case): threads along cols, rows chunked
__global__ void colsum_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                              int rows, int cols, int row_chunk) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    int r0 = blockIdx.y * row_chunk;
    int r1 = r0 + row_chunk < rows ? r0 + row_chunk : rows;
    float s = 0.f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        float v = x[(int64_t)r * cols + c];
        if (mask && !(mask[(int64_t)r * cols + c] > 0.f)) v = 0.f;
        s += v;
    }
    unsafeAtomicAdd(out + c, s);
}

// ---- NonLocal helpers ------------------------------------------------------------------------------------
// wcat = [Wt;Wp;Wg] (3C x C), bcat = [bt;bp;bg]
__global__ void nl_pack_kernel(const float* __restrict__ wt, const float* __restrict__ wp, const float* __restrict__ wg,
                               const float* __restrict__ bt, const float* __restrict__ bp, const float* __restrict__ bg,
                               float* __restrict__ wcat, float* __restrict__ bcat, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cc = C * C;
    if (i < 3 * cc) {
        int which = i / cc, j = i - which * cc;
        wcat[i] = which == 0 ? wt[j] : (which == 1 ? wp[j] : wg[j]);
    }
    if (bcat && i < 3 * C) {
        int which = i / C, j = i - which * C;
        bcat[i] = which == 0 ? bt[j] : (which == 1 ? bp[j] : bg[j]);
    }
}
__global__ void nl_unpack_kernel(const float* __restrict__ dwcat, const float* __restrict__ dbcat, float* __restrict__ dwt,
                                 float* __restrict__ dwp, float* __restrict__ dwg, float* __restrict__ dbt,
                                 float* __restrict__ dbp, float* __restrict__ dbg, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cc = C * C;
    if (i < 3 * cc) {
        int which = i / cc, j = i - which * cc;
        float v = dwcat[i];
        if (which == 0) dwt[j] = v; else if (which == 1) dwp[j] = v; else dwg[j] = v;
    }
    if (i < 3 * C) {
        int which = i / C, j = i - which * C;
        float v = dbcat[i];
        if (which == 0) dbt[j] = v; else if (which == 1) dbp[j] = v; else dbg[j] = v;
    }
}

// in-place row softmax of S [rows, L]: one 256-thread workgroup per row, the row is held in registers
// (L <= 256*NL_ROW_REGS) so HBM sees one read and one write per element.
constexpr int NL_ROW_REGS = 20;   // up to 5120 columns
__global__ __launch_bounds__(256) void nl_softmax_rows_kernel(float* __restrict__ S, int L) {
    float* row = S + (int64_t)blockIdx.x * L;
    float v[NL_ROW_REGS];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NL_ROW_REGS; ++i) {
        int j = threadIdx.x + i * 256;
        v[i] = j < L ? row[j] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    __shared__ float red[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL_ROW_REGS; ++i) {
        int j = threadIdx.x + i * 256;
        v[i] = j < L ? expf(v[i] - mx) : 0.f;
        s += v[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < NL_ROW_REGS; ++i) {
        int j = threadIdx.x + i * 256;
        if (j < L) row[j] = v[i] * inv;
    }
}
// delta[z][i] = sum_c a[z][c][i] * b[z][c][i]  ([B, C, L] operands): the row sums sum_j P_ij dP_ij of the softmax backward
// equal dy_i . y_i (y = P g is the forward output), so they never need the L x L matrices.  64 positions x 4 channel
// quarters per workgroup.
__global__ __launch_bounds__(256) void nl_rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ delta, int C, int L) {
    __shared__ float red[3][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + li, z = blockIdx.y;
    const int cq = (C + 3) / 4, c0 = q * cq, c1 = c0 + cq < C ? c0 + cq : C;
    float s = 0.f;
    if (i < L) {
        const float* pa = a + ((int64_t)z * C + c0) * L + i;
        const float* pb = b + ((int64_t)z * C + c0) * L + i;
#pragma unroll 8
        for (int c = c0; c < c1; ++c) { s += pa[0] * pb[0]; pa += L; pb += L; }
    }
    if (q > 0) red[q - 1][li] = s;
    __syncthreads();
    if (q == 0 && i < L) delta[(int64_t)z * L + i] = ((s + red[0][li]) + red[1][li]) + red[2][li];
}

__global__ void relu_inplace_kernel(float* __restrict__ x, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = fmaxf(x[i], 0.f);
}

__global__ void mask_inplace_kernel(float* __restrict__ x, const float* __restrict__ m, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) if (!(m[i] > 0.f)) x[i] = 0.f;
}

static inline EpiPlain epi_plain(float* c, int64_t sz, int64_t ld) {
    EpiPlain e;
    e.c = c; e.sz = sz; e.ld = ld; e.bias_m = nullptr; e.bias_n = nullptr; e.residual = nullptr; e.mask = nullptr;
    e.relu = 0; e.accumulate = 0; e.atomic = 0;
    return e;
}

// GEMM whose output region (zero_elems floats from e.c) may be cut along K when the tile count alone cannot
// fill the chip: the output is zeroed and the slices add atomically (bias/residual enter with slice 0).
// out[o] = epilogue(sum_s slabs[s * stride + o]) over the GEMM's output elements o = z*sz + m*ld + n: the second half of a
// slab-mode split-K (EpiPlain::slab_stride).  One thread per 4 consecutive n (16-byte reads when the row allows).
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, int64_t stride,
                                                               int M, int N, int batch, EpiPlain e) {
    const int nq = (N + 3) / 4;
    const int64_t total = (int64_t)batch * M * nq;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % nq);
    const int m = (int)((i / nq) % M);
    const int z = (int)(i / ((int64_t)nq * M));
    const int n0 = q * 4;
    const int64_t o = z * e.sz + (int64_t)m * e.ld + n0;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = n0 + 3 < N;          // float4_u promises 4-byte alignment only: rows of any width take the 16-byte read
#pragma unroll 4
    for (int k = 0; k < nslabs; ++k) {
        const float* p = slabs + (int64_t)k * stride + o;
        if (vec) {
            const float4_u v = *reinterpret_cast<const float4_u*>(p);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        } else {
            for (int t = 0; t < 4 && n0 + t < N; ++t) s[t] += p[t];
        }
    }
    for (int t = 0; t < 4 && n0 + t < N; ++t) {
        float v = s[t];
        if (e.bias_m) v += e.bias_m[m];
        if (e.bias_n) v += e.bias_n[n0 + t];
        if (e.residual) v += e.residual[o + t];
        if (e.accumulate) v += e.c[o + t];
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.mask && !(e.mask[o + t] > 0.f)) v = 0.f;
        e.c[o + t] = v;
    }
}

// Split-K with per-slice slabs when the planner wants K slices and the scratch holds them: the GEMM stores raw partial
// sums, gemm_slab_reduce_kernel applies `e` (bias, residual, accumulate, ReLU, mask all allowed).  *done = false: nothing was
// launched (no split planned, scratch too small, exact-fp32 / forced-plan mode) and the caller takes its other path.
template <class LA, class LB>
static inline int32_t gemm_try_slabs(int M, int N, int K, int kb, int batch, LA la, LB lb, const EpiPlain& e, hipStream_t st,
                                     float* slabs, int64_t slab_floats, bool* done) {
    *done = false;
    static const char* plan_env = getenv("ARFE_GEMM_PLAN");        // a forced plan bypasses the planner replicated below
    if (!slabs || gemm_use_f32() || e.scale || plan_env) return ARFE_OK;
    const int ksteps = (kb < 1 ? 1 : kb) * cdiv(K, gemm_bk());
    // the slice count the launcher will settle on (same planner calls as launch_gemm_auto)
    const int req = pick_splitk(M, N, batch, ksteps);
    int sk_large = req;
    const bool large = plan_large_tile(M, N, batch, ksteps, &sk_large);
    int final_sk = large ? sk_large : req;
    if (final_sk > ksteps) final_sk = ksteps;
    const int nslabs = final_sk > 1 ? cdiv(ksteps, cdiv(ksteps, final_sk)) : 1;
    const int64_t region = (((int64_t)(batch - 1) * e.sz + (int64_t)(M - 1) * e.ld + N) + 3) & ~(int64_t)3;
    if (nslabs <= 1 || (int64_t)nslabs * region > slab_floats) return ARFE_OK;
    EpiPlain part = e;
    part.c = slabs; part.slab_stride = region;
    part.bias_m = part.bias_n = part.residual = part.mask = nullptr; part.relu = part.accumulate = part.atomic = 0;
    int32_t rc = launch_gemm_auto(M, N, K, kb, batch, req, la, lb, part, st);
    if (rc) return rc;
    const int64_t total = (int64_t)batch * M * ((N + 3) / 4);
    hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, (const float*)slabs, nslabs,
                       region, M, N, batch, e);
    ARFE_CHECK_LAUNCH();
    *done = true;
    return ARFE_OK;
}

// slabs / slab_floats: optional scratch for the atomic-free split-K above; otherwise (or when it is too small for the planned
// slice count) the output is zeroed and the slices add atomically.
template <class LA, class LB>
static inline int32_t gemm_autosplit(int M, int N, int K, int kb, int batch, LA la, LB lb, EpiPlain e,
                                     size_t zero_elems, hipStream_t st, float* slabs = nullptr, int64_t slab_floats = 0) {
    const int ksteps = (kb < 1 ? 1 : kb) * cdiv(K, gemm_bk());
    {
        bool done;
        int32_t rc = gemm_try_slabs(M, N, K, kb, batch, la, lb, e, st, slabs, slab_floats, &done);
        if (rc || done) return rc;
    }
    int splitk = (e.relu || e.accumulate) ? 1 : pick_splitk(M, N, batch, ksteps);
    if (splitk > 1) {
        if (hipMemsetAsync(e.c, 0, sizeof(float) * zero_elems, st) != hipSuccess) {
            set_error("gemm: memset failed");
            return ARFE_ERR_LAUNCH;
        }
        e.atomic = 1;
    }
    return launch_gemm_auto(M, N, K, kb, batch, splitk, la, lb, e, st);
}

int32_t conv3x3_unpermute_dw(int C, const float* dwp, float* dw, hipStream_t stream) {
    const int total = 9 * C * C;
    hipLaunchKernelGGL(conv_dw_unpermute_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, dwp, dw, C);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// 8*49 zero floats per device, allocated on the first conv of that device (the padding taps of ConvGatherB point here).
// Filled by a blocking copy from pageable host memory: it has completed when hipMemcpy returns, so the block is valid
// for whichever stream launches first (a hipMemset on the null stream is not ordered against torch's non-blocking
// side stream, which is where the first conv of a step runs).
static const float* conv_zeros() {
    static std::mutex mu;
    static const float* blocks[ARFE_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ARFE_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!blocks[dev]) {
        static const float host_zeros[8 * 49] = {};
        float* p = nullptr;
        if (hipMalloc(&p, sizeof(host_zeros)) != hipSuccess) return nullptr;
        if (hipMemcpy(p, host_zeros, sizeof(host_zeros), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return nullptr; }
        blocks[dev] = p;
    }
    return blocks[dev];
}

}  // namespace arfe

using namespace arfe;

// =========================================================================================================
// nn.Linear
// =========================================================================================================
extern "C" int64_t arfe_linear_workspace_bytes(int32_t R, int32_t in_f, int32_t out_f) {
    // up to 16 K-slices of the larger of the two activation-sized results (y [R,out], dx [R,in]) or of dw [out,in] when that
    // is small; capped: a call whose plan needs more falls back to atomic partial sums
    if (R < 0 || in_f <= 0 || out_f <= 0) return -1;
    int64_t m = (int64_t)R * (in_f > out_f ? in_f : out_f);
    const int64_t dw = (int64_t)in_f * out_f;
    if (dw <= ((int64_t)4 << 20) && dw > m) m = dw;
    int64_t bytes = 16 * m * (int64_t)sizeof(float);
    const int64_t cap = (int64_t)256 << 20;
    return (bytes < cap ? bytes : cap) + 64;
}

extern "C" int32_t arfe_linear_fwd(int32_t R, int32_t in_f, int32_t out_f, const float* x, const float* w,
                                   const float* b, int32_t relu, float* y, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    ARFE_CHECK_ARG(R >= 0 && in_f > 0 && out_f > 0, ARFE_ERR_ARG, "linear_fwd: bad shape R=%d in=%d out=%d", R, in_f, out_f);
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && w && y, ARFE_ERR_ARG, "linear_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    // y[R,out] = x[R,in] . w[out,in]^T : A = x (k contiguous), B(n,k) = w[n*in + k] (k contiguous)
    LoadRowMajorK la{x, 0, 0, in_f, R, in_f};
    LoadRowMajorK lb{w, 0, 0, in_f, out_f, in_f};
    int ksteps = cdiv(in_f, gemm_bk());
    int splitk = pick_splitk(R, out_f, 1, ksteps);
    EpiPlain e = epi_plain(y, 0, out_f);
    e.bias_n = b;
    if (workspace) {                 // split-K through per-slice slabs: bias + ReLU ride in the reduction, no memset
        EpiPlain fin = e;
        fin.relu = relu;
        bool done;
        int32_t rc = gemm_try_slabs(R, out_f, in_f, 1, 1, la, lb, fin, st, (float*)workspace, workspace_bytes / (int64_t)sizeof(float), &done);
        if (rc || done) return rc;
    }
    if (splitk > 1) {
        // split-K: atomic partial sums into a zeroed output, activation in a second tiny pass
        if (hipMemsetAsync(y, 0, sizeof(float) * (size_t)R * out_f, st) != hipSuccess) {
            set_error("linear_fwd: memset failed");
            return ARFE_ERR_LAUNCH;
        }
        e.atomic = 1;
        int32_t rc = launch_gemm_auto(R, out_f, in_f, 1, 1, splitk, la, lb, e, st);
        if (rc) return rc;
        if (relu) {
            int64_t n = (int64_t)R * out_f;
            hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096)), dim3(256), 0, st, y, n);
            ARFE_CHECK_LAUNCH();
        }
        return ARFE_OK;
    }
    e.relu = relu;
    return launch_gemm_auto(R, out_f, in_f, 1, 1, 1, la, lb, e, st);
}

extern "C" int32_t arfe_linear_bwd(int32_t R, int32_t in_f, int32_t out_f, const float* x, const float* w,
                                   const float* dy, float* dx, int32_t accumulate_dx, int32_t mask_dx_by_x, float* dw,
                                   float* db, void* workspace, int64_t workspace_bytes, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && in_f > 0 && out_f > 0, ARFE_ERR_ARG, "linear_bwd: bad shape");
    hipStream_t st = as_stream(stream);
    float* slabs = (float*)workspace;
    const int64_t slab_floats = workspace ? workspace_bytes / (int64_t)sizeof(float) : 0;
    bool dw_by_slabs = false;
    if (R > 0 && dw && slabs && x && dy) {
        // dw[out,in] = dy^T . x through slabs when the planner splits K and the scratch holds the slices (no pre-zeroed dw)
        LoadRowMajorI la{dy, 0, 0, out_f, out_f, R};
        LoadRowMajorI lb{x, 0, 0, in_f, in_f, R};
        EpiPlain e = epi_plain(dw, 0, in_f);
        int32_t rcw = gemm_try_slabs(out_f, in_f, R, 1, 1, la, lb, e, st, slabs, slab_floats, &dw_by_slabs);
        if (rcw) return rcw;
    }
    {
        ZeroList z;
        if (!dw_by_slabs) z.add(dw, (int64_t)out_f * in_f);
        z.add(db, out_f);
        int32_t zr = zero_buffers(z, st);
        if (zr) return zr;
    }
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && w && dy, ARFE_ERR_ARG, "linear_bwd: null pointer");
    int32_t rc;
    bool dx_done = false;
    if (dx && slabs) {
        LoadRowMajorK la{dy, 0, 0, out_f, R, out_f};
        LoadRowMajorI lb{w, 0, 0, in_f, in_f, out_f};
        EpiPlain e = epi_plain(dx, 0, in_f);
        e.accumulate = accumulate_dx;
        if (mask_dx_by_x) e.mask = x;
        rc = gemm_try_slabs(R, in_f, out_f, 1, 1, la, lb, e, st, slabs, slab_floats, &dx_done);
        if (rc) return rc;
    }
    if (dx && !dx_done) {
        // dx[R,in] = dy[R,out] . w[out,in] : A = dy (k=out contiguous), B(n=in,k=out) = w[k*in + n]
        LoadRowMajorK la{dy, 0, 0, out_f, R, out_f};
        LoadRowMajorI lb{w, 0, 0, in_f, in_f, out_f};
        int splitk = pick_splitk(R, in_f, 1, cdiv(out_f, gemm_bk()));
        EpiPlain e = epi_plain(dx, 0, in_f);
        if (splitk > 1) {
            if (!accumulate_dx && hipMemsetAsync(dx, 0, sizeof(float) * (size_t)R * in_f, st) != hipSuccess)
                return ARFE_ERR_LAUNCH;
            e.atomic = 1;
        } else {
            e.accumulate = accumulate_dx;
            if (mask_dx_by_x) e.mask = x;
        }
        rc = launch_gemm_auto(R, in_f, out_f, 1, 1, splitk, la, lb, e, st);
        if (rc) return rc;
        if (splitk > 1 && mask_dx_by_x) {
            int64_t n = (int64_t)R * in_f;
            hipLaunchKernelGGL(mask_inplace_kernel, dim3((unsigned)(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096)), dim3(256), 0, st,
                               dx, x, n);
            ARFE_CHECK_LAUNCH();
        }
    }
    if (dw && !dw_by_slabs) {
        // dw[out,in] = dy^T[out,R] . x[R,in] : A(m=out,k=r) = dy[k*out + m] (m contiguous), B(n=in,k=r) = x[k*in+n]
        LoadRowMajorI la{dy, 0, 0, out_f, out_f, R};
        LoadRowMajorI lb{x, 0, 0, in_f, in_f, R};
        int splitk = pick_splitk(out_f, in_f, 1, cdiv(R, gemm_bk()));
        EpiPlain e = epi_plain(dw, 0, in_f);
        e.atomic = splitk > 1;
        rc = launch_gemm_auto(out_f, in_f, R, 1, 1, splitk, la, lb, e, st);
        if (rc) return rc;
    }
    if (db) {
        int chunk = 16;                        // 256 workgroups at R = 1024, out_f = 1024
        dim3 grid(cdiv(out_f, 256), cdiv(R, chunk));
        hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, dy, (const float*)nullptr, db, R, out_f, chunk);
        ARFE_CHECK_LAUNCH();
    }
    return ARFE_OK;
}

// =========================================================================================================
// 3x3 conv on RoI tiles
// =========================================================================================================
extern "C" int64_t arfe_conv3x3_packed_bytes(int32_t C) {
    // wp[tap][co][ci] followed by wd[tap][ci][co] (flipped/transposed for the data gradient), then the same two matrices
    // as bf16 (hi plane | lo plane) for the LDS-DMA staging of the stage-once kernels, then once more as planes of halves
    return (int64_t)5 * 9 * C * C * sizeof(float);
}
extern "C" int64_t arfe_conv3x3_workspace_bytes(int32_t C) {
    // permuted weight-gradient accumulator of the generic path, or the per-slice slabs of the stage-once kernel
    // (256 workgroups x 9 x 128 x 64 floats = 75.5 MB whenever C is a multiple of 128)
    const int64_t slabs = C % WS_CO == 0 ? (int64_t)(256 / ((C / WS_CO) * (C / WS_CI)) > 0 ? 256 / ((C / WS_CO) * (C / WS_CI)) : 1) : 1;
    return slabs * 9 * C * C * (int64_t)sizeof(float);
}

extern "C" int32_t arfe_conv3x3_pack_weights(int32_t C, const float* w, void* packed, void* stream) {
    ARFE_CHECK_ARG(C > 0 && w && packed, ARFE_ERR_ARG, "conv3x3_pack_weights: bad arguments");
    float* wp = (float*)packed;
    float* wd = wp + (size_t)9 * C * C;
    int total = 9 * C * C;
    unsigned short* ph = (unsigned short*)(wd + (size_t)9 * C * C);
    unsigned short* pd = ph + (size_t)2 * 9 * C * C;
    unsigned short* fh = pd + (size_t)2 * 9 * C * C;
    unsigned short* fd = fh + (size_t)9 * C * C;
    hipLaunchKernelGGL(conv_w_permute_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, wp, wd, C,
                       gemm_use_f32() ? 0 : 1, ph, pd, fh, fd);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_conv3x3_fwd(int32_t R, int32_t C, const float* x, int64_t x_roi_stride, const void* packed,
                                    const float* b, int32_t relu, float* y, int64_t y_roi_stride, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0, ARFE_ERR_ARG, "conv3x3_fwd: bad shape");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && packed && y, ARFE_ERR_ARG, "conv3x3_fwd: null pointer");
    const float* wp = (const float*)packed;
    LoadSplitK la{{wp, 0, (int64_t)C * C, C, C, C}};     // A(m=co,k=ci) for tap kb, pre-split words
    ConvGatherB lb{x, x_roi_stride, R * 49, C, conv_zeros()};
    ARFE_CHECK_ARG(lb.zeros, ARFE_ERR_LAUNCH, "conv3x3: cannot allocate the padding block");
    EpiConv e{y, y_roi_stride, b, relu, nullptr, 0};
    {
        int32_t rc_so;
        const unsigned short* ph = (const unsigned short*)(wp + (size_t)2 * 9 * C * C);
        const unsigned short* fh = ph + (size_t)4 * 9 * C * C;
        if (launch_conv_stage_once(R, C, ph, fh, x, x_roi_stride, e, as_stream(stream), &rc_so)) return rc_so;
    }
    return launch_gemm_auto(C, R * 49, C, 9, 1, 1, la, lb, e, as_stream(stream), /*exact=*/false, /*kb_inner=*/true);
}

// lh = ReLU(conv(x; w) + b) and z = ori + ori * (lw + lh) in one launch: the AR-RFF head's second gating conv with the fusion
// (arfe_rff_fuse_fwd) in its epilogue.  Same results as the two calls; falls back to them where the stage-once kernel does not apply.
extern "C" int32_t arfe_conv3x3_fwd_fuse(int32_t R, int32_t C, const float* x, int64_t x_roi_stride, const void* packed,
                                         const float* b, const float* ori, int64_t ori_roi_stride, const float* lw, float* lh,
                                         float* z, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0, ARFE_ERR_ARG, "conv3x3_fwd_fuse: bad shape");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && packed && ori && lw && lh && z, ARFE_ERR_ARG, "conv3x3_fwd_fuse: null pointer");
    {
        const float* wp = (const float*)packed;
        const unsigned short* ph = (const unsigned short*)(wp + (size_t)2 * 9 * C * C);
        const unsigned short* fh = ph + (size_t)4 * 9 * C * C;
        EpiConvFuseZ e{lh, b, ori, ori_roi_stride, lw, z, C};
        int32_t rc_so;
        if (launch_conv_stage_once(R, C, ph, fh, x, x_roi_stride, e, as_stream(stream), &rc_so)) return rc_so;
    }
    int32_t rc = arfe_conv3x3_fwd(R, C, x, x_roi_stride, packed, b, 1, lh, (int64_t)C * 49, stream);
    if (rc) return rc;
    return arfe_rff_fuse_fwd(R, C, ori, ori_roi_stride, lw, lh, z, stream);
}

extern "C" int32_t arfe_conv3x3_bwd_data(int32_t R, int32_t C, const void* packed, const float* dy,
                                         int64_t dy_roi_stride, float* dx, int64_t dx_roi_stride,
                                         const float* dx_mask, int64_t dx_mask_roi_stride, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0, ARFE_ERR_ARG, "conv3x3_bwd_data: bad shape");
    if (R == 0) return ARFE_OK;
    ARFE_CHECK_ARG(packed && dy && dx, ARFE_ERR_ARG, "conv3x3_bwd_data: null pointer");
    const float* wd = (const float*)packed + (size_t)9 * C * C;
    LoadSplitK la{{wd, 0, (int64_t)C * C, C, C, C}};      // A(m=ci,k=co) = w[co][ci][8-tap], pre-split words
    ConvGatherB lb{dy, dy_roi_stride, R * 49, C, conv_zeros()};
    ARFE_CHECK_ARG(lb.zeros, ARFE_ERR_LAUNCH, "conv3x3: cannot allocate the padding block");
    EpiConv e{dx, dx_roi_stride, nullptr, 0, dx_mask, dx_mask_roi_stride};
    {
        int32_t rc_so;
        const unsigned short* pd = (const unsigned short*)((const float*)packed + (size_t)2 * 9 * C * C) + (size_t)2 * 9 * C * C;
        const unsigned short* fd = pd + (size_t)2 * 9 * C * C + (size_t)9 * C * C;
        if (launch_conv_stage_once(R, C, pd, fd, dy, dy_roi_stride, e, as_stream(stream), &rc_so)) return rc_so;
    }
    return launch_gemm_auto(C, R * 49, C, 9, 1, 1, la, lb, e, as_stream(stream), /*exact=*/false, /*kb_inner=*/true);
}

extern "C" int32_t arfe_conv3x3_bwd_weight(int32_t R, int32_t C, const float* x, int64_t x_roi_stride,
                                           const float* dy, int64_t dy_roi_stride, float* dw, float* db,
                                           void* workspace, void* stream) {
    ARFE_CHECK_ARG(R >= 0 && C > 0 && dw && db, ARFE_ERR_ARG, "conv3x3_bwd_weight: bad arguments");
    hipStream_t st = as_stream(stream);
    if (R == 0) {
        ZeroList z;
        z.add(db, C);
        z.add(dw, (int64_t)9 * C * C);
        return zero_buffers(z, st);
    }
    ARFE_CHECK_ARG(x && dy && workspace, ARFE_ERR_ARG, "conv3x3_bwd_weight: null pointer");
    float* dwp = (float*)workspace;
    int total = 9 * C * C;
    static const char* so_sw = getenv("ARFE_CONV_STAGE_ONCE");
    const bool stage_once = !(so_sw && so_sw[0] == '0') && !gemm_use_f32() && conv_wgrad_stage_once_slabs(R, C) > 0;
    {
        ZeroList z;
        z.add(db, C);
        if (!stage_once) z.add(dwp, total);           // the stage-once kernel stores whole slabs: nothing to pre-zero
        int32_t zr = zero_buffers(z, st);
        if (zr) return zr;
    }
    // K per tap: R x (valid pixel rows) x 8 slots.  With the large-tile kernel every tap gets its own length and a share
    // of the ~250 workgroups proportional to it; the other kernels walk 7 rows for every tap (`uniform`).
    const int bk = gemm_bk();
    const int steps7 = cdiv(R * 56, bk), steps6 = cdiv(R * 48, bk);
    int sk_probe = 64;
    const bool zvar = !gemm_use_f32() && plan_large_tile(C, C, 9, steps7, &sk_probe, /*caller_plans_k=*/true);
    WgradA la{dy, dy_roi_stride, C, R, zvar ? 0 : 1};
    WgradB lb{x, x_roi_stride, C, R, zvar ? 0 : 1};
    EpiPlain e = epi_plain(dwp, (int64_t)C * C, C);
    e.atomic = 1;
    int32_t rc;
    if (stage_once && launch_conv_wgrad_stage_once(R, C, x, x_roi_stride, dy, dy_roi_stride, dwp, dw, db, st, &rc)) {
        // (the nine taps in one kernel, operands staged once per pixel chunk: conv_wgrad_stage_once.hpp; it also sums db
        // and its slab reduction writes dw in the parameter's layout)
        return rc;
    } else if (zvar) {
        GemmDims zv;
        const int tiles = cdiv(C, XL_BM) * cdiv(C, XL_BN);
        const int total_steps = 3 * steps7 + 6 * steps6;
        const int budget = 252 / tiles > 9 ? 252 / tiles : 9;            // workgroups over the 9 taps (~ one per CU)
        auto slices = [&](int st_z) {
            int sp = (int)((int64_t)budget * st_z / total_steps);
            const int cap = st_z / 8 > 1 ? st_z / 8 : 1;                  // at least 8 K-steps per slice
            return sp < 1 ? 1 : (sp > cap ? cap : sp);
        };
        zv.zk_class = 0x38u;                                              // taps 3,4,5 (dy = 0): 7 pixel rows
        zv.zk_steps0 = steps6; zv.zk_splits0 = slices(steps6);
        zv.zk_steps1 = steps7; zv.zk_splits1 = slices(steps7);
        // Which (tap, slice) pairs share an XCD: the nine taps read the SAME x and dy rows of a K range, so the pairs are
        // ordered by where their slice starts in K and each XCD (workgroup id modulo 8) takes a contiguous run of that order -
        // its L2 then serves the other taps' reads of a range (class-major order: every tap's slice on another XCD, x and dy
        // crossed the fabric nine times, 843 MB per launch by the PMC counters).
        {
            struct Pair { double key; int z, s; };
            Pair pr[256];
            int np = 0;
            for (int z = 0; z < 9; ++z) {
                const int sp = ((zv.zk_class >> z) & 1u) ? zv.zk_splits1 : zv.zk_splits0;
                for (int sl = 0; sl < sp; ++sl) pr[np++] = Pair{(double)sl / sp, z, sl};
            }
            if (np <= 256 && zv.zk_splits0 < 256 && zv.zk_splits1 < 256) {
                std::stable_sort(pr, pr + np, [](const Pair& a, const Pair& b) { return a.key < b.key; });
                int start = 0;
                for (int c = 0; c < 8; ++c) {
                    const int cnt = (np - c + 7) / 8;                 // workgroups y = c, c + 8, ... < np
                    for (int r = 0; r < cnt; ++r)
                        zv.zk_pair[8 * r + c] = (unsigned short)((pr[start + r].z << 8) | pr[start + r].s);
                    start += cnt;
                }
                zv.zk_on = 2;
            }
        }
        rc = launch_gemm_auto(C, C, steps7 * bk, 1, 9, 64, la, lb, e, st, false, false, &zv);
    } else {
        const int Kp = R * 56;
        rc = launch_gemm_auto(C, C, Kp, 1, 9, pick_splitk(C, C, 9, cdiv(Kp, bk)), la, lb, e, st);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(conv_dw_unpermute_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, dwp, dw, C);
    ARFE_CHECK_LAUNCH();
    int chunk = 64;
    dim3 grid(C, cdiv(R, chunk));
    hipLaunchKernelGGL(reduce_mid_kernel, grid, dim3(256), 0, st, dy, (const float*)nullptr, db, R, dy_roi_stride, 49, chunk);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// =========================================================================================================
// NonLocal2D
// =========================================================================================================
constexpr int NL_SLABS = 8;
static inline int64_t nl_slab_region(int B, int C, int L) {
    return ((((int64_t)(B - 1) * 3 + 1) * C * L) + 3) & ~(int64_t)3;
}
static inline int64_t nl_slab_floats(int B, int C, int L) { return (int64_t)NL_SLABS * nl_slab_region(B, C, L); }
static inline float* nl_slab_base(void* workspace, int B, int C, int L) {
    int64_t f = (int64_t)2 * (3 * C * C + 3 * C) + (int64_t)B * 3 * C * L + (int64_t)B * C * L + (int64_t)B * L;
    f = (f + 3) & ~(int64_t)3;
    return (float*)workspace + f;
}
extern "C" int64_t arfe_nonlocal_workspace_bytes(int32_t B, int32_t C, int32_t L) {
    // wcat (3C*C) + bcat (3C) + dwcat (3C*C) + dbcat (3C) + dtpg [B,3C,L] + dy [B,C,L] + delta [B,L]
    // + NL_SLABS partial-sum slabs for the K = L contractions' split-K (each spans one [C, L] slice of a [B, 3C, L] tensor)
    int64_t f = (int64_t)2 * (3 * C * C + 3 * C) + (int64_t)B * 3 * C * L + (int64_t)B * C * L + (int64_t)B * L;
    f = (f + 3) & ~(int64_t)3;
    return (f + nl_slab_floats(B, C, L)) * (int64_t)sizeof(float) + 256;
}

extern "C" int32_t arfe_nonlocal_fwd(int32_t B, int32_t C, int32_t L, const float* m, const float* w_theta,
                                     const float* b_theta, const float* w_phi, const float* b_phi, const float* w_g,
                                     const float* b_g, const float* w_out, const float* b_out, float* tpg, float* P,
                                     float* y, float* bsf, void* workspace, void* stream) {
    ARFE_CHECK_ARG(B > 0 && C > 0 && L > 0, ARFE_ERR_ARG, "nonlocal_fwd: bad shape B=%d C=%d L=%d", B, C, L);
    ARFE_CHECK_ARG(L <= 256 * NL_ROW_REGS, ARFE_ERR_ARG, "nonlocal_fwd: L=%d exceeds the row-softmax limit %d", L,
                   256 * NL_ROW_REGS);
    ARFE_CHECK_ARG(m && w_theta && w_phi && w_g && w_out && tpg && P && y && bsf && workspace, ARFE_ERR_ARG,
                   "nonlocal_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    float* wcat = (float*)workspace;
    float* bcat = wcat + (size_t)3 * C * C;
    hipLaunchKernelGGL(nl_pack_kernel, dim3(cdiv(3 * C * C, 256)), dim3(256), 0, st, w_theta, w_phi, w_g, b_theta,
                       b_phi, b_g, wcat, bcat, C);
    ARFE_CHECK_LAUNCH();
    const int64_t CL = (int64_t)C * L, LL = (int64_t)L * L;
    int32_t rc;
    {   // tpg[z] [3C,L] = wcat [3C,C] . m[z] [C,L] + bcat
        LoadRowMajorK la{wcat, 0, 0, C, 3 * C, C};
        LoadRowMajorI lb{m, CL, 0, L, L, C};
        EpiPlain e = epi_plain(tpg, 3 * CL, L);
        e.bias_m = bcat;
        // theta and phi enter the logits: exact fp32 (see the S product below); 1.65 GFLOP/img
        rc = launch_gemm_auto(3 * C, L, C, 1, B, 1, la, lb, e, st, /*exact=*/true);
        if (rc) return rc;
    }
    {   // S[z][i][j] = sum_c theta[c][i] phi[c][j]
        LoadRowMajorI la{tpg, 3 * CL, 0, L, L, C};
        LoadRowMajorI lb{tpg + CL, 3 * CL, 0, L, L, C};
        EpiPlain e = epi_plain(P, LL, L);
        // the logits feed exp(): keep this one contraction on the exact-fp32 MFMA so that the softmax is
        // fp32-accurate for any weight scale (bf16x3's ~1e-5 |theta||phi| error is amplified by peaky rows)
        rc = launch_gemm_auto(L, L, C, 1, B, 1, la, lb, e, st, /*exact=*/true);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(nl_softmax_rows_kernel, dim3(B * L), dim3(256), 0, st, P, L);
    ARFE_CHECK_LAUNCH();
    {   // y[z][c][i] = sum_j g[c][j] P[i][j]
        LoadRowMajorK la{tpg + 2 * CL, 3 * CL, 0, L, C, L};
        LoadRowMajorK lb{P, LL, 0, L, L, L};
        EpiPlain e = epi_plain(y, CL, L);
        rc = gemm_autosplit(C, L, L, 1, B, la, lb, e, (size_t)B * CL, st, nl_slab_base(workspace, B, C, L), nl_slab_floats(B, C, L));
        if (rc) return rc;
    }
    {   // bsf[z] = m[z] + w_out . y[z] + b_out
        LoadRowMajorK la{w_out, 0, 0, C, C, C};
        LoadRowMajorI lb{y, CL, 0, L, L, C};
        EpiPlain e = epi_plain(bsf, CL, L);
        e.bias_m = b_out;
        e.residual = m;
        rc = launch_gemm_auto(C, L, C, 1, B, 1, la, lb, e, st);
        if (rc) return rc;
    }
    return ARFE_OK;
}

// Workspace layout shared by the three parts of the backward (arfe_nonlocal_workspace_bytes):
//   wcat [3C*C] | bcat [3C] | dwcat [3C*C] | dbcat [3C] | dtpg [B,3C,L] | dy [B,C,L] | delta [B,L]
namespace {
struct NlWs { float *wcat, *bcat, *dwcat, *dbcat, *dtpg, *dy, *delta; };
NlWs nl_ws(void* workspace, int B, int C, int L) {
    NlWs w;
    const int64_t CL = (int64_t)C * L;
    w.wcat = (float*)workspace;
    w.bcat = w.wcat + (size_t)3 * C * C;
    w.dwcat = w.bcat + (size_t)3 * C;
    w.dbcat = w.dwcat + (size_t)3 * C * C;
    w.dtpg = w.dbcat + (size_t)3 * C;
    w.dy = w.dtpg + (size_t)B * 3 * CL;
    w.delta = w.dy + (size_t)B * CL;
    return w;
}
}  // namespace

// dw_out, db_out: need only dbsf and the forward's y, i.e. nothing of the data path below
extern "C" int32_t arfe_nonlocal_bwd_wout(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* y, float* dw_out,
                                          float* db_out, void* stream) {
    ARFE_CHECK_ARG(B > 0 && C > 0 && L > 0, ARFE_ERR_ARG, "nonlocal_bwd_wout: bad shape");
    ARFE_CHECK_ARG(dbsf && y && dw_out && db_out, ARFE_ERR_ARG, "nonlocal_bwd_wout: null pointer");
    hipStream_t st = as_stream(stream);
    const int64_t CL = (int64_t)C * L;
    int32_t rc;
    {
        ZeroList z;
        z.add(dw_out, (int64_t)C * C);
        z.add(db_out, C);
        rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    const int kstepsL = cdiv(L, gemm_bk());
    // dw_out[co][ci] = sum_{z,l} dbsf[z][co][l] y[z][ci][l]   (k-batched over images, split-K)
    LoadRowMajorK la{dbsf, 0, CL, L, C, L};
    LoadRowMajorK lb{y, 0, CL, L, C, L};
    EpiPlain e = epi_plain(dw_out, 0, C);
    e.atomic = 1;
    rc = launch_gemm_auto(C, C, L, B, 1, pick_splitk(C, C, 1, B * kstepsL), la, lb, e, st);
    if (rc) return rc;
    dim3 grid(C, B);
    hipLaunchKernelGGL(reduce_mid_kernel, grid, dim3(256), 0, st, dbsf, (const float*)nullptr, db_out, B, CL, L, 1);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// the data path: dm = dbsf + [Wt;Wp;Wg]^T d(tpg); leaves d(tpg) in the workspace for arfe_nonlocal_bwd_wcat.
// parts: 1 = everything up to d(tpg) (the three projections' gradients), 2 = the last contraction (dm), 3 = both.  After part 1 the
// parameter gradients (arfe_nonlocal_bwd_wcat) have all they need: a caller starts them on another stream BEFORE part 2.
static int32_t nl_bwd_data_parts(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                                 const float* w_phi, const float* w_g, const float* w_out, const float* tpg,
                                 const float* P, const float* y, float* dP, float* dm, void* workspace,
                                 void* stream, int parts) {
    ARFE_CHECK_ARG(B > 0 && C > 0 && L > 0 && L <= 256 * NL_ROW_REGS, ARFE_ERR_ARG, "nonlocal_bwd: bad shape");
    ARFE_CHECK_ARG(dbsf && w_theta && w_phi && w_g && w_out && tpg && P && y && dP && dm && workspace, ARFE_ERR_ARG,
                   "nonlocal_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    const int64_t CL = (int64_t)C * L, LL = (int64_t)L * L;
    const NlWs ws = nl_ws(workspace, B, C, L);
    float *wcat = ws.wcat, *dtpg = ws.dtpg, *dy = ws.dy, *delta = ws.delta;
    int32_t rc;
    if (parts & 1) {
    hipLaunchKernelGGL(nl_pack_kernel, dim3(cdiv(3 * C * C, 256)), dim3(256), 0, st, w_theta, w_phi, w_g,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, wcat, (float*)nullptr, C);
    ARFE_CHECK_LAUNCH();
    {
        ZeroList z;
        z.add(dtpg, (int64_t)B * 3 * CL);
        rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    {   // dy[z][ci][l] = sum_co w_out[co][ci] dbsf[z][co][l]
        LoadRowMajorI la{w_out, 0, 0, C, C, C};
        LoadRowMajorI lb{dbsf, CL, 0, L, L, C};
        EpiPlain e = epi_plain(dy, CL, L);
        rc = launch_gemm_auto(C, L, C, 1, B, 1, la, lb, e, st);
        if (rc) return rc;
    }
    {   // dS[z][i][j] = P[i][j] * (dP[i][j] - delta[i]),  dP[i][j] = sum_c dy[c][i] g[c][j],  delta[i] = sum_j P dP = dy_i . y_i:
        // the softmax backward rides in the epilogue of the dP contraction, dP itself is never stored
        hipLaunchKernelGGL(nl_rowdot_kernel, dim3(cdiv(L, 64), B), dim3(256), 0, st, (const float*)dy, y, delta, C, L);
        ARFE_CHECK_LAUNCH();
        LoadRowMajorI la{dy, CL, 0, L, L, C};
        LoadRowMajorI lb{tpg + 2 * CL, 3 * CL, 0, L, L, C};
        EpiPlain e = epi_plain(dP, LL, L);
        e.scale = P; e.rowsub = delta; e.rowsub_sz = L;
        rc = launch_gemm_auto(L, L, C, 1, B, 1, la, lb, e, st);
        if (rc) return rc;
    }
    {   // dg[z][c][j] = sum_i dy[c][i] P[i][j]   (before dP is overwritten nothing depends on order; P is read-only)
        LoadRowMajorK la{dy, CL, 0, L, C, L};
        LoadRowMajorI lb{P, LL, 0, L, L, L};
        EpiPlain e = epi_plain(dtpg + 2 * CL, 3 * CL, L);
        rc = gemm_autosplit(C, L, L, 1, B, la, lb, e, 0, st, nl_slab_base(workspace, B, C, L), nl_slab_floats(B, C, L));
        if (rc) return rc;
    }
    {   // dtheta[z][c][i] = sum_j dS[i][j] phi[c][j]
        LoadRowMajorK la{tpg + CL, 3 * CL, 0, L, C, L};
        LoadRowMajorK lb{dP, LL, 0, L, L, L};
        EpiPlain e = epi_plain(dtpg, 3 * CL, L);
        rc = gemm_autosplit(C, L, L, 1, B, la, lb, e, 0, st, nl_slab_base(workspace, B, C, L), nl_slab_floats(B, C, L));
        if (rc) return rc;
    }
    {   // dphi[z][c][j] = sum_i dS[i][j] theta[c][i]
        LoadRowMajorK la{tpg, 3 * CL, 0, L, C, L};
        LoadRowMajorI lb{dP, LL, 0, L, L, L};
        EpiPlain e = epi_plain(dtpg + CL, 3 * CL, L);
        rc = gemm_autosplit(C, L, L, 1, B, la, lb, e, 0, st, nl_slab_base(workspace, B, C, L), nl_slab_floats(B, C, L));
        if (rc) return rc;
    }
    }
    if (parts & 2) {   // dm[z] = dbsf[z] + wcat^T . dtpg[z]
        LoadRowMajorI la{wcat, 0, 0, C, C, 3 * C};
        LoadRowMajorI lb{dtpg, 3 * CL, 0, L, L, 3 * C};
        EpiPlain e = epi_plain(dm, CL, L);
        e.residual = dbsf;
        rc = gemm_autosplit(C, L, 3 * C, 1, B, la, lb, e, (size_t)B * CL, st);
        if (rc) return rc;
    }
    return ARFE_OK;
}
extern "C" int32_t arfe_nonlocal_bwd_data(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                                          const float* w_phi, const float* w_g, const float* w_out, const float* tpg,
                                          const float* P, const float* y, float* dP, float* dm, void* workspace,
                                          void* stream) {
    return nl_bwd_data_parts(B, C, L, dbsf, w_theta, w_phi, w_g, w_out, tpg, P, y, dP, dm, workspace, stream, 3);
}
// arfe_nonlocal_bwd_data in two calls (same arguments): _dtpg leaves d(theta, phi, g) in the workspace, _dm finishes dm.
extern "C" int32_t arfe_nonlocal_bwd_dtpg(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                                          const float* w_phi, const float* w_g, const float* w_out, const float* tpg,
                                          const float* P, const float* y, float* dP, float* dm, void* workspace,
                                          void* stream) {
    return nl_bwd_data_parts(B, C, L, dbsf, w_theta, w_phi, w_g, w_out, tpg, P, y, dP, dm, workspace, stream, 1);
}
extern "C" int32_t arfe_nonlocal_bwd_dm(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                                        const float* w_phi, const float* w_g, const float* w_out, const float* tpg,
                                        const float* P, const float* y, float* dP, float* dm, void* workspace,
                                        void* stream) {
    return nl_bwd_data_parts(B, C, L, dbsf, w_theta, w_phi, w_g, w_out, tpg, P, y, dP, dm, workspace, stream, 2);
}

// dw / db of theta, phi, g from the d(tpg) the data path left in the workspace
extern "C" int32_t arfe_nonlocal_bwd_wcat(int32_t B, int32_t C, int32_t L, const float* m, float* dw_theta, float* db_theta,
                                          float* dw_phi, float* db_phi, float* dw_g, float* db_g, void* workspace,
                                          void* stream) {
    ARFE_CHECK_ARG(B > 0 && C > 0 && L > 0, ARFE_ERR_ARG, "nonlocal_bwd_wcat: bad shape");
    ARFE_CHECK_ARG(m && dw_theta && db_theta && dw_phi && db_phi && dw_g && db_g && workspace, ARFE_ERR_ARG,
                   "nonlocal_bwd_wcat: null pointer");
    hipStream_t st = as_stream(stream);
    const int64_t CL = (int64_t)C * L;
    const NlWs ws = nl_ws(workspace, B, C, L);
    float *dwcat = ws.dwcat, *dbcat = ws.dbcat, *dtpg = ws.dtpg;
    int32_t rc;
    {
        ZeroList z;
        z.add(dwcat, (int64_t)3 * C * C + 3 * C);
        rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    const int kstepsL = cdiv(L, gemm_bk());
    {   // dwcat[3C][C] = sum_{z,l} dtpg[z][.][l] m[z][.][l] ; dbcat = rowsum(dtpg)
        LoadRowMajorK la{dtpg, 0, 3 * CL, L, 3 * C, L};
        LoadRowMajorK lb{m, 0, CL, L, C, L};
        EpiPlain e = epi_plain(dwcat, 0, C);
        // per-slice slabs + fixed-order reduce (the data path, the slabs' other user, is through when this runs): theta's and
        // phi's weight gradients are sums under heavy cancellation, and atomic partial sums made them differ by 1e-5 run to run
        bool by_slabs = false;
        rc = gemm_try_slabs(3 * C, C, L, B, 1, la, lb, e, st, nl_slab_base(workspace, B, C, L), nl_slab_floats(B, C, L), &by_slabs);
        if (rc) return rc;
        if (!by_slabs) {
            e.atomic = 1;
            rc = launch_gemm_auto(3 * C, C, L, B, 1, pick_splitk(3 * C, C, 1, B * kstepsL), la, lb, e, st);
            if (rc) return rc;
        }
        dim3 grid(3 * C, B);
        hipLaunchKernelGGL(reduce_mid_kernel, grid, dim3(256), 0, st, dtpg, (const float*)nullptr, dbcat, B, 3 * CL, L, 1);
        ARFE_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(nl_unpack_kernel, dim3(cdiv(3 * C * C, 256)), dim3(256), 0, st, dwcat, dbcat, dw_theta, dw_phi, dw_g,
                       db_theta, db_phi, db_g, C);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_nonlocal_bwd(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* m,
                                     const float* w_theta, const float* w_phi, const float* w_g, const float* w_out,
                                     const float* tpg, const float* P, const float* y, float* dP, float* dm,
                                     float* dw_theta, float* db_theta, float* dw_phi, float* db_phi, float* dw_g,
                                     float* db_g, float* dw_out, float* db_out, void* workspace, void* stream) {
    int32_t rc = arfe_nonlocal_bwd_wout(B, C, L, dbsf, y, dw_out, db_out, stream);
    if (rc) return rc;
    rc = arfe_nonlocal_bwd_data(B, C, L, dbsf, w_theta, w_phi, w_g, w_out, tpg, P, y, dP, dm, workspace, stream);
    if (rc) return rc;
    return arfe_nonlocal_bwd_wcat(B, C, L, m, dw_theta, db_theta, dw_phi, db_phi, dw_g, db_g, workspace, stream);
}
