// Weight gradient of the 3x3 conv on [R, C, 7, 7] RoI tiles, "stage once" form:
//   dWp[tap][co][ci] = sum_{r, py, px} dy[r][co][py][px] * x[r][ci][py + dy(tap)][px + dx(tap)]        (0 outside the 7x7)
// The generic path runs nine GEMMs (one per tap) that each gather, split and store BOTH operands again; that staging, not
// the MFMAs, bounds it.  Here a workgroup owns a 128 (co) x 64 (ci) block of ALL NINE taps (9 x 4 accumulator blocks per
// wave) over a slice of the pixel rows, and per K-step of four pixel rows (k-octet = one row of 7 pixels + a zero slot):
//   * dy rows are staged once: [128 co][4 octets];
//   * x rows once with a one-row halo, in THREE column-shifted copies made in registers before the split
//     (dx = -1, 0, +1; zeros shifted in): [3][64 ci][6 octets].  A tap's row shift dy is then a whole k-octet
//     (octet lq + dy + 1 of copy dx + 1) and a lane whose shifted row leaves the RoI reads a zero octet.
// Partial sums of the row slices: every workgroup stores its 9 x 128 x 64 block into ITS slice's slab of the workspace
// (plain stores) and conv_wgrad_reduce_kernel sums the slabs straight into dw[co][ci][tap] - 75 MB of float atomics into
// a zeroed dWp took 50 of the kernel's 240 us (tools/probe/conv_dbg.sh), and the sum is now in a fixed order.
// The bias gradient db[co] = sum_{r, p} dy[r][co][p] rides along: the threads that stage the dy rows of the ci-block-0
// workgroups add up what they fetch (fp32, before the split) and add their totals to the zeroed db once at the end, so
// the separate 51 MB reduction pass over dy is not needed.
#pragma once
#include "gemm_bf16x3.hpp"

namespace arfe {

constexpr int WS_CO = 128, WS_CI = 64;
constexpr int WS_A_LD = 32;                         // dy image row: 4 octets (64 B), swizzled like the GEMM images
constexpr int WS_B_LD = 48;                         // x image row: 6 octets (96 B)
constexpr int WS_A_IMG = WS_CO * WS_A_LD;           // elements per (hi | lo) image
constexpr int WS_B_IMG = WS_CI * WS_B_LD;           // per dx copy and (hi | lo)
constexpr int WS_BUF = 2 * WS_A_IMG + 6 * WS_B_IMG; // one buffer: A hi, A lo, B[3][hi, lo]
constexpr int WS_ZERO = WS_B_IMG + 16 * WS_B_LD + 8; // zero region: a lane that selects it reads (hi, lo) x (j = 0, 1) at the same
                                                     // constant offsets as from an image
constexpr int WS_LDS_BYTES = (2 * WS_BUF + WS_ZERO) * 2;

struct WgradStageDims { int R, C, rows_total, rows_per_slice, tiles_ci, dbg; };   // dbg: probe switches (ARFE_CONV_DBG)

// PREC == 1: half-operand mode (gemm_bf16x3.hpp) - operands converted to half, one f16 MFMA per block, "lo" images unused
template <int PREC>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_stage_once_kernel(WgradStageDims d, const float* __restrict__ x,
                                                                          int64_t x_roi_stride, const float* __restrict__ dy,
                                                                          int64_t dy_roi_stride, float* __restrict__ dwp,
                                                                          float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) unsigned short ws_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    // grid = (row slices, output blocks): workgroups go round-robin to the 8 XCDs by their linear id, so with the slice as
    // the fast index the eight output blocks of one row slice share an XCD and its L2 serves their common dy / x rows
    const int co0 = ((int)blockIdx.y / d.tiles_ci) * WS_CO, ci0 = ((int)blockIdx.y % d.tiles_ci) * WS_CI;
    const int row_begin = blockIdx.x * d.rows_per_slice;
    int row_end = row_begin + d.rows_per_slice;
    if (row_end > d.rows_total) row_end = d.rows_total;
    if (row_begin >= row_end) return;
    const int nsteps = (row_end - row_begin + 3) >> 2;
    auto img_a = [&](int buf, int lo) -> unsigned short* { return ws_lds + buf * WS_BUF + lo * WS_A_IMG; };
    auto img_b = [&](int buf, int dxi, int lo) -> unsigned short* {
        return ws_lds + buf * WS_BUF + 2 * WS_A_IMG + (dxi * 2 + lo) * WS_B_IMG;
    };
    unsigned short* zero_base = ws_lds + 2 * WS_BUF;
    for (int i = tid; i < WS_ZERO; i += 512) zero_base[i] = 0;

    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging maps ----
    // the pixel row a thread stages is WAVE-UNIFORM (dy: waves 2q, 2q+1 take octet q of the 128 channels; x: wave o takes
    // octet o of the 64 channels), so the row's RoI / py arithmetic, its validity and the "row 6" window shift are scalar
    // work and uniform branches instead of per-lane selects
    const int a_co = tid & 127, a_q = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int a_off = a_co * WS_A_LD + 8 * (a_q ^ xs_swz(a_co));
    const bool b_on = tid < WS_CI * 6;
    const int b_ci = tid & 63, b_o = __builtin_amdgcn_readfirstlane(b_on ? tid >> 6 : 0);   // x: octet o = row (first row - 1 + o)
    const int b_off = b_ci * WS_B_LD + 8 * b_o;
    float ra[8], rb[8];
    float bias_sum = 0.f;                                    // this thread's share of db[co0 + a_co] (ci block 0 only)
    auto fetch = [&](int step) {
        const int g0 = row_begin + 4 * step;
        {
            const int g = g0 + a_q;
            const bool ok = g < row_end;
            const int gc = ok ? g : row_end - 1;
            const int r = gc / 7, py = gc - r * 7;
            // rows 0..5: the 8-float window starts at the row; row 6: read pixels 41..48 and shift by one so that the
            // window never leaves the channel's 49 floats
            const float* q = dy + (int64_t)r * dy_roi_stride + (co0 + a_co) * 49 + (py < 6 ? py * 7 : 41);
            float t[8];
            ld8_contig(q, t);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 7; ++j) ra[j] = 0.f;
            } else if (py < 6) {
#pragma unroll
                for (int j = 0; j < 7; ++j) ra[j] = t[j];
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) ra[j] = t[j + 1];
            }
            ra[7] = 0.f;
            bias_sum += ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((ra[4] + ra[5]) + ra[6]);
        }
        if (b_on) {
            const int g = g0 - 1 + b_o;
            const bool ok = g >= row_begin - 1 && g >= 0 && g < d.rows_total;
            const int gc = g < 0 ? 0 : (g >= d.rows_total ? d.rows_total - 1 : g);
            const int r = gc / 7, py = gc - r * 7;
            const float* q = x + (int64_t)r * x_roi_stride + (ci0 + b_ci) * 49 + (py < 6 ? py * 7 : 41);
            float t[8];
            ld8_contig(q, t);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 7; ++j) rb[j] = 0.f;
            } else if (py < 6) {
#pragma unroll
                for (int j = 0; j < 7; ++j) rb[j] = t[j];
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) rb[j] = t[j + 1];
            }
            rb[7] = 0.f;
        }
    };
    auto stash = [&](int buf) {
        bf16x8 hi, lo;
        if constexpr (PREC == 1) { hi = cvt8_f16(ra); lo = hi; }
        else split8(ra, hi, lo);
        *reinterpret_cast<bf16x8*>(img_a(buf, 0) + a_off) = hi;
        if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_a(buf, 1) + a_off) = lo;
        if (b_on) {
            // three column-shifted copies: slot px of copy dx holds x[px + dx] (0 where px + dx leaves the row; slot 7 = 0).
            // The split is element-wise, so the row is split ONCE and the packed bf16 words are shifted (v_alignbit):
            // copy -1 = {0, e0..e5, 0}, copy +1 = {e1..e6, 0, 0}.
            if constexpr (PREC == 1) { hi = cvt8_f16(rb); lo = hi; }
            else split8(rb, hi, lo);
            *reinterpret_cast<bf16x8*>(img_b(buf, 1, 0) + b_off) = hi;
            if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_b(buf, 1, 1) + b_off) = lo;
            auto shifted = [&](const bf16x8& v, bf16x8& m1, bf16x8& p1) {
                const u32x4 w = __builtin_bit_cast(u32x4, v);
                const unsigned t1 = __builtin_amdgcn_alignbit(w[1], w[0], 16), t2 = __builtin_amdgcn_alignbit(w[2], w[1], 16),
                               t3 = __builtin_amdgcn_alignbit(w[3], w[2], 16);
                const u32x4 a = {w[0] << 16, t1, t2, w[2] >> 16}, b = {t1, t2, t3, 0u};
                m1 = __builtin_bit_cast(bf16x8, a);
                p1 = __builtin_bit_cast(bf16x8, b);
            };
            bf16x8 hm, hp, lm, lp;
            shifted(hi, hm, hp);
            *reinterpret_cast<bf16x8*>(img_b(buf, 0, 0) + b_off) = hm;
            *reinterpret_cast<bf16x8*>(img_b(buf, 2, 0) + b_off) = hp;
            if constexpr (PREC == 0) {
                shifted(lo, lm, lp);
                *reinterpret_cast<bf16x8*>(img_b(buf, 0, 1) + b_off) = lm;
                *reinterpret_cast<bf16x8*>(img_b(buf, 2, 1) + b_off) = lp;
            }
        }
    };
    // ---- fragment geometry ----
    const int a_frag = 8 * (lq ^ xs_swz(l15));           // A rows: wm*32 + i*16 + l15 (bases are multiples of 16)
    const int b_frag = (wn * 32 + l15) * WS_B_LD + 8 * lq;  // B rows: wn*32 + j*16 + l15, octet lq + dy + 1
    auto mma = [&](int buf, int step) {
        // this lane's k-octet is pixel row g = row_begin + 4*step + lq: a tap with dy is valid when py + dy stays in [0, 7)
        const int g = row_begin + 4 * step + lq;
        const int py = g % 7;
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = (wm * 32 + i * 16 + l15) * WS_A_LD + a_frag;
            ah[i] = *reinterpret_cast<const bf16x8*>(img_a(buf, 0) + off);
            if constexpr (PREC == 0) al[i] = *reinterpret_cast<const bf16x8*>(img_a(buf, 1) + off);
            else al[i] = ah[i];
        }
        // B fragments of tap t + 1 are read before the twelve MFMAs of tap t are issued (register double buffer)
        bf16x8 bh[2][2], bl[2][2];
        auto read_b = [&](int t, int slot) {
            const int dyy = t / 3 - 1, dxi = t % 3;
            const bool ok = (unsigned)(py + dyy) < 7u;
            // ONE address select per tap: hi / lo and the two column blocks sit at constant offsets from it (ds_read
            // immediates), in an image as in the zero region
            const unsigned short* p = ok ? img_b(buf, dxi, 0) + b_frag + 8 * (dyy + 1) : zero_base;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[slot][j] = *reinterpret_cast<const bf16x8*>(p + j * 16 * WS_B_LD);
                if constexpr (PREC == 0) bl[slot][j] = *reinterpret_cast<const bf16x8*>(p + WS_B_IMG + j * 16 * WS_B_LD);
                else bl[slot][j] = bh[slot][j];
            }
        };
        read_b(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) read_b(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PREC == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[i]),
                                                                              __builtin_bit_cast(f16x8, bh[t & 1][j]), acc[t][i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[t & 1][j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[t & 1][j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[t & 1][j], acc[t][i][j], 0, 0, 0);
            }
        }
    };

    fetch(0);
    stash(0);
    if (nsteps > 1) fetch(1);
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps && !(d.dbg & 4)) stash(buf ^ 1);
        if (s + 2 < nsteps && !(d.dbg & 4)) fetch(s + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (!(d.dbg & 2)) mma(buf, s);
        __syncthreads();
        buf ^= 1;
    }

    if (db && ci0 == 0) unsafeAtomicAdd(db + co0 + a_co, bias_sum);
    if (d.dbg & 1) { if (acc[0][0][0][0] == 123.456f) dwp[0] = acc[1][1][1][1]; return; }
    // ---- epilogue: D layout lane -> column l15 (ci), rows 4*lq + r (co) ----
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = ci0 + wn * 32 + j * 16 + l15;
                float* slab = dwp + (int64_t)blockIdx.x * 9 * d.C * d.C;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = co0 + wm * 32 + i * 16 + 4 * lq + r;
                    slab[((int64_t)t * d.C + m) * d.C + n] = acc[t][i][j][r];
                }
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same block (128 co x 64 ci x nine taps per workgroup) with the K axis in IMAGE-ROW-MAJOR order.  A K-step is
// image row py of FOUR RoIs (k-octet j = RoI 4G + j: 7 pixels + a zero slot) instead of four consecutive rows of one RoI, and
// the steps walk py = 0..6 of a group of four RoIs before the next group.  Then
//   * whether a tap's source row py + dy exists is the same for all four octets: the three taps that look up are SKIPPED at
//     py = 0 and the three that look down at py = 6 (wave-uniform branches) - 6 of 63 (tap, row) pairs, 9.5 % of the MFMAs,
//     which the row-of-one-RoI order had to issue on zero octets - and there is no per-lane address select left;
//   * the x rows live in a ROLLING window: image rows py-1, py, py+1 of the four RoIs are three of four row slots per RoI
//     (slot = global step & 3, octet = slot * 4 + j), each step stages ONE new row per RoI (row step + 2) instead of four rows
//     plus two halo rows: a third less x staging (loads, splits, the three column-shifted copies).
// LDS: A as before (2 x (hi, lo) x 128 x 64 B = 32 KB); B = 3 shifted copies x (hi, lo) x 64 ci x 16 octets, 288-byte rows
// (conflict-free b128 fragment reads for lane = (ci row, RoI)): 110.6 KB.
constexpr int W2_B_LD = 144;                        // x image row: 16 octets + 16 pad elements (72 dwords: see above)
constexpr int W2_B_IMG = WS_CI * W2_B_LD;
constexpr int W2_LDS_BYTES = (2 * 2 * WS_A_IMG + 6 * W2_B_IMG) * 2;

struct Wgrad2Dims { int R, C, steps_total, steps_per_slice, tiles_ci; };

template <int PREC>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_rows_kernel(Wgrad2Dims d, const float* __restrict__ x, int64_t x_roi_stride,
                                                                    const float* __restrict__ dy, int64_t dy_roi_stride,
                                                                    float* __restrict__ dwp, float* __restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) unsigned short w2_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    const int co0 = ((int)blockIdx.y / d.tiles_ci) * WS_CO, ci0 = ((int)blockIdx.y % d.tiles_ci) * WS_CI;
    const int n_begin = blockIdx.x * d.steps_per_slice;
    int n_end = n_begin + d.steps_per_slice;
    if (n_end > d.steps_total) n_end = d.steps_total;
    if (n_begin >= n_end) return;
    const int nsteps = n_end - n_begin;
    auto img_a = [&](int buf, int lo) -> unsigned short* { return w2_lds + (buf * 2 + lo) * WS_A_IMG; };
    auto img_b = [&](int dxi, int lo) -> unsigned short* { return w2_lds + 4 * WS_A_IMG + (dxi * 2 + lo) * W2_B_IMG; };

    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging maps: the RoI (k-octet) a thread stages is wave-uniform ----
    const int a_co = tid & 127, a_j = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int a_off = a_co * WS_A_LD + 8 * (a_j ^ xs_swz(a_co));
    const bool b_on = tid < WS_CI * 4;
    const int b_ci = tid & 63, b_j = __builtin_amdgcn_readfirstlane(b_on ? tid >> 6 : 0);
    const int b_off = b_ci * W2_B_LD + 8 * b_j;          // + 32 * slot
    float ra[8], rb[8];
    float bias_sum = 0.f;
    // one image row of 7 floats (+ zero slot) of channel `ch` of RoI r at row py; zeros when !ok
    auto load_row = [&](const float* base, int64_t roi_stride, int ch, int r, int py, bool ok, float (&v)[8]) {
        const float* q = base + (int64_t)r * roi_stride + ch * 49 + (py < 6 ? py * 7 : 41);
        float t[8];
        ld8_contig(q, t);
        if (!ok) {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = 0.f;
        } else if (py < 6) {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = t[j];
        } else {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = t[j + 1];
        }
        v[7] = 0.f;
    };
    auto fetch_a = [&](int n) {          // dy row of global step n (only steps of this slice are ever staged)
        const int G = n / 7, py = n - G * 7, r = 4 * G + a_j;
        const bool ok = n < n_end && r < d.R;
        load_row(dy, dy_roi_stride, co0 + a_co, ok ? r : 0, py, ok, ra);
        bias_sum += ((ra[0] + ra[1]) + (ra[2] + ra[3])) + ((ra[4] + ra[5]) + ra[6]);
    };
    auto fetch_b_to = [&](int n, float (&v)[8]) {   // x row of global step n: any row of the tensor (the window reaches one row past the slice)
        if (!b_on) return;
        const bool in = n >= 0 && n < d.steps_total;
        const int nc = in ? n : 0;
        const int G = nc / 7, py = nc - G * 7, r = 4 * G + b_j;
        const bool ok = in && r < d.R;
        load_row(x, x_roi_stride, ci0 + b_ci, ok ? r : 0, py, ok, v);
    };
    auto fetch_b = [&](int n) { fetch_b_to(n, rb); };
    auto stash_a = [&](int buf) {
        bf16x8 hi, lo;
        if constexpr (PREC == 1) { hi = cvt8_f16(ra); lo = hi; }
        else split8(ra, hi, lo);
        *reinterpret_cast<bf16x8*>(img_a(buf, 0) + a_off) = hi;
        if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_a(buf, 1) + a_off) = lo;
    };
    auto stash_b_from = [&](int n, const float (&v)[8]) {          // into row slot n & 3 (the slot of row n - 4, last read two steps ago)
        if (!b_on) return;
        const int o = b_off + 32 * (n & 3);
        bf16x8 hi, lo;
        if constexpr (PREC == 1) { hi = cvt8_f16(v); lo = hi; }
        else split8(v, hi, lo);
        *reinterpret_cast<bf16x8*>(img_b(1, 0) + o) = hi;
        if constexpr (PREC == 0) *reinterpret_cast<bf16x8*>(img_b(1, 1) + o) = lo;
        auto shifted = [&](const bf16x8& v, bf16x8& m1, bf16x8& p1) {        // copy -1 = {0, e0..e5, 0}, copy +1 = {e1..e6, 0, 0}
            const u32x4 w = __builtin_bit_cast(u32x4, v);
            const unsigned t1 = __builtin_amdgcn_alignbit(w[1], w[0], 16), t2 = __builtin_amdgcn_alignbit(w[2], w[1], 16),
                           t3 = __builtin_amdgcn_alignbit(w[3], w[2], 16);
            const u32x4 a = {w[0] << 16, t1, t2, w[2] >> 16}, b = {t1, t2, t3, 0u};
            m1 = __builtin_bit_cast(bf16x8, a);
            p1 = __builtin_bit_cast(bf16x8, b);
        };
        bf16x8 hm, hp, lm, lp;
        shifted(hi, hm, hp);
        *reinterpret_cast<bf16x8*>(img_b(0, 0) + o) = hm;
        *reinterpret_cast<bf16x8*>(img_b(2, 0) + o) = hp;
        if constexpr (PREC == 0) {
            shifted(lo, lm, lp);
            *reinterpret_cast<bf16x8*>(img_b(0, 1) + o) = lm;
            *reinterpret_cast<bf16x8*>(img_b(2, 1) + o) = lp;
        }
    };
    auto stash_b = [&](int n) { stash_b_from(n, rb); };
    // ---- fragment geometry ----
    const int a_frag = 8 * (lq ^ xs_swz(l15));                 // A rows: wm*32 + i*16 + l15
    const int b_frag = (wn * 32 + l15) * W2_B_LD + 8 * lq;     // B rows: wn*32 + j*16 + l15, octet slot*4 + lq
    auto mma = [&](int buf, int n) {
        const int py = n % 7;                                  // wave-uniform
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = (wm * 32 + i * 16 + l15) * WS_A_LD + a_frag;
            ah[i] = *reinterpret_cast<const bf16x8*>(img_a(buf, 0) + off);
            if constexpr (PREC == 0) al[i] = *reinterpret_cast<const bf16x8*>(img_a(buf, 1) + off);
            else al[i] = ah[i];
        }
        bf16x8 bh[2][2], bl[2][2];
        auto read_b = [&](int t, int slot) {                   // fragments of tap t: row n + dy in its row slot, copy dx + 1
            const int dyy = t / 3 - 1, dxi = t % 3;
            const unsigned short* p = img_b(dxi, 0) + b_frag + 32 * ((n + dyy) & 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[slot][j] = *reinterpret_cast<const bf16x8*>(p + j * 16 * W2_B_LD);
                if constexpr (PREC == 0) bl[slot][j] = *reinterpret_cast<const bf16x8*>(p + W2_B_IMG + j * 16 * W2_B_LD);
                else bl[slot][j] = bh[slot][j];
            }
        };
        read_b(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) read_b(t + 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if ((t < 3 && py == 0) || (t >= 6 && py == 6)) continue;          // the source row is outside the 7x7 map for all four RoIs
            if constexpr (PREC == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[i]),
                                                                              __builtin_bit_cast(f16x8, bh[t & 1][j]), acc[t][i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[t & 1][j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[t & 1][j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[t & 1][j], acc[t][i][j], 0, 0, 0);
            }
        }
    };

    // ---- prologue: the window of the first step (rows n-1, n, n+1), its dy row, the registers of the next stashes ----
    {
        float w0[8], w1[8];                       // three rows in flight, then three stashes: one memory round trip
        fetch_b_to(n_begin - 1, w0);
        fetch_b_to(n_begin, w1);
        fetch_b(n_begin + 1);
        fetch_a(n_begin);
        stash_b_from(n_begin - 1, w0);
        stash_b_from(n_begin, w1);
        stash_b(n_begin + 1);
    }
    stash_a(0);
    if (nsteps > 1) fetch_a(n_begin + 1);
    fetch_b(n_begin + 2);
    __syncthreads();
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int n = n_begin + s;
        if (s + 1 < nsteps) {
            stash_a(buf ^ 1);                     // dy row n + 1
            stash_b(n + 2);                       // x row n + 2: the third row of step n + 1's window
            if (s + 2 < nsteps) fetch_a(n + 2);
            fetch_b(n + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(buf, n);
        __syncthreads();
        buf ^= 1;
    }

    if (db && ci0 == 0) unsafeAtomicAdd(db + co0 + a_co, bias_sum);
    // ---- epilogue: D layout lane -> column l15 (ci), rows 4*lq + r (co) ----
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = ci0 + wn * 32 + j * 16 + l15;
                float* slab = dwp + (int64_t)blockIdx.x * 9 * d.C * d.C;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = co0 + wm * 32 + i * 16 + 4 * lq + r;
                    slab[((int64_t)t * d.C + m) * d.C + n] = acc[t][i][j][r];
                }
            }
}

// dw[co][ci][tap] = sum over the slabs of slab[s][tap][co][ci]: four ci per thread (16-byte reads along a slab row)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, int C,
                                                                float* __restrict__ dw) {
    const int64_t per = (int64_t)9 * C * C;
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (q >= per) return;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p = slabs + q;
    int k = 0;
    for (; k + 8 <= nslabs; k += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f4*>(p + (int64_t)(k + u) * per);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nslabs; ++k) s += *reinterpret_cast<const f4*>(p + (int64_t)k * per);
    const int tap = (int)(q / ((int64_t)C * C));
    const int64_t pair = q - (int64_t)tap * C * C;          // co * C + ci (C is a multiple of 4: the quad stays in one row)
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[(pair + e) * 9 + tap] = s[e];
}

// slabs the launch below will write for (R, C), 0 when the kernel does not apply: the caller sizes the workspace with it
// ARFE_WGRAD_ROWS=0: the first-generation K order (four consecutive rows of one RoI per K-step)
static inline bool wgrad_rows_order() {
    static const bool v = []() { const char* e = getenv("ARFE_WGRAD_ROWS"); return !(e && e[0] == '0'); }();
    return v;
}
static inline void wgrad_rows_plan(int R, int C, Wgrad2Dims& d, int& nslices) {
    d.R = R; d.C = C; d.tiles_ci = C / WS_CI;
    d.steps_total = 7 * cdiv(R, 4);
    const int tiles = (C / WS_CO) * d.tiles_ci;
    nslices = 256 / tiles;
    if (nslices < 1) nslices = 1;
    d.steps_per_slice = cdiv(d.steps_total, nslices);
    nslices = cdiv(d.steps_total, d.steps_per_slice);
}
static inline int conv_wgrad_stage_once_slabs(int R, int C) {
    if (C % WS_CO != 0 || R < 64) return 0;
    if (wgrad_rows_order()) {
        Wgrad2Dims d; int ns;
        wgrad_rows_plan(R, C, d, ns);
        return ns;
    }
    const int tiles = (C / WS_CO) * (C / WS_CI);
    int nslices = 256 / tiles;
    if (nslices < 1) nslices = 1;
    const int rows_total = R * 7;
    const int rows_per_slice = ((cdiv(rows_total, nslices) + 3) / 4) * 4;
    return cdiv(rows_total, rows_per_slice);
}

// -> true when the kernel ran (C a multiple of 128, bf16x3 mode, ARFE_CONV_STAGE_ONCE != 0): dw is complete (slabs in
// `dwp`, which must hold conv_wgrad_stage_once_slabs(R, C) x 9 C C floats); db must be zeroed before and receives the
// bias gradient
static inline bool launch_conv_wgrad_stage_once(int R, int C, const float* x, int64_t x_roi_stride, const float* dy,
                                                int64_t dy_roi_stride, float* dwp, float* dw, float* db, hipStream_t stream,
                                                int32_t* rc) {
    static const char* sw = getenv("ARFE_CONV_STAGE_ONCE");
    *rc = ARFE_OK;
    if ((sw && sw[0] == '0') || gemm_use_f32() || C % WS_CO != 0 || R < 64) return false;
    if (wgrad_rows_order()) {
        static DeviceOnce attr2;
        if (!attr2.run([]() { return hipFuncSetAttribute((const void*)conv3x3_wgrad_rows_kernel<0>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES) == hipSuccess &&
                                     hipFuncSetAttribute((const void*)conv3x3_wgrad_rows_kernel<1>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES) == hipSuccess; })) {
            set_error("conv3x3 wgrad: cannot reserve %d bytes of LDS", W2_LDS_BYTES);
            *rc = ARFE_ERR_LAUNCH;
            return true;
        }
        Wgrad2Dims d2; int ns;
        wgrad_rows_plan(R, C, d2, ns);
        const int tiles2 = (C / WS_CO) * d2.tiles_ci;
        if (gemm_half_operands())
            hipLaunchKernelGGL(conv3x3_wgrad_rows_kernel<1>, dim3(ns, tiles2), dim3(512), W2_LDS_BYTES, stream, d2, x, x_roi_stride, dy,
                               dy_roi_stride, dwp, db);
        else
            hipLaunchKernelGGL(conv3x3_wgrad_rows_kernel<0>, dim3(ns, tiles2), dim3(512), W2_LDS_BYTES, stream, d2, x, x_roi_stride, dy,
                               dy_roi_stride, dwp, db);
        if (hipGetLastError() != hipSuccess) { set_error("conv3x3 wgrad launch failed"); *rc = ARFE_ERR_LAUNCH; return true; }
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(9 * C * C / 4, 256)), dim3(256), 0, stream, (const float*)dwp, ns, C, dw);
        if (hipGetLastError() != hipSuccess) { set_error("conv3x3 wgrad reduce launch failed"); *rc = ARFE_ERR_LAUNCH; }
        return true;
    }
    static DeviceOnce attr;
    if (!attr.run([]() { return hipFuncSetAttribute((const void*)conv3x3_wgrad_stage_once_kernel<0>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) == hipSuccess &&
                                hipFuncSetAttribute((const void*)conv3x3_wgrad_stage_once_kernel<1>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) == hipSuccess; })) {
        set_error("conv3x3 wgrad: cannot reserve %d bytes of LDS", WS_LDS_BYTES);
        *rc = ARFE_ERR_LAUNCH;
        return true;
    }
    WgradStageDims d;
    // probe switches (skip the epilogue / MFMA phase / staging: wrong results on purpose) exist only in a build with
    // -DARFE_PROBES (tools/probe/build_variant.sh); the shipped library ignores ARFE_CONV_DBG
#ifdef ARFE_PROBES
    static const int dbg = []() { const char* e = getenv("ARFE_CONV_DBG"); return e ? atoi(e) : 0; }();
#else
    constexpr int dbg = 0;
#endif
    d.R = R; d.C = C; d.rows_total = R * 7; d.tiles_ci = C / WS_CI; d.dbg = dbg;
    const int tiles = (C / WS_CO) * d.tiles_ci;
    int nslices = 256 / tiles;
    if (nslices < 1) nslices = 1;
    d.rows_per_slice = ((cdiv(d.rows_total, nslices) + 3) / 4) * 4;
    nslices = cdiv(d.rows_total, d.rows_per_slice);
    if (gemm_half_operands())
        hipLaunchKernelGGL(conv3x3_wgrad_stage_once_kernel<1>, dim3(nslices, tiles), dim3(512), WS_LDS_BYTES, stream, d, x,
                           x_roi_stride, dy, dy_roi_stride, dwp, db);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_stage_once_kernel<0>, dim3(nslices, tiles), dim3(512), WS_LDS_BYTES, stream, d, x,
                           x_roi_stride, dy, dy_roi_stride, dwp, db);
    if (hipGetLastError() != hipSuccess) { set_error("conv3x3 wgrad stage-once launch failed"); *rc = ARFE_ERR_LAUNCH; return true; }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(9 * C * C / 4, 256)), dim3(256), 0, stream, (const float*)dwp, nslices, C, dw);
    if (hipGetLastError() != hipSuccess) { set_error("conv3x3 wgrad reduce launch failed"); *rc = ARFE_ERR_LAUNCH; }
    return true;
}

}  // namespace arfe
