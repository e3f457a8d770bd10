// Exact-fp32 tiled GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain,
// 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).  One kernel template serves every dense contraction of the
// AR path -- NonLocal 1x1 convs / theta^T phi / P g^T and their gradients, the AR-RFF 3x3 convs (implicit
// GEMM, shifted-tap loaders) and the FC layers -- through small loader/epilogue functors.
//
//   C[m][n] (+)= sum_k A(m,k) * B(k,n)          tile 128x128x16, 256 threads = 4 waves (2x2), wave = 64x64
//
//   * LDS images are k-major (As[k][m], Bs[k][n], row pad 4) so the MFMA operand fetch
//     (lane l -> element [k = l>>5][m = l&31]) is one conflict-free ds_read_b32 per operand.
//   * global -> register -> LDS double buffering: the loads of K-step t+1 are in flight while step t runs
//     on the matrix pipe; one barrier per K-step.
//   * blockIdx.x = output tile (XCD-aware remap: tiles that share an A panel land on one XCD's L2),
//     blockIdx.y = split-K slice (atomic epilogue), blockIdx.z = batch.
//   * K may be "batched": K = kb_count * k_inner, the loaders see (kb, k) so a reduction over images
//     (weight gradients) needs no extra pass.
#pragma once
#include "common.hpp"

namespace arfe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 16;
constexpr int GEMM_LDS_LD = 132;  // 128 + 4 pad (floats)

struct GemmDims {
    int M, N, K;        // K = inner extent per k-batch
    int kb_count;       // number of k-batches folded into the reduction (>= 1)
    int tiles_m, tiles_n;
    int splitk;         // gridDim.y
    int ksteps_total;   // kb_count * ceil(K / BK)
    int kb_inner;       // 1: iterate k-batches innermost
    // optional per-batch reduction length (large-tile kernel only): batch z belongs to class (zk_class >> z) & 1 and runs
    // zk_steps<class> K-steps cut into zk_splits<class> slices (gridDim.y = the larger count).  Scalars, not arrays: an
    // array indexed by blockIdx.z would move the whole argument struct to scratch memory.
    int zk_on = 0;
    unsigned zk_class = 0;
    int zk_steps0 = 0, zk_steps1 = 0, zk_splits0 = 0, zk_splits1 = 0;
    // zk_on == 2: gridDim.y indexes this table of (batch << 8 | slice) pairs instead of the class-major order, so that the
    // caller decides which pairs share an XCD (workgroup id modulo 8) and with it an L2
    unsigned short zk_pair[256] = {};
};

// ---- operand loaders: return A(m,k) / B(k,n) for batch z and k-batch kb, 0 outside the matrix ------------
// `kfast` tells the kernel which index is contiguous in memory so that consecutive lanes read consecutive
// addresses.

// Besides the scalar operator() (used by the exact-fp32 kernel) every loader offers the octet interface of the
// bf16x3 kernel: `Row row(z, i)` caches what depends only on the row, `load8(row, z, kb, k0, v)` fetches 8 consecutive k.
__device__ __forceinline__ void zero8(float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
}
__device__ __forceinline__ void ld8_contig(const float* q, float (&v)[8]) {
    const float4_u a = *reinterpret_cast<const float4_u*>(q);
    const float4_u b = *reinterpret_cast<const float4_u*>(q + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// Octet loads are branch-free: rows are clamped into the matrix by row() (an `ok` flag zeroes them afterwards) and
// FULLK (K a multiple of the K-step, decided once per launch) removes every k bound check, so all loads of a
// K-step are issued back to back instead of sitting in divergent branches.
struct LoadRowMajorK {   // element(i, k) = p[z*sz + kb*skb + i*ld + k]     (k contiguous)
    const float* p; int64_t sz, skb; int64_t ld; int rows, K;
    static constexpr bool kfast = true;
    __device__ __forceinline__ float operator()(int z, int kb, int i, int k) const {
        return (i < rows && k < K) ? p[z * sz + kb * skb + (int64_t)i * ld + k] : 0.f;
    }
    struct Row { const float* q; bool ok; };
    __device__ __forceinline__ Row row(int z, int i) const {
        return Row{p + z * sz + (int64_t)(i < rows ? i : rows - 1) * ld, i < rows};
    }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int, int kb, int k0, float (&v)[8]) const {
        const float* q = r.q + kb * skb;
        if (FULLK) {
            // rows past the matrix were clamped by row(): they yield copies of the last row, whose products only reach
            // output rows/columns the epilogue never stores - no zeroing (8 selects per octet on the MFMA issue port)
            ld8_contig(q + k0, v);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const float t = q[k < K ? k : K - 1];
                v[j] = (r.ok && k < K) ? t : 0.f;
            }
        }
    }
};
struct LoadRowMajorI {   // element(i, k) = p[z*sz + kb*skb + k*ld + i]     (i contiguous)
    const float* p; int64_t sz, skb; int64_t ld; int rows, K;
    static constexpr bool kfast = false;
    __device__ __forceinline__ float operator()(int z, int kb, int i, int k) const {
        return (i < rows && k < K) ? p[z * sz + kb * skb + (int64_t)k * ld + i] : 0.f;
    }
    struct Row { const float* q; bool ok; };
    __device__ __forceinline__ Row row(int z, int i) const { return Row{p + z * sz + (i < rows ? i : rows - 1), i < rows}; }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int, int kb, int k0, float (&v)[8]) const {
        const float* q = r.q + kb * skb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const float t = q[(int64_t)(FULLK || k < K ? k : K - 1) * ld];
            v[j] = FULLK ? t : ((r.ok && k < K) ? t : 0.f);     // FULLK: clamped rows need no zeroing (see LoadRowMajorK)
        }
    }
};

// ---- epilogues ------------------------------------------------------------------------------------------
// Read-only auxiliary operands of an epilogue (bias, residual, masks, scale) are loaded through the constant address
// space: the compiler may then move those loads ahead of the epilogue's earlier stores.  Through a plain pointer every
// such load may alias the output and has to wait for the store before it - one memory round trip per output element.
// (None of these tensors is written by the kernel that reads it; `accumulate` reads the output itself and stays plain.)
typedef const float __attribute__((address_space(4)))* epi_cptr;
__device__ __forceinline__ float epi_ld(const float* p, int64_t i) { return ((epi_cptr)(uintptr_t)p)[i]; }
template <class E> struct epi_has_pre { static constexpr bool value = false; };
template <class E> struct epi_has_rowbias { static constexpr bool value = false; };
struct EpiPlain {   // C[z*sz + m*ld + n] ; optional bias per row (m) or per column (n), optional ReLU,
                    // optional residual add (same layout as C), beta (0: overwrite, 1: accumulate)
    float* c; int64_t sz, ld;
    const float* bias_m; const float* bias_n; const float* residual;
    const float* mask;      // optional: result is zeroed where mask[o] <= 0 (ReLU backward of the consumer), non-atomic only
    int relu; int accumulate; int atomic;
    // optional softmax backward folded into the producer of dP (no split-K): C = scale[o] * (v - rowsub[z*rowsub_sz + m]),
    // i.e. dS = P * (dP - delta) with delta the per-row sum of P * dP (computed from the outputs, see nl_rowdot_kernel)
    const float* scale = nullptr; const float* rowsub = nullptr; int64_t rowsub_sz = 0;
    // split-K without atomics: slice s stores its raw partial sums into slab s (c + s * slab_stride, selected by the
    // kernel through epi_select_slice) and gemm_slab_reduce_kernel sums the slabs, then applies bias / residual / ReLU /
    // mask.  0 = off (atomic adds into a zeroed c, or no split at all).
    int64_t slab_stride = 0;
    __device__ __forceinline__ void operator()(int z, int m, int n, float v, bool first_split) const {
        int64_t o = z * sz + (int64_t)m * ld + n;
        if (slab_stride) { c[o] = v; return; }
        if (first_split) {
            if (bias_m) v += epi_ld(bias_m, m);
            if (bias_n) v += epi_ld(bias_n, n);
            if (residual) v += epi_ld(residual, o);
        }
        if (scale) v = epi_ld(scale, o) * (v - epi_ld(rowsub, z * rowsub_sz + m));
        if (atomic) { unsafeAtomicAdd(c + o, v); return; }
        if (accumulate) v += c[o];
        if (relu) v = fmaxf(v, 0.f);
        if (mask && !(epi_ld(mask, o) > 0.f)) v = 0.f;
        c[o] = v;
    }
    // two-phase form for epilogues that read auxiliary tensors (scale / rowsub): a kernel first collects the operands of a
    // batch of elements, then stores - a load issued between two stores of `c` cannot be moved ahead of them (possible
    // alias), so the element-wise form above serialises one memory round trip per element.
    __device__ __forceinline__ bool wants_pre() const { return scale != nullptr; }
    __device__ __forceinline__ void pre(int z, int m, int n, float& s, float& d) const {
        s = epi_ld(scale, z * sz + (int64_t)m * ld + n);
        d = epi_ld(rowsub, z * rowsub_sz + m);
    }
    __device__ __forceinline__ void store_pre(int z, int m, int n, float v, float s, float d) const {
        c[z * sz + (int64_t)m * ld + n] = s * (v - d);
    }
};
template <> struct epi_has_pre<EpiPlain> { static constexpr bool value = true; };
// hook called by the split-K kernels with their K-slice index before the epilogue (a no-op for other epilogues)
template <class EPI> __device__ __forceinline__ void epi_select_slice(EPI&, int) {}
__device__ __forceinline__ void epi_select_slice(EpiPlain& e, int slice) {
    if (e.slab_stride) e.c += (int64_t)slice * e.slab_stride;
}

template <class LA, class LB, class EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmDims d, LA la, LB lb, EPI epi) {
    __shared__ float As[2][GEMM_BK][GEMM_LDS_LD];
    __shared__ float Bs[2][GEMM_BK][GEMM_LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;

    // XCD-aware tile remap (bijective for any tile count): consecutive ids of one XCD walk along n
    // inside one m-panel so the A panel stays in that XCD's L2.
    int ntiles = d.tiles_m * d.tiles_n;
    int bid = blockIdx.x;
    {
        int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tm = bid / d.tiles_n, tn = bid % d.tiles_n;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    // split-K range in units of K-steps over the (kb, k) space
    const int ksteps_per_kb = (d.K + GEMM_BK - 1) / GEMM_BK;
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

    // staging: operand tile = 128 rows x 2 k-octets = 256 octets -> one octet (8 consecutive k of one row) per thread
    // and operand, fetched with the branch-free octet loaders (vector loads where k is contiguous)
    int a_row, a_oct, b_row, b_oct;
    if (LA::kfast) { a_oct = tid & 1; a_row = tid >> 1; } else { a_row = tid & 127; a_oct = tid >> 7; }
    if (LB::kfast) { b_oct = tid & 1; b_row = tid >> 1; } else { b_row = tid & 127; b_oct = tid >> 7; }
    const typename LA::Row arow = la.row(z, m0 + a_row);
    const typename LB::Row brow = lb.row(z, n0 + b_row);
    float ra[8], rb[8];
    auto fetch = [&](int step) {
        const int kb = d.kb_inner ? step % d.kb_count : step / ksteps_per_kb;
        const int k0 = (d.kb_inner ? step / d.kb_count : step - kb * ksteps_per_kb) * GEMM_BK;
        if (k0 + GEMM_BK <= d.K) {
            la.template load8<true>(arow, z, kb, k0 + 8 * a_oct, ra);
            lb.template load8<true>(brow, z, kb, k0 + 8 * b_oct, rb);
        } else {
            la.template load8<false>(arow, z, kb, k0 + 8 * a_oct, ra);
            lb.template load8<false>(brow, z, kb, k0 + 8 * b_oct, rb);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            As[buf][8 * a_oct + j][a_row] = ra[j];
            Bs[buf][8 * b_oct + j][b_row] = rb[j];
        }
    };

    if (s_begin < s_end) {
        fetch(s_begin);
        stash(0);
        __syncthreads();
        int buf = 0;
        for (int s = s_begin; s < s_end; ++s) {
            const bool more = (s + 1 < s_end);
            if (more) fetch(s + 1);
            const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
            for (int kk = 0; kk < GEMM_BK; kk += 2) {
                float a0 = As[buf][kk + kh][wm * 64 + l31];
                float a1 = As[buf][kk + kh][wm * 64 + 32 + l31];
                float b0 = Bs[buf][kk + kh][wn * 64 + l31];
                float b1 = Bs[buf][kk + kh][wn * 64 + 32 + l31];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (more) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    } else if (!first_split) {
        return;
    }

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
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

template <class LA, class LB, class EPI>
static inline int32_t launch_gemm(int M, int N, int K, int kb_count, int batch, int splitk, LA la, LB lb, EPI epi,
                                  hipStream_t stream, bool kb_inner = false) {
    if (M <= 0 || N <= 0 || batch <= 0) return ARFE_OK;
    GemmDims d;
    d.M = M; d.N = N; d.K = K; d.kb_count = kb_count < 1 ? 1 : kb_count;
    d.kb_inner = kb_inner ? 1 : 0;
    d.tiles_m = cdiv(M, GEMM_BM); d.tiles_n = cdiv(N, GEMM_BN);
    d.ksteps_total = d.kb_count * cdiv(K, GEMM_BK);
    if (splitk < 1) splitk = 1;
    if (splitk > d.ksteps_total && d.ksteps_total > 0) splitk = d.ksteps_total;
    d.splitk = splitk;
    dim3 grid(d.tiles_m * d.tiles_n, splitk, batch);
    hipLaunchKernelGGL((gemm_f32_kernel<LA, LB, EPI>), grid, dim3(256), 0, stream, d, la, lb, epi);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// how many K-slices to cut so that about one wave of workgroups per CU pair is in flight (256 CUs, 2 WG/CU)
// K-slices for the 128x128 kernels, from a cost model fitted to tools/gemm_sweep.py on MI355X (us), round 2 (the
// double-buffered kernel; profiles/r02_gemm_sweep.txt):
//   rounds * (K-steps per slice * t_step + 20 fixed)  +  atomic bytes / 1.3 TB/s
//   t_step = 1.1 us while every workgroup has a CU to itself (<= 256), 1.65 us with two per CU; rounds = workgroups / 512
//   beyond one full wave of two workgroups per CU (fractional: the tail overlaps the next round)
// Small outputs (cls / reg / bias-sized) take many thin slices, large ones few (their atomic traffic dominates).
static inline int pick_splitk(int M, int N, int batch, int ksteps) {
    const int64_t tiles = (int64_t)cdiv(M, GEMM_BM) * cdiv(N, GEMM_BN) * batch;
    if (tiles >= 384 || ksteps < 4) return 1;
    const double out_bytes = 4.0 * (double)M * (double)N * (double)batch;
    double best = 1e30;
    int best_s = 1;
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128};
    for (int s : cand) {
        if (s > 1 && s * 2 > ksteps) break;                 // at least 2 K-steps per slice
        const double wgs = (double)tiles * s;
        const double t_step = wgs <= 256.0 ? 1.1 : 1.65;
        const double rounds = wgs <= 512.0 ? 1.0 : wgs / 512.0;
        const double t = rounds * ((double)cdiv(ksteps, s) * t_step + 20.0) + (s > 1 ? out_bytes * s / 1.3e6 : 0.0);
        if (t < best) { best = t; best_s = s; }
    }
    return best_s;
}

}  // namespace arfe
