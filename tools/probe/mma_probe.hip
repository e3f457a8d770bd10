// Micro-benchmark of the 256x224 kernel's inner loop (tools/probe, not part of the library): how long 72 K-steps of
// 84 v_mfma_f32_16x16x32_bf16 per wave take with (0) operands held in registers, (1) + the kernel's LDS fragment reads,
// (2) + the per-step workgroup barrier, (3) + B fragments single-buffered as the compiler scheduled them before.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mma_probe.bin tools/probe/mma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int XS_LD = 32, BM = 256, BN = 224, BUF = 2 * (BM + BN) * XS_LD;
__device__ __forceinline__ int xs_swz(int r) { return ((0x1320 >> (((r >> 2) & 3) * 4)) & 3) ^ ((r >> 1) & 1); }

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
// MODE 4: + split of 4 staged octets per thread and the 8 ds_write_b128 of a K-step (values from registers)
// MODE 5: + the global loads of the next K-step (A: 2 x 2 dwordx4 rows, B: 2 x 8 dword loads, pixel-contiguous)
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int steps, const float* __restrict__ src = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    for (int i = tid; i < 2 * BUF; i += 512) lds[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    f32x4 acc[4][7];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 7; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag = 8 * (lq ^ xs_swz(l15));
    bf16x8 ah[4], al[4], bh[2], bl[2];
    for (int t = 0; t < 4; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(lds + (wm * 64 + t * 16 + l15) * XS_LD + frag);
        al[t] = *reinterpret_cast<const bf16x8*>(lds + BM * XS_LD + (wm * 64 + t * 16 + l15) * XS_LD + frag);
    }
    bh[0] = bh[1] = ah[0]; bl[0] = bl[1] = al[1];
    float ra[2][8], rb[2][8];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) { ra[i][e] = 1.f + 0.001f * (tid + e); rb[i][e] = 2.f + 0.002f * (tid + e + i); }
    const int a_off0 = (tid >> 2) * XS_LD + 8 * ((tid & 3) ^ xs_swz(tid >> 2)), a_off1 = a_off0 + 128 * XS_LD;
    const int br0 = tid & 255, br1 = br0 < BN ? br0 : BN - 1;
    const int b_off0 = br1 * XS_LD + 8 * ((tid >> 8) ^ xs_swz(br1)), b_off1 = br1 * XS_LD + 8 * (((tid >> 8) + 2) ^ xs_swz(br1));
    const float* gA = src ? src + (size_t)blockIdx.x * 65536 + (tid >> 2) * 2304 + (tid & 3) * 8 : nullptr;
    const float* gB = src ? src + 32 * 1024 * 1024 + (size_t)(blockIdx.x * 224 + br1) : nullptr;
    int buf = 0;
    for (int s = 0; s < steps; ++s) {
        auto stash = [&]() {
            unsigned short* W = lds + (buf ^ 1) * BUF;
            bf16x8 hi, lo;
            split8(ra[0], hi, lo);
            *reinterpret_cast<bf16x8*>(W + a_off0) = hi; *reinterpret_cast<bf16x8*>(W + BM * XS_LD + a_off0) = lo;
            split8(ra[1], hi, lo);
            *reinterpret_cast<bf16x8*>(W + a_off1) = hi; *reinterpret_cast<bf16x8*>(W + BM * XS_LD + a_off1) = lo;
            split8(rb[0], hi, lo);
            if (br0 < BN) { *reinterpret_cast<bf16x8*>(W + 2 * BM * XS_LD + b_off0) = hi; *reinterpret_cast<bf16x8*>(W + 2 * BM * XS_LD + BN * XS_LD + b_off0) = lo; }
            split8(rb[1], hi, lo);
            if (br0 < BN) { *reinterpret_cast<bf16x8*>(W + 2 * BM * XS_LD + b_off1) = hi; *reinterpret_cast<bf16x8*>(W + 2 * BM * XS_LD + BN * XS_LD + b_off1) = lo; }
            if (MODE == 5 || MODE == 7) {
                const float* pa = gA + (s & 63) * 32;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 v0 = *reinterpret_cast<const float4*>(pa + i * 128 * 2304), v1 = *reinterpret_cast<const float4*>(pa + i * 128 * 2304 + 4);
                    ra[i][0] = v0.x; ra[i][1] = v0.y; ra[i][2] = v0.z; ra[i][3] = v0.w; ra[i][4] = v1.x; ra[i][5] = v1.y; ra[i][6] = v1.z; ra[i][7] = v1.w;
                }
                const float* pb = gB + (size_t)((s & 31) * 32 + (tid >> 8) * 8) * 50176;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) rb[i][e] = pb[(size_t)(i * 16 + e) * 50176];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) { ra[i][e] += 1.f; rb[i][e] += 1.f; }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        const bool late = (MODE == 6 || MODE == 7) && wave >= 4;      // SIMD partner waves take stash and mma in opposite order
        if (MODE >= 4 && !late) stash();
        const unsigned short* A0 = lds + buf * BUF, *A1 = A0 + BM * XS_LD, *B0 = A0 + 2 * BM * XS_LD, *B1 = B0 + BN * XS_LD;
        if (MODE >= 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = (wm * 64 + t * 16 + l15) * XS_LD + frag;
                ah[t] = *reinterpret_cast<const bf16x8*>(A0 + off);
                al[t] = *reinterpret_cast<const bf16x8*>(A1 + off);
            }
            const int off = (wn * 112 + l15) * XS_LD + frag;
            bh[0] = *reinterpret_cast<const bf16x8*>(B0 + off);
            bl[0] = *reinterpret_cast<const bf16x8*>(B1 + off);
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (MODE >= 1 && MODE != 3 && j + 1 < 7) {
                const int off = (wn * 112 + (j + 1) * 16 + l15) * XS_LD + frag;
                bh[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(B0 + off);
                bl[(j + 1) & 1] = *reinterpret_cast<const bf16x8*>(B1 + off);
            }
            if (MODE == 3 && j > 0) {
                const int off = (wn * 112 + j * 16 + l15) * XS_LD + frag;
                bh[j & 1] = *reinterpret_cast<const bf16x8*>(B0 + off);
                bl[j & 1] = *reinterpret_cast<const bf16x8*>(B1 + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j & 1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j & 1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j & 1], acc[i][j], 0, 0, 0);
        }
        if (late) stash();
        if (MODE >= 2) { __syncthreads(); buf ^= 1; }
        if (MODE >= 4 && s + 1 == steps) { float t = 0.f; for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) t += ra[i][e] + rb[i][e]; if (t == 1.2345f) out[tid] = t; }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 7; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run(const char* what, float* out, int wgs, const float* src = nullptr) {
    const int lds_bytes = 2 * BUF * 2, steps = 72;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), lds_bytes, 0, out, steps, src);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), lds_bytes, 0, out, steps, src);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("mode %d (%s): %.1f us per launch of %d workgroups, %.1f ns per K-step, MFMA-only bound at 16 cyc x 84 x 2 waves = 2688 cyc\n",
           MODE, what, us, wgs, us * 1e3 / steps);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    run<0>("MFMAs on register operands", out, 224);
    run<1>("+ LDS fragment reads (B double-buffered)", out, 224);
    run<2>("+ barrier per K-step", out, 224);
    run<3>("barrier, B single-buffered", out, 224);
    run<0>("MFMAs on register operands, 256 workgroups", out, 256);
    float* src;
    hipMalloc(&src, (size_t)(32 * 1024 * 1024 + 1024 * 50176 + 65536) * 4);
    hipMemset(src, 0, (size_t)(32 * 1024 * 1024 + 1024 * 50176 + 65536) * 4);
    run<4>("barrier + split and LDS stores of register data", out, 224);
    run<5>("barrier + split + LDS stores + global loads one step ahead", out, 224, src);
    run<6>("as mode 4, partner waves in opposite order", out, 224);
    run<7>("as mode 5, partner waves in opposite order", out, 224, src);
    return 0;
}
