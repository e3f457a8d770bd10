// Shared helpers for the libarfe_hip kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/arfe_hip.h"

namespace arfe {

// ---- error plumbing (no exceptions cross the C ABI) -------------------------------------
void set_error(const char* fmt, ...);

#define ARFE_CHECK_ARG(cond, code, ...)        \
    do {                                       \
        if (!(cond)) {                         \
            ::arfe::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

#define ARFE_CHECK_LAUNCH()                                                   \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) {                                              \
            ::arfe::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,  \
                              hipGetErrorString(e__));                        \
            return ARFE_ERR_LAUNCH;                                           \
        }                                                                     \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- index semantics shared by the AR-FPN kernels -----------------
// PyTorch legacy 'nearest': src = min((int)floorf(dst * scale), in-1), scale = (float)in/out in fp32.
__device__ __forceinline__ int nearest_src(int dst, int in_size, float scale) {
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}
__host__ __device__ __forceinline__ float nearest_scale(int in_size, int out_size) {
    return (float)in_size / (float)out_size;
}
// first dst index d in [0,out] with nearest_src(d) >= s  (the map is monotone non-decreasing)
__device__ __forceinline__ int nearest_dst_begin(int s, int in_size, int out_size, float scale) {
    if (s <= 0) return 0;
    if (s > in_size - 1) return out_size;
    int d = (int)ceilf((float)s / scale);
    if (d > out_size) d = out_size;
    if (d < 0) d = 0;
    while (d > 0 && nearest_src(d - 1, in_size, scale) >= s) --d;
    while (d < out_size && nearest_src(d, in_size, scale) < s) ++d;
    return d;
}
// adaptive max pool window of output index o: [floor(o*in/out), ceil((o+1)*in/out))
__device__ __forceinline__ int pool_start(int o, int in_size, int out_size) {
    return (int)(((int64_t)o * in_size) / out_size);
}
__device__ __forceinline__ int pool_end(int o, int in_size, int out_size) {
    return (int)((((int64_t)(o + 1)) * in_size + out_size - 1) / out_size);
}

// ---- wave-level reductions (wave = 64 lanes) --------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte vector with only 4-byte alignment promised (rows of odd width are not 16-B aligned;
// the hardware runs in unaligned-access mode, the type keeps the compiler honest).
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

struct Levels {
    int n;        // number of pyramid levels
    int r;        // refine level
    int B, C;
    int H[ARFE_MAX_LEVELS], W[ARFE_MAX_LEVELS];
    // log2 of the exact power-of-two size ratio to the refine level in BOTH dimensions:
    //   sh > 0: level is finer  (H = Hr << sh)   sh < 0: coarser (Hr = H << -sh)   sh == 0: same size
    //   SH_GENERAL: no such ratio -> kernels take the generic index path (ragged / unpadded pyramids)
    int sh[ARFE_MAX_LEVELS];
    int half;     // the per-level activation arrays hold IEEE halves (arfe_pyramid_t::storage)
};
constexpr int SH_GENERAL = 99;
static inline int pow2_shift(int big, int small) {   // big == small << s ? s : -1
    for (int s = 0; s < 16; ++s) if ((small << s) == big) return s;
    return -1;
}
static inline Levels to_levels(const arfe_pyramid_t* p) {
    Levels l;
    l.n = p->num_levels; l.r = p->refine_level; l.B = p->batch; l.C = p->channels;
    l.half = p->storage == ARFE_STORAGE_F16 ? 1 : 0;
    for (int i = 0; i < ARFE_MAX_LEVELS; ++i) { l.H[i] = p->height[i]; l.W[i] = p->width[i]; l.sh[i] = SH_GENERAL; }
    const int hr = p->height[l.r], wr = p->width[l.r];
    for (int i = 0; i < l.n; ++i) {
        int a = pow2_shift(l.H[i], hr), b = pow2_shift(l.W[i], wr);
        if (a >= 0 && a == b) { l.sh[i] = a; continue; }
        a = pow2_shift(hr, l.H[i]); b = pow2_shift(wr, l.W[i]);
        if (a >= 0 && a == b) l.sh[i] = -a;
    }
    return l;
}
int32_t check_pyramid(const arfe_pyramid_t* p);

// One launch that zero-fills up to 8 buffers of 4-byte words (a hipMemsetAsync per buffer is one ~6 us launch each and
// the hot path needs up to six in a row).  Entries with a null pointer or zero length are skipped.
struct ZeroList {
    void* p[8]; int64_t words[8]; int n;
    ZeroList() : n(0) {}
    void add(void* ptr, int64_t nwords) { if (ptr && nwords > 0 && n < 8) { p[n] = ptr; words[n] = nwords; ++n; } }
};
int32_t zero_buffers(const ZeroList& z, hipStream_t stream);
// dwp[tap][co][ci] -> dw[co][ci][tap] (gemm_ops.hip)
int32_t conv3x3_unpermute_dw(int C, const float* dwp, float* dw, hipStream_t stream);
bool gemm_use_f32();
bool gemm_half_operands();
void set_operand_precision(int mode);

// Function attributes (the opt-in to >64 KB of dynamic LDS) and small constant blocks belong to ONE device: a process
// that drives several GPUs must repeat them per device.  DeviceOnce is a flag per device ordinal; two threads racing
// on the same device both run the (idempotent) body.
constexpr int ARFE_MAX_DEVICES = 64;
struct DeviceOnce {
    std::atomic<unsigned char> done[ARFE_MAX_DEVICES];
    template <class F> bool run(F&& body) {          // body() -> bool ok
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ARFE_MAX_DEVICES) return body();
        if (done[dev].load(std::memory_order_acquire)) return true;
        if (!body()) return false;
        done[dev].store(1, std::memory_order_release);
        return true;
    }
};

template <int N>
struct PtrArray { const float* p[N]; };
template <int N>
struct MutPtrArray { float* p[N]; };

}  // namespace arfe
