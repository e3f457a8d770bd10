// C-ABI plumbing: error strings, version, argument validation shared by all entry points.
#include <stdarg.h>
#include <stdlib.h>

#include "common.hpp"

namespace arfe {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(256) void zero_buffers_kernel(ZeroList z) {
    const int b = blockIdx.y;
    if (b >= z.n) return;
    uint32_t* p = (uint32_t*)z.p[b];
    const int64_t n = z.words[b];
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    const bool aligned = ((uintptr_t)p & 15) == 0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (aligned && i + 3 < n) {
            *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e) p[i + e] = 0u;
        }
    }
}

int32_t zero_buffers(const ZeroList& z, hipStream_t stream) {
    if (z.n == 0) return ARFE_OK;
    int64_t mx = 0;
    for (int i = 0; i < z.n; ++i) mx = z.words[i] > mx ? z.words[i] : mx;
    int gx = (int)cdiv64(mx, 256 * 4 * 4);            // ~4 quads per thread for the largest buffer
    gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
    hipLaunchKernelGGL(zero_buffers_kernel, dim3(gx, z.n), dim3(256), 0, stream, z);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// operand precision of the dense contractions, PER THREAD (the autograd engine runs the backward on its own thread: the host
// wrappers set the mode around every call they make): 0 = fp32 operands (bf16x3 / bf16x6 split), 1 = half operands
static thread_local int g_operand_precision = 0;
bool gemm_half_operands() { return g_operand_precision == 1; }
void set_operand_precision(int mode) { g_operand_precision = mode == 1 ? 1 : 0; }

bool gemm_use_f32() {
    // process-level switch, read once (a function-local static is initialised under the C++11 guard, so the main and
    // the autograd thread may both arrive first): ARFE_GEMM_F32=1 -> exact-fp32 MFMA
    static const bool on = []() { const char* e = getenv("ARFE_GEMM_F32"); return e && e[0] == '1'; }();
    return on;
}

int32_t check_pyramid(const arfe_pyramid_t* p) {
    if (!p) { set_error("pyramid descriptor is null"); return ARFE_ERR_ARG; }
    if (p->num_levels < 1 || p->num_levels > ARFE_MAX_LEVELS) {
        set_error("num_levels=%d outside [1,%d]", p->num_levels, ARFE_MAX_LEVELS);
        return ARFE_ERR_ARG;
    }
    if (p->refine_level < 0 || p->refine_level >= p->num_levels) {
        set_error("refine_level=%d outside [0,%d)", p->refine_level, p->num_levels);
        return ARFE_ERR_ARG;
    }
    if (p->storage != ARFE_STORAGE_F32 && p->storage != ARFE_STORAGE_F16) {
        set_error("pyramid storage=%d (0 = fp32, 1 = half)", p->storage);
        return ARFE_ERR_ARG;
    }
    if (p->batch < 1 || p->channels < 1) {
        set_error("batch=%d channels=%d must be positive", p->batch, p->channels);
        return ARFE_ERR_ARG;
    }
    for (int i = 0; i < p->num_levels; ++i)
        if (p->height[i] < 1 || p->width[i] < 1) {
            set_error("level %d has empty size %dx%d", i, p->height[i], p->width[i]);
            return ARFE_ERR_ARG;
        }
    return ARFE_OK;
}

// fp16 storage <-> fp32 compute (the fp16 mode of the reference, mmdet/core/fp16): 8 elements per thread, round to
// nearest even on the way down.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void cast_h2f_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    const bool aligned = (((uintptr_t)src & 15) | ((uintptr_t)dst & 31)) == 0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (aligned && i + 7 < n) {
            const half8 h = *reinterpret_cast<const half8*>(src + i);
            *reinterpret_cast<float8*>(dst + i) = __builtin_convertvector(h, float8);
        } else {
            for (int e = 0; e < 8 && i + e < n; ++e) dst[i + e] = (float)src[i + e];
        }
    }
}
__global__ __launch_bounds__(256) void cast_f2h_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    const bool aligned = (((uintptr_t)src & 31) | ((uintptr_t)dst & 15)) == 0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (aligned && i + 7 < n) {
            const float8 f = *reinterpret_cast<const float8*>(src + i);
            *reinterpret_cast<half8*>(dst + i) = __builtin_convertvector(f, half8);
        } else {
            for (int e = 0; e < 8 && i + e < n; ++e) dst[i + e] = (_Float16)src[i + e];
        }
    }
}

// many small tensors in ONE launch (the parameters of a module that runs in fp16 mode: 14 - 28 tensors per boundary and
// direction, each a 2 - 7 us launch on its own): blockIdx.y = tensor, grid-stride over its elements
constexpr int CAST_MULTI_MAX = 48;
struct CastList { const void* src[CAST_MULTI_MAX]; void* dst[CAST_MULTI_MAX]; int64_t n[CAST_MULTI_MAX]; };
template <bool TO_HALF>
__global__ __launch_bounds__(256) void cast_multi_kernel(CastList l) {
    const int t = blockIdx.y;
    const int64_t n = l.n[t];
    const int64_t stride = (int64_t)gridDim.x * 256 * 8;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if constexpr (TO_HALF) {
            const float* s = (const float*)l.src[t];
            _Float16* d = (_Float16*)l.dst[t];
            if ((((uintptr_t)s & 31) | ((uintptr_t)d & 15)) == 0 && i + 7 < n) {
                *reinterpret_cast<half8*>(d + i) = __builtin_convertvector(*reinterpret_cast<const float8*>(s + i), half8);
            } else {
                for (int e = 0; e < 8 && i + e < n; ++e) d[i + e] = (_Float16)s[i + e];
            }
        } else {
            const _Float16* s = (const _Float16*)l.src[t];
            float* d = (float*)l.dst[t];
            if ((((uintptr_t)s & 15) | ((uintptr_t)d & 31)) == 0 && i + 7 < n) {
                *reinterpret_cast<float8*>(d + i) = __builtin_convertvector(*reinterpret_cast<const half8*>(s + i), float8);
            } else {
                for (int e = 0; e < 8 && i + e < n; ++e) d[i + e] = (float)s[i + e];
            }
        }
    }
}

}  // namespace arfe

extern "C" int32_t arfe_cast_multi(int32_t count, const void* const* src, void* const* dst, const int64_t* n, int32_t to_half,
                                   void* stream) {
    using namespace arfe;
    ARFE_CHECK_ARG(count >= 0 && (count == 0 || (src && dst && n)), ARFE_ERR_ARG, "cast_multi: bad arguments");
    for (int base = 0; base < count; base += CAST_MULTI_MAX) {
        CastList l;
        int m = 0;
        int64_t mx = 0;
        for (int i = base; i < count && m < CAST_MULTI_MAX; ++i) {
            ARFE_CHECK_ARG(n[i] >= 0, ARFE_ERR_ARG, "cast_multi: tensor %d has %lld elements", i, (long long)n[i]);
            if (n[i] == 0) continue;
            ARFE_CHECK_ARG(src[i] && dst[i], ARFE_ERR_ARG, "cast_multi: null pointer (tensor %d)", i);
            l.src[m] = src[i]; l.dst[m] = dst[i]; l.n[m] = n[i];
            mx = n[i] > mx ? n[i] : mx;
            ++m;
        }
        if (m == 0) continue;
        int64_t gx = cdiv64(mx, 256 * 8 * 2);
        gx = gx < 1 ? 1 : (gx > 2048 ? 2048 : gx);
        if (to_half) hipLaunchKernelGGL(cast_multi_kernel<true>, dim3((unsigned)gx, m), dim3(256), 0, as_stream(stream), l);
        else hipLaunchKernelGGL(cast_multi_kernel<false>, dim3((unsigned)gx, m), dim3(256), 0, as_stream(stream), l);
        ARFE_CHECK_LAUNCH();
    }
    return ARFE_OK;
}

extern "C" int32_t arfe_cast_f16_to_f32(int64_t n, const void* src, float* dst, void* stream) {
    using namespace arfe;
    ARFE_CHECK_ARG(n >= 0, ARFE_ERR_ARG, "cast_f16_to_f32: n=%lld", (long long)n);
    if (n == 0) return ARFE_OK;
    ARFE_CHECK_ARG(src && dst, ARFE_ERR_ARG, "cast_f16_to_f32: null pointer");
    int64_t g = cdiv64(n, 256 * 8 * 2);
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    hipLaunchKernelGGL(cast_h2f_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (const _Float16*)src, dst, n);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
extern "C" int32_t arfe_cast_f32_to_f16(int64_t n, const float* src, void* dst, void* stream) {
    using namespace arfe;
    ARFE_CHECK_ARG(n >= 0, ARFE_ERR_ARG, "cast_f32_to_f16: n=%lld", (long long)n);
    if (n == 0) return ARFE_OK;
    ARFE_CHECK_ARG(src && dst, ARFE_ERR_ARG, "cast_f32_to_f16: null pointer");
    int64_t g = cdiv64(n, 256 * 8 * 2);
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    hipLaunchKernelGGL(cast_f2h_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), src, (_Float16*)dst, n);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_set_operand_precision(int32_t mode) {
    if (mode != 0 && mode != 1) { arfe::set_error("set_operand_precision: mode %d (0 = fp32 operands, 1 = half operands)", mode); return ARFE_ERR_ARG; }
    arfe::set_operand_precision(mode);
    return ARFE_OK;
}
extern "C" int32_t arfe_get_operand_precision(void) { return arfe::gemm_half_operands() ? 1 : 0; }
extern "C" const char* arfe_last_error(void) { return arfe::g_err; }
extern "C" int32_t arfe_abi_version(void) { return ARFE_ABI_VERSION; }
