// Base FPN (reference necks/fpn.py): lateral 1x1 convs, top-down nearest add, 3x3 output convs on
// whole feature maps, stride-2 subsample for the extra level.  The convolutions go through the same GEMM template as the
// rest of the path (gemm_bf16x3.hpp): only the operand loaders are new.
//
//   conv1x1       y[b][co][p]  = sum_ci W[co][ci] x[b][ci][p] + bias          M = Cout, N = HW,  K = Cin,  batch = B
//   conv3x3 map   y[b][co][p]  = sum_{tap,ci} Wp[tap][co][ci] x[b][ci][p + shift(tap)] (zero padded)
//                                                                             M = C,    N = HW,  K = C per tap x 9 taps
//   weight grads  K runs over the pixels of every image (k-batched over b, split-K, atomic epilogue)
#include "gemm_bf16x3.hpp"

namespace arfe {

static inline EpiPlain epi_plain(float* c, int64_t sz, int64_t ld) {
    EpiPlain e;
    e.c = c; e.sz = sz; e.ld = ld; e.bias_m = nullptr; e.bias_n = nullptr; e.residual = nullptr; e.mask = nullptr;
    e.relu = 0; e.accumulate = 0; e.atomic = 0;
    return e;
}

// B(n = pixel of image z, k = ci) under tap kb: x[z][ci][(y+dy)*W + x+dx], 0 outside the map
struct ConvMapB {
    const float* x; int64_t img_stride; int H, W, HW, Cin;
    static constexpr bool kfast = false;
    __device__ __forceinline__ float operator()(int z, int kb, int n, int k) const {
        if (n >= HW || k >= Cin) return 0.f;
        const int y = n / W, xx = n - y * W;
        const int yy = y + kb / 3 - 1, xs = xx + kb % 3 - 1;
        if ((unsigned)yy >= (unsigned)H || (unsigned)xs >= (unsigned)W) return 0.f;
        return x[z * img_stride + (int64_t)k * HW + yy * W + xs];
    }
    struct Row { const float* xb; int y, x; bool ok; };
    __device__ __forceinline__ Row row(int z, int n) const {
        Row r;
        r.ok = n < HW;
        const int nn = r.ok ? n : HW - 1;
        r.y = nn / W; r.x = nn - r.y * W;
        r.xb = x + z * img_stride;
        return r;
    }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int, int kb, int k0, float (&v)[8]) const {
        const int yy = r.y + kb / 3 - 1, xs = r.x + kb % 3 - 1;
        const bool inside = (unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W;
        // padding taps read the centre pixel (a valid address) and are zeroed by a select
        const float* q = r.xb + (inside ? yy * W + xs : r.y * W + r.x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const float t = q[(int64_t)(FULLK || k < Cin ? k : Cin - 1) * HW];
            v[j] = (inside && (FULLK || (r.ok && k < Cin))) ? t : 0.f;
        }
    }
};

// B(n = ci, k = pixel p of image kb) under tap z: x[kb][ci][p + off(z)], 0 where the shifted pixel is padding
struct WgradMapB {
    const float* x; int64_t img_stride; int H, W, HW, Cin;
    static constexpr bool kfast = true;
    __device__ __forceinline__ float operator()(int z, int kb, int n, int k) const {
        if (n >= Cin || k >= HW) return 0.f;
        const int y = k / W, xx = k - y * W;
        const int yy = y + z / 3 - 1, xs = xx + z % 3 - 1;
        if ((unsigned)yy >= (unsigned)H || (unsigned)xs >= (unsigned)W) return 0.f;
        return x[kb * img_stride + (int64_t)n * HW + yy * W + xs];
    }
    struct Row { const float* q; bool ok; };
    __device__ __forceinline__ Row row(int, int n) const { return Row{x + (int64_t)(n < Cin ? n : Cin - 1) * HW, n < Cin}; }
    template <bool FULLK>
    __device__ __forceinline__ void load8(const Row& r, int z, int kb, int k0, float (&v)[8]) const {
        const int dy = z / 3 - 1, dx = z % 3 - 1;
        const int off = dy * W + dx;
        const float* q = r.q + kb * img_stride;
        int y = k0 / W, xx = k0 - y * W;
        const int s0 = k0 + off;
        const bool whole = s0 >= 0 && s0 + 7 < HW;      // the 8-float window lies inside this channel's plane
        float t[8];
        if (whole) {
            ld8_contig(q + s0, t);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int a = s0 + j;
                t[j] = q[a < 0 ? 0 : (a >= HW ? HW - 1 : a)];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = r.ok && (FULLK || k0 + j < HW) && (unsigned)(y + dy) < (unsigned)H &&
                            (unsigned)(xx + dx) < (unsigned)W;
            v[j] = ok ? t[j] : 0.f;
            if (++xx == W) { xx = 0; ++y; }
        }
    }
};

// lat[b][c][Y][X] += top[b][c][nearest(Y), nearest(X)]  (F.interpolate(size=...), legacy nearest index)
__global__ void nearest_add_fwd_kernel(int64_t total, int h, int w, int H, int W, const float* __restrict__ top,
                                       const float* __restrict__ lat, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int X = (int)(i % W);
    const int64_t r = i / W;
    const int Y = (int)(r % H);
    const int64_t bc = r / H;
    const float ys = nearest_scale(h, H), xs = nearest_scale(w, W);
    out[i] = lat[i] + top[(bc * h + nearest_src(Y, h, ys)) * w + nearest_src(X, w, xs)];
}
// dtop[b][c][y][x] += sum of dlat over the destination pixels that read (y, x)
__global__ void nearest_add_bwd_kernel(int64_t total, int h, int w, int H, int W, const float* __restrict__ dlat,
                                       float* __restrict__ dtop) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w);
    const int64_t r = i / w;
    const int y = (int)(r % h);
    const int64_t bc = r / h;
    const float ys = nearest_scale(h, H), xs = nearest_scale(w, W);
    const int Y0 = nearest_dst_begin(y, h, H, ys), Y1 = nearest_dst_begin(y + 1, h, H, ys);
    const int X0 = nearest_dst_begin(x, w, W, xs), X1 = nearest_dst_begin(x + 1, w, W, xs);
    float s = 0.f;
    for (int Y = Y0; Y < Y1; ++Y)
        for (int X = X0; X < X1; ++X) s += dlat[(bc * H + Y) * W + X];
    dtop[i] += s;
}
// max_pool2d(kernel 1, stride 2): out[y][x] = in[2y][2x]; backward scatters into a zeroed map
__global__ void subsample2_kernel(int64_t total, int h, int w, int H, int W, const float* __restrict__ in,
                                  float* __restrict__ out, int backward) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (!backward) {            // total = B*C*h*w (output elements)
        const int x = (int)(i % w);
        const int64_t r = i / w;
        const int y = (int)(r % h);
        const int64_t bc = r / h;
        out[i] = in[(bc * H + 2 * y) * W + 2 * x];
    } else {                    // total = B*C*H*W (input-gradient elements); `in` is the output gradient [h][w]
        const int X = (int)(i % W);
        const int64_t r = i / W;
        const int Y = (int)(r % H);
        const int64_t bc = r / H;
        out[i] = ((Y & 1) == 0 && (X & 1) == 0) ? in[(bc * h + (Y >> 1)) * w + (X >> 1)] : 0.f;
    }
}

__global__ void rowsum_kb_kernel(const float* __restrict__ x, float* __restrict__ out, int batch, int64_t batch_stride,
                                 int inner, int chunk) {
    // out[row] += sum_{b, i in chunk} x[b*batch_stride + row*inner + i]; grid = (rows, chunks of the row, images)
    const int row = blockIdx.x, b = blockIdx.z;
    const int i0 = blockIdx.y * chunk, i1 = i0 + chunk < inner ? i0 + chunk : inner;
    const float* p = x + b * batch_stride + (int64_t)row * inner;
    float s = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) s += p[i];
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out + row, (part[0] + part[1]) + (part[2] + part[3]));
    (void)batch;
}
static inline void launch_rowsum(const float* x, float* out, int rows, int batch, int64_t batch_stride, int inner,
                                 hipStream_t st) {
    const int chunk = 4096;
    hipLaunchKernelGGL(rowsum_kb_kernel, dim3(rows, cdiv(inner, chunk), batch), dim3(256), 0, st, x, out, batch, batch_stride,
                       inner, chunk);
}

}  // namespace arfe

using namespace arfe;

extern "C" int32_t arfe_conv1x1_fwd(int32_t B, int32_t Cin, int32_t Cout, int32_t HW, const float* x, const float* w,
                                    const float* b, float* y, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && Cin > 0 && Cout > 0 && HW >= 0, ARFE_ERR_ARG, "conv1x1_fwd: bad shape");
    if (B == 0 || HW == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && w && y, ARFE_ERR_ARG, "conv1x1_fwd: null pointer");
    LoadRowMajorK la{w, 0, 0, Cin, Cout, Cin};
    LoadRowMajorI lb{x, (int64_t)Cin * HW, 0, HW, HW, Cin};
    EpiPlain e = epi_plain(y, (int64_t)Cout * HW, HW);
    e.bias_m = b;
    return launch_gemm_auto(Cout, HW, Cin, 1, B, 1, la, lb, e, as_stream(stream));
}

extern "C" int32_t arfe_conv1x1_bwd_data(int32_t B, int32_t Cin, int32_t Cout, int32_t HW, const float* w, const float* dy,
                                         float* dx, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && Cin > 0 && Cout > 0 && HW >= 0, ARFE_ERR_ARG, "conv1x1_bwd_data: bad shape");
    if (B == 0 || HW == 0) return ARFE_OK;
    ARFE_CHECK_ARG(w && dy && dx, ARFE_ERR_ARG, "conv1x1_bwd_data: null pointer");
    // dx[b][ci][p] = sum_co W[co][ci] dy[b][co][p]:  A(m=ci, k=co) = W[co*Cin + ci]
    LoadRowMajorI la{w, 0, 0, Cin, Cin, Cout};
    LoadRowMajorI lb{dy, (int64_t)Cout * HW, 0, HW, HW, Cout};
    EpiPlain e = epi_plain(dx, (int64_t)Cin * HW, HW);
    return launch_gemm_auto(Cin, HW, Cout, 1, B, 1, la, lb, e, as_stream(stream));
}

extern "C" int32_t arfe_conv1x1_bwd_weight(int32_t B, int32_t Cin, int32_t Cout, int32_t HW, const float* x,
                                           const float* dy, float* dw, float* db, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && Cin > 0 && Cout > 0 && HW >= 0 && dw && db, ARFE_ERR_ARG, "conv1x1_bwd_weight: bad arguments");
    hipStream_t st = as_stream(stream);
    {
        ZeroList z;
        z.add(dw, (int64_t)Cout * Cin);
        z.add(db, Cout);
        int32_t rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    if (B == 0 || HW == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && dy, ARFE_ERR_ARG, "conv1x1_bwd_weight: null pointer");
    // dW[co][ci] = sum_{b,p} dy[b][co][p] x[b][ci][p]   (k-batched over images, split-K, atomic epilogue)
    LoadRowMajorK la{dy, 0, (int64_t)Cout * HW, HW, Cout, HW};
    LoadRowMajorK lb{x, 0, (int64_t)Cin * HW, HW, Cin, HW};
    EpiPlain e = epi_plain(dw, 0, Cin);
    e.atomic = 1;
    const int ksteps = B * cdiv(HW, gemm_bk());
    int32_t rc = launch_gemm_auto(Cout, Cin, HW, B, 1, pick_splitk(Cout, Cin, 1, ksteps), la, lb, e, st);
    if (rc) return rc;
    launch_rowsum(dy, db, Cout, B, (int64_t)Cout * HW, HW, st);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

// packed = arfe_conv3x3_pack_weights(C, w): wp[tap][co][ci] then wd[tap][ci][co] (flipped / transposed)
extern "C" int32_t arfe_convmap3x3_fwd(int32_t B, int32_t C, int32_t H, int32_t W, const float* x, const void* packed,
                                       const float* b, float* y, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && C > 0 && H >= 0 && W >= 0, ARFE_ERR_ARG, "convmap3x3_fwd: bad shape");
    if (B == 0 || H == 0 || W == 0) return ARFE_OK;
    ARFE_CHECK_ARG(x && packed && y, ARFE_ERR_ARG, "convmap3x3_fwd: null pointer");
    const int HW = H * W;
    const float* wp = (const float*)packed;
    LoadSplitK la{{wp, 0, (int64_t)C * C, C, C, C}};
    ConvMapB lb{x, (int64_t)C * HW, H, W, HW, C};
    EpiPlain e = epi_plain(y, (int64_t)C * HW, HW);
    e.bias_m = b;
    return launch_gemm_auto(C, HW, C, 9, B, 1, la, lb, e, as_stream(stream), false, /*kb_inner=*/true);
}

extern "C" int32_t arfe_convmap3x3_bwd_data(int32_t B, int32_t C, int32_t H, int32_t W, const void* packed,
                                            const float* dy, float* dx, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && C > 0 && H >= 0 && W >= 0, ARFE_ERR_ARG, "convmap3x3_bwd_data: bad shape");
    if (B == 0 || H == 0 || W == 0) return ARFE_OK;
    ARFE_CHECK_ARG(packed && dy && dx, ARFE_ERR_ARG, "convmap3x3_bwd_data: null pointer");
    const int HW = H * W;
    const float* wd = (const float*)packed + (size_t)9 * C * C;
    LoadSplitK la{{wd, 0, (int64_t)C * C, C, C, C}};
    ConvMapB lb{dy, (int64_t)C * HW, H, W, HW, C};
    EpiPlain e = epi_plain(dx, (int64_t)C * HW, HW);
    return launch_gemm_auto(C, HW, C, 9, B, 1, la, lb, e, as_stream(stream), false, /*kb_inner=*/true);
}

extern "C" int32_t arfe_convmap3x3_bwd_weight(int32_t B, int32_t C, int32_t H, int32_t W, const float* x, const float* dy,
                                              float* dw, float* db, void* workspace, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && C > 0 && H >= 0 && W >= 0 && dw && db, ARFE_ERR_ARG, "convmap3x3_bwd_weight: bad arguments");
    hipStream_t st = as_stream(stream);
    const int HW = H * W;
    if (B == 0 || HW == 0) {
        ZeroList z;
        z.add(dw, (int64_t)9 * C * C);
        z.add(db, C);
        return zero_buffers(z, st);
    }
    ARFE_CHECK_ARG(x && dy && workspace, ARFE_ERR_ARG, "convmap3x3_bwd_weight: null pointer");
    float* dwp = (float*)workspace;                 // [tap][co][ci], arfe_conv3x3_workspace_bytes(C)
    {
        ZeroList z;
        z.add(dwp, (int64_t)9 * C * C);
        z.add(db, C);
        int32_t rc = zero_buffers(z, st);
        if (rc) return rc;
    }
    // dWp[tap][co][ci] = sum_{b,p} dy[b][co][p] x[b][ci][p + shift(tap)]: batch = tap, k-batches = images
    LoadRowMajorK la{dy, 0, (int64_t)C * HW, HW, C, HW};
    WgradMapB lb{x, (int64_t)C * HW, H, W, HW, C};
    EpiPlain e = epi_plain(dwp, (int64_t)C * C, C);
    e.atomic = 1;
    const int ksteps = B * cdiv(HW, gemm_bk());
    int32_t rc = launch_gemm_auto(C, C, HW, B, 9, pick_splitk(C, C, 9, ksteps), la, lb, e, st);
    if (rc) return rc;
    rc = conv3x3_unpermute_dw(C, dwp, dw, st);
    if (rc) return rc;
    launch_rowsum(dy, db, C, B, (int64_t)C * HW, HW, st);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_nearest_add_fwd(int32_t B, int32_t C, int32_t h, int32_t w, const float* top, int32_t H, int32_t W,
                                        const float* lat, float* out, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, ARFE_ERR_ARG, "nearest_add_fwd: bad shape");
    if (B == 0) return ARFE_OK;
    ARFE_CHECK_ARG(top && lat && out, ARFE_ERR_ARG, "nearest_add_fwd: null pointer");
    const int64_t total = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(nearest_add_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), total, h, w,
                       H, W, top, lat, out);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_nearest_add_bwd(int32_t B, int32_t C, int32_t h, int32_t w, float* dtop, int32_t H, int32_t W,
                                        const float* dlat, void* stream) {
    ARFE_CHECK_ARG(B >= 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, ARFE_ERR_ARG, "nearest_add_bwd: bad shape");
    if (B == 0) return ARFE_OK;
    ARFE_CHECK_ARG(dtop && dlat, ARFE_ERR_ARG, "nearest_add_bwd: null pointer");
    const int64_t total = (int64_t)B * C * h * w;
    hipLaunchKernelGGL(nearest_add_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), total, h, w,
                       H, W, dlat, dtop);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}

extern "C" int32_t arfe_subsample2(int32_t B, int32_t C, int32_t H, int32_t W, const float* in, float* out, int32_t backward,
                                   void* stream) {
    ARFE_CHECK_ARG(B >= 0 && C > 0 && H > 0 && W > 0, ARFE_ERR_ARG, "subsample2: bad shape");
    if (B == 0) return ARFE_OK;
    ARFE_CHECK_ARG(in && out, ARFE_ERR_ARG, "subsample2: null pointer");
    const int h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1;
    const int64_t total = backward ? (int64_t)B * C * H * W : (int64_t)B * C * h * w;
    hipLaunchKernelGGL(subsample2_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), total, h, w, H, W,
                       in, out, backward);
    ARFE_CHECK_LAUNCH();
    return ARFE_OK;
}
