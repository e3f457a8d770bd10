/* libarfe_hip — C ABI of the MI355X-native (gfx950) AR-FPN / AR-RFF forward+backward hot path.
 *
 * The reference (Fanzhongjie/ARFE, a mmdetection 2.0 fork) has NO native boundary on this path: it is
 * nn.Modules calling ATen ops.  This header is therefore the NEW boundary one level below
 * the reference's Python plugin surface; every entry point names the reference code it replaces
 * (paths relative to the reference root).  The host-side mirror of the plugin surface (same class names,
 * ctor arguments, state_dict keys) lives in arfe_amd/ and calls these functions through ctypes.
 *
 * Conventions
 *   - every function returns 0 (ARFE_OK) or a negative ARFE_ERR_* code; arfe_last_error() gives a
 *     thread-local message.  No exception crosses the ABI.
 *   - all pointers are raw DEVICE addresses of contiguous fp32 NCHW tensors (int32/int64 where stated).
 *     The CALLER allocates every input, output and workspace buffer; the library holds no global mutable
 *     state, is re-entrant, enqueues only on `stream` (a hipStream_t) and never synchronises the device.
 *   - the device is selected by the caller (hipSetDevice / torch device guard).
 *   - gradients are WRITTEN (not accumulated) unless stated.
 */
#ifndef ARFE_HIP_H
#define ARFE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARFE_ABI_VERSION 3   /* 2: arfe_linear_fwd / _bwd take an optional split-K scratch (workspace, workspace_bytes)
                              * 3: arfe_set_operand_precision, arfe_fpn_gather_gate_fwd; the conv3x3 pack holds half planes too */
#define ARFE_MAX_LEVELS 8

#define ARFE_OK 0
#define ARFE_ERR_ARG (-1)     /* bad shape / null pointer / unsupported size */
#define ARFE_ERR_LAUNCH (-2)  /* hip launch error */

/* Pyramid geometry shared by the AR-FPN entry points: level i is [batch, channels, height[i], width[i]]. */
typedef struct arfe_pyramid {
    int32_t num_levels;    /* <= ARFE_MAX_LEVELS */
    int32_t refine_level;  /* WFPNDualSpatial(refine_level=2) */
    int32_t batch;
    int32_t channels;
    int32_t height[ARFE_MAX_LEVELS];
    int32_t width[ARFE_MAX_LEVELS];
    int32_t storage;       /* ARFE_STORAGE_F32 (0) or ARFE_STORAGE_F16: what the PER-LEVEL activation arrays of the AR-FPN entry
                            * points (x, out, gout, dx) hold.  With F16 (the reference's fp16 mode: a half pyramid in, a half
                            * pyramid out) those pointers address IEEE halves - pass them through a cast - and the kernels
                            * convert on load / store; every other tensor (gate maps, m, bsf, dm, weights, gradients of the
                            * weights) stays fp32 and so does the arithmetic.  arfe_fpn_half_storage_supported says whether a
                            * pyramid's kernels have half instantiations (otherwise the caller converts, as in round 2). */
} arfe_pyramid_t;
#define ARFE_STORAGE_F32 0
#define ARFE_STORAGE_F16 1
int32_t arfe_fpn_half_storage_supported(const arfe_pyramid_t* pyr);

const char* arfe_last_error(void);
int32_t arfe_abi_version(void);

/* Operand precision of the dense contractions (3x3 / 1x1 convs, linear layers, the NonLocal products), PER CALLING THREAD:
 *   0 (default)  fp32 operands: every product is the three-term bf16 split on the matrix pipe (six terms for the softmax
 *                logits), ~1e-5 relative - the fp32 reference's arithmetic to parity tolerance;
 *   1            half operands: each operand is rounded to IEEE half on its way into LDS and every product is ONE f16 MFMA with
 *                fp32 accumulation - the arithmetic of the reference's fp16 mode (mmdet/core/fp16/hooks.py:86-105:
 *                model.half()), exact for tensors that are stored as half.
 * The mode is thread-local (no state shared between threads); host code sets it around the calls of a module that runs in
 * fp16 mode, in the forward and again in the backward (autograd's thread).  ARFE_GEMM_F32=1 overrides both. */
int32_t arfe_set_operand_precision(int32_t mode);
int32_t arfe_get_operand_precision(void);

/* ======================================================================================================
 * AR-FPN  (mmdet/models/necks/wfpn_dual_spatial.py: WFPNDualSpatial.forward :99-137)
 * ==================================================================================================== */

/* gather (:102-113): m = (((((0+G0)+G1)+G2)+G3)+G4)/num_levels, G_i = adaptive_max_pool2d (i < refine) or
 * nearest (i >= refine) of x[i] to the refine level's size.  argmax (optional, may be NULL):
 * [refine_level][B,C,hr,wr] int32 flat (y*w+x) index of each window's max, first max in row-major order
 * (F.adaptive_max_pool2d tie-break), consumed by arfe_fpn_input_bwd. */
int32_t arfe_fpn_gather_fwd(const arfe_pyramid_t* pyr, const float* const* x, float* m, int32_t* argmax,
                            void* stream);

/* gate maps (:119-121 + mmcv ConvModule's ReLU): a_i = tanh(relu(conv3x3(x_i; w1_i) + b1_i)),
 * b_i likewise with (w2_i, b2_i); w*_i is [1,C,3,3], b*_i is [1]; a_i, b_i are [B,1,h_i,w_i].
 * workspace: arfe_fpn_gatemap_workspace_bytes(pyr) bytes (the 18 per-(gate, tap) product maps + packed weights). */
int64_t arfe_fpn_gatemap_workspace_bytes(const arfe_pyramid_t* pyr);
int32_t arfe_fpn_gatemap_fwd(const arfe_pyramid_t* pyr, const float* const* x, const float* const* w1,
                             const float* const* b1, const float* const* w2, const float* const* b2,
                             float* const* a, float* const* b, void* workspace, void* stream);

/* gather (:102-113) and gate maps (:119-121) from ONE read of the pyramid: same results as arfe_fpn_gather_fwd followed by
 * arfe_fpn_gatemap_fwd (m and argmax bit-identical).  workspace: arfe_fpn_gather_gate_workspace_bytes(pyr) bytes.  Pyramids the
 * fused kernel does not take (pooled levels whose ratio to the refine level is not 2 or 4, widths not a multiple of 4,
 * channels not a multiple of 4) run the two separate ops inside. */
int64_t arfe_fpn_gather_gate_workspace_bytes(const arfe_pyramid_t* pyr);
int32_t arfe_fpn_gather_gate_fwd(const arfe_pyramid_t* pyr, const float* const* x, const float* const* w1,
                                 const float* const* b1, const float* const* w2, const float* const* b2, float* m,
                                 int32_t* argmax, float* const* a, float* const* b, void* workspace, void* stream);

/* scatter + gate + residual (:125,135): out_i = x_i + nearest(bsf -> h_i x w_i) * (a_i + b_i). */
int32_t arfe_fpn_scatter_fwd(const arfe_pyramid_t* pyr, const float* const* x, const float* bsf,
                             const float* const* a, const float* const* b, float* const* out, void* stream);

/* backward of the scatter stage (autograd of :125,135), given g_i = dL/dout_i:
 *   dm_i  = sum_c g_i[c] * U_i(bsf)[c]            da_i = dm_i (1 - a_i^2) [a_i > 0]   (same for b)
 *   dbsf  = sum_i nearest^T( g_i * (a_i + b_i) )
 * outputs: dpre_a[i], dpre_b[i] ([B,1,h_i,w_i], gradient w.r.t. the conv pre-activation) and dbsf [B,C,hr,wr]. */
int32_t arfe_fpn_scatter_bwd(const arfe_pyramid_t* pyr, const float* const* gout, const float* bsf,
                             const float* const* a, const float* const* b, float* const* dpre_a,
                             float* const* dpre_b, float* dbsf, void* stream);

/* dx_i = g_i + conv3x3^T(dpre_a_i; w1_i) + conv3x3^T(dpre_b_i; w2_i) + gather^T(dm / num_levels)
 * (dm = gradient w.r.t. the gathered mean m, [B,C,hr,wr]; argmax from arfe_fpn_gather_fwd). */
int32_t arfe_fpn_input_bwd(const arfe_pyramid_t* pyr, const float* const* gout, const float* const* dpre_a,
                           const float* const* dpre_b, const float* const* w1, const float* const* w2,
                           const float* dm, const int32_t* argmax, float* const* dx, void* stream);

/* gate-conv parameter gradients: dw1_i[c,tap] = sum x_i[c, y+dy, x+dx] dpre_a_i[y,x], db1_i = sum dpre_a_i
 * (same for 2).  dw*_i [1,C,3,3], db*_i [1]. */
int32_t arfe_fpn_gate_wgrad(const arfe_pyramid_t* pyr, const float* const* x, const float* const* dpre_a,
                            const float* const* dpre_b, float* const* dw1, float* const* db1,
                            float* const* dw2, float* const* db2, void* stream);

/* ======================================================================================================
 * NonLocal2D refine  (mmdet/ops/non_local.py :63-70, :78-105; reduction=1, use_scale=False,
 * mode='embedded_gaussian').  L = h*w of the refine level.
 *   tpg  = [Wt;Wp;Wg] m + [bt;bp;bg]        [B,3C,L]   (theta | phi | g)
 *   P    = softmax_rows(theta^T phi)        [B,L,L]    (materialised: 4 B * L^2 per image; on exact-fp32
 *                                                        MFMA the block is compute-bound, recompute costs more)
 *   y    = g P^T  (y[c,i] = sum_j P[i,j] g[c,j])        [B,C,L]
 *   bsf  = m + Wo y + bo                                [B,C,L]
 * Saved for backward (caller-allocated): tpg, P, y.
 * ==================================================================================================== */
int64_t arfe_nonlocal_workspace_bytes(int32_t B, int32_t C, int32_t L);   /* fwd and bwd scratch */

int32_t arfe_nonlocal_fwd(int32_t B, int32_t C, int32_t L, const float* m, const float* w_theta,
                          const float* b_theta, const float* w_phi, const float* b_phi, const float* w_g,
                          const float* b_g, const float* w_out, const float* b_out, float* tpg, float* P,
                          float* y, float* bsf, void* workspace, void* stream);

/* dm = dbsf + [Wt;Wp;Wg]^T d(tpg); all eight parameter gradients are written. `dP` is an [B,L,L] scratch. */
int32_t arfe_nonlocal_bwd(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* m,
                          const float* w_theta, const float* w_phi, const float* w_g, const float* w_out,
                          const float* tpg, const float* P, const float* y, float* dP, float* dm,
                          float* dw_theta, float* db_theta, float* dw_phi, float* db_phi, float* dw_g,
                          float* db_g, float* dw_out, float* db_out, void* workspace, void* stream);

/* The same backward in three calls, so that a caller can put the parameter gradients on a second stream beside the data
 * path: _wout (dw_out, db_out: needs dbsf and y only), _data (dm; leaves d(theta|phi|g) in the workspace) and _wcat
 * (the six theta / phi / g gradients from that workspace, after _data).  arfe_nonlocal_bwd == the three in a row. */
int32_t arfe_nonlocal_bwd_wout(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* y, float* dw_out,
                               float* db_out, void* stream);
int32_t arfe_nonlocal_bwd_data(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                               const float* w_phi, const float* w_g, const float* w_out, const float* tpg, const float* P,
                               const float* y, float* dP, float* dm, void* workspace, void* stream);
/* arfe_nonlocal_bwd_data in two calls with the same arguments (ABI 3): _dtpg runs everything up to the gradients of the three
 * projections (left in the workspace: all that arfe_nonlocal_bwd_wcat needs, so a caller starts it on another stream here),
 * _dm the last contraction. */
int32_t arfe_nonlocal_bwd_dtpg(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                               const float* w_phi, const float* w_g, const float* w_out, const float* tpg, const float* P,
                               const float* y, float* dP, float* dm, void* workspace, void* stream);
int32_t arfe_nonlocal_bwd_dm(int32_t B, int32_t C, int32_t L, const float* dbsf, const float* w_theta,
                               const float* w_phi, const float* w_g, const float* w_out, const float* tpg, const float* P,
                               const float* y, float* dP, float* dm, void* workspace, void* stream);
int32_t arfe_nonlocal_bwd_wcat(int32_t B, int32_t C, int32_t L, const float* m, float* dw_theta, float* db_theta,
                               float* dw_phi, float* db_phi, float* dw_g, float* db_g, void* workspace, void* stream);

/* ======================================================================================================
 * AR-RFF
 * ==================================================================================================== */

/* get_adaptive_scale_rois(rois, facs=1) (mmdet/models/utils/additional.py:38-71): rois [R,5] = (b,x1,y1,x2,y2)
 * -> h_rois, w_rois [R,5]; same fp32 op order as the reference (bit-exact). */
int32_t arfe_rff_make_rois(int32_t R, const float* rois, float* h_rois, float* w_rois, void* stream);

/* SingleRoIExtractor.map_roi_levels (roi_extractors/single_level.py:53-93):
 * clamp(floor(log2(sqrt((x2-x1)(y2-y1)) / finest_scale + 1e-6)), 0, num_levels-1) as int64, bit-exact with the
 * reference's fp32 torch-CPU result (the device compares against the fp32 values at which torch's
 * floor(log2(.)) steps; no transcendental is evaluated). num_levels <= 5. */
int32_t arfe_map_roi_levels(int32_t R, const float* rois, int32_t num_levels, float finest_scale,
                            int64_t* levels, void* stream);

/* The AR-RFF glue of _bbox_forward (standard_roi_head.py:138-150) in one launch: arfe_rff_make_rois followed by
 * arfe_map_roi_levels of the original, the w- and the h-RoI set (each enlarged RoI gets its own level).  Results are
 * bit-identical to the separate calls. */
int32_t arfe_rff_roi_sets(int32_t R, const float* rois, int32_t num_levels, float finest_scale, float* h_rois,
                          float* w_rois, int64_t* levels_ori, int64_t* levels_w, int64_t* levels_h, void* stream);

/* RoIAlign v2 (aligned=True, sample_num=0 -> adaptive grid; mmdet/ops/roi_align/src/cpu/roi_align_v2.cpp
 * :107-191 fwd, :247-332 bwd; CUDA kernels src/cuda/roi_align_kernel_v2.cu:63-129,180-) fused with the
 * per-level dispatch of SingleRoIExtractor.forward (single_level.py:109-152) for `nsets` RoI sets at once:
 * RoI k of set s samples level levels[s][k] with spatial scale 1/strides[level] and is written at
 * out + k*out_roi_stride + (s*C + c)*49 + p, i.e. the three sets of the AR-RFF path fill the [R, 3C, 7, 7]
 * concat of standard_roi_head.py:155 in place.  pooled must be 7.
 * workspace: arfe_roialign_fwd_workspace_bytes(nsets, R) (per-RoI separable weight tables). */
int64_t arfe_roialign_fwd_workspace_bytes(int32_t nsets, int32_t R);
int32_t arfe_roialign_fwd(const arfe_pyramid_t* pyr, const float* const* feats, const int32_t* strides,
                          int32_t nsets, int32_t R, const float* const* rois, const int64_t* const* levels,
                          int32_t pooled, int32_t sample_num, float* out, int64_t out_roi_stride,
                          void* workspace, void* stream);

/* The forward in two parts: _prepare is the RoI-only part (processing order, weight tables; needs no feature map, same
 * workspace as arfe_roialign_fwd); _group extracts the RoIs of processing groups [group_first, group_first + group_count) on a
 * prepared workspace.  Group g holds RoI set (g + 1) % nsets - for the AR-RFF sets (ori, w, h): w, h, ori - so a caller starts
 * the gating conv of a set's channel slice while the next set is still being extracted.  arfe_roialign_fwd = _prepare + every
 * group. */
int32_t arfe_roialign_fwd_prepare(const arfe_pyramid_t* pyr, const int32_t* strides, int32_t nsets, int32_t R,
                                  const float* const* rois, const int64_t* const* levels, int32_t pooled, int32_t sample_num,
                                  void* workspace, void* stream);
int32_t arfe_roialign_fwd_group(const arfe_pyramid_t* pyr, const float* const* feats, const int32_t* strides, int32_t nsets,
                                int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                int32_t sample_num, float* out, int64_t out_roi_stride, const void* workspace,
                                int32_t group_first, int32_t group_count, void* stream);

/* The same forward with HALF feature maps and a HALF output (the extractor of a model in the reference's fp16 mode reads
 * the half maps a neck handed on and returns half RoI features, single_level.py:109 @force_fp32(out_fp16=True)): the kernel
 * converts on load / store, the arithmetic is the fp32 one - results equal a half -> float pass, arfe_roialign_fwd and a float ->
 * half pass.  out_roi_stride in elements. */
int32_t arfe_roialign_fwd_h(const arfe_pyramid_t* pyr, const void* const* feats, const int32_t* strides, int32_t nsets,
                            int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                            int32_t sample_num, void* out, int64_t out_roi_stride, void* workspace, void* stream);
/* ... and its backward on a workspace filled by arfe_roialign_bwd_prepare: half gradient in, half feature-map gradients out
 * (every element written). */
int32_t arfe_roialign_bwd_prepared_h(const arfe_pyramid_t* pyr, void* const* dfeats, const int32_t* strides, int32_t nsets,
                                     int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                     int32_t sample_num, const void* gout, int64_t gout_roi_stride, const void* workspace,
                                     void* stream);

/* dfeats[i] = sum over all `nsets` RoI sets of RoIAlign^T(gout slice of that set): every element of every
 * dfeats[i] is written by the call (a 16x16 tile whose RoI list is long is zeroed and summed by a few workgroups
 * with float atomics, every other tile - the empty ones included - is stored exactly once).  Set s reads its upstream
 * gradient at gout + k*gout_roi_stride + (s*C + c)*49 + p (the [R, nsets*C, 7, 7] concat gradient).
 * RoIs are binned per 16x16 feature tile in `workspace` (arfe_roialign_bwd_workspace_bytes). pooled must be 7. */
int64_t arfe_roialign_bwd_workspace_bytes(const arfe_pyramid_t* pyr, int32_t nsets, int32_t R);
int32_t arfe_roialign_bwd(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides, int32_t nsets,
                          int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                          int32_t sample_num, const float* gout, int64_t gout_roi_stride, void* workspace,
                          void* stream);
/* The same backward in two calls: _prepare does everything that depends on the RoIs alone (tile lists, per-RoI
 * weight tables) into `workspace`, so a caller can run it beside the forward RoIAlign on another stream; _prepared
 * consumes that workspace (same pyr / strides / RoI sets) once the upstream gradient exists.
 * arfe_roialign_bwd == _prepare followed by _prepared on one stream. */
int32_t arfe_roialign_bwd_prepare(const arfe_pyramid_t* pyr, const int32_t* strides, int32_t nsets, int32_t R,
                                  const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                  int32_t sample_num, void* workspace, void* stream);
int32_t arfe_roialign_bwd_prepared(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides, int32_t nsets,
                                   int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                   int32_t sample_num, const float* gout, int64_t gout_roi_stride,
                                   const void* workspace, void* stream);
/* _prepared itself in two calls: _prezero zeroes the (few) tiles of dfeats that are summed with atomics and needs nothing but
 * the prepared workspace - a caller that allocates dfeats at forward time runs it there, off the backward's critical path;
 * _apply is the scatter itself (every other element of dfeats is stored exactly once).  _prepared == _prezero + _apply. */
int32_t arfe_roialign_bwd_prezero(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides, int32_t nsets,
                                  int32_t R, const void* workspace, void* stream);
int32_t arfe_roialign_bwd_apply(const arfe_pyramid_t* pyr, float* const* dfeats, const int32_t* strides, int32_t nsets,
                                int32_t R, const float* const* rois, const int64_t* const* levels, int32_t pooled,
                                int32_t sample_num, const float* gout, int64_t gout_roi_stride, const void* workspace,
                                void* stream);

/* 3x3 conv, padding 1, on RoI features [R, C, 7, 7] (MultiBBoxHead wh_conv/hh_conv/final_conv,
 * bbox_heads/multirois_bbox_head.py:50-78,172-173,183) as an implicit GEMM on the matrix cores.
 * Weights w [C,C,3,3] are re-laid once per optimizer step into `packed` (arfe_conv3x3_packed_bytes(C) bytes:
 * [tap][co][ci] for the forward pass, [tap][ci][co] flipped for the data gradient; each as pre-split (bf16 hi, lo) words
 * for the generic kernels and as two bf16 planes for the LDS-DMA staging of the stage-once kernels).
 * x / y / dy / dx are addressed as base + r*roi_stride + c*49 + p, so a channel slice of the [R,3C,7,7]
 * concat is passed without a copy. */
int64_t arfe_conv3x3_packed_bytes(int32_t C);
int64_t arfe_conv3x3_workspace_bytes(int32_t C);   /* scratch of arfe_conv3x3_bwd_weight: 9 C C floats, or the per-slice
                                                     * partial-sum slabs of the stage-once kernel (75.5 MB, C % 128 == 0) */
int32_t arfe_conv3x3_pack_weights(int32_t C, const float* w, void* packed, void* stream);
/* y = act(conv(x; w) + b) */
int32_t arfe_conv3x3_fwd(int32_t R, int32_t C, const float* x, int64_t x_roi_stride, const void* packed,
                         const float* b, int32_t relu, float* y, int64_t y_roi_stride, void* stream);
/* ABI 3: the second gating conv of the AR-RFF head with the fusion in its epilogue (multirois_bbox_head.py:172-181):
 *   lh = ReLU(conv(x; w) + b)     z = ori + ori * (lw + lh)
 * = arfe_conv3x3_fwd(relu) followed by arfe_rff_fuse_fwd, one launch; lh and z are [R, C, 7, 7] contiguous. */
int32_t arfe_conv3x3_fwd_fuse(int32_t R, int32_t C, const float* x, int64_t x_roi_stride, const void* packed,
                              const float* b, const float* ori, int64_t ori_roi_stride, const float* lw, float* lh,
                              float* z, void* stream);
/* ReLU backward is folded into the PRODUCER of each gradient (epilogue masks), so dy is used as is.
 * dx = conv^T(dy) * [dx_mask > 0]   (dx_mask optional: the post-ReLU tensor that fed this conv's input, addressed
 * like dx with its own roi stride) */
int32_t arfe_conv3x3_bwd_data(int32_t R, int32_t C, const void* packed, const float* dy, int64_t dy_roi_stride,
                              float* dx, int64_t dx_roi_stride, const float* dx_mask, int64_t dx_mask_roi_stride,
                              void* stream);
/* dw = x (*) dy  [C,C,3,3], db = sum dy  [C] */
int32_t arfe_conv3x3_bwd_weight(int32_t R, int32_t C, const float* x, int64_t x_roi_stride, const float* dy,
                                int64_t dy_roi_stride, float* dw, float* db, void* workspace, void* stream);

/* fusion (multirois_bbox_head.py:175-182): z = ori + ori * (lw + lh); ori is a slice (roi stride given),
 * lw, lh, z are dense [R,C,7,7]. */
int32_t arfe_rff_fuse_fwd(int32_t R, int32_t C, const float* ori, int64_t ori_roi_stride, const float* lw,
                          const float* lh, float* z, void* stream);
/* d_ori = dz (1 + lw + lh) ; dlw = dz * ori * [lw > 0] ; dlh = dz * ori * [lh > 0]  (lw, lh are the post-ReLU conv
 * outputs, so the ReLU backward of wh_conv / hh_conv is applied here).  d_ori is written with d_ori_roi_stride. */
int32_t arfe_rff_fuse_bwd(int32_t R, int32_t C, const float* ori, int64_t ori_roi_stride, const float* lw,
                          const float* lh, const float* dz, float* d_ori, int64_t d_ori_roi_stride,
                          float* dlw, float* dlh, void* stream);

/* nn.Linear (shared_fcs / fc_cls / fc_reg, multirois_bbox_head.py:203-235): y = act(x W^T + b),
 * x [R,in], W [out,in], y [R,out].
 * workspace (optional, may be NULL / 0 bytes): scratch of arfe_linear_workspace_bytes(R, in, out) bytes.  With it a
 * contraction whose plan cuts K stores per-slice partial sums there and reduces them in a fixed order (bias, ReLU, mask and
 * accumulate applied by the reduction); without it (or when it is too small) the slices add atomically into the output. */
int64_t arfe_linear_workspace_bytes(int32_t R, int32_t in_f, int32_t out_f);
int32_t arfe_linear_fwd(int32_t R, int32_t in_f, int32_t out_f, const float* x, const float* w, const float* b,
                        int32_t relu, float* y, void* workspace, int64_t workspace_bytes, void* stream);
/* dx = (dy W) (* [x > 0] when mask_dx_by_x: x is then the post-ReLU output of the previous layer, so the mask is
 * that layer's ReLU backward); written, or accumulated when accumulate_dx != 0; dx may be NULL.
 * dw = dy^T x, db = colsum(dy). */
int32_t arfe_linear_bwd(int32_t R, int32_t in_f, int32_t out_f, const float* x, const float* w,
                        const float* dy, float* dx, int32_t accumulate_dx, int32_t mask_dx_by_x, float* dw,
                        float* db, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- base FPN: mmdet/models/necks/fpn.py:166-201 ----------------------------------------------
 * The lateral 1x1 convs (fpn.py:117-124,170-173), the top-down `laterals[i-1] += F.interpolate(laterals[i], size=...)`
 * (fpn.py:180-188, mode nearest), the 3x3 output convs (fpn.py:125-133,192-194) and the stride-2 extra levels
 * `F.max_pool2d(outs[-1], 1, stride=2)` (fpn.py:199-201).  All tensors contiguous NCHW fp32 device memory. */
/* y[B,Cout,HW] = W[Cout,Cin] x[B,Cin,HW] + b (b may be NULL) */
int32_t arfe_conv1x1_fwd(int32_t B, int32_t Cin, int32_t Cout, int32_t HW, const float* x, const float* w,
                         const float* b, float* y, void* stream);
int32_t arfe_conv1x1_bwd_data(int32_t B, int32_t Cin, int32_t Cout, int32_t HW, const float* w, const float* dy,
                              float* dx, void* stream);
int32_t arfe_conv1x1_bwd_weight(int32_t B, int32_t Cin, int32_t Cout, int32_t HW, const float* x, const float* dy,
                                float* dw, float* db, void* stream);
/* 3x3, padding 1, C -> C on whole maps; `packed` from arfe_conv3x3_pack_weights; workspace arfe_conv3x3_workspace_bytes(C) */
int32_t arfe_convmap3x3_fwd(int32_t B, int32_t C, int32_t H, int32_t W, const float* x, const void* packed,
                            const float* b, float* y, void* stream);
int32_t arfe_convmap3x3_bwd_data(int32_t B, int32_t C, int32_t H, int32_t W, const void* packed, const float* dy,
                                 float* dx, void* stream);
int32_t arfe_convmap3x3_bwd_weight(int32_t B, int32_t C, int32_t H, int32_t W, const float* x, const float* dy,
                                   float* dw, float* db, void* workspace, void* stream);
/* out[B,C,H,W] = lat + nearest(top[B,C,h,w] -> H x W) (out may alias lat); backward: dtop += nearest^T(dlat) */
int32_t arfe_nearest_add_fwd(int32_t B, int32_t C, int32_t h, int32_t w, const float* top, int32_t H, int32_t W,
                             const float* lat, float* out, void* stream);
int32_t arfe_nearest_add_bwd(int32_t B, int32_t C, int32_t h, int32_t w, float* dtop, int32_t H, int32_t W,
                             const float* dlat, void* stream);
/* backward == 0: out[B,C,(H-1)/2+1,(W-1)/2+1] = in[B,C,H,W][::2, ::2]; backward != 0: out[B,C,H,W] = scatter of in */
int32_t arfe_subsample2(int32_t B, int32_t C, int32_t H, int32_t W, const float* in, float* out, int32_t backward,
                        void* stream);

/* `count` tensors converted by one launch (to_half: fp32 -> half, else half -> fp32); src[i] / dst[i] hold n[i] elements.  The
 * three arrays are HOST arrays read at launch time.  Same rounding as the single-tensor forms below. */
int32_t arfe_cast_multi(int32_t count, const void* const* src, void* const* dst, const int64_t* n, int32_t to_half,
                        void* stream);

/* ---- fp16 mode: mmdet/core/fp16/utils.py:7-23 cast_tensor_type ------------------------------------
 * The reference's fp16 mode (hooks.py:86-105 wrap_fp16_model, decorators.py:8-63 auto_fp16, :66-123 force_fp32) stores
 * parameters and activations as IEEE half; this library computes in fp32, so the module boundary converts with these
 * two kernels (round to nearest even, inf/nan preserved, like Tensor.to).  src/dst: n contiguous elements. */
int32_t arfe_cast_f16_to_f32(int64_t n, const void* src, float* dst, void* stream);
int32_t arfe_cast_f32_to_f16(int64_t n, const float* src, void* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARFE_HIP_H */
