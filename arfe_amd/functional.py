"""Thin host-side wrappers over the C ABI (device memory, streams and autograd plumbing only).

Raw ops (`*_op`) allocate outputs/workspaces with torch and call libarfe_hip; the three
`torch.autograd.Function`s wire them into AR-FPN, the 3-RoI extractor and the AR-RFF head.
Nothing here computes on the CPU or through PyTorch kernels.
"""
import ctypes
import os

import torch

from . import hip
from .hip import Pyramid, check, ptr, ptr_array, stream

_i32 = ctypes.c_int32


class _Probe:
    """Optional HIP-event brackets around single kernel launches (bench.py's live roofline timing).
    Events are recorded on the launch stream (torch's current stream == the stream handed to the C ABI)."""
    enabled = False
    only = None         # optional set of probe names: the others stay silent (each bracket costs two event records)
    records = {}

    def __init__(self, name):
        self.name = name
        self.on = _Probe.enabled and (_Probe.only is None or name in _Probe.only)

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record(torch.cuda.current_stream())
            _Probe.records.setdefault(self.name, []).append((self.e0, self.e1))
        return False


def probe(name):
    return _Probe(name)


def probe_enable(flag=True, only=None):
    _Probe.enabled = flag
    _Probe.only = set(only) if only else None
    _Probe.records = {}


def probe_results_ms():
    """name -> list of per-launch milliseconds (call after a device synchronise)."""
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in _Probe.records.items()}


# Parity tooling: when a dict is installed here the forward passes drop references to the tensors that hold their
# discrete choices (pooling argmax, gate maps, ReLU outputs) into it, so a parity test can evaluate its CPU checker under
# the same choices.  None (the default) costs nothing.
_decision_tap = None


def tap_decisions(target):
    """target: a dict to fill during the next forward passes, or None to switch the tap off."""
    global _decision_tap
    _decision_tap = target


class lib_precision:
    """`with lib_precision(half_ops):` - the library's dense contractions take half operands (one f16 MFMA per product, the
    reference's fp16 mode) for the calls made inside, on THIS thread (arfe_set_operand_precision is per thread: autograd runs
    the backward on its own thread, so every Function sets the mode it recorded in the forward again in its backward)."""

    def __init__(self, half_ops):
        self.half_ops = bool(half_ops)

    def __enter__(self):
        if self.half_ops:
            check(hip.lib().arfe_set_operand_precision(1), "arfe_set_operand_precision")
        return self

    def __exit__(self, *exc):
        if self.half_ops:
            hip.lib().arfe_set_operand_precision(0)
        return False


def _empty_like_levels(xs):
    return [torch.empty_like(x) for x in xs]


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


_side_stream = {}
_GATE_SEQ = os.environ.get('ARFE_GATE_SEQ', '1') != '0'         # 0: the two gating convs side by side on two streams (rounds 1-2)
_GATE_FUSE = os.environ.get('ARFE_GATE_FUSE', '0') == '1'       # 1: gating convs in a row, fusion in the second one's epilogue (measured slower)
_ONE_STREAM = os.environ.get('ARFE_ONE_STREAM', '0') == '1'      # measurement aid: everything on the current stream


def side_stream(device):
    """A second HIP stream per device for the parameter-gradient kernels: they depend only on the incoming gradient
    and a saved activation, never on each other's results, so they run beside the data-gradient chain and fill the
    CUs a 196-workgroup GEMM leaves idle (256 CUs)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _side_stream:
        _side_stream[key] = torch.cuda.Stream(device=device)
    return _side_stream[key]



def on_side(device, fn):
    """Run fn() on the side stream, ordered after everything already enqueued on the current stream."""
    if device.type != "cuda":
        raise hip.ArfeHipError(f"expected tensors on the GPU, got device {device} (there is no CPU path)")
    if _ONE_STREAM:
        return fn()
    main = torch.cuda.current_stream()
    side = side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        return fn()


def join_side(device):
    if _ONE_STREAM:
        return
    torch.cuda.current_stream().wait_stream(side_stream(device))


def side_reads(device, *tensors):
    """Tensors allocated on the CURRENT stream that kernels already enqueued on the side stream still read after the caller
    returns (a backward that leaves its weight gradients running, join=False): `record_stream` tells the caching allocator
    not to hand their blocks out again before the side stream has passed this point.  This replaces the rule "hold a `keep`
    tuple until the next join" that every such caller had to remember (the tuples stay: they cost nothing)."""
    side = side_stream(device)
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(side)
    return tensors


# ---------------------------------------------------------------------------------------------------------
# AR-FPN raw ops
# ---------------------------------------------------------------------------------------------------------

def _req_levels(xs):
    """pyramid levels: contiguous GPU tensors of ONE storage dtype, fp32 or (fp16 mode, eligible pyramids) half"""
    dt = xs[0].dtype
    if dt not in (torch.float32, torch.float16):
        raise hip.ArfeHipError(f"pyramid levels must be fp32 or half, got {dt}")
    hip.require_cuda(dt, *xs)


def fpn_half_storage_supported(xs, refine_level):
    """True when every AR-FPN kernel has a half-storage instantiation for this pyramid (arfe_fpn_half_storage_supported)."""
    pyr = Pyramid.of(xs, refine_level)
    return bool(hip.lib().arfe_fpn_half_storage_supported(ctypes.byref(pyr)))


def fpn_gather_op(xs, refine_level, want_argmax=True):
    hip.require_f32_cuda(*xs)
    pyr = Pyramid.of(xs, refine_level)
    B, C = xs[0].shape[:2]
    hr, wr = xs[refine_level].shape[2:]
    m = torch.empty(B, C, hr, wr, device=xs[0].device)
    argmax = None
    if want_argmax and refine_level > 0:
        argmax = torch.empty(refine_level, B, C, hr, wr, dtype=torch.int32, device=xs[0].device)
    with probe('fpn_gather_fwd'):
        check(hip.lib().arfe_fpn_gather_fwd(ctypes.byref(pyr), ptr_array(xs), ptr(m), ptr(argmax), stream()),
              "arfe_fpn_gather_fwd")
    return m, argmax


def fpn_gatemap_op(xs, w1, b1, w2, b2, refine_level=0):
    hip.require_f32_cuda(*xs, *w1, *b1, *w2, *b2)
    pyr = Pyramid.of(xs, refine_level)
    dev = xs[0].device
    a = [torch.empty(x.shape[0], 1, x.shape[2], x.shape[3], device=dev) for x in xs]
    b = [torch.empty(x.shape[0], 1, x.shape[2], x.shape[3], device=dev) for x in xs]
    ws = _ws(hip.lib().arfe_fpn_gatemap_workspace_bytes(ctypes.byref(pyr)), dev)
    with probe('fpn_gatemap_fwd'):
        check(hip.lib().arfe_fpn_gatemap_fwd(ctypes.byref(pyr), ptr_array(xs), ptr_array(w1), ptr_array(b1),
                                             ptr_array(w2), ptr_array(b2), ptr_array(a), ptr_array(b), ptr(ws),
                                             stream()), "arfe_fpn_gatemap_fwd")
    return a, b


def fpn_gather_gate_op(xs, w1, b1, w2, b2, refine_level, want_argmax=True):
    """-> (m, argmax, a, b): gather and both gate maps from ONE read of the pyramid (arfe_fpn_gather_gate_fwd)."""
    _req_levels(xs)
    hip.require_f32_cuda(*w1, *b1, *w2, *b2)
    pyr = Pyramid.of(xs, refine_level)
    B, C = xs[0].shape[:2]
    hr, wr = xs[refine_level].shape[2:]
    dev = xs[0].device
    m = torch.empty(B, C, hr, wr, device=dev)
    argmax = None
    if want_argmax and refine_level > 0:
        argmax = torch.empty(refine_level, B, C, hr, wr, dtype=torch.int32, device=dev)
    a = [torch.empty(x.shape[0], 1, x.shape[2], x.shape[3], device=dev) for x in xs]
    b = [torch.empty(x.shape[0], 1, x.shape[2], x.shape[3], device=dev) for x in xs]
    ws = _ws(hip.lib().arfe_fpn_gather_gate_workspace_bytes(ctypes.byref(pyr)), dev)
    with probe('fpn_gather_gate_fwd'):
        check(hip.lib().arfe_fpn_gather_gate_fwd(ctypes.byref(pyr), ptr_array(xs), ptr_array(w1), ptr_array(b1), ptr_array(w2),
                                                 ptr_array(b2), ptr(m), ptr(argmax), ptr_array(a), ptr_array(b), ptr(ws),
                                                 stream()), "arfe_fpn_gather_gate_fwd")
    return m, argmax, a, b


def fpn_scatter_op(xs, bsf, a, b, refine_level):
    _req_levels(xs)
    hip.require_f32_cuda(bsf, *a, *b)
    pyr = Pyramid.of(xs, refine_level)
    outs = _empty_like_levels(xs)
    with probe('fpn_scatter_fwd'):
        check(hip.lib().arfe_fpn_scatter_fwd(ctypes.byref(pyr), ptr_array(xs), ptr(bsf), ptr_array(a), ptr_array(b),
                                             ptr_array(outs), stream()), "arfe_fpn_scatter_fwd")
    return outs


def fpn_scatter_bwd_op(gouts, bsf, a, b, refine_level):
    _req_levels(gouts)
    hip.require_f32_cuda(bsf, *a, *b)
    pyr = Pyramid.of(gouts, refine_level)
    dpa = [torch.empty_like(t) for t in a]
    dpb = [torch.empty_like(t) for t in b]
    dbsf = torch.empty_like(bsf)
    with probe('fpn_scatter_bwd'):
        check(hip.lib().arfe_fpn_scatter_bwd(ctypes.byref(pyr), ptr_array(gouts), ptr(bsf), ptr_array(a), ptr_array(b),
                                             ptr_array(dpa), ptr_array(dpb), ptr(dbsf), stream()),
              "arfe_fpn_scatter_bwd")
    return dpa, dpb, dbsf


def fpn_input_bwd_op(gouts, dpa, dpb, w1, w2, dm, argmax, refine_level):
    _req_levels(gouts)
    hip.require_f32_cuda(*dpa, *dpb, *w1, *w2, dm)
    pyr = Pyramid.of(gouts, refine_level)
    dxs = _empty_like_levels(gouts)
    with probe('fpn_input_bwd'):
        check(hip.lib().arfe_fpn_input_bwd(ctypes.byref(pyr), ptr_array(gouts), ptr_array(dpa), ptr_array(dpb),
                                           ptr_array(w1), ptr_array(w2), ptr(dm), ptr(argmax), ptr_array(dxs),
                                           stream()), "arfe_fpn_input_bwd")
    return dxs


def fpn_gate_wgrad_op(xs, dpa, dpb, refine_level=0):
    _req_levels(xs)
    hip.require_f32_cuda(*dpa, *dpb)
    pyr = Pyramid.of(xs, refine_level)
    dev, C = xs[0].device, xs[0].shape[1]
    n = len(xs)
    dw1 = [torch.empty(1, C, 3, 3, device=dev) for _ in range(n)]
    dw2 = [torch.empty(1, C, 3, 3, device=dev) for _ in range(n)]
    db1 = [torch.empty(1, device=dev) for _ in range(n)]
    db2 = [torch.empty(1, device=dev) for _ in range(n)]
    with probe('fpn_gate_wgrad'):
        check(hip.lib().arfe_fpn_gate_wgrad(ctypes.byref(pyr), ptr_array(xs), ptr_array(dpa), ptr_array(dpb),
                                            ptr_array(dw1), ptr_array(db1), ptr_array(dw2), ptr_array(db2), stream()),
              "arfe_fpn_gate_wgrad")
    return dw1, db1, dw2, db2


# ---------------------------------------------------------------------------------------------------------
# NonLocal2D raw ops
# ---------------------------------------------------------------------------------------------------------

def nonlocal_fwd_op(m, wt, bt, wp, bp, wg, bg, wo, bo):
    hip.require_f32_cuda(m, wt, bt, wp, bp, wg, bg, wo, bo)
    B, C, h, w = m.shape
    L = h * w
    dev = m.device
    tpg = torch.empty(B, 3 * C, L, device=dev)
    P = torch.empty(B, L, L, device=dev)
    y = torch.empty(B, C, L, device=dev)
    bsf = torch.empty_like(m)
    ws = _ws(hip.lib().arfe_nonlocal_workspace_bytes(B, C, L), dev)
    with probe('nonlocal_fwd'):
        check(hip.lib().arfe_nonlocal_fwd(B, C, L, ptr(m), ptr(wt), ptr(bt), ptr(wp), ptr(bp), ptr(wg), ptr(bg), ptr(wo),
                                          ptr(bo), ptr(tpg), ptr(P), ptr(y), ptr(bsf), ptr(ws), stream()),
              "arfe_nonlocal_fwd")
    return bsf, (tpg, P, y)


def nonlocal_bwd_op(dbsf, m, wt, wp, wg, wo, saved, split_streams=False):
    """-> dm, (dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo).
    split_streams: the two parameter-gradient parts go to the side stream (dw_out beside the data path, the theta / phi / g
    gradients after it - beside whatever the caller enqueues next on the current stream); the CALLER joins the side stream."""
    tpg, P, y = saved
    hip.require_f32_cuda(dbsf, m, wt, wp, wg, wo, tpg, P, y)
    B, C, h, w = m.shape
    L = h * w
    dev = m.device
    dP = torch.empty_like(P)
    dm = torch.empty_like(m)
    dwt, dwp, dwg, dwo = (torch.empty_like(t) for t in (wt, wp, wg, wo))
    dbt, dbp, dbg, dbo = (torch.empty(C, device=dev) for _ in range(4))
    ws = _ws(hip.lib().arfe_nonlocal_workspace_bytes(B, C, L), dev)
    if not split_streams:
        with probe('nonlocal_bwd'):
            check(hip.lib().arfe_nonlocal_bwd(B, C, L, ptr(dbsf), ptr(m), ptr(wt), ptr(wp), ptr(wg), ptr(wo), ptr(tpg), ptr(P),
                                              ptr(y), ptr(dP), ptr(dm), ptr(dwt), ptr(dbt), ptr(dwp), ptr(dbp), ptr(dwg),
                                              ptr(dbg), ptr(dwo), ptr(dbo), ptr(ws), stream()), "arfe_nonlocal_bwd")
        return dm, (dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo)
    L_ = hip.lib()
    on_side(dev, lambda: check(L_.arfe_nonlocal_bwd_wout(B, C, L, ptr(dbsf), ptr(y), ptr(dwo), ptr(dbo), stream()),
                               "arfe_nonlocal_bwd_wout"))
    args = (B, C, L, ptr(dbsf), ptr(wt), ptr(wp), ptr(wg), ptr(wo), ptr(tpg), ptr(P), ptr(y), ptr(dP), ptr(dm), ptr(ws))
    with probe('nonlocal_bwd'):
        check(L_.arfe_nonlocal_bwd_dtpg(*args, stream()), "arfe_nonlocal_bwd_dtpg")
    # the projections' parameter gradients need d(theta, phi, g) only: they go to the side stream BEFORE the last contraction
    # of the data path is enqueued (behind it they were the last kernel of the step to finish)
    on_side(dev, lambda: check(L_.arfe_nonlocal_bwd_wcat(B, C, L, ptr(m), ptr(dwt), ptr(dbt), ptr(dwp), ptr(dbp), ptr(dwg),
                                                         ptr(dbg), ptr(ws), stream()), "arfe_nonlocal_bwd_wcat"))
    with probe('nonlocal_bwd'):
        check(L_.arfe_nonlocal_bwd_dm(*args, stream()), "arfe_nonlocal_bwd_dm")
    # (ws, dP and the outputs were allocated on the current stream and are used on the side stream: the caller's join
    # orders every later use; record_stream keeps the allocator from re-issuing them before the side stream is through)
    side_reads(dev, ws, dP, dbsf, y, m, dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo)
    return dm, (dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo), (ws, dP)


# ---------------------------------------------------------------------------------------------------------
# AR-FPN autograd
# ---------------------------------------------------------------------------------------------------------

class ARFPNFunction(torch.autograd.Function):
    """WFPNDualSpatial.forward as one autograd node.
    args: refine_level, n, x_0..x_{n-1}, w1_0.., b1_0.., w2_0.., b2_0.., wt, bt, wp, bp, wg, bg, wo, bo"""

    @staticmethod
    def forward(ctx, refine_level, n, half_ops, *t):
        ctx.half_ops = half_ops
        with lib_precision(half_ops):
            return ARFPNFunction._forward(ctx, refine_level, n, *t)

    @staticmethod
    def _forward(ctx, refine_level, n, *t):
        xs = [v.contiguous() for v in t[0:n]]
        w1, b1 = list(t[n:2 * n]), list(t[2 * n:3 * n])
        w2, b2 = list(t[3 * n:4 * n]), list(t[4 * n:5 * n])
        wt, bt, wp, bp, wg, bg, wo, bo = t[5 * n:5 * n + 8]
        w1, b1, w2, b2 = ([v.contiguous() for v in l] for l in (w1, b1, w2, b2))
        need_grad = any(ctx.needs_input_grad)
        dev = xs[0].device
        # (round 2 ran the gate maps on the side stream beside gather + NonLocal: two reads of the pyramid.  The fused op
        # reads it once, in front of the NonLocal block.)
        m, argmax, a, b = fpn_gather_gate_op(xs, w1, b1, w2, b2, refine_level, want_argmax=need_grad)
        bsf, saved = nonlocal_fwd_op(m, wt, bt, wp, bp, wg, bg, wo, bo)
        outs = fpn_scatter_op(xs, bsf, a, b, refine_level)
        if _decision_tap is not None:
            _decision_tap.update(argmax=argmax, gate_a=a, gate_b=b)
        ctx.refine_level, ctx.n = refine_level, n
        if need_grad:
            ctx.save_for_backward(*xs, *w1, *w2, wt, wp, wg, wo, m, bsf, *a, *b, *saved,
                                  *( [argmax] if argmax is not None else []))
            ctx.has_argmax = argmax is not None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        with lib_precision(ctx.half_ops):
            return ARFPNFunction._backward(ctx, *gouts)

    @staticmethod
    def _backward(ctx, *gouts):
        n, r = ctx.n, ctx.refine_level
        s = ctx.saved_tensors
        xs, w1, w2 = list(s[0:n]), list(s[n:2 * n]), list(s[2 * n:3 * n])
        wt, wp, wg, wo, m, bsf = s[3 * n:3 * n + 6]
        a, b = list(s[3 * n + 6:4 * n + 6]), list(s[4 * n + 6:5 * n + 6])
        saved = s[5 * n + 6:5 * n + 9]
        argmax = s[5 * n + 9] if ctx.has_argmax else None
        gouts = [g.contiguous() if g is not None else torch.zeros_like(x) for g, x in zip(gouts, xs)]
        dev = xs[0].device
        dpa, dpb, dbsf = fpn_scatter_bwd_op(gouts, bsf, a, b, r)
        dw1, db1, dw2, db2 = on_side(dev, lambda: fpn_gate_wgrad_op(xs, dpa, dpb, r))   # beside the NonLocal backward
        # the NonLocal parameter gradients (five small latency-bound launches, 0.12 ms) leave the critical chain:
        # dw_out runs beside the data path, the theta / phi / g gradients beside the HBM-bound input gradient below
        dm, (dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo), keep = nonlocal_bwd_op(dbsf, m, wt, wp, wg, wo, saved, split_streams=True)
        dxs = fpn_input_bwd_op(gouts, dpa, dpb, w1, w2, dm, argmax, r)
        join_side(dev)
        del keep
        return (None, None, None, *dxs, *dw1, *db1, *dw2, *db2, dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo)


def arfpn(xs, w1, b1, w2, b2, wt, bt, wp, bp, wg, bg, wo, bo, refine_level=2, half_ops=False):
    n = len(xs)
    return ARFPNFunction.apply(refine_level, n, bool(half_ops), *xs, *w1, *b1, *w2, *b2, wt, bt, wp, bp, wg, bg, wo, bo)


# ---------------------------------------------------------------------------------------------------------
# AR-RFF RoI side
# ---------------------------------------------------------------------------------------------------------

def make_rois_op(rois):
    hip.require_f32_cuda(rois)
    h_rois, w_rois = torch.empty_like(rois), torch.empty_like(rois)
    check(hip.lib().arfe_rff_make_rois(rois.shape[0], ptr(rois), ptr(h_rois), ptr(w_rois), stream()),
          "arfe_rff_make_rois")
    return h_rois, w_rois


def map_roi_levels_op(rois, num_levels, finest_scale=56):
    hip.require_f32_cuda(rois)
    lv = torch.empty(rois.shape[0], dtype=torch.int64, device=rois.device)
    check(hip.lib().arfe_map_roi_levels(rois.shape[0], ptr(rois), num_levels, float(finest_scale), ptr(lv), stream()),
          "arfe_map_roi_levels")
    return lv


def rff_roi_sets_op(rois, num_levels, finest_scale=56):
    """-> (h_rois, w_rois, [levels_ori, levels_w, levels_h]): get_adaptive_scale_rois(rois, 1) and the level of every
    RoI of the three sets, one launch."""
    hip.require_f32_cuda(rois)
    R, dev = rois.shape[0], rois.device
    h_rois, w_rois = torch.empty_like(rois), torch.empty_like(rois)
    lv = [torch.empty(R, dtype=torch.int64, device=dev) for _ in range(3)]
    check(hip.lib().arfe_rff_roi_sets(R, ptr(rois), num_levels, float(finest_scale), ptr(h_rois), ptr(w_rois), ptr(lv[0]),
                                      ptr(lv[1]), ptr(lv[2]), stream()), "arfe_rff_roi_sets")
    return h_rois, w_rois, lv


def _stride_array(strides):
    arr = (_i32 * hip.MAX_LEVELS)()
    for i, s in enumerate(strides):
        arr[i] = int(s)
    return arr


def roialign_fwd_op(feats, strides, roi_sets, level_sets, out, pooled=7, sample_num=0):
    """RoIAlign of every set into its channel slice of `out` [R, nsets*C, 7, 7] (one launch for all sets).
    Half feature maps + a half `out` take the half-storage instantiation (fp16 mode)."""
    half = feats[0].dtype == torch.float16
    hip.require_cuda(torch.float16 if half else torch.float32, *feats, out)
    hip.require_f32_cuda(*roi_sets)
    pyr = Pyramid.of(feats, 0)
    nsets, R = len(roi_sets), roi_sets[0].shape[0]
    ws = _ws(hip.lib().arfe_roialign_fwd_workspace_bytes(nsets, R), out.device)
    fn = hip.lib().arfe_roialign_fwd_h if half else hip.lib().arfe_roialign_fwd
    with probe('roialign_fwd'):
        check(fn(ctypes.byref(pyr), ptr_array(feats), _stride_array(strides), nsets, R, ptr_array(roi_sets),
                 ptr_array(level_sets), pooled, sample_num, ptr(out), out.stride(0), ptr(ws), stream()), "arfe_roialign_fwd")


def roialign_fwd_prepare_op(feats, strides, roi_sets, level_sets, pooled=7, sample_num=0):
    """RoI-only part of the forward (processing order, weight tables) -> workspace for roialign_fwd_group_op."""
    pyr = Pyramid.of(feats, 0)
    nsets, R = len(roi_sets), roi_sets[0].shape[0]
    ws = _ws(hip.lib().arfe_roialign_fwd_workspace_bytes(nsets, R), feats[0].device)
    check(hip.lib().arfe_roialign_fwd_prepare(ctypes.byref(pyr), _stride_array(strides), nsets, R, ptr_array(roi_sets),
                                              ptr_array(level_sets), pooled, sample_num, ptr(ws), stream()),
          "arfe_roialign_fwd_prepare")
    return ws


def roialign_fwd_group_op(feats, strides, roi_sets, level_sets, out, ws, group_first, group_count, pooled=7, sample_num=0):
    """Extraction of processing groups [group_first, +group_count) (group g = RoI set (g + 1) % nsets) into `out`."""
    hip.require_f32_cuda(*feats, *roi_sets, out)
    pyr = Pyramid.of(feats, 0)
    nsets, R = len(roi_sets), roi_sets[0].shape[0]
    with probe('roialign_fwd'):
        check(hip.lib().arfe_roialign_fwd_group(ctypes.byref(pyr), ptr_array(feats), _stride_array(strides), nsets, R,
                                                ptr_array(roi_sets), ptr_array(level_sets), pooled, sample_num, ptr(out),
                                                out.stride(0), ptr(ws), group_first, group_count, stream()),
              "arfe_roialign_fwd_group")


def roialign_bwd_prepare_op(feats, strides, roi_sets, level_sets, pooled=7, sample_num=0):
    """The RoI-only part of the backward (tile lists, weight tables) -> workspace tensor; launched on the CURRENT stream
    (RoIExtractFunction.forward puts it on the side stream, beside the forward RoIAlign)."""
    pyr = Pyramid.of(feats, 0)
    nsets, R = len(roi_sets), roi_sets[0].shape[0]
    ws = _ws(hip.lib().arfe_roialign_bwd_workspace_bytes(ctypes.byref(pyr), nsets, R), feats[0].device)
    check(hip.lib().arfe_roialign_bwd_prepare(ctypes.byref(pyr), _stride_array(strides), nsets, R, ptr_array(roi_sets),
                                              ptr_array(level_sets), pooled, sample_num, ptr(ws), stream()),
          "arfe_roialign_bwd_prepare")
    return ws


def roialign_bwd_prezero_op(shapes, strides, roi_sets, prepared):
    """Allocates d(feats) and zeroes the tiles the scatter sums with atomics (current stream; needs `prepared`, the workspace of
    roialign_bwd_prepare_op, ordered before it).  -> dfeats for roialign_bwd_op(..., prezeroed=dfeats)."""
    dev = prepared.device
    dfeats = [torch.empty(shape, device=dev) for shape in shapes]
    pyr = Pyramid.of(dfeats, 0)
    nsets, R = len(roi_sets), roi_sets[0].shape[0]
    check(hip.lib().arfe_roialign_bwd_prezero(ctypes.byref(pyr), ptr_array(dfeats), _stride_array(strides), nsets, R,
                                              ptr(prepared), stream()), "arfe_roialign_bwd_prezero")
    return dfeats


def roialign_bwd_op(shapes, strides, roi_sets, level_sets, gout, pooled=7, sample_num=0, prepared=None, prezeroed=None):
    """d(feats) of all RoI sets at once; every element is written by the kernel (no zero-fill, no atomics).
    prepared: the workspace of roialign_bwd_prepare_op (already ordered before the current stream), or None.
    prezeroed: the dfeats of roialign_bwd_prezero_op (same prepared workspace, ordered before the current stream)."""
    dev = gout.device
    nsets, R = len(roi_sets), roi_sets[0].shape[0]
    if gout.dtype == torch.float16:          # fp16 mode: half gradient in, half feature-map gradients out
        dfeats = [torch.empty(shape, dtype=torch.float16, device=dev) for shape in shapes]
        pyr = Pyramid.of(dfeats, 0)
        if prepared is None:
            prepared = _ws(hip.lib().arfe_roialign_bwd_workspace_bytes(ctypes.byref(pyr), nsets, R), dev)
            check(hip.lib().arfe_roialign_bwd_prepare(ctypes.byref(pyr), _stride_array(strides), nsets, R, ptr_array(roi_sets),
                                                      ptr_array(level_sets), pooled, sample_num, ptr(prepared), stream()),
                  "arfe_roialign_bwd_prepare")
        with probe('roialign_bwd'):
            check(hip.lib().arfe_roialign_bwd_prepared_h(ctypes.byref(pyr), ptr_array(dfeats), _stride_array(strides), nsets, R,
                                                         ptr_array(roi_sets), ptr_array(level_sets), pooled, sample_num,
                                                         ptr(gout), gout.stride(0), ptr(prepared), stream()),
                  "arfe_roialign_bwd_prepared_h")
        return dfeats
    if prezeroed is not None:
        dfeats = list(prezeroed)
        pyr = Pyramid.of(dfeats, 0)
        with probe('roialign_bwd'):
            check(hip.lib().arfe_roialign_bwd_apply(ctypes.byref(pyr), ptr_array(dfeats), _stride_array(strides), nsets, R,
                                                    ptr_array(roi_sets), ptr_array(level_sets), pooled, sample_num, ptr(gout),
                                                    gout.stride(0), ptr(prepared), stream()), "arfe_roialign_bwd_apply")
        return dfeats
    dfeats = [torch.empty(shape, device=dev) for shape in shapes]
    pyr = Pyramid.of(dfeats, 0)
    with probe('roialign_bwd'):
        if prepared is None:
            ws = _ws(hip.lib().arfe_roialign_bwd_workspace_bytes(ctypes.byref(pyr), nsets, R), dev)
            check(hip.lib().arfe_roialign_bwd(ctypes.byref(pyr), ptr_array(dfeats), _stride_array(strides), nsets, R,
                                              ptr_array(roi_sets), ptr_array(level_sets), pooled, sample_num, ptr(gout),
                                              gout.stride(0), ptr(ws), stream()), "arfe_roialign_bwd")
        else:
            check(hip.lib().arfe_roialign_bwd_prepared(ctypes.byref(pyr), ptr_array(dfeats), _stride_array(strides), nsets,
                                                       R, ptr_array(roi_sets), ptr_array(level_sets), pooled, sample_num,
                                                       ptr(gout), gout.stride(0), ptr(prepared), stream()),
                  "arfe_roialign_bwd_prepared")
    return dfeats


class RoIExtractFunction(torch.autograd.Function):
    """SingleRoIExtractor.forward for `nsets` RoI sets at once; set s fills channels [s*C, (s+1)*C) of the
    [R, nsets*C, P, P] output (the concat of standard_roi_head.py:155 without a copy).
    args: strides, finest_scale, pooled, sample_num, nsets, have_levels, rois_0..rois_{nsets-1},
          [levels_0..levels_{nsets-1} when have_levels], feat_0..feat_{n-1}"""

    @staticmethod
    def forward(ctx, strides, finest_scale, pooled, sample_num, nsets, have_levels, *t):
        roi_sets = [r.contiguous() for r in t[:nsets]]
        nl = nsets if have_levels else 0
        feats = [f.contiguous() for f in t[nsets + nl:]]
        R, C = roi_sets[0].shape[0], feats[0].shape[1]
        dev = feats[0].device
        out = torch.empty(R, nsets * C, pooled, pooled, device=dev, dtype=feats[0].dtype)      # (half maps -> half features)
        if have_levels:
            levels = [l.contiguous() for l in t[nsets:nsets + nl]]
        else:
            levels = [map_roi_levels_op(rois, len(feats), finest_scale) for rois in roi_sets]
        need_grad = any(ctx.needs_input_grad[6 + nsets + nl:])
        prepared, ctx.prep_event = None, None
        if R > 0:
            if need_grad:
                # the backward's RoI-only half (tile lists, weight tables: six small dependent launches) runs on the
                # side stream beside the forward kernel instead of in front of the backward kernel
                prepared = on_side(dev, lambda: roialign_bwd_prepare_op(feats, strides, roi_sets, levels, pooled, sample_num))
                ctx.prep_event = side_stream(dev).record_event()
            roialign_fwd_op(feats, strides, roi_sets, levels, out, pooled, sample_num)
        ctx.cfg = (tuple(strides), pooled, sample_num, nsets, C)
        ctx.shapes = [f.shape for f in feats]
        ctx.has_prepared = prepared is not None
        ctx.save_for_backward(*roi_sets, *levels, *([prepared] if prepared is not None else []))
        return out

    @staticmethod
    def backward(ctx, gout):
        strides, pooled, sample_num, nsets, C = ctx.cfg
        s = ctx.saved_tensors
        roi_sets, levels = s[:nsets], s[nsets:2 * nsets]
        prepared = s[2 * nsets] if ctx.has_prepared else None
        gout = gout.contiguous()
        if prepared is not None:
            torch.cuda.current_stream().wait_event(ctx.prep_event)
        dfeats = roialign_bwd_op(ctx.shapes, strides, list(roi_sets), list(levels), gout, pooled, sample_num, prepared)
        nl = len(ctx.needs_input_grad) - 6 - nsets - len(dfeats)
        return (None, None, None, None, None, None, *([None] * (nsets + nl)), *dfeats)


def roi_extract(feats, roi_sets, strides, finest_scale=56, pooled=7, sample_num=0, levels=None):
    """levels: optional precomputed int64 level of every RoI of every set (rff_roi_sets_op); else map_roi_levels per set."""
    lv = list(levels) if levels is not None else []
    return RoIExtractFunction.apply(tuple(strides), finest_scale, pooled, sample_num, len(roi_sets), levels is not None,
                                    *roi_sets, *lv, *feats)


# ---------------------------------------------------------------------------------------------------------
# AR-RFF head raw ops + autograd
# ---------------------------------------------------------------------------------------------------------

def conv3x3_pack_op(w):
    C = w.shape[0]
    packed = _ws(hip.lib().arfe_conv3x3_packed_bytes(C), w.device)
    check(hip.lib().arfe_conv3x3_pack_weights(C, ptr(w), ptr(packed), stream()), "arfe_conv3x3_pack_weights")
    return packed


def conv3x3_fwd_op(x_view, roi_stride, R, C, packed, b, relu):
    y = torch.empty(R, C, 7, 7, device=b.device)
    with probe('conv3x3_fwd'):
        check(hip.lib().arfe_conv3x3_fwd(R, C, x_view, roi_stride, ptr(packed), ptr(b), int(relu), ptr(y), C * 49,
                                         stream()), "arfe_conv3x3_fwd")
    return y


def conv3x3_fwd_fuse_op(x_view, roi_stride, R, C, packed, b, ori_view, ori_roi_stride, lw):
    """-> (lh, z): lh = ReLU(conv(x) + b), z = ori + ori * (lw + lh) (the fusion rides in the conv's epilogue)."""
    lh = torch.empty(R, C, 7, 7, device=b.device)
    z = torch.empty(R, C, 7, 7, device=b.device)
    with probe('conv3x3_fwd'):
        check(hip.lib().arfe_conv3x3_fwd_fuse(R, C, x_view, roi_stride, ptr(packed), ptr(b), ori_view, ori_roi_stride, ptr(lw),
                                              ptr(lh), ptr(z), stream()), "arfe_conv3x3_fwd_fuse")
    return lh, z


def conv3x3_bwd_data_op(R, C, packed, dy, dx_view, dx_roi_stride, dx_mask=0, dx_mask_stride=0):
    """dx = conv^T(dy) (* [dx_mask > 0]); dy must already carry this conv's ReLU backward."""
    with probe('conv3x3_bwd_data'):
        check(hip.lib().arfe_conv3x3_bwd_data(R, C, ptr(packed), ptr(dy), C * 49, dx_view, dx_roi_stride, dx_mask,
                                              dx_mask_stride, stream()), "arfe_conv3x3_bwd_data")


def conv3x3_bwd_weight_op(x_view, x_roi_stride, R, C, w, dy, ws):
    dw = torch.empty_like(w)
    db = torch.empty(C, device=w.device)
    with probe('conv3x3_bwd_weight'):
        check(hip.lib().arfe_conv3x3_bwd_weight(R, C, x_view, x_roi_stride, ptr(dy), C * 49, ptr(dw), ptr(db), ptr(ws),
                                                stream()), "arfe_conv3x3_bwd_weight")
    return dw, db


def _linear_ws(R, in_f, out_f, device):
    """split-K scratch of the linear ops (per-slice partial sums, reduced in a fixed order instead of atomics)."""
    return _ws(hip.lib().arfe_linear_workspace_bytes(R, in_f, out_f), device)


def linear_fwd_op(x, w, b, relu):
    R, out_f, in_f = x.shape[0], w.shape[0], w.shape[1]
    y = torch.empty(R, out_f, device=x.device)
    ws = _linear_ws(R, in_f, out_f, x.device)
    check(hip.lib().arfe_linear_fwd(R, in_f, out_f, ptr(x), ptr(w), ptr(b), int(relu), ptr(y), ptr(ws), ws.numel(), stream()),
          "arfe_linear_fwd")
    return y


def linear_bwd_data_op(x, w, dy, dx=None, accumulate_dx=False, mask_dx_by_x=False):
    """dx = dy W (* [x > 0]); dy is used as is (ReLU backward is applied by the producer of dy)."""
    R, out_f, in_f = x.shape[0], w.shape[0], w.shape[1]
    if dx is None:
        dx = torch.empty(R, in_f, device=x.device)
    ws = _linear_ws(R, in_f, out_f, x.device)
    check(hip.lib().arfe_linear_bwd(R, in_f, out_f, ptr(x), ptr(w), ptr(dy), ptr(dx), int(accumulate_dx),
                                    int(mask_dx_by_x), 0, 0, ptr(ws), ws.numel(), stream()), "arfe_linear_bwd")
    return dx


# ---- gradient sinks: a parameter gradient produced straight into its consumer's buffer ------------------------------
# The fused AR-RFF node returns all its parameter gradients at once, when its backward is through.  A data-parallel reducer that
# wants a LARGE gradient on the wire earlier (fc1's weight is 51 of the head's 64 MB and exists 2 ms before the step ends)
# registers a sink for that parameter: the kernel then writes the gradient into sink.buffer() and sink.ready() is called on the
# stream that produced it; the node returns None for that parameter (the buffer is the reducer's p.grad already).
_grad_sinks = {}        # id(parameter) -> (weakref to the parameter, sink)


def register_grad_sink(param, sink):
    """sink: object with buffer() -> tensor shaped like param (fp32, contiguous) and ready() (called under the producing stream).
    sink=None removes the registration.  The registry is keyed by the Parameter OBJECT (its id, validated through a weak
    reference at lookup), not by its storage address: swapping `p.data` neither loses the sink nor lets a stale entry match
    an unrelated weight at a reused address, and a dead parameter's entry can never be hit."""
    import weakref
    if sink is None:
        _grad_sinks.pop(id(param), None)
    else:
        _grad_sinks[id(param)] = (weakref.ref(param), sink)


def grad_sink_of(t):
    """The sink registered for exactly this tensor object (an autograd leaf handed to a Function's forward), or None."""
    e = _grad_sinks.get(id(t))
    if e is None:
        return None
    if e[0]() is not t:             # the id was recycled by another object: the registration is dead
        _grad_sinks.pop(id(t), None)
        return None
    return e[1]


def linear_bwd_weight_op(x, w, dy, dw=None):
    R, out_f, in_f = x.shape[0], w.shape[0], w.shape[1]
    if dw is None:
        dw = torch.empty_like(w)
    db = torch.empty(out_f, device=x.device)
    ws = _linear_ws(R, in_f, out_f, x.device)
    check(hip.lib().arfe_linear_bwd(R, in_f, out_f, ptr(x), ptr(w), ptr(dy), 0, 0, 0, ptr(dw), ptr(db), ptr(ws), ws.numel(),
                                    stream()), "arfe_linear_bwd")
    return dw, db


def rff_prepack(w_wh, w_hh, w_f):
    """The three conv weights in the kernels' layouts, launched on the side stream: a caller that knows the head is
    coming (StandardRoIHead._bbox_forward) issues this before the RoI extraction, so the re-layout runs beside the
    RoIAlign kernel instead of in front of the first conv.  The head joins the side stream before it reads them."""
    ws = [w.detach().contiguous() for w in (w_wh, w_hh, w_f)]
    hip.require_f32_cuda(*ws)
    return on_side(ws[0].device, lambda: tuple(conv3x3_pack_op(w) for w in ws))


def stack_predictors(wc, bc, wr, br):
    """[fc_cls; fc_reg] as one weight / bias (the two predictors run as one contraction)."""
    return torch.cat([wc, wr], 0), torch.cat([bc, br], 0)


def _rff_head_forward(x, params, packs, gate=None):
    """MultiBBoxHead.forward on contiguous fp32 tensors -> (cls, reg, saved tensors for _rff_head_backward).
    packs: (pk_wh, pk_hh, pk_f) already ordered before the current stream, or None (packed here).
    gate: (lw, lh) when the caller has launched the two gating convs already (ordered before the current stream)."""
    w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br = params
    R, C3 = x.shape[0], x.shape[1]
    C = C3 // 3
    assert C3 == 3 * C and x.shape[2:] == (7, 7), "AR-RFF head expects [R, 3C, 7, 7]"
    wcr = bcr = None
    if packs is None:
        pk_wh, pk_hh, pk_f = conv3x3_pack_op(w_wh), conv3x3_pack_op(w_hh), conv3x3_pack_op(w_f)
    elif len(packs) == 5:
        pk_wh, pk_hh, pk_f, wcr, bcr = packs
    else:
        pk_wh, pk_hh, pk_f = packs
    rs = C3 * 49
    base = x.data_ptr()
    z = None
    if gate is not None:
        lw, lh = gate
    elif _GATE_FUSE:
        # the two gating convs one after the other on the current stream, the fusion in the second one's epilogue (no join, no
        # fusion launch) - measured 4.000 ms against 3.980 ms per step: the epilogue is a burst of all workgroups at once
        lw = conv3x3_fwd_op(base + 4 * C * 49, rs, R, C, pk_wh, b_wh, True)
        lh, z = conv3x3_fwd_fuse_op(base + 4 * 2 * C * 49, rs, R, C, pk_hh, b_hh, base, rs, lw)
    elif _GATE_SEQ:
        # the two gating convs one after the other on the current stream.  Side by side on two streams they took what two in a
        # row take (each workgroup needs a CU's whole LDS, so the second launch only gets the CUs the first one leaves: 137 +
        # 251 us against 2 x 131) and cost a join: 3.856 ms against 3.857 ms per step either way, one queue less to wait for
        lw = conv3x3_fwd_op(base + 4 * C * 49, rs, R, C, pk_wh, b_wh, True)
        lh = conv3x3_fwd_op(base + 4 * 2 * C * 49, rs, R, C, pk_hh, b_hh, True)
    else:
        # the two gating convs are independent: two 224-workgroup launches side by side pack the 256 CUs
        lh = on_side(x.device, lambda: conv3x3_fwd_op(base + 4 * 2 * C * 49, rs, R, C, pk_hh, b_hh, True))
        lw = conv3x3_fwd_op(base + 4 * C * 49, rs, R, C, pk_wh, b_wh, True)
        join_side(x.device)
    if z is None:
        z = torch.empty(R, C, 7, 7, device=x.device)
        check(hip.lib().arfe_rff_fuse_fwd(R, C, base, rs, ptr(lw), ptr(lh), ptr(z), stream()), "arfe_rff_fuse_fwd")
    z2 = conv3x3_fwd_op(z.data_ptr(), C * 49, R, C, pk_f, b_f, True)
    f1 = linear_fwd_op(z2.view(R, C * 49), w1, b1, True)
    f2 = linear_fwd_op(f1, w2, b2, True)
    # the two predictors are small and latency-bound: ONE contraction over the stacked weights [classes + 4*classes, fc],
    # cls / reg are the column blocks of its output (the backward stacks the two incoming gradients the same way)
    if wcr is None:
        wcr, bcr = stack_predictors(wc, bc, wr, br)
    ycr = linear_fwd_op(f2, wcr, bcr, False)
    nc = wc.shape[0]
    cls, reg = ycr[:, :nc], ycr[:, nc:]
    if _decision_tap is not None:
        _decision_tap.update(lw=lw, lh=lh, z2=z2, f1=f1, f2=f2)
    return cls, reg, (x, w_wh, w_hh, w_f, w1, w2, wcr, bc, lw, lh, z, z2, f1, f2, pk_wh, pk_hh, pk_f)


def _rff_head_backward(saved, dcls, dreg, join=True, defer_gate_wgrads=False, sink1=None):
    """-> (dx [R,3C,7,7], 14 parameter gradients in forward-argument order, keep-alive tensors[, late]).
    defer_gate_wgrads: do not launch the weight gradients of wh_conv / hh_conv; `late()` launches them on the side stream
    (ordered after everything enqueued on the current stream at THAT moment) and fills their four slots of the list.
    Every ReLU backward is folded into the epilogue of the kernel that PRODUCES the gradient.  Data gradients form the
    critical chain on the current stream; each parameter gradient is launched on the side stream as soon as its incoming
    gradient exists.  join=False leaves the side stream (the three conv weight gradients, 0.65 ms of matrix-pipe work
    nothing downstream reads) running: the caller joins after it has enqueued more data-path work."""
    x, w_wh, w_hh, w_f, w1, w2, wcr, bc, lw, lh, z, z2, f1, f2, pk_wh, pk_hh, pk_f = saved
    nc = bc.shape[0]                  # the stacked predictor weights split after the classifier's rows
    R, C3 = x.shape[0], x.shape[1]
    C = C3 // 3
    dev = x.device
    rs = C3 * 49
    nr = wcr.shape[0] - nc
    dycr = torch.cat([dcls if dcls is not None else torch.zeros(R, nc, device=dev),
                      dreg if dreg is not None else torch.zeros(R, nr, device=dev)], 1)
    ws = _ws(hip.lib().arfe_conv3x3_workspace_bytes(C), dev)
    z2f = z2.view(R, C * 49)

    def side_do(fn):
        return on_side(dev, fn)

    dwcr, dbcr = side_do(lambda: linear_bwd_weight_op(f2, wcr, dycr))
    dwc, dbc, dwr, dbr = dwcr[:nc], dbcr[:nc], dwcr[nc:], dbcr[nc:]
    df2 = linear_bwd_data_op(f2, wcr, dycr, mask_dx_by_x=True)                             # df2 *= [f2>0]
    dw2, db2 = side_do(lambda: linear_bwd_weight_op(f1, w2, df2))
    df1 = linear_bwd_data_op(f1, w2, df2, mask_dx_by_x=True)                               # df1 *= [f1>0]
    if sink1 is not None:           # fc1's gradient sink, looked up by the forward on the parameter object itself
        def fc1_wgrad():                            # straight into the reducer's bucket, on the wire from the side stream
            out = linear_bwd_weight_op(z2f, w1, df1, dw=sink1.buffer())
            sink1.ready()
            return None, out[1]
        dw1, db1 = side_do(fc1_wgrad)
    else:
        dw1, db1 = side_do(lambda: linear_bwd_weight_op(z2f, w1, df1))
    dz2 = linear_bwd_data_op(z2f, w1, df1, mask_dx_by_x=True)                              # dz2 *= [z2>0]
    dwf, dbf = side_do(lambda: conv3x3_bwd_weight_op(z.data_ptr(), C * 49, R, C, w_f, dz2, ws))
    dz = torch.empty(R, C, 7, 7, device=dev)
    conv3x3_bwd_data_op(R, C, pk_f, dz2, dz.data_ptr(), C * 49)
    dx = torch.empty_like(x)
    dlw = torch.empty(R, C, 7, 7, device=dev)
    dlh = torch.empty(R, C, 7, 7, device=dev)
    check(hip.lib().arfe_rff_fuse_bwd(R, C, x.data_ptr(), rs, ptr(lw), ptr(lh), ptr(dz), dx.data_ptr(), rs, ptr(dlw),
                                      ptr(dlh), stream()), "arfe_rff_fuse_bwd")
    # side stream: the two weight gradients back to back; main stream: the two data gradients back to back (each pair
    # of concurrent kernels fills the CUs the other leaves idle; three kernels in a row on one stream did not)
    pg = [None, None, None, None, dwf, dbf, dw1, db1, dw2, db2, dwc, dbc, dwr, dbr]

    def gate_wgrads():
        pg[0], pg[1] = side_do(lambda: conv3x3_bwd_weight_op(x.data_ptr() + 4 * C * 49, rs, R, C, w_wh, dlw, ws))
        pg[2], pg[3] = side_do(lambda: conv3x3_bwd_weight_op(x.data_ptr() + 8 * C * 49, rs, R, C, w_hh, dlh, ws))

    if not defer_gate_wgrads:
        gate_wgrads()
    conv3x3_bwd_data_op(R, C, pk_wh, dlw, dx.data_ptr() + 4 * C * 49, rs)
    conv3x3_bwd_data_op(R, C, pk_hh, dlh, dx.data_ptr() + 8 * C * 49, rs)
    if join:
        join_side(dev)
    # (dz2, dz, dlw, dlh, ws are read by side-stream kernels that may still be queued when join=False: they were allocated
    # on the current stream, whose allocator only re-issues them to work enqueued after the caller's join)
    keep = (dz2, dz, dlw, dlh, ws, df1, df2, dycr)
    if not join:
        side_reads(dev, *keep, x, z, z2f, f1, f2)
    if defer_gate_wgrads:
        return dx, pg, keep, gate_wgrads
    return dx, pg, keep


class RFFHeadFunction(torch.autograd.Function):
    """MultiBBoxHead.forward (num_shared_fcs=2): x [R,3C,7,7] -> (cls_score, bbox_pred).
    args: x, w_wh, b_wh, w_hh, b_hh, w_f, b_f, w_fc1, b_fc1, w_fc2, b_fc2, w_cls, b_cls, w_reg, b_reg,
          [pk_wh, pk_hh, pk_f from rff_prepack of the same weights, or None]"""

    @staticmethod
    def forward(ctx, x, w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br, pk_wh=None, pk_hh=None,
                pk_f=None, half_ops=False):
        ctx.half_ops = bool(half_ops)
        with lib_precision(ctx.half_ops):
            return RFFHeadFunction._forward(ctx, x, w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br, pk_wh,
                                            pk_hh, pk_f)

    @staticmethod
    def _forward(ctx, x, w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br, pk_wh, pk_hh, pk_f):
        x = x.contiguous()
        ctx.sink1 = grad_sink_of(w1) if w1.is_contiguous() else None
        params = [p.contiguous() for p in (w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br)]
        hip.require_f32_cuda(x, *params)
        packs = None
        if pk_wh is not None:
            join_side(x.device)          # the packs were launched on the side stream
            packs = (pk_wh, pk_hh, pk_f)
        cls, reg, saved = _rff_head_forward(x, params, packs)
        ctx.save_for_backward(*saved)
        return cls, reg

    @staticmethod
    def backward(ctx, dcls, dreg):
        with lib_precision(ctx.half_ops):
            dx, pg, _keep = _rff_head_backward(ctx.saved_tensors, dcls, dreg, join=True, sink1=ctx.sink1)
        return (dx, *pg, None, None, None, None)


def rff_head(x, w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br, packed=None, half_ops=False):
    pk = tuple(packed) if packed is not None else (None, None, None)
    return RFFHeadFunction.apply(x, w_wh, b_wh, w_hh, b_hh, w_f, b_f, w1, b1, w2, b2, wc, bc, wr, br, *pk, bool(half_ops))


class RFFExtractHeadFunction(torch.autograd.Function):
    """The AR-RFF body of `_bbox_forward` as ONE autograd node: 3-set RoIAlign concat (RoIExtractFunction) followed by
    MultiBBoxHead.forward (RFFHeadFunction) -> (cls_score, bbox_pred, bbox_feats).  Same kernels, same results; what the
    fusion buys is scheduling the two separate nodes cannot express without breaking autograd's contract that a node's
    returned gradients are ready on its stream:
      * backward: the conv / FC weight gradients (matrix-pipe bound, read by nothing downstream) stay on the side stream
        while the RoIAlign backward (latency bound) runs on the main one; the streams join once, at the end;
      * forward: weight re-layout and the RoI-only half of the RoIAlign backward run beside the RoIAlign forward.
    args: strides, pooled, sample_num, nfeat, feat_0..feat_{n-1}, rois, w_rois, h_rois, lv_ori, lv_w, lv_h, 14 head params"""

    @staticmethod
    def forward(ctx, strides, pooled, sample_num, nfeat, *t):
        feats = [f.contiguous() for f in t[:nfeat]]
        roi_sets = [r.contiguous() for r in t[nfeat:nfeat + 3]]
        levels = [l.contiguous() for l in t[nfeat + 3:nfeat + 6]]
        w1_leaf = t[nfeat + 6 + 6]
        ctx.sink1 = grad_sink_of(w1_leaf) if w1_leaf.is_contiguous() else None
        params = [p.contiguous() for p in t[nfeat + 6:nfeat + 20]]
        hip.require_f32_cuda(*feats, *roi_sets, *params)
        dev = feats[0].device
        R, C = roi_sets[0].shape[0], feats[0].shape[1]
        ctx.set_materialize_grads(False)
        packs = on_side(dev, lambda: (*(conv3x3_pack_op(w) for w in (params[0], params[2], params[4])),
                                      *stack_predictors(*params[10:14])))
        x = torch.empty(R, 3 * C, pooled, pooled, device=dev)
        prepared, ctx.prep_event = None, None
        if R > 0:
            if any(ctx.needs_input_grad[4:4 + nfeat]):
                shapes = [f.shape for f in feats]

                def prep():
                    ws = roialign_bwd_prepare_op(feats, strides, roi_sets, levels, pooled, sample_num)
                    # d(feats) exists from here on (183 MB at the bench size for the length of the step): the few tiles the
                    # backward sums with atomics are zeroed now, beside the forward, not in front of the backward kernel
                    return ws, roialign_bwd_prezero_op(shapes, strides, roi_sets, ws)
                prepared, ctx.dfeats = on_side(dev, prep)
                ctx.prep_event = side_stream(dev).record_event()
        gate = None
        if R > 0 and os.environ.get('ARFE_ROI_PIPELINE', '0') == '1':
            # Round 3 experiment, OFF by default (measured: no gain).  The three RoI sets are extracted one after the other
            # (w, h, ori: arfe_roialign_fwd_group) on this stream and each gating conv starts on the side stream as soon as
            # ITS set's channel slice is complete.  The extraction is vector-issue bound and so is the conv kernel's operand
            # staging (the MFMAs share that issue port): side by side the two just time-slice the CUs - extraction of a set
            # 100 -> 224 us, a gating conv 150 -> 270 us, 607 us from first extraction to last conv against 604 us in sequence.
            main, side = torch.cuda.current_stream(), side_stream(dev)
            ws_f = roialign_fwd_prepare_op(feats, strides, roi_sets, levels, pooled, sample_num)
            rs, base = 3 * C * 49, x.data_ptr()
            x.record_stream(side)
            roialign_fwd_group_op(feats, strides, roi_sets, levels, x, ws_f, 0, 1, pooled, sample_num)       # set 1: w-RoIs
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                lw = conv3x3_fwd_op(base + 4 * C * 49, rs, R, C, packs[0], params[1], True)
            roialign_fwd_group_op(feats, strides, roi_sets, levels, x, ws_f, 1, 1, pooled, sample_num)       # set 2: h-RoIs
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                lh = conv3x3_fwd_op(base + 4 * 2 * C * 49, rs, R, C, packs[1], params[3], True)
            roialign_fwd_group_op(feats, strides, roi_sets, levels, x, ws_f, 2, 1, pooled, sample_num)       # set 0: the RoIs
            gate = (lw, lh)
        elif R > 0:
            roialign_fwd_op(feats, strides, roi_sets, levels, x, pooled, sample_num)
        join_side(dev)                    # packs, the gating convs (and the prepare, long finished) before the head goes on
        cls, reg, saved = _rff_head_forward(x, params, packs, gate)
        ctx.cfg = (tuple(strides), pooled, sample_num, nfeat)
        ctx.shapes = [f.shape for f in feats]
        ctx.has_prepared = prepared is not None
        ctx.save_for_backward(*saved, *roi_sets, *levels, *([prepared] if prepared is not None else []))
        return cls, reg, x

    @staticmethod
    def backward(ctx, dcls, dreg, dfeat):
        strides, pooled, sample_num, nfeat = ctx.cfg
        s = ctx.saved_tensors
        saved, roi_sets, levels = s[:17], list(s[17:20]), list(s[20:23])
        prepared = s[23] if ctx.has_prepared else None
        dev = saved[0].device
        # (Measured: holding the gate convs' weight gradients back until the data-gradient chain is through, so that they
        # share the chip with the RoIAlign backward only, is SLOWER - that kernel's two contractions run on the fp32 matrix
        # pipe, and the two kinds of MFMA work just queue: 620 + 549 us side by side against 450 + 2 x 217 alone.)
        dx, pg, keep = _rff_head_backward(saved, dcls, dreg, join=False, sink1=ctx.sink1)
        dfeats = [None] * nfeat
        if any(ctx.needs_input_grad[4:4 + nfeat]):
            if dfeat is not None:             # a consumer of bbox_feats other than the head (not on the AR path)
                dx = dx + dfeat
            if prepared is not None:
                torch.cuda.current_stream().wait_event(ctx.prep_event)
            if roi_sets[0].shape[0] > 0:
                pz = getattr(ctx, 'dfeats', None) if prepared is not None else None
                ctx.dfeats = None             # a second backward (retain_graph) must not scatter into tensors already handed out
                dfeats = roialign_bwd_op(ctx.shapes, strides, roi_sets, levels, dx, pooled, sample_num, prepared, prezeroed=pz)
            else:
                dfeats = [torch.zeros(shape, device=dev) for shape in ctx.shapes]
        join_side(dev)                        # the parameter gradients
        del keep
        return (None, None, None, None, *dfeats, None, None, None, None, None, None, *pg)


def rff_extract_head(feats, roi_sets, levels, params, strides, pooled=7, sample_num=0):
    """-> (cls_score, bbox_pred, bbox_feats); roi_sets / levels in concat order (ori, w, h), params as RFFHeadFunction's."""
    return RFFExtractHeadFunction.apply(tuple(strides), pooled, sample_num, len(feats), *feats, *roi_sets, *levels, *params)


# ---------------------------------------------------------------------------------------------------------
# base FPN (reference necks/fpn.py): per-op autograd nodes, composed by necks.FPN
# ---------------------------------------------------------------------------------------------------------

class Conv1x1Function(torch.autograd.Function):
    """ConvModule(cin, cout, 1, act_cfg=None): y = W x + b on [B, Cin, H, W] (fpn.py:117-124)."""

    @staticmethod
    def forward(ctx, x, w, b, half_ops=False):
        x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
        hip.require_f32_cuda(x, w, b)
        B, cin, H, W = x.shape
        cout = w.shape[0]
        y = torch.empty(B, cout, H, W, device=x.device)
        ctx.half_ops = bool(half_ops)
        with lib_precision(ctx.half_ops):
            check(hip.lib().arfe_conv1x1_fwd(B, cin, cout, H * W, ptr(x), ptr(w), ptr(b), ptr(y), stream()), "arfe_conv1x1_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, cin, H, W = x.shape
        cout = w.shape[0]
        dx = None
        with lib_precision(ctx.half_ops):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                check(hip.lib().arfe_conv1x1_bwd_data(B, cin, cout, H * W, ptr(w), ptr(dy), ptr(dx), stream()),
                      "arfe_conv1x1_bwd_data")
            dw, db = torch.empty_like(w), torch.empty(cout, device=x.device)
            check(hip.lib().arfe_conv1x1_bwd_weight(B, cin, cout, H * W, ptr(x), ptr(dy), ptr(dw), ptr(db), stream()),
                  "arfe_conv1x1_bwd_weight")
        return dx, dw, db, None


class ConvMap3x3Function(torch.autograd.Function):
    """ConvModule(c, c, 3, padding=1, act_cfg=None) on whole feature maps (fpn.py:125-133)."""

    @staticmethod
    def forward(ctx, x, w, b, half_ops=False):
        x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
        hip.require_f32_cuda(x, w, b)
        B, C, H, W = x.shape
        if w.shape != (C, C, 3, 3):
            raise hip.ArfeHipError(f"conv3x3 on maps needs square C -> C weights, got {tuple(w.shape)} for C={C}")
        pk = conv3x3_pack_op(w)
        y = torch.empty_like(x)
        ctx.half_ops = bool(half_ops)
        with lib_precision(ctx.half_ops):
            check(hip.lib().arfe_convmap3x3_fwd(B, C, H, W, ptr(x), ptr(pk), ptr(b), ptr(y), stream()), "arfe_convmap3x3_fwd")
        ctx.save_for_backward(x, w, pk)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pk = ctx.saved_tensors
        dy = dy.contiguous()
        B, C, H, W = x.shape
        dx = None
        with lib_precision(ctx.half_ops):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                check(hip.lib().arfe_convmap3x3_bwd_data(B, C, H, W, ptr(pk), ptr(dy), ptr(dx), stream()),
                      "arfe_convmap3x3_bwd_data")
            dw, db = torch.empty_like(w), torch.empty(C, device=x.device)
            ws = _ws(hip.lib().arfe_conv3x3_workspace_bytes(C), x.device)
            check(hip.lib().arfe_convmap3x3_bwd_weight(B, C, H, W, ptr(x), ptr(dy), ptr(dw), ptr(db), ptr(ws), stream()),
                  "arfe_convmap3x3_bwd_weight")
        return dx, dw, db, None


class NearestAddFunction(torch.autograd.Function):
    """lat + F.interpolate(top, size=lat.shape[2:], mode='nearest') (fpn.py:180-188)."""

    @staticmethod
    def forward(ctx, lat, top):
        lat, top = lat.contiguous(), top.contiguous()
        hip.require_f32_cuda(lat, top)
        B, C, H, W = lat.shape
        h, w = top.shape[2:]
        out = torch.empty_like(lat)
        check(hip.lib().arfe_nearest_add_fwd(B, C, h, w, ptr(top), H, W, ptr(lat), ptr(out), stream()), "arfe_nearest_add_fwd")
        ctx.shapes = (B, C, h, w, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, h, w, H, W = ctx.shapes
        g = g.contiguous()
        dtop = torch.zeros(B, C, h, w, device=g.device)
        check(hip.lib().arfe_nearest_add_bwd(B, C, h, w, ptr(dtop), H, W, ptr(g), stream()), "arfe_nearest_add_bwd")
        return g, dtop


class Subsample2Function(torch.autograd.Function):
    """F.max_pool2d(x, 1, stride=2) (fpn.py:199-201)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        hip.require_f32_cuda(x)
        B, C, H, W = x.shape
        out = torch.empty(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=x.device)
        check(hip.lib().arfe_subsample2(B, C, H, W, ptr(x), ptr(out), 0, stream()), "arfe_subsample2")
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        g = g.contiguous()
        dx = torch.empty(B, C, H, W, device=g.device)
        check(hip.lib().arfe_subsample2(B, C, H, W, ptr(g), ptr(dx), 1, stream()), "arfe_subsample2")
        return dx


def conv1x1(x, w, b, half_ops=False):
    return Conv1x1Function.apply(x, w, b, bool(half_ops))


def convmap3x3(x, w, b, half_ops=False):
    return ConvMap3x3Function.apply(x, w, b, bool(half_ops))


def nearest_add(lat, top):
    return NearestAddFunction.apply(lat, top)


def subsample2(x):
    return Subsample2Function.apply(x)


# ---------------------------------------------------------------------------------------------------------
# fp16 mode: half storage at the module boundary, fp32 inside the library
# ---------------------------------------------------------------------------------------------------------
def _cast_op(x, to_half):
    x = x.contiguous()
    if not x.is_cuda:
        raise hip.ArfeHipError(f"the HIP cast kernels need a GPU tensor, got {x.device}")
    want = torch.float32 if to_half else torch.float16
    if x.dtype != want:
        raise hip.ArfeHipError(f"cast expects {want}, got {x.dtype}")
    out = torch.empty(x.shape, dtype=torch.float16 if to_half else torch.float32, device=x.device)
    fn = hip.lib().arfe_cast_f32_to_f16 if to_half else hip.lib().arfe_cast_f16_to_f32
    check(fn(x.numel(), ptr(x), ptr(out), stream()), "arfe_cast")
    return out


class CastFunction(torch.autograd.Function):
    """Tensor.to(half) / Tensor.to(float) of cast_tensor_type (mmdet/core/fp16/utils.py:7-23); the gradient takes the
    inverse conversion, as autograd's does."""

    @staticmethod
    def forward(ctx, x, to_half):
        ctx.to_half = to_half
        return _cast_op(x, to_half)

    @staticmethod
    def backward(ctx, g):
        return _cast_op(g, not ctx.to_half), None


def _cast_multi_op(ts, to_half):
    """one launch for a list of contiguous GPU tensors -> list of converted tensors"""
    ts = [t.contiguous() for t in ts]
    want = torch.float32 if to_half else torch.float16
    for t in ts:
        if not t.is_cuda or t.dtype != want:
            raise hip.ArfeHipError(f"cast expects {want} GPU tensors, got {t.dtype} on {t.device}")
    outs = [torch.empty(t.shape, dtype=torch.float16 if to_half else torch.float32, device=t.device) for t in ts]
    k = len(ts)
    if k == 0:
        return outs
    src = (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
    dst = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
    n = (ctypes.c_int64 * k)(*[t.numel() for t in ts])
    check(hip.lib().arfe_cast_multi(k, src, dst, n, int(to_half), stream()), "arfe_cast_multi")
    return outs


class MultiCastFunction(torch.autograd.Function):
    """cast_tensor_type over a LIST of tensors in one launch each way (the parameters of a module in fp16 mode)."""

    @staticmethod
    def forward(ctx, to_half, *ts):
        ctx.to_half = to_half
        return tuple(_cast_multi_op(ts, to_half))

    @staticmethod
    def backward(ctx, *gs):
        idx = [i for i, g in enumerate(gs) if g is not None]
        conv = _cast_multi_op([gs[i] for i in idx], not ctx.to_half)
        out = [None] * len(gs)
        for i, c in zip(idx, conv):
            out[i] = c
        return (None, *out)


def to_f32_list(ts):
    """[half tensors] -> [float tensors], one launch (float entries pass through)."""
    ts = list(ts)
    idx = [i for i, t in enumerate(ts) if t is not None and t.dtype == torch.float16]
    if len(idx) <= 1:
        return [to_f32(t) for t in ts]
    conv = MultiCastFunction.apply(False, *[ts[i] for i in idx])
    for i, c in zip(idx, conv):
        ts[i] = c
    return ts


def to_f32(t):
    """half -> float through the library (float tensors pass through)."""
    if t is None or t.dtype == torch.float32:
        return t
    return CastFunction.apply(t, False)


def to_f16(t):
    if t is None or t.dtype == torch.float16:
        return t
    return CastFunction.apply(t, True)


def storage_dtype(tensors, params, what):
    """The dtype a reference module would compute in: inputs and parameters must agree (fp32, or half after
    wrap_fp16_model); a mismatch is the RuntimeError torch's conv / linear raise in the reference."""
    dts = {t.dtype for t in list(tensors) + list(params) if t is not None}
    if len(dts) != 1 or next(iter(dts)) not in (torch.float32, torch.float16):
        raise RuntimeError(f"{what}: input and parameter dtypes should be the same fp32 or fp16 type, got "
                           f"{sorted(str(d) for d in dts)}")
    return next(iter(dts))


def f32_boundary(tensors, params, what):
    """-> (tensors, params, finish): fp32 views of a module's activations and parameters for the library, and
    `finish(out)` returning the outputs in the module's storage dtype (half in fp16 mode)."""
    dt = storage_dtype(tensors, params, what)
    if dt == torch.float32:
        def same(out):
            return out
        same.half_ops = False
        return list(tensors), list(params), same
    def finish(out):
        if isinstance(out, (tuple, list)):
            return type(out)(to_f16(o) for o in out)
        return to_f16(out)
    # half storage = the reference's fp16 mode: the module's contractions take half operands (pass finish.half_ops on)
    finish.half_ops = os.environ.get('ARFE_FP16_F32_PRODUCTS', '0') != '1'
    return [to_f32(t) for t in tensors], to_f32_list(params), finish
