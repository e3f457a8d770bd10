"""TEST INFRASTRUCTURE — CPU restatement (oracle) of ARFE's AR-FPN / AR-RFF hot path.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; the product package `arfe_amd` never does (it fails loudly without its HIP library).

Every function restates, in plain PyTorch-CPU fp32 (the reference's own arithmetic type),
what one reference function computes and cites it (paths relative to /root/reference):

  fpn_gather        mmdet/models/necks/wfpn_dual_spatial.py:102-113
  nonlocal_refine   mmdet/ops/non_local.py:63-70,78-105 (embedded_gaussian, use_scale=False)
  gate_maps         mmdet/models/necks/wfpn_dual_spatial.py:120-121 (+ ConvModule ReLU, B.1)
  fpn_scatter       mmdet/models/necks/wfpn_dual_spatial.py:125,135
  make_rois         mmdet/models/utils/additional.py:38-71  (facs = 1)
  map_roi_levels    mmdet/models/roi_heads/roi_extractors/single_level.py:53-93
  roi_extract3      .../single_level.py:109-152 + standard_roi_head.py:138-155 (intended body)
  rff_head          mmdet/models/roi_heads/bbox_heads/multirois_bbox_head.py:165-235

Pinning: tests/test_oracle_golden.py checks each of these against golden vectors generated
by importing the unmodified reference files in the build container (oracle/gen_golden.py,
fixtures in tests/golden/*.npz) and against the compiled reference RoIAlign (oracle/_ref).
The third-party `mmcv` layer underneath (ConvModule = conv + bias -> ReLU) is absent from
/root/reference and is restated, i.e. parity at that one boundary is "unpinned" (SURVEY §8c).
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------
# decisions: the discrete choices of a forward pass (test tooling, not part of the reference)
# ----------------------------------------------------------------------------------------
# Every gradient of the path is a smooth function of the inputs GIVEN the discrete choices the forward made: which ReLUs
# are open (gate convs, wh/hh/final convs, the two shared FCs) and which pixel of a pooling window is the maximum.  Two
# fp32 evaluations round a pre-activation of magnitude ~1e-7 to different signs a few times per million elements, and
# such an element then carries its FULL gradient in one evaluation and none in the other.  To separate that from
# arithmetic error the parity tests (i) COUNT the choices that differ between the reference evaluation and the device,
# and (ii) compare gradients against this restatement evaluated UNDER THE DEVICE'S CHOICES (`decisions=`), where every
# element is comparable in max-norm.  With decisions=None every function is the plain restatement of the reference.

class Decisions:
    """argmax: [r][B,C,hr,wr] flat h*w indices (adaptive_max_pool2d's return_indices layout) for the pooled levels;
    gate_a / gate_b: per level bool [B,1,h,w] (gate conv pre-activation > 0); lw, lh, z2: bool [R,C,7,7]; f1, f2: bool [R,fc]."""
    FIELDS = ('argmax', 'gate_a', 'gate_b', 'lw', 'lh', 'z2', 'f1', 'f2')

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw.get(k))

    @staticmethod
    def _items(v):
        if v is None:
            return []
        return list(v) if isinstance(v, (list, tuple)) else [v]

    def count(self):
        return sum(int(t.numel()) for k in self.FIELDS for t in self._items(getattr(self, k)))

    def differing(self, other):
        """-> {field: number of choices that differ} (fields present on both sides)."""
        out = {}
        for k in self.FIELDS:
            a, b = self._items(getattr(self, k)), self._items(getattr(other, k))
            if a and b:
                out[k] = sum(int((x.cpu().long() != y.cpu().long()).sum()) for x, y in zip(a, b))
        return out


class _ReLUGiven(torch.autograd.Function):
    """relu(x) with the open/closed choice GIVEN: y = x * mask, dy/dx = mask."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def relu_d(x, mask=None):
    if mask is None:
        return F.relu(x)
    return _ReLUGiven.apply(x, mask.to(x.dtype).view(x.shape))


def decisions_of(xs, parts_neck, parts_head, refine_level):
    """The choices a free-running evaluation of this restatement made (from the return_parts dictionaries)."""
    hr, wr = xs[refine_level].shape[2:]
    am = [F.adaptive_max_pool2d(x.detach(), (hr, wr), return_indices=True)[1] for x in xs[:refine_level]]
    return Decisions(argmax=am, gate_a=[a.detach() > 0 for a in parts_neck['a']],
                     gate_b=[b.detach() > 0 for b in parts_neck['b']],
                     **{k: parts_head[k].detach() > 0 for k in ('lw', 'lh', 'z2', 'f1', 'f2')})


# ----------------------------------------------------------------------------------------
# index semantics
# ----------------------------------------------------------------------------------------

def nearest_src_index(out_size, in_size):
    """PyTorch legacy mode='nearest' (F.interpolate(size=...)): src = min(floor(dst*scale), in-1)
    with scale = float32(in)/float32(out) evaluated in fp32 (wfpn_dual_spatial.py:110,125)."""
    scale = np.float32(in_size) / np.float32(out_size)
    d = np.arange(out_size, dtype=np.float32)
    return np.minimum(np.floor(d * scale).astype(np.int64), in_size - 1)


def adaptive_pool_windows(out_size, in_size):
    """F.adaptive_max_pool2d windows: [floor(o*in/out), ceil((o+1)*in/out)) (integer math)."""
    o = np.arange(out_size, dtype=np.int64)
    start = (o * in_size) // out_size
    end = -((-(o + 1) * in_size) // out_size)
    return start, end


def fpn_gather(xs, refine_level, explicit=False, argmax=None):
    """m = (((((0+G0)+G1)+G2)+G3)+G4)/5 ; G_i = adaptive-max (i<r) or nearest (i>=r) to level r's size.
    argmax (decision injection, see `Decisions`): per pooled level the flat h*w index of the window maximum, in the layout
    of adaptive_max_pool2d(return_indices=True); the pooled value is then READ at that index instead of searched for."""
    hr, wr = xs[refine_level].shape[2:]
    acc = 0
    for i, x in enumerate(xs):
        h, w = x.shape[2:]
        if i < refine_level:
            if argmax is not None:
                g = x.flatten(2).gather(2, argmax[i].long().flatten(2)).view(x.shape[0], x.shape[1], hr, wr)
            elif explicit:
                ys, ye = adaptive_pool_windows(hr, h)
                xs_, xe = adaptive_pool_windows(wr, w)
                rows = []
                for p in range(hr):
                    cols = [x[:, :, ys[p]:ye[p], xs_[q]:xe[q]].amax(dim=(2, 3)) for q in range(wr)]
                    rows.append(torch.stack(cols, dim=-1))
                g = torch.stack(rows, dim=-2)
            else:
                g = F.adaptive_max_pool2d(x, output_size=(hr, wr))
        else:
            if explicit:
                iy = torch.from_numpy(nearest_src_index(hr, h))
                ix = torch.from_numpy(nearest_src_index(wr, w))
                g = x[:, :, iy][:, :, :, ix]
            else:
                g = F.interpolate(x, size=(hr, wr), mode='nearest')
        acc = acc + g
    return acc / len(xs)


def nonlocal_refine(m, p, return_parts=False):
    """bsf = m + Wo * (softmax_rows(theta^T phi) g^T) + bo ; 1x1 convs with bias, no activation."""
    n, c, h, w = m.shape
    g = F.conv2d(m, p['refine.g.conv.weight'], p['refine.g.conv.bias']).view(n, c, -1)
    th = F.conv2d(m, p['refine.theta.conv.weight'], p['refine.theta.conv.bias']).view(n, c, -1)
    ph = F.conv2d(m, p['refine.phi.conv.weight'], p['refine.phi.conv.bias']).view(n, c, -1)
    pw = torch.matmul(th.permute(0, 2, 1), ph).softmax(dim=-1)          # [n, L, L]
    y = torch.matmul(pw, g.permute(0, 2, 1))                            # [n, L, c]
    y = y.permute(0, 2, 1).contiguous().reshape(n, c, h, w)
    bsf = m + F.conv2d(y, p['refine.conv_out.conv.weight'], p['refine.conv_out.conv.bias'])
    if return_parts:
        return bsf, dict(theta=th, phi=ph, g=g, y=y)
    return bsf


def gate_maps(x, w1, b1, w2, b2, masks=None):
    """a = tanh(relu(conv3x3(x;W1)+b1)), b likewise with W2 — zero padding 1, [B,1,h,w] each.
    masks = (mask_a, mask_b): ReLU decisions injected (see `Decisions`)."""
    ma, mb = masks if masks is not None else (None, None)
    a = torch.tanh(relu_d(F.conv2d(x, w1, b1, padding=1), ma))
    b = torch.tanh(relu_d(F.conv2d(x, w2, b2, padding=1), mb))
    return a, b


def fpn_scatter(x, bsf, a, b, explicit=False):
    """out = x + nearest(bsf -> (h,w)) * (a + b)   (nearest both up and down)."""
    h, w = x.shape[2:]
    if explicit:
        iy = torch.from_numpy(nearest_src_index(h, bsf.shape[2]))
        ix = torch.from_numpy(nearest_src_index(w, bsf.shape[3]))
        u = bsf[:, :, iy][:, :, :, ix]
    else:
        u = F.interpolate(bsf, size=[h, w])
    return x + u * (a + b)


def arfpn_forward(xs, p, refine_level=2, explicit=False, return_parts=False, decisions=None):
    """Whole AR-FPN (WFPNDualSpatial.forward, wfpn_dual_spatial.py:99-137). `p` = state_dict.
    decisions: optional `Decisions` of another implementation to evaluate under (None = the reference's own)."""
    dz = decisions
    m = fpn_gather(xs, refine_level, explicit, argmax=dz.argmax if dz is not None else None)
    bsf, parts = nonlocal_refine(m, p, return_parts=True)
    outs, As, Bs = [], [], []
    for i, x in enumerate(xs):
        a, b = gate_maps(x, p[f'reduce_convs.{i}.conv.weight'], p[f'reduce_convs.{i}.conv.bias'],
                         p[f'reduce_convs2.{i}.conv.weight'], p[f'reduce_convs2.{i}.conv.bias'],
                         masks=(dz.gate_a[i], dz.gate_b[i]) if dz is not None else None)
        outs.append(fpn_scatter(x, bsf, a, b, explicit))
        As.append(a)
        Bs.append(b)
    if return_parts:
        parts.update(m=m, bsf=bsf, a=As, b=Bs)
        return tuple(outs), parts
    return tuple(outs)


# ----------------------------------------------------------------------------------------
# base FPN -- restated from necks/fpn.py
# ----------------------------------------------------------------------------------------

def fpn_params(in_channels, out_channels=256, seed=0):
    """state_dict-shaped params of FPN(in_channels, out_channels, num_outs) with add_extra_convs=False:
    lateral_convs.i.conv (1x1) and fpn_convs.i.conv (3x3, padding 1), xavier-uniform, zero bias (fpn.py:117-138,161-164)."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for i, cin in enumerate(in_channels):
        p[f'lateral_convs.{i}.conv.weight'] = _xavier_uniform((out_channels, cin, 1, 1), gen)
        p[f'lateral_convs.{i}.conv.bias'] = torch.zeros(out_channels)
        p[f'fpn_convs.{i}.conv.weight'] = _xavier_uniform((out_channels, out_channels, 3, 3), gen)
        p[f'fpn_convs.{i}.conv.bias'] = torch.zeros(out_channels)
    return p


def fpn_forward(inputs, p, num_outs):
    """FPN.forward with start_level=0, end_level=-1, add_extra_convs=False, act/norm None (fpn.py:166-201):
    laterals = conv1x1(x_i)+b; top-down: lat[i-1] += nearest(lat[i], size of lat[i-1]); outs = conv3x3(lat_i)+b;
    extra levels = max_pool2d(outs[-1], 1, stride=2)."""
    n = len(inputs)
    lat = [F.conv2d(inputs[i], p[f'lateral_convs.{i}.conv.weight'], p[f'lateral_convs.{i}.conv.bias']) for i in range(n)]
    for i in range(n - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
    outs = [F.conv2d(lat[i], p[f'fpn_convs.{i}.conv.weight'], p[f'fpn_convs.{i}.conv.bias'], padding=1) for i in range(n)]
    for _ in range(num_outs - n):
        outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    return tuple(outs)


# ----------------------------------------------------------------------------------------
# AR-RFF
# ----------------------------------------------------------------------------------------

def make_rois(rois, facs=1.0):
    """get_adaptive_scale_rois: returns (h_rois, w_rois), same op order as additional.py:38-71."""
    cx = ((rois[:, 1] + rois[:, 3]) * 0.5).view(-1, 1)
    cy = ((rois[:, 2] + rois[:, 4]) * 0.5).view(-1, 1)
    rw = (rois[:, 3] - rois[:, 1] + 1.0).view(-1, 1)
    rh = (rois[:, 4] - rois[:, 2] + 1.0).view(-1, 1)
    lo = torch.ones_like(rw) * 0.1
    large_h = rh * ((rw / rh) * facs + 1.0)
    large_w = rw * ((rh / rw) * facs + 1.0)
    b = rois[:, 0].view(-1, 1)
    h_rois = torch.cat((b, torch.max(cx - rw * 0.5, lo), torch.max(cy - large_h * 0.5, lo),
                        cx + rw * 0.5, cy + large_h * 0.5), dim=-1)
    w_rois = torch.cat((b, torch.max(cx - large_w * 0.5, lo), torch.max(cy - large_h * 0.5, lo),
                        cx + large_w * 0.5, cy + large_h * 0.5), dim=-1)
    return h_rois, w_rois


def map_roi_levels(rois, num_levels=5, finest_scale=56):
    """floor(log2(sqrt(w*h)/56 + 1e-6)) clamped to [0, num_levels-1], int64 (single_level.py:68-92)."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvl = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvl.clamp(min=0, max=num_levels - 1).long()


# fp32 values v = scale/56 + 1e-6 at which torch-CPU's floor(log2(v)) steps to 1,2,3,4.
# 2 and 4 are exact; just below 8 and 16 the fp32 log2 result rounds UP to the integer, so the
# step sits one ulp early.  tests/test_oracle_golden.py sweeps +-64 ulp around each and the
# HIP kernel compares against exactly these constants (no transcendental on the device).
LEVEL_STEPS_F32 = np.array([2.0, 4.0, np.nextafter(np.float32(8), np.float32(0)),
                            np.nextafter(np.float32(16), np.float32(0))], dtype=np.float32)


def map_roi_levels_steps(rois, num_levels=5, finest_scale=56):
    """Same mapping without log2: count the steps v has reached (restated, used to pin the device form)."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    v = scale / finest_scale + 1e-6
    steps = torch.from_numpy(LEVEL_STEPS_F32[:num_levels - 1])
    return (v.view(-1, 1) >= steps.view(1, -1)).sum(dim=1).long()


_lib = None


def c_oracle():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, '_build', 'liboracle.so')
        if not os.path.exists(path):
            from oracle.build_ref import build_c_oracle
            build_c_oracle()
        _lib = ctypes.CDLL(path)
        _lib.oracle_roi_align_fwd.restype = ctypes.c_int32
        _lib.oracle_roi_align_bwd.restype = ctypes.c_int32
        _lib.oracle_set_threads.argtypes = [ctypes.c_int]
        _lib.oracle_set_threads.restype = None
    return _lib


def set_roi_align_threads(n):
    """Threads of the C RoIAlign restatement (forward split over RoIs, backward over channels: results are bit-identical
    for every thread count)."""
    c_oracle().oracle_set_threads(int(n))


def _fp(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def roi_align_fwd(feat, rois, scale, out_size=7, sample_num=0, out=None, chan_offset=0):
    """RoIAlign v2 forward through the C restatement (oracle/roi_align_oracle.c)."""
    feat = feat.contiguous().float()
    rois = rois.contiguous().float()
    B, C, H, W = feat.shape
    R = rois.shape[0]
    if out is None:
        out = torch.zeros(R, C, out_size, out_size)
    ctot = out.shape[1]
    view = out[:, chan_offset:chan_offset + C]
    rc = c_oracle().oracle_roi_align_fwd(
        _fp(feat), B, C, H, W, _fp(rois), R, ctypes.c_float(scale), out_size, out_size, sample_num,
        ctypes.cast(view.data_ptr(), ctypes.POINTER(ctypes.c_float)),
        ctypes.c_int64(ctot * out_size * out_size), ctypes.c_int64(out_size * out_size))
    if rc != 0:
        raise RuntimeError(f"RoIAlign: roi {-rc - 1} has negative size")
    return out


def roi_align_bwd(gout, rois, scale, feat_shape, sample_num=0):
    gout = gout.float()
    rois = rois.contiguous().float()
    B, C, H, W = feat_shape
    R, _, P, _ = gout.shape
    assert gout.stride(3) == 1 and gout.stride(2) == P
    d = torch.zeros(B, C, H, W)
    rc = c_oracle().oracle_roi_align_bwd(
        _fp(gout), ctypes.c_int64(gout.stride(0)), ctypes.c_int64(gout.stride(1)), _fp(rois), R,
        ctypes.c_float(scale), P, P, sample_num, _fp(d), B, C, H, W)
    if rc != 0:
        raise RuntimeError(f"RoIAlign: roi {-rc - 1} has negative size")
    return d


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, scale, out_size):
        ctx.save_for_backward(rois)
        ctx.scale, ctx.shape = scale, feat.shape
        return roi_align_fwd(feat, rois, scale, out_size)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        return roi_align_bwd(g.contiguous(), rois, ctx.scale, ctx.shape), None, None, None


def roi_extract(feats, rois, strides=(4, 8, 16, 32, 64), out_size=7, finest_scale=56):
    """SingleRoIExtractor.forward: per-RoI level choice, RoIAlign on that level (differentiable)."""
    R = rois.shape[0]
    C = feats[0].shape[1]
    lv = map_roi_levels(rois, len(feats), finest_scale)
    out = feats[0].new_zeros(R, C, out_size, out_size)
    for i, f in enumerate(feats):
        idx = (lv == i).nonzero(as_tuple=True)[0]
        if idx.numel():
            out = out.index_copy(0, idx, _RoIAlignFn.apply(f, rois[idx], 1.0 / strides[i], out_size))
    return out


def roi_extract3(feats, rois, strides=(4, 8, 16, 32, 64), out_size=7):
    """[ori | w-RoI | h-RoI] channel concat, [R, 3C, 7, 7] (standard_roi_head.py:138-155)."""
    h_rois, w_rois = make_rois(rois, 1.0)
    return torch.cat([roi_extract(feats, rois, strides, out_size),
                      roi_extract(feats, w_rois, strides, out_size),
                      roi_extract(feats, h_rois, strides, out_size)], dim=1)


def rff_head(x, p, return_parts=False, decisions=None):
    """MultiRoIsBBoxHead.forward: x [R,3C,7,7] -> (cls [R,81], reg [R,320]). `p` = state_dict.
    decisions: optional `Decisions` (ReLU masks of lw, lh, z2, f1, f2) to evaluate under."""
    dz = decisions
    mk = (lambda k: getattr(dz, k)) if dz is not None else (lambda k: None)
    c = p['wh_conv.conv.weight'].shape[0]
    ori, fw, fh = x[:, :c], x[:, c:2 * c], x[:, 2 * c:]
    lw = relu_d(F.conv2d(fw, p['wh_conv.conv.weight'], p['wh_conv.conv.bias'], padding=1), mk('lw'))
    lh = relu_d(F.conv2d(fh, p['hh_conv.conv.weight'], p['hh_conv.conv.bias'], padding=1), mk('lh'))
    z = ori + ori * (lw + lh)
    z2 = relu_d(F.conv2d(z, p['final_conv.conv.weight'], p['final_conv.conv.bias'], padding=1), mk('z2'))
    f1 = relu_d(F.linear(z2.flatten(1), p['shared_fcs.0.weight'], p['shared_fcs.0.bias']), mk('f1'))
    f2 = relu_d(F.linear(f1, p['shared_fcs.1.weight'], p['shared_fcs.1.bias']), mk('f2'))
    cls = F.linear(f2, p['fc_cls.weight'], p['fc_cls.bias'])
    reg = F.linear(f2, p['fc_reg.weight'], p['fc_reg.bias'])
    if return_parts:
        return cls, reg, dict(lw=lw, lh=lh, z=z, z2=z2, f1=f1, f2=f2)
    return cls, reg


# ----------------------------------------------------------------------------------------
# fp16 mode
# ----------------------------------------------------------------------------------------
class _HalfStore(torch.autograd.Function):
    """x.half() as a storage step: the value is rounded to IEEE half (kept in an fp32 tensor here) and so is the
    gradient that flows back through it (cast_tensor_type, mmdet/core/fp16/utils.py:7-23, is Tensor.to)."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


def half_store(t):
    return _HalfStore.apply(t)


def fp16_mode_forward(xs, p_fpn, p_neck, p_head, rois, num_outs=5, refine_level=2, decisions=None):
    """The path after wrap_fp16_model (mmdet/core/fp16/hooks.py:86-105): parameters hold half values (pass them as
    the fp32 images of those halves), FPN.forward casts its inputs to half (@auto_fp16, fpn.py:166), the neck and the
    head run on half tensors, the RoI extractor computes in fp32 and returns half (@force_fp32(feats), out_fp16,
    single_level.py:109).  This restatement computes every module in fp32 and rounds to half exactly where a tensor
    is STORED between modules - the accuracy contract of the HIP fp16 mode (the reference additionally rounds inside
    its half kernels; tests bound that difference against fixtures made by the reference itself).
    -> (pyramid, feats, bbox_feats, cls, reg), fp32 tensors holding half-representable values."""
    pyr = [half_store(t) for t in fpn_forward([half_store(x) for x in xs], p_fpn, num_outs)]
    feats = [half_store(t) for t in arfpn_forward(pyr, p_neck, refine_level, decisions=decisions)]
    bbox_feats = half_store(roi_extract3(feats, rois))
    cls, reg = rff_head(bbox_feats, p_head, decisions=decisions)
    return pyr, feats, bbox_feats, half_store(cls), half_store(reg)


# ----------------------------------------------------------------------------------------
# parameter construction (same shapes/names/init as the reference modules)
# ----------------------------------------------------------------------------------------

def _xavier_uniform(shape, gen):
    fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
    fan_in = shape[1] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[1]
    bound = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def arfpn_params(channels=256, num_levels=5, seed=0):
    """state_dict-shaped params, xavier-uniform convs / zero biases (wfpn_dual_spatial.py:94-97)."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for name in ('reduce_convs', 'reduce_convs2'):
        for i in range(num_levels):
            p[f'{name}.{i}.conv.weight'] = _xavier_uniform((1, channels, 3, 3), gen)
            p[f'{name}.{i}.conv.bias'] = torch.zeros(1)
    for name in ('g', 'theta', 'phi', 'conv_out'):
        p[f'refine.{name}.conv.weight'] = _xavier_uniform((channels, channels, 1, 1), gen)
        p[f'refine.{name}.conv.bias'] = torch.zeros(channels)
    return p


def rff_params(channels=256, fc=1024, num_classes=80, roi=7, seed=0):
    """MultiRoIsBBoxHead params (multirois_bbox_head.py:153-163, bbox_head.py:68-75)."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for name in ('hh_conv', 'wh_conv', 'final_conv'):
        p[f'{name}.conv.weight'] = _xavier_uniform((channels, channels, 3, 3), gen)
        p[f'{name}.conv.bias'] = torch.zeros(channels)
    p['shared_fcs.0.weight'] = _xavier_uniform((fc, channels * roi * roi), gen)
    p['shared_fcs.0.bias'] = torch.zeros(fc)
    p['shared_fcs.1.weight'] = _xavier_uniform((fc, fc), gen)
    p['shared_fcs.1.bias'] = torch.zeros(fc)
    p['fc_cls.weight'] = torch.randn((num_classes + 1, fc), generator=gen) * 0.01
    p['fc_cls.bias'] = torch.zeros(num_classes + 1)
    p['fc_reg.weight'] = torch.randn((4 * num_classes, fc), generator=gen) * 0.001
    p['fc_reg.bias'] = torch.zeros(4 * num_classes)
    return p


# ----------------------------------------------------------------------------------------
# synthetic workload of SURVEY §8(d) (shared by tests and bench so CPU and GPU see the same data)
# ----------------------------------------------------------------------------------------
PYRAMID_1333x800 = ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21))
STRIDES = (4, 8, 16, 32, 64)


def synth_pyramid(batch, channels=256, sizes=PYRAMID_1333x800, seed=1234):
    gen = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, channels, h, w, generator=gen) for h, w in sizes]


def synth_rois(batch, per_img=512, img_w=1344, img_h=800, seed=1234):
    """x1,y1 ~ U(0,0.8*(W,H)); w,h = exp(U(ln16, ln512)) clipped to the image; batch idx = repeat."""
    gen = torch.Generator().manual_seed(seed + 7)
    n = batch * per_img
    x1 = torch.rand(n, generator=gen) * 0.8 * img_w
    y1 = torch.rand(n, generator=gen) * 0.8 * img_h
    lo, hi = float(np.log(16.0)), float(np.log(512.0))
    w = torch.exp(lo + torch.rand(n, generator=gen) * (hi - lo))
    h = torch.exp(lo + torch.rand(n, generator=gen) * (hi - lo))
    x2 = torch.minimum(x1 + w, torch.tensor(float(img_w - 1)))
    y2 = torch.minimum(y1 + h, torch.tensor(float(img_h - 1)))
    b = torch.arange(batch, dtype=torch.float32).repeat_interleave(per_img)
    return torch.stack([b, x1, y1, x2, y2], dim=1)
