"""Box arithmetic of the detector harness in plain PyTorch: the callers on either side of the
AR-FPN / AR-RFF hot path.  Host-side logic only - nothing here is a hot-path kernel.

Reference behaviour (paths under mmdet/):
    bbox_overlaps / BboxOverlaps2D     core/bbox/iou_calculators/iou2d_calculator.py:6-125
    bbox2delta / delta2bbox / coder    core/bbox/coder/delta_xywh_bbox_coder.py:9-172
    AssignResult, MaxIoUAssigner       core/bbox/assigners/assign_result.py, max_iou_assigner.py:9-212
    SamplingResult, RandomSampler      core/bbox/samplers/sampling_result.py, base_sampler.py:31-110, random_sampler.py:8-70
    nms / batched_nms                  ops/nms/nms_wrapper.py:7-48,119-170 (greedy NMS of ops/nms/src/cpu/nms_cpu.cpp:5-68)
"""
import math

import torch

from ..registry import (BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS, build_iou_calculator)


# ------------------------------------------------------------------------------------------------ overlaps
def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """IoU (or intersection over the FIRST box's area, mode='iof') between two sets of (x1, y1, x2, y2) boxes;
    [rows, cols], or [rows] pairwise when is_aligned.  No +1 on widths (mmdet 2.0 convention), no epsilon."""
    assert mode in ('iou', 'iof')
    assert bboxes1.size(-1) == 4 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 4 or bboxes2.size(0) == 0
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new_empty((rows, 1) if is_aligned else (rows, cols))
    a = bboxes1 if is_aligned else bboxes1[:, None, :]
    b = bboxes2
    wh = (torch.min(a[..., 2:], b[..., 2:]) - torch.max(a[..., :2], b[..., :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area1 = (bboxes1[:, 2] - bboxes1[:, 0]) * (bboxes1[:, 3] - bboxes1[:, 1])
    if not is_aligned:
        area1 = area1[:, None]
    if mode == 'iof':
        return inter / area1
    area2 = (bboxes2[:, 2] - bboxes2[:, 0]) * (bboxes2[:, 3] - bboxes2[:, 1])
    return inter / (area1 + area2 - inter)


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in (0, 4, 5) and bboxes2.size(-1) in (0, 4, 5)
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        return bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + '()'


# ------------------------------------------------------------------------------------------------ delta coder
def _cxcywh(b):
    return (b[..., 0] + b[..., 2]) * 0.5, (b[..., 1] + b[..., 3]) * 0.5, b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """(dx, dy, dw, dh) of gt w.r.t. proposals, normalised: ((g - p) / size, log(size ratio) - mean) / std."""
    assert proposals.size() == gt.size()
    px, py, pw, ph = _cxcywh(proposals.float())
    gx, gy, gw, gh = _cxcywh(gt.float())
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return (deltas - deltas.new_tensor(means).unsqueeze(0)) / deltas.new_tensor(stds).unsqueeze(0)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None, wh_ratio_clip=16 / 1000):
    """Inverse of bbox2delta for deltas [N, 4*k] (k class-specific boxes per roi): dw, dh clamped to
    +-|log(wh_ratio_clip)|, boxes clamped to max_shape = (H, W) when given."""
    k = deltas.size(1) // 4
    d = deltas * deltas.new_tensor(stds).repeat(1, k) + deltas.new_tensor(means).repeat(1, k)
    dx, dy, dw, dh = d[:, 0::4], d[:, 1::4], d[:, 2::4], d[:, 3::4]
    lim = abs(math.log(wh_ratio_clip))
    dw, dh = dw.clamp(min=-lim, max=lim), dh.clamp(min=-lim, max=lim)
    px, py, pw, ph = (t.unsqueeze(1).expand_as(dx) for t in _cxcywh(rois))
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    if max_shape is not None:
        x1, x2 = x1.clamp(min=0, max=max_shape[1]), x2.clamp(min=0, max=max_shape[1])
        y1, y2 = y1.clamp(min=0, max=max_shape[0]), y2.clamp(min=0, max=max_shape[0])
    return torch.stack([x1, y1, x2, y2], dim=-1).view_as(deltas)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.)):
        self.means, self.stds = target_means, target_stds

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip)


# ------------------------------------------------------------------------------------------------ assigner
class AssignResult:
    """gt_inds: -1 ignore, 0 background, i+1 = assigned to gt i."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        """Prepends the ground-truth boxes as perfectly matched predictions (assign_result.py:191-206)."""
        n = len(gt_labels)
        self.gt_inds = torch.cat([torch.arange(1, n + 1, dtype=torch.long, device=gt_labels.device), self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(n), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    """Order of the rules matters (max_iou_assigner.py:60-106): background below neg_iou_thr, then the best gt when it
    reaches pos_iou_thr, then - match_low_quality - every gt claims the box(es) it overlaps most (all ties with
    gt_max_assign_all), later gts overriding earlier ones."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr = gt_max_assign_all, ignore_iof_thr
        self.ignore_wrt_candidates, self.match_low_quality = ignore_wrt_candidates, match_low_quality
        self.gpu_assign_thr = gpu_assign_thr
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        on_cpu = self.gpu_assign_thr > 0 and gt_bboxes.shape[0] > self.gpu_assign_thr
        device = bboxes.device
        if on_cpu:
            bboxes, gt_bboxes = bboxes.cpu(), gt_bboxes.cpu()
            gt_bboxes_ignore = None if gt_bboxes_ignore is None else gt_bboxes_ignore.cpu()
            gt_labels = None if gt_labels is None else gt_labels.cpu()
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                and bboxes.numel() > 0):
            if self.ignore_wrt_candidates:
                ign = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)[0]
            else:
                ign = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)[0]
            overlaps[:, ign > self.ignore_iof_thr] = -1
        res = self.assign_wrt_overlaps(overlaps, gt_labels)
        if on_cpu:
            res.gt_inds, res.max_overlaps = res.gt_inds.to(device), res.max_overlaps.to(device)
            if res.labels is not None:
                res.labels = res.labels.to(device)
        return res

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        k, n = overlaps.size(0), overlaps.size(1)
        gt_inds = overlaps.new_full((n, ), -1, dtype=torch.long)
        if k == 0 or n == 0:
            if k == 0:
                gt_inds[:] = 0
            labels = None if gt_labels is None else overlaps.new_full((n, ), -1, dtype=torch.long)
            return AssignResult(k, gt_inds, overlaps.new_zeros((n, )), labels=labels)
        max_ov, argmax_ov = overlaps.max(dim=0)              # best gt of every box
        gt_max, gt_argmax = overlaps.max(dim=1)              # best box of every gt
        if isinstance(self.neg_iou_thr, float):
            gt_inds[(max_ov >= 0) & (max_ov < self.neg_iou_thr)] = 0
        elif isinstance(self.neg_iou_thr, tuple):
            assert len(self.neg_iou_thr) == 2
            gt_inds[(max_ov >= self.neg_iou_thr[0]) & (max_ov < self.neg_iou_thr[1])] = 0
        pos = max_ov >= self.pos_iou_thr
        gt_inds[pos] = argmax_ov[pos] + 1
        if self.match_low_quality:
            # the reference walks the gts in order and lets gt i overwrite: per box, the LAST claiming gt wins
            claims = gt_max >= self.min_pos_iou
            if self.gt_max_assign_all:
                cand = (overlaps == gt_max[:, None]) & claims[:, None]
            else:
                cand = torch.zeros_like(overlaps, dtype=torch.bool)
                cand[torch.arange(k, device=overlaps.device), gt_argmax] = claims
            last = (cand.long() * torch.arange(1, k + 1, device=overlaps.device)[:, None]).max(dim=0)[0]
            gt_inds = torch.where(last > 0, last, gt_inds)
        labels = None
        if gt_labels is not None:
            labels = gt_inds.new_full((n, ), -1)
            pos_idx = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(1)
            if pos_idx.numel() > 0:
                labels[pos_idx] = gt_labels[gt_inds[pos_idx] - 1]
        return AssignResult(k, gt_inds, max_ov, labels=labels)


# ------------------------------------------------------------------------------------------------ sampler
class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """`num` boxes per image, at most num*pos_fraction positives, drawn with torch.randperm on the boxes' device
    (random_sampler.py:22-45); ground-truth boxes join the candidates first when add_gt_as_proposals."""

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction, self.neg_pos_ub = num, pos_fraction, neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals

    @staticmethod
    def random_choice(gallery, num):
        assert len(gallery) >= num
        return gallery[torch.randperm(gallery.numel(), device=gallery.device)[:num]]

    def _pick(self, mask, num_expected):
        inds = torch.nonzero(mask, as_tuple=False)
        if inds.numel() != 0:
            inds = inds.squeeze(1)
        return inds if inds.numel() <= num_expected else self.random_choice(inds, num_expected)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        if bboxes.dim() < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0], ), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8), gt_flags])
        pos_inds = self._pick(assign_result.gt_inds > 0, int(self.num * self.pos_fraction)).unique()
        num_neg = self.num - pos_inds.numel()
        if self.neg_pos_ub >= 0:
            num_neg = min(num_neg, int(self.neg_pos_ub * max(1, pos_inds.numel())))
        neg_inds = self._pick(assign_result.gt_inds == 0, num_neg).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


# ------------------------------------------------------------------------------------------------ NMS
def nms(dets, iou_thr):
    """Greedy NMS of [N, 5] (x1, y1, x2, y2, score) -> (kept dets, kept indices in descending score order): a box is
    dropped when a KEPT box of higher score overlaps it by more than iou_thr (strict, nms_cpu.cpp:55).
    Evaluated without a per-box loop: with boxes sorted by score and S[i, j] = (IoU > thr, i before j), iterate
    keep <- not any_i(S[i, :] & keep[i]) from keep = all; the fixed point is the greedy result (a box's fate only
    depends on boxes before it, so position p is final after p rounds; real lists settle in a handful)."""
    if dets.shape[0] == 0:
        return dets, dets.new_zeros(0, dtype=torch.long)
    order = dets[:, 4].sort(descending=True, stable=True)[1]
    boxes = dets[order, :4]
    supp = torch.triu(bbox_overlaps(boxes, boxes) > iou_thr, diagonal=1)
    keep = torch.ones(boxes.shape[0], dtype=torch.bool, device=dets.device)
    for _ in range(boxes.shape[0]):
        new = ~(supp & keep[:, None]).any(dim=0)
        if torch.equal(new, keep):
            break
        keep = new
    inds = order[keep]
    return dets[inds], inds


def batched_nms(bboxes, scores, inds, nms_cfg, class_agnostic=False):
    """NMS inside each group `inds` (level or class id) in one call: every group is shifted into its own coordinate
    range first (nms_wrapper.py:119-160).  -> ([k, 5] kept boxes with scores, keep indices)."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    kind = cfg.pop('type', 'nms')
    if kind != 'nms':
        raise NotImplementedError(f"nms type {kind}: the harness implements hard NMS")
    if bboxes.numel() == 0:
        return torch.cat([bboxes, scores[:, None]], -1), bboxes.new_zeros(0, dtype=torch.long)
    if class_agnostic:
        dets, keep = nms(torch.cat([bboxes, scores[:, None]], -1), **cfg)
        return torch.cat([bboxes[keep], dets[:, -1:]], -1), keep
    # The reference shifts every group into its own coordinate range and runs ONE greedy pass over all boxes; boxes of
    # different groups can then never suppress each other, so the same result comes from one pass PER GROUP on the shifted
    # coordinates (same arithmetic) - the all-pairs IoU of 5 x 2000 RPN boxes is 5 x [2000, 2000] instead of [10000, 10000].
    shifted = bboxes + (inds.to(bboxes) * (bboxes.max() + 1))[:, None]
    kept = []
    for g in torch.unique(inds).tolist():
        sel = torch.nonzero(inds == g, as_tuple=False).squeeze(1)
        _, k = nms(torch.cat([shifted[sel], scores[sel, None]], -1), **cfg)
        kept.append(sel[k])
    keep = torch.cat(kept).sort()[0]
    # descending score, ties by ascending index: the order of the single pass (stable sort of the scores)
    keep = keep[scores[keep].sort(descending=True, stable=True)[1]]
    return torch.cat([bboxes[keep], scores[keep, None]], -1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """Test-time selection of detections (core/post_processing/bbox_nms.py:6-60): boxes [n, 4] or [n, 4 * classes], scores
    [n, classes + 1] with the background LAST (ignored); every (box, class) pair above score_thr competes in a per-class hard NMS,
    the best max_num survive.  -> ([k, 5] boxes with scores in descending score order, [k] zero-based labels)."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(-1, num_classes, 4)
    scores = multi_scores[:, :-1]
    valid = scores > score_thr
    bboxes = bboxes[valid]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid]
    labels = valid.nonzero(as_tuple=False)[:, 1]
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0, ), dtype=torch.long)
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]


def bbox2result(bboxes, labels, num_classes):
    """[n, 5] detections + labels -> one float32 numpy array [n_c, 5] per class (core/bbox/transforms.py:73-89)."""
    import numpy as np
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    b, l = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [b[l == i, :] for i in range(num_classes)]
