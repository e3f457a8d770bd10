"""RPN head of the detector harness (reference mmdet/models/dense_heads/rpn_head.py:12-120 on top of
anchor_head.py:14-553): 3x3 conv + ReLU, 1x1 objectness / delta convs per level (plain PyTorch-ROCm convolutions),
anchor targets (assign, sample, encode), sigmoid-CE + L1 losses and proposal generation (top-k, decode, per-level NMS)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..cnn import normal_init
from ..registry import (HEADS, build_anchor_generator, build_assigner, build_bbox_coder, build_loss, build_sampler)
from .anchors import anchor_inside_flags, images_to_levels, unmap
from .boxes import batched_nms


def _get(cfg, key, default=None):
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


@HEADS.register_module()
class RPNHead(nn.Module):
    def __init__(self, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8, 16, 32], ratios=[0.5, 1.0, 2.0],
                                       strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=(.0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0), train_cfg=None, test_cfg=None):
        super().__init__()
        self.in_channels, self.feat_channels, self.num_classes = in_channels, feat_channels, 1
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in ('FocalLoss', 'GHMC')
        self.cls_out_channels = 1 if self.use_sigmoid_cls else 2
        self.reg_decoded_bbox = reg_decoded_bbox
        self.background_label = 0                      # RPN: label 1 = object (rpn_head.py:14-16)
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls, self.loss_bbox = build_loss(loss_cls), build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(_get(self.train_cfg, 'assigner'))
            sampler_cfg = _get(self.train_cfg, 'sampler') if self.sampling else None
            if sampler_cfg is None:
                raise NotImplementedError("the RPN of the AR configs samples its anchors (RandomSampler)")
            self.sampler = build_sampler(sampler_cfg)
        self.fp16_enabled = False
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 1)

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            normal_init(m, std=0.01)

    def forward_single(self, x):
        x = F.relu(self.rpn_conv(x), inplace=True)
        return self.rpn_cls(x), self.rpn_reg(x)

    def forward(self, feats):
        outs = [self.forward_single(x) for x in feats]
        return [o[0] for o in outs], [o[1] for o in outs]

    # ------------------------------------------------------------------------------------------- targets
    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        """the same anchors for every image, valid flags per image from its pad_shape (anchor_head.py:117-148)."""
        multi_level = self.anchor_generator.grid_anchors(featmap_sizes, device)
        anchor_list = [multi_level for _ in img_metas]
        valid_flag_list = [self.anchor_generator.valid_flags(featmap_sizes, m['pad_shape'], device) for m in img_metas]
        return anchor_list, valid_flag_list

    def _get_targets_single(self, flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, gt_labels, img_meta):
        inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                     _get(self.train_cfg, 'allowed_border'))
        if not inside.any():
            return (None, ) * 6
        anchors = flat_anchors[inside, :]
        assign_result = self.assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore, None if self.sampling else gt_labels)
        sampling = self.sampler.sample(assign_result, anchors, gt_bboxes)
        n = anchors.shape[0]
        bbox_targets, bbox_weights = torch.zeros_like(anchors), torch.zeros_like(anchors)
        labels = anchors.new_full((n, ), self.background_label, dtype=torch.long)
        label_weights = anchors.new_zeros(n, dtype=torch.float)
        pos_inds, neg_inds = sampling.pos_inds, sampling.neg_inds
        if len(pos_inds) > 0:
            bbox_targets[pos_inds, :] = (sampling.pos_gt_bboxes if self.reg_decoded_bbox else
                                         self.bbox_coder.encode(sampling.pos_bboxes, sampling.pos_gt_bboxes))
            bbox_weights[pos_inds, :] = 1.0
            labels[pos_inds] = 1 if gt_labels is None else gt_labels[sampling.pos_assigned_gt_inds]
            pw = _get(self.train_cfg, 'pos_weight')
            label_weights[pos_inds] = 1.0 if pw <= 0 else pw
        if len(neg_inds) > 0:
            label_weights[neg_inds] = 1.0
        total = flat_anchors.size(0)
        return (unmap(labels, total, inside, fill=self.num_classes), unmap(label_weights, total, inside),
                unmap(bbox_targets, total, inside), unmap(bbox_weights, total, inside), pos_inds, neg_inds)

    def get_targets(self, anchor_list, valid_flag_list, gt_bboxes_list, img_metas, gt_bboxes_ignore_list=None,
                    gt_labels_list=None):
        num_imgs = len(img_metas)
        assert len(anchor_list) == len(valid_flag_list) == num_imgs
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        ignore = gt_bboxes_ignore_list or [None] * num_imgs
        glabels = gt_labels_list or [None] * num_imgs
        per_img = [self._get_targets_single(torch.cat(anchor_list[i]), torch.cat(valid_flag_list[i]), gt_bboxes_list[i],
                                            ignore[i], glabels[i], img_metas[i]) for i in range(num_imgs)]
        if any(r[0] is None for r in per_img):
            return None
        num_pos = sum(max(r[4].numel(), 1) for r in per_img)
        num_neg = sum(max(r[5].numel(), 1) for r in per_img)
        by_level = [images_to_levels([r[j] for r in per_img], num_level_anchors) for j in range(4)]
        return by_level[0], by_level[1], by_level[2], by_level[3], num_pos, num_neg

    # ------------------------------------------------------------------------------------------- loss
    def loss_single(self, cls_score, bbox_pred, anchors, labels, label_weights, bbox_targets, bbox_weights, num_total_samples):
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels.reshape(-1), label_weights.reshape(-1), avg_factor=num_total_samples)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4)
        if self.reg_decoded_bbox:
            bbox_pred = self.bbox_coder.decode(anchors.reshape(-1, 4), bbox_pred)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets.reshape(-1, 4), bbox_weights.reshape(-1, 4),
                                   avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        cls_scores, bbox_preds = [c.float() for c in cls_scores], [b.float() for b in bbox_preds]   # @force_fp32
        featmap_sizes = [f.size()[-2:] for f in cls_scores]
        assert len(featmap_sizes) == self.anchor_generator.num_levels
        device = cls_scores[0].device
        anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=device)
        targets = self.get_targets(anchor_list, valid_flag_list, gt_bboxes, img_metas, gt_bboxes_ignore_list=gt_bboxes_ignore)
        if targets is None:
            return None
        labels, label_weights, bbox_targets, bbox_weights, num_pos, num_neg = targets
        num_total = num_pos + num_neg if self.sampling else num_pos
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        all_anchors = images_to_levels([torch.cat(a) for a in anchor_list], num_level_anchors)
        per_level = [self.loss_single(cls_scores[i], bbox_preds[i], all_anchors[i], labels[i], label_weights[i],
                                      bbox_targets[i], bbox_weights[i], num_total) for i in range(len(cls_scores))]
        return dict(loss_rpn_cls=[p[0] for p in per_level], loss_rpn_bbox=[p[1] for p in per_level])

    # ------------------------------------------------------------------------------------------- proposals
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False):
        assert len(cls_scores) == len(bbox_preds)
        device = cls_scores[0].device
        mlvl_anchors = self.anchor_generator.grid_anchors([c.shape[-2:] for c in cls_scores], device=device)
        out = []
        for img_id, meta in enumerate(img_metas):
            cs = [c[img_id].detach().float() for c in cls_scores]
            bp = [b[img_id].detach().float() for b in bbox_preds]
            out.append(self._get_bboxes_single(cs, bp, mlvl_anchors, meta['img_shape'], meta.get('scale_factor', 1.0), cfg, rescale))
        return out

    def _get_bboxes_single(self, cls_scores, bbox_preds, mlvl_anchors, img_shape, scale_factor, cfg, rescale=False):
        cfg = self.test_cfg if cfg is None else cfg
        level_ids, scores_l, preds_l, anchors_l = [], [], [], []
        for idx, (score, pred) in enumerate(zip(cls_scores, bbox_preds)):
            assert score.size()[-2:] == pred.size()[-2:]
            score = score.permute(1, 2, 0)
            if self.use_sigmoid_cls:
                scores = score.reshape(-1).sigmoid()
            else:
                scores = score.reshape(-1, 2).softmax(dim=1)[:, :-1].reshape(-1)   # background is the LAST class since mmdet 2.0
            pred = pred.permute(1, 2, 0).reshape(-1, 4)
            anchors = mlvl_anchors[idx]
            nms_pre = _get(cfg, 'nms_pre')
            if nms_pre > 0 and scores.shape[0] > nms_pre:
                ranked, rank = scores.sort(descending=True)
                top = rank[:nms_pre]
                scores, pred, anchors = ranked[:nms_pre], pred[top, :], anchors[top, :]
            scores_l.append(scores); preds_l.append(pred); anchors_l.append(anchors)
            level_ids.append(scores.new_full((scores.size(0), ), idx, dtype=torch.long))
        scores, ids = torch.cat(scores_l), torch.cat(level_ids)
        proposals = self.bbox_coder.decode(torch.cat(anchors_l), torch.cat(preds_l), max_shape=img_shape)
        min_size = _get(cfg, 'min_bbox_size')
        if min_size > 0:
            ok = ((proposals[:, 2] - proposals[:, 0] >= min_size) & (proposals[:, 3] - proposals[:, 1] >= min_size))
            if not bool(ok.all()):
                proposals, scores, ids = proposals[ok], scores[ok], ids[ok]
        dets, _ = batched_nms(proposals, scores, ids, dict(type='nms', iou_thr=_get(cfg, 'nms_thr')))
        return dets[:_get(cfg, 'nms_post')]
