"""AR-RFF: SingleRoIExtractor, MultiBBoxHead / MultiRoIsBBoxHead and the `_bbox_forward` glue of
StandardRoIHead, with the reference's registry names, constructor arguments and state_dict keys
(roi_extractors/single_level.py:10-152, bbox_heads/multirois_bbox_head.py:11-251, bbox_heads/bbox_head.py:14-75,
standard_roi_head.py:135-170 [intended body, :138-155], models/utils/additional.py:38-71)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from . import functional as Fn
from .cnn import ConvModule, xavier_init
from .fp16 import force_fp32
from .registry import (HEADS, ROI_EXTRACTORS, build_assigner, build_bbox_coder, build_head, build_loss,
                       build_roi_extractor, build_sampler)
from .roi_align import RoIAlign


def get_adaptive_scale_rois(rois, facs=1):
    """-> (adaptive_h_rois, adaptive_w_rois), additional.py:38-71 (facs must be 1 on the AR path)."""
    if facs != 1:
        raise NotImplementedError("the AR-RFF path calls get_adaptive_scale_rois(rois, 1)")
    return Fn.make_rois_op(rois.contiguous())


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def build_roi_layers(self, layer_cfg, featmap_strides):
        cfg = layer_cfg.copy()
        layer_type = cfg.pop('type')
        if layer_type != 'RoIAlign':
            raise NotImplementedError(f"roi_layer {layer_type}: only RoIAlign is on the AR path")
        return nn.ModuleList([RoIAlign(spatial_scale=1 / s, **cfg) for s in featmap_strides])

    def map_roi_levels(self, rois, num_levels):
        return Fn.map_roi_levels_op(rois.contiguous(), num_levels, self.finest_scale)

    def roi_rescale(self, rois, scale_factor):
        """single_level.py:95-107: the box scaled about its centre (RoI bookkeeping on [R, 5], plain tensor ops)."""
        cx = (rois[:, 1] + rois[:, 3]) * 0.5
        cy = (rois[:, 2] + rois[:, 4]) * 0.5
        new_w = (rois[:, 3] - rois[:, 1]) * scale_factor
        new_h = (rois[:, 4] - rois[:, 2]) * scale_factor
        return torch.stack((rois[:, 0], cx - new_w * 0.5, cy - new_h * 0.5, cx + new_w * 0.5, cy + new_h * 0.5), dim=-1)

    @force_fp32(apply_to=('feats', ), out_fp16=True)
    def forward(self, feats, rois, roi_scale_factor=None, lvl=None, replace_rois=None):
        """single_level.py:109-152.  The fork's three options (unused by the AR configs) are host-side RoI bookkeeping in front
        of the same RoIAlign launch: `replace_rois` - other boxes decide the levels; `lvl` - the levels shifted up or down and
        clamped; `roi_scale_factor` - the boxes rescaled about their centres AFTER the levels were taken from the originals."""
        if roi_scale_factor is None and lvl is None and replace_rois is None:
            return self._extract(feats, [rois])
        feats = list(feats)[:self.num_inputs]
        rois = rois.float().contiguous()
        num_levels = len(feats)
        src = replace_rois.float().contiguous() if replace_rois is not None else rois
        levels = self.map_roi_levels(src, num_levels)            # (one level: all zeros, the reference's shortcut)
        if lvl is not None:
            levels = (levels + lvl).clamp(min=0, max=num_levels - 1).long()
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor).contiguous()
        return self._extract(feats, [rois], [levels.contiguous()])

    @force_fp32(apply_to=('feats', ), out_fp16=True)
    def extract_sets(self, feats, roi_sets):
        """RoIAlign of several RoI sets into one channel-concatenated tensor (set s -> channels [s*C,(s+1)*C));
        fp32 inside and half out in fp16 mode, like `forward` (single_level.py:109)."""
        return self._extract(feats, roi_sets)

    def extract_ar_sets(self, feats, rois):
        """The AR-RFF body of `_bbox_forward` (standard_roi_head.py:138-155): get_adaptive_scale_rois(rois, 1), the level
        of every RoI of the three sets (one launch for both) and the [ori, lw, lh] channel concat.
        fp16 mode (single_level.py:109 @force_fp32(apply_to=('feats',), out_fp16=True)): half maps in, half features out.
        The kernels read / write half storage directly and compute in fp32 - the values of the decorator's
        half -> float -> RoIAlign -> half chain without its two conversion passes over the pyramid."""
        feats = list(feats)[:self.num_inputs]
        if self.fp16_enabled and all(f.is_cuda and f.dtype == torch.float16 for f in feats):
            rois = rois.float().contiguous()
            lh_rois, lw_rois, (lv_ori, lv_w, lv_h) = Fn.rff_roi_sets_op(rois, len(feats), self.finest_scale)
            return self._extract(feats, [rois, lw_rois, lh_rois], [lv_ori, lv_w, lv_h])
        return self._extract_ar_sets_f32(feats, rois)

    @force_fp32(apply_to=('feats', ), out_fp16=True)
    def _extract_ar_sets_f32(self, feats, rois):
        feats = list(feats)[:self.num_inputs]
        rois = rois.float().contiguous()
        lh_rois, lw_rois, (lv_ori, lv_w, lv_h) = Fn.rff_roi_sets_op(rois, len(feats), self.finest_scale)
        return self._extract(feats, [rois, lw_rois, lh_rois], [lv_ori, lv_w, lv_h])

    def _extract(self, feats, roi_sets, levels=None):
        layer = self.roi_layers[0]
        feats = list(feats)[:self.num_inputs]
        return Fn.roi_extract(feats, [r.float() for r in roi_sets], self.featmap_strides[:len(feats)], self.finest_scale,
                              layer.out_size[0], layer.sample_num, levels)


@HEADS.register_module()
class BBoxHead(nn.Module):
    """Constructor surface of the reference base head (bbox_head.py:19-65); losses/targets are out of scope."""

    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=256,
                 num_classes=80, bbox_coder=None, reg_class_agnostic=False, reg_decoded_bbox=False, loss_cls=None,
                 loss_bbox=None):
        super().__init__()
        assert with_cls or with_reg
        if with_avg_pool or not (with_cls and with_reg):
            raise NotImplementedError("AR-RFF head: with_avg_pool=False, with_cls=with_reg=True")
        self.with_avg_pool = with_avg_pool
        self.with_cls = with_cls
        self.with_reg = with_reg
        self.roi_feat_size = _pair(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.reg_class_agnostic = reg_class_agnostic
        self.reg_decoded_bbox = reg_decoded_bbox
        self.fp16_enabled = False
        self.bbox_coder_cfg, self.loss_cls_cfg, self.loss_bbox_cfg = bbox_coder, loss_cls, loss_bbox
        # training side (the harness): built only when the config names them, as the hot-path users
        # construct the head without coder / losses
        from . import detector  # noqa: F401  (registers the coder / loss classes)
        self.bbox_coder = build_bbox_coder(bbox_coder) if bbox_coder is not None else None
        self.loss_cls = build_loss(loss_cls) if loss_cls is not None else None
        self.loss_bbox = build_loss(loss_bbox) if loss_bbox is not None else None
        in_ch = in_channels * self.roi_feat_area
        self.fc_cls = nn.Linear(in_ch, num_classes + 1)
        self.fc_reg = nn.Linear(in_ch, 4 if reg_class_agnostic else 4 * num_classes)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)

    # ---- training targets / losses / box refinement (bbox_head.py:86-156, :158-205, :262-352): plain PyTorch host logic ----
    def _get_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg):
        """positives first, then negatives: labels (background = num_classes), label weights, encoded box targets."""
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        n = num_pos + num_neg
        labels = pos_bboxes.new_full((n, ), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros(n)
        bbox_targets, bbox_weights = pos_bboxes.new_zeros(n, 4), pos_bboxes.new_zeros(n, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels
            pw = cfg['pos_weight'] if isinstance(cfg, dict) else cfg.pos_weight
            label_weights[:num_pos] = 1.0 if pw <= 0 else pw
            bbox_targets[:num_pos, :] = (pos_gt_bboxes if self.reg_decoded_bbox else
                                         self.bbox_coder.encode(pos_bboxes, pos_gt_bboxes))
            bbox_weights[:num_pos, :] = 1.0
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, bbox_weights

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        per_img = [self._get_target_single(r.pos_bboxes, r.neg_bboxes, r.pos_gt_bboxes, r.pos_gt_labels, rcnn_train_cfg)
                   for r in sampling_results]
        cols = tuple([p[j] for p in per_img] for j in range(4))
        return tuple(torch.cat(c, 0) for c in cols) if concat else cols

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights, reduction_override=None):
        from .detector.losses import accuracy
        losses = dict()
        if cls_score is not None:
            cls_score = cls_score.float()                         # @force_fp32(apply_to=('cls_score', 'bbox_pred'))
            avg_factor = max(torch.sum(label_weights > 0).float().item(), 1.)
            if cls_score.numel() > 0:
                losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg_factor,
                                                   reduction_override=reduction_override)
                losses['acc'] = accuracy(cls_score, labels)
        if bbox_pred is not None:
            bbox_pred = bbox_pred.float()
            pos = (labels >= 0) & (labels < self.num_classes)
            if pos.any():
                if self.reg_decoded_bbox:
                    bbox_pred = self.bbox_coder.decode(rois[:, 1:], bbox_pred)
                if self.reg_class_agnostic:
                    pos_pred = bbox_pred.view(bbox_pred.size(0), 4)[pos]
                else:
                    pos_pred = bbox_pred.view(bbox_pred.size(0), -1, 4)[pos, labels[pos]]
                losses['loss_bbox'] = self.loss_bbox(pos_pred, bbox_targets[pos], bbox_weights[pos],
                                                     avg_factor=bbox_targets.size(0), reduction_override=reduction_override)
            else:
                losses['loss_bbox'] = bbox_pred.sum() * 0
        return losses

    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """Test-time decode (bbox_head.py:206-243): softmax scores (a list of stage scores is averaged first), every class's box
        decoded from its roi and clipped to the image, optionally mapped back to the original image scale; with a test cfg the
        per-class NMS follows (-> [k, 5] detections, [k] labels), without one the raw (boxes, scores)."""
        from .detector.boxes import multiclass_nms
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scores = F.softmax(cls_score.float(), dim=1) if cls_score is not None else None
        if bbox_pred is not None:
            bboxes = self.bbox_coder.decode(rois[:, 1:], bbox_pred.float(), max_shape=img_shape)
        else:
            # the reference "clamps" here with bboxes[:, [0, 2]].clamp_(...) - an in-place op on the COPY that advanced indexing
            # returns, i.e. a no-op (bbox_head.py:222-225): the rois come back unclipped, and so they do here (pinned by the fixture)
            bboxes = rois[:, 1:].clone()
        if rescale:
            if isinstance(scale_factor, float):
                bboxes = bboxes / scale_factor
            else:
                sf = bboxes.new_tensor(scale_factor)
                bboxes = (bboxes.view(bboxes.size(0), -1, 4) / sf).view(bboxes.size(0), -1)
        if cfg is None:
            return bboxes, scores
        get = cfg.get if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        return multiclass_nms(bboxes, scores, get('score_thr'), get('nms'), get('max_per_img'))

    def regress_by_class(self, rois, label, bbox_pred, img_meta):
        """the box of the given class applied to each roi ([n, 4] or [n, 5] with the batch column kept); Cascade R-CNN."""
        assert rois.size(1) in (4, 5), repr(rois.shape)
        bbox_pred = bbox_pred.float()
        if not self.reg_class_agnostic:
            idx = label * 4
            bbox_pred = torch.gather(bbox_pred, 1, torch.stack((idx, idx + 1, idx + 2, idx + 3), 1))
        assert bbox_pred.size(1) == 4
        if rois.size(1) == 4:
            return self.bbox_coder.decode(rois, bbox_pred, max_shape=img_meta['img_shape'])
        boxes = self.bbox_coder.decode(rois[:, 1:], bbox_pred, max_shape=img_meta['img_shape'])
        return torch.cat((rois[:, [0]], boxes), dim=1)

    def refine_bboxes(self, rois, labels, bbox_preds, pos_is_gts, img_metas):
        """per image: regress every roi by its (target) class, drop the rois that were ground-truth boxes (they lead the
        positives of each image) -> the next cascade stage's proposals."""
        img_ids = rois[:, 0].long().unique(sorted=True)
        assert img_ids.numel() <= len(img_metas)
        out = []
        for i in range(len(img_metas)):
            inds = torch.nonzero(rois[:, 0] == i, as_tuple=False).squeeze(dim=1)
            boxes = self.regress_by_class(rois[inds, 1:], labels[inds], bbox_preds[inds], img_metas[i])
            keep = pos_is_gts[i].new_ones(inds.numel())
            keep[:len(pos_is_gts[i])] = 1 - pos_is_gts[i]
            out.append(boxes[keep.type(torch.bool)])
        return out


@HEADS.register_module()
class MultiBBoxHead(BBoxHead):
    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                 num_reg_fcs=0, conv_out_channels=256, fc_out_channels=1024, num_ws_convs=2, num_ws_fcs=2,
                 conv_cfg=None, norm_cfg=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert (num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0)
        if (num_shared_convs, num_cls_convs, num_cls_fcs, num_reg_convs, num_reg_fcs) != (0, 0, 0, 0, 0) \
                or num_shared_fcs != 2:
            raise NotImplementedError("the AR-RFF configuration is MultiRoIsBBoxHead: 2 shared FCs, no branch convs/FCs")
        if self.roi_feat_size != (7, 7):
            raise NotImplementedError("the HIP conv kernels are specialised for 7x7 RoI features")
        if conv_out_channels != self.in_channels:
            raise ValueError("conv_out_channels must equal in_channels (the head slices x by conv_out_channels)")
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs = num_cls_convs, num_cls_fcs
        self.num_reg_convs, self.num_reg_fcs = num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self.num_ws_convs, self.num_ws_fcs = num_ws_convs, num_ws_fcs
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        c = self.in_channels
        self.hh_conv = ConvModule(c, c, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, inplace=False)
        self.wh_conv = ConvModule(c, c, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, inplace=False)
        self.final_conv = ConvModule(c, c, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, inplace=False)
        self.shared_convs = nn.ModuleList()
        self.shared_fcs = nn.ModuleList([nn.Linear(c * self.roi_feat_area, fc_out_channels),
                                         nn.Linear(fc_out_channels, fc_out_channels)])
        self.cls_convs, self.cls_fcs = nn.ModuleList(), nn.ModuleList()
        self.reg_convs, self.reg_fcs = nn.ModuleList(), nn.ModuleList()
        self.shared_out_channels = self.cls_last_dim = self.reg_last_dim = fc_out_channels
        self.relu = nn.ReLU(inplace=True)
        self.fc_cls = nn.Linear(self.cls_last_dim, self.num_classes + 1)
        self.fc_reg = nn.Linear(self.reg_last_dim, 4 if self.reg_class_agnostic else 4 * self.num_classes)

    def init_weights(self):
        super().init_weights()
        for module_list in [self.shared_fcs, self.cls_fcs, self.reg_fcs]:
            for m in module_list.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def prepack(self):
        """Launch the re-layout of the three conv weights on the side stream (fp32 mode; returns None in fp16 mode, where
        the parameters first pass the half -> float boundary inside forward).  Pass the result to forward(x, packed)."""
        ws = (self.wh_conv.conv.weight, self.hh_conv.conv.weight, self.final_conv.conv.weight)
        if any(w.dtype != torch.float32 or not w.is_cuda for w in ws):
            return None
        return Fn.rff_prepack(*ws)

    def _params(self):
        return [self.wh_conv.conv.weight, self.wh_conv.conv.bias, self.hh_conv.conv.weight, self.hh_conv.conv.bias,
                self.final_conv.conv.weight, self.final_conv.conv.bias, self.shared_fcs[0].weight,
                self.shared_fcs[0].bias, self.shared_fcs[1].weight, self.shared_fcs[1].bias, self.fc_cls.weight,
                self.fc_cls.bias, self.fc_reg.weight, self.fc_reg.bias]

    def forward(self, x, packed=None):
        # no @auto_fp16 on this forward in the reference (multirois_bbox_head.py:165): in fp16 mode x arrives as half
        # from the extractor (out_fp16) and meets half parameters
        params = self._params()
        (x,), p, finish = Fn.f32_boundary([x], params, 'MultiBBoxHead')
        return finish(Fn.rff_head(x, *p, packed=packed if x.dtype == torch.float32 else None, half_ops=finish.half_ops))


@HEADS.register_module()
class MultiRoIsBBoxHead(MultiBBoxHead):
    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                         num_reg_fcs=0, fc_out_channels=fc_out_channels, *args, **kwargs)


def ar_rff_bbox_forward(bbox_roi_extractor, bbox_head, x, rois):
    """The intended AR-RFF body of `_bbox_forward` (standard_roi_head.py:138-155, cascade_roi_head.py:126-135):
    get_adaptive_scale_rois, three RoIAlign sets, channel concat, MultiBBoxHead -> dict(cls_score, bbox_pred, bbox_feats).
    In fp32 with the stock extractor / head classes the whole body is one autograd node (Fn.rff_extract_head: same
    kernels, better stream overlap); otherwise (fp16 mode: the extractor is forced to fp32 and rounds its output to half,
    subclasses) the extractor and the head run as the two modules they are."""
    n = bbox_roi_extractor.num_inputs
    feats = list(x)[:n]
    fusable = (type(bbox_roi_extractor) is SingleRoIExtractor and isinstance(bbox_head, MultiBBoxHead)
               and type(bbox_head).forward is MultiBBoxHead.forward
               and all(f.dtype == torch.float32 and f.is_cuda for f in feats)
               and all(p.dtype == torch.float32 for p in bbox_head._params()))
    if fusable:
        rois = rois.float().contiguous()
        lh_rois, lw_rois, (lv_ori, lv_w, lv_h) = Fn.rff_roi_sets_op(rois, len(feats), bbox_roi_extractor.finest_scale)
        layer = bbox_roi_extractor.roi_layers[0]
        cls_score, bbox_pred, bbox_feats = Fn.rff_extract_head(
            feats, [rois, lw_rois, lh_rois], [lv_ori, lv_w, lv_h], bbox_head._params(),
            bbox_roi_extractor.featmap_strides[:len(feats)], layer.out_size[0], layer.sample_num)
    else:
        packed = bbox_head.prepack()      # weight re-layout on the side stream, beside the RoIAlign kernel
        # get_adaptive_scale_rois + torch.cat([ori, lw, lh], dim=1) of the reference, produced in place by the fused extractor
        bbox_feats = bbox_roi_extractor.extract_ar_sets(feats, rois)
        cls_score, bbox_pred = bbox_head(bbox_feats, packed)
    return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats)


def cascade_refine_labels(bbox_head, labels, cls_score):
    """The class whose box refines each RoI for the next cascade stage (cascade_roi_head.py:274-280 passes the target labels).
    Background targets carry label == num_classes; with a class-SPECIFIC regressor (reg_class_agnostic=False, the BBoxHead
    default) regress_by_class would gather columns 4*num_classes.. of a [n, 4*num_classes] prediction - an index error on the
    CPU and a device-side assert on the GPU.  The reference at HEAD has that hole (its cascade configs are all class-agnostic,
    where the labels are not used); background RoIs take the best foreground class instead, as upstream mmdetection does."""
    if bbox_head.reg_class_agnostic:
        return labels
    nc = bbox_head.num_classes
    return torch.where(labels == nc, cls_score[:, :nc].argmax(1), labels)


@HEADS.register_module()
class StandardRoIHead(nn.Module):
    """Only the bbox feature path of the reference's StandardRoIHead is in scope: `_bbox_forward` with the
    INTENDED AR-RFF body (standard_roi_head.py:138-155, a string literal at the reference's HEAD).  Assigners,
    samplers, losses, NMS and mask branches are the surrounding framework (OUT)."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 shared_head=None, train_cfg=None, test_cfg=None):
        super().__init__()
        if mask_roi_extractor is not None or mask_head is not None or shared_head is not None:
            raise NotImplementedError("mask / shared heads are outside the AR-RFF path")
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.bbox_head = build_head(bbox_head)
        self.init_assigner_sampler()

    @property
    def with_bbox(self):
        return True

    def init_assigner_sampler(self):
        """standard_roi_head.py:22-29."""
        self.bbox_assigner = self.bbox_sampler = None
        if self.train_cfg:
            from . import detector  # noqa: F401  (registers assigner / sampler classes)
            self.bbox_assigner = build_assigner(self.train_cfg['assigner'])
            self.bbox_sampler = build_sampler(self.train_cfg['sampler'])

    def init_weights(self, pretrained=None):
        self.bbox_roi_extractor.init_weights()
        self.bbox_head.init_weights()

    def _bbox_forward(self, x, rois):
        return ar_rff_bbox_forward(self.bbox_roi_extractor, self.bbox_head, x, rois)

    def forward_dummy(self, x, proposals):
        from .bbox import bbox2roi
        r = self._bbox_forward(x, bbox2roi([proposals]))
        return (r['cls_score'], r['bbox_pred'])

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None):
        """standard_roi_head.py:64-133 (bbox branch): assign + sample per image, the AR-RFF body on the sampled RoIs,
        targets and losses.  Host logic around the hot path."""
        if self.bbox_assigner is None:
            raise RuntimeError("StandardRoIHead.forward_train needs train_cfg (assigner / sampler)")
        num_imgs = len(img_metas)
        ignore = gt_bboxes_ignore if gt_bboxes_ignore is not None else [None] * num_imgs
        sampling_results = []
        for i in range(num_imgs):
            assign_result = self.bbox_assigner.assign(proposal_list[i], gt_bboxes[i], ignore[i], gt_labels[i])
            sampling_results.append(self.bbox_sampler.sample(assign_result, proposal_list[i], gt_bboxes[i], gt_labels[i]))
        losses = dict()
        losses.update(self._bbox_forward_train(x, sampling_results, gt_bboxes, gt_labels, img_metas)['loss_bbox'])
        return losses

    def _bbox_forward_train(self, x, sampling_results, gt_bboxes, gt_labels, img_metas):
        """standard_roi_head.py:172-190 with the two-value head of the AR path (HEAD passes a third `multi_cls` output of
        an ablation head through)."""
        from .bbox import bbox2roi
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(x, rois)
        bbox_targets = self.bbox_head.get_targets(sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        bbox_results.update(loss_bbox=self.bbox_head.loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                                          *bbox_targets))
        return bbox_results


    # ---- test time (test_mixins.py:52-71, standard_roi_head.py:266-285; bbox branch only) ----
    def simple_test_bboxes(self, x, img_metas, proposals, rcnn_test_cfg, rescale=False):
        from .bbox import bbox2roi
        rois = bbox2roi(proposals)
        res = self._bbox_forward(x, rois)
        return self.bbox_head.get_bboxes(rois, res['cls_score'], res['bbox_pred'], img_metas[0]['img_shape'],
                                         img_metas[0]['scale_factor'], rescale=rescale, cfg=rcnn_test_cfg)

    def simple_test(self, x, proposal_list, img_metas, proposals=None, rescale=False):
        """one image (the reference's test loop runs samples_per_gpu = 1) -> per-class [n_c, 5] arrays"""
        from .detector.boxes import bbox2result
        det_bboxes, det_labels = self.simple_test_bboxes(x, img_metas, proposal_list, self.test_cfg, rescale=rescale)
        return bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)


@HEADS.register_module()
class CascadeRoIHead(nn.Module):
    """The bbox feature path of the reference's CascadeRoIHead (cascade_roi_head.py:16-60 constructor / per-stage module
    lists, :126-149 `_bbox_forward(stage, x, rois)` with the INTENDED AR-RFF body, a string literal at the reference's HEAD
    exactly like StandardRoIHead's): one RoI extractor and one MultiRoIsBBoxHead per stage, each stage one pass of the
    same HIP path.  Box refinement between stages, assigners, samplers, losses and masks are the surrounding framework."""

    def __init__(self, num_stages, stage_loss_weights, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None,
                 mask_head=None, shared_head=None, train_cfg=None, test_cfg=None):
        super().__init__()
        assert bbox_roi_extractor is not None
        assert bbox_head is not None
        assert shared_head is None, 'Shared head is not supported in Cascade RCNN anymore'
        if mask_roi_extractor is not None or mask_head is not None:
            raise NotImplementedError("mask heads are outside the AR-RFF path")
        self.num_stages = num_stages
        self.stage_loss_weights = stage_loss_weights
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_bbox_head(bbox_roi_extractor, bbox_head)
        self.init_assigner_sampler()

    def init_assigner_sampler(self):
        """one assigner / sampler per stage (cascade_roi_head.py:83-91)."""
        self.bbox_assigner, self.bbox_sampler = [], []
        if self.train_cfg is not None:
            from . import detector  # noqa: F401
            for cfg in self.train_cfg:
                self.bbox_assigner.append(build_assigner(cfg['assigner']))
                self.bbox_sampler.append(build_sampler(cfg['sampler']))

    def init_bbox_head(self, bbox_roi_extractor, bbox_head):
        """One extractor and one head per stage; a single config dict stands for every stage."""
        def per_stage(cfg, what):
            cfgs = list(cfg) if isinstance(cfg, (list, tuple)) else [cfg] * self.num_stages
            assert len(cfgs) == self.num_stages, f"{what}: {len(cfgs)} configs for {self.num_stages} stages"
            return cfgs
        self.bbox_roi_extractor = nn.ModuleList(build_roi_extractor(c) for c in per_stage(bbox_roi_extractor, 'bbox_roi_extractor'))
        self.bbox_head = nn.ModuleList(build_head(c) for c in per_stage(bbox_head, 'bbox_head'))

    @property
    def with_bbox(self):
        return True

    def init_weights(self, pretrained=None):
        for i in range(self.num_stages):
            self.bbox_roi_extractor[i].init_weights()
            self.bbox_head[i].init_weights()

    def _bbox_forward(self, stage, x, rois):
        bbox_roi_extractor = self.bbox_roi_extractor[stage]
        bbox_head = self.bbox_head[stage]
        return ar_rff_bbox_forward(bbox_roi_extractor, bbox_head, x, rois)

    def forward_dummy(self, x, proposals):
        """bbox part of cascade_roi_head.py:101-112: every stage on the same proposals."""
        from .bbox import bbox2roi
        outs = ()
        rois = bbox2roi([proposals])
        for i in range(self.num_stages):
            r = self._bbox_forward(i, x, rois)
            outs = outs + (r['cls_score'], r['bbox_pred'])
        return outs

    def _bbox_forward_train(self, stage, x, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg):
        """cascade_roi_head.py:151-166."""
        from .bbox import bbox2roi
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(stage, x, rois)
        bbox_targets = self.bbox_head[stage].get_targets(sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg)
        loss_bbox = self.bbox_head[stage].loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois, *bbox_targets)
        bbox_results.update(loss_bbox=loss_bbox, rois=rois, bbox_targets=bbox_targets)
        return bbox_results

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None):
        """cascade_roi_head.py:192-283 (bbox branch): every stage assigns / samples the current proposals, runs the AR-RFF
        body, weighs its losses by stage_loss_weights and (but for the last) refines the boxes for the next stage."""
        if not self.bbox_assigner:
            raise RuntimeError("CascadeRoIHead.forward_train needs train_cfg (one assigner / sampler per stage)")
        losses = dict()
        num_imgs = len(img_metas)
        ignore = gt_bboxes_ignore if gt_bboxes_ignore is not None else [None] * num_imgs
        for i in range(self.num_stages):
            self.current_stage = i
            lw = self.stage_loss_weights[i]
            sampling_results = []
            for j in range(num_imgs):
                assign_result = self.bbox_assigner[i].assign(proposal_list[j], gt_bboxes[j], ignore[j], gt_labels[j])
                sampling_results.append(self.bbox_sampler[i].sample(assign_result, proposal_list[j], gt_bboxes[j], gt_labels[j]))
            bbox_results = self._bbox_forward_train(i, x, sampling_results, gt_bboxes, gt_labels, self.train_cfg[i])
            for name, value in bbox_results['loss_bbox'].items():
                losses[f's{i}.{name}'] = value * lw if 'loss' in name else value
            if i < self.num_stages - 1:
                pos_is_gts = [res.pos_is_gt for res in sampling_results]
                with torch.no_grad():
                    roi_labels = cascade_refine_labels(self.bbox_head[i], bbox_results['bbox_targets'][0],
                                                       bbox_results['cls_score'])
                    proposal_list = self.bbox_head[i].refine_bboxes(bbox_results['rois'], roi_labels,
                                                                    bbox_results['bbox_pred'], pos_is_gts, img_metas)
        return losses

    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        """cascade_roi_head.py:285-349, bbox branch: every stage on the boxes the previous one regressed (by its best class,
        background included - the regressors are class-agnostic in the cascade configs), the stages' class scores averaged, the
        last stage's boxes decoded -> per-class [n_c, 5] arrays."""
        from .bbox import bbox2roi
        from .detector.boxes import bbox2result
        rois = bbox2roi(proposal_list)
        ms_scores, res = [], None
        for i in range(self.num_stages):
            res = self._bbox_forward(i, x, rois)
            ms_scores.append(res['cls_score'].float())
            if i < self.num_stages - 1:
                label = res['cls_score'].argmax(dim=1)
                label = cascade_refine_labels(self.bbox_head[i], label, res['cls_score'])
                rois = self.bbox_head[i].regress_by_class(rois, label, res['bbox_pred'], img_metas[0])
        cls_score = sum(ms_scores) / self.num_stages
        det_bboxes, det_labels = self.bbox_head[-1].get_bboxes(rois, cls_score, res['bbox_pred'], img_metas[0]['img_shape'],
                                                               img_metas[0]['scale_factor'], rescale=rescale, cfg=self.test_cfg)
        return bbox2result(det_bboxes, det_labels, self.bbox_head[-1].num_classes)
