"""AR-RFF: SingleRoIExtractor, MultiBBoxHead / MultiRoIsBBoxHead and the `_bbox_forward` glue of
StandardRoIHead, with the reference's registry names, constructor arguments and state_dict keys
(roi_extractors/single_level.py:10-152, bbox_heads/multirois_bbox_head.py:11-251, bbox_heads/bbox_head.py:14-75,
standard_roi_head.py:135-170 [intended body, :138-155], models/utils/additional.py:38-71)."""
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import functional as Fn
from .cnn import ConvModule, xavier_init
from .fp16 import force_fp32
from .registry import HEADS, ROI_EXTRACTORS, build_head, build_roi_extractor
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

    @force_fp32(apply_to=('feats', ), out_fp16=True)
    def forward(self, feats, rois, roi_scale_factor=None, lvl=None, replace_rois=None):
        if roi_scale_factor is not None or lvl is not None or replace_rois is not None:
            raise NotImplementedError("roi_scale_factor / lvl / replace_rois are unused by the AR-RFF path")
        return self._extract(feats, [rois])

    @force_fp32(apply_to=('feats', ), out_fp16=True)
    def extract_sets(self, feats, roi_sets):
        """RoIAlign of several RoI sets into one channel-concatenated tensor (set s -> channels [s*C,(s+1)*C));
        fp32 inside and half out in fp16 mode, like `forward` (single_level.py:109)."""
        return self._extract(feats, roi_sets)

    @force_fp32(apply_to=('feats', ), out_fp16=True)
    def extract_ar_sets(self, feats, rois):
        """The AR-RFF body of `_bbox_forward` (standard_roi_head.py:138-155): get_adaptive_scale_rois(rois, 1), the level
        of every RoI of the three sets (one launch for both) and the [ori, lw, lh] channel concat."""
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
        in_ch = in_channels * self.roi_feat_area
        self.fc_cls = nn.Linear(in_ch, num_classes + 1)
        self.fc_reg = nn.Linear(in_ch, 4 if reg_class_agnostic else 4 * num_classes)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_reg.bias, 0)


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
        return finish(Fn.rff_head(x, *p, packed=packed if x.dtype == torch.float32 else None))


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

    @property
    def with_bbox(self):
        return True

    def init_weights(self, pretrained=None):
        self.bbox_roi_extractor.init_weights()
        self.bbox_head.init_weights()

    def _bbox_forward(self, x, rois):
        return ar_rff_bbox_forward(self.bbox_roi_extractor, self.bbox_head, x, rois)

    def forward_dummy(self, x, proposals):
        from .bbox import bbox2roi
        r = self._bbox_forward(x, bbox2roi([proposals]))
        return (r['cls_score'], r['bbox_pred'])

    def forward_train(self, *a, **k):
        raise NotImplementedError("target assignment / losses are outside the AR-FPN + AR-RFF hot path")


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

    def forward_train(self, *a, **k):
        raise NotImplementedError("target assignment / box refinement / losses are outside the AR-FPN + AR-RFF hot path")
