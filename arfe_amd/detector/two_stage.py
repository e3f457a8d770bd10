"""Two-stage detector assembly of the harness (reference mmdet/models/detectors/two_stage.py:66-217, faster_rcnn.py,
cascade_rcnn.py:6-34): backbone -> neck (a LIST of neck configs becomes a Sequential: [FPN, WFPNDualSpatial]) -> RPN ->
RoI head, `forward_train` returning the loss dict, `forward_dummy` for FLOP counting.  Inference post-processing
(bbox2result, multiclass NMS, test-time augmentation) stays out of scope."""
import torch
import torch.nn as nn

from ..registry import DETECTORS, build_backbone, build_head, build_neck


@DETECTORS.register_module()
class TwoStageDetector(nn.Module):
    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        self.rpn_head = self.roi_head = None
        if rpn_head is not None:
            rpn_train_cfg = train_cfg['rpn'] if train_cfg is not None else None
            self.rpn_head = build_head(dict(rpn_head, train_cfg=rpn_train_cfg, test_cfg=test_cfg['rpn'] if test_cfg else None))
        if roi_head is not None:
            rcnn_train_cfg = train_cfg['rcnn'] if train_cfg is not None else None
            self.roi_head = build_head(dict(roi_head, train_cfg=rcnn_train_cfg, test_cfg=test_cfg['rcnn'] if test_cfg else None))
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return self.neck is not None

    @property
    def with_rpn(self):
        return self.rpn_head is not None

    @property
    def with_roi_head(self):
        return self.roi_head is not None

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            for m in (self.neck if isinstance(self.neck, nn.Sequential) else [self.neck]):
                m.init_weights()
        if self.with_rpn:
            self.rpn_head.init_weights()
        if self.with_roi_head:
            self.roi_head.init_weights(pretrained)

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def forward_dummy(self, img):
        outs = ()
        x = self.extract_feat(img)
        if self.with_rpn:
            outs = outs + tuple(self.rpn_head(x))
        proposals = torch.randn(1000, 4).to(img.device)
        return outs + tuple(self.roi_head.forward_dummy(x, proposals))

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None,
                      **kwargs):
        x = self.extract_feat(img)
        losses = dict()
        if self.with_rpn:
            rpn_outs = self.rpn_head(x)
            losses.update(self.rpn_head.loss(*rpn_outs, gt_bboxes, img_metas, gt_bboxes_ignore=gt_bboxes_ignore))
            proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg['rpn'] if self.test_cfg else None)
            proposal_list = self.rpn_head.get_bboxes(*rpn_outs, img_metas, cfg=proposal_cfg)
        else:
            proposal_list = proposals
        losses.update(self.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore,
                                                  gt_masks, **kwargs))
        return losses

    # ---- test time (detectors/test_mixins.py:27-30, two_stage.py:236-330, base.py:100-135; no augmentation, no masks) ----
    def simple_test_rpn(self, x, img_metas):
        return self.rpn_head.get_bboxes(*self.rpn_head(x), img_metas)

    @torch.no_grad()
    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        """one image [1, 3, H, W] -> per-class [n_c, 5] detections (x1, y1, x2, y2, score)"""
        x = self.extract_feat(img)
        proposal_list = self.simple_test_rpn(x, img_metas) if proposals is None else proposals
        return self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in ((imgs, 'imgs'), (img_metas, 'img_metas')):
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        if len(imgs) != 1:
            raise NotImplementedError("test-time augmentation is outside the harness")
        assert imgs[0].size(0) == 1, "the reference tests one image per GPU"
        if 'proposals' in kwargs:
            kwargs['proposals'] = kwargs['proposals'][0]
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if not return_loss:
            return self.forward_test(img, img_metas, **kwargs)
        return self.forward_train(img, img_metas, **kwargs)


@DETECTORS.register_module()
class FasterRCNN(TwoStageDetector):
    pass


@DETECTORS.register_module()
class CascadeRCNN(TwoStageDetector):
    pass


def parse_losses(losses):
    """the scalar to back-propagate and the log values (mmdet/apis/train.py:35-55, without the logging all-reduce)."""
    log_vars = {}
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f'{name} is not a tensor or list of tensors')
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    return loss, {k: v.item() for k, v in log_vars.items()}


def synthetic_batch(batch=2, height=800, width=1344, num_classes=80, max_gt=20, seed=0, device='cpu'):
    """COCO-shaped synthetic input: N(0,1) images [B, 3, H, W], 1..max_gt random boxes / labels per image."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, height, width, generator=g)
    gt_bboxes, gt_labels, metas = [], [], []
    for _ in range(batch):
        n = int(torch.randint(1, max_gt + 1, (1, ), generator=g))
        wh = torch.exp(torch.rand(n, 2, generator=g) * (torch.log(torch.tensor(400.0)) - torch.log(torch.tensor(24.0))) +
                       torch.log(torch.tensor(24.0)))
        xy = torch.rand(n, 2, generator=g) * (torch.tensor([width, height], dtype=torch.float32) - wh).clamp(min=1.0)
        gt_bboxes.append(torch.cat([xy, torch.minimum(xy + wh, torch.tensor([width - 1.0, height - 1.0]))], 1))
        gt_labels.append(torch.randint(0, num_classes, (n, ), generator=g))
        metas.append(dict(img_shape=(height, width, 3), pad_shape=(height, width, 3), ori_shape=(height, width, 3),
                          scale_factor=1.0, flip=False))
    return (img.to(device), metas, [b.to(device) for b in gt_bboxes], [l.to(device) for l in gt_labels])
