"""Detector harness around the AR-FPN / AR-RFF hot path: ResNet backbone, RPN, anchors, box coder,
assigner, sampler, losses and the two-stage assembly, as plain PyTorch-ROCm host code with the reference's registry
names, constructor arguments and state_dict keys - so that BASELINE configs #2-#5 (Faster / Cascade R-CNN with AR-FPN and
AR-RFF) build from the reference's config files and run forward_train end to end on synthetic COCO-shaped input.
The hot path inside (necks, RoI extractor, bbox head) is the HIP library; nothing here replaces or shadows it."""
from .anchors import AnchorGenerator, anchor_inside_flags, images_to_levels, unmap
from .boxes import (AssignResult, BboxOverlaps2D, DeltaXYWHBBoxCoder, MaxIoUAssigner, RandomSampler, SamplingResult,
                    batched_nms, bbox2delta, bbox2result, bbox_overlaps, delta2bbox, multiclass_nms, nms)
from .losses import CrossEntropyLoss, L1Loss, SmoothL1Loss, accuracy
from .resnet import Bottleneck, ResNet
from .rpn_head import RPNHead
from .two_stage import CascadeRCNN, FasterRCNN, TwoStageDetector, parse_losses, synthetic_batch

__all__ = ['AnchorGenerator', 'anchor_inside_flags', 'images_to_levels', 'unmap', 'AssignResult', 'BboxOverlaps2D',
           'DeltaXYWHBBoxCoder', 'MaxIoUAssigner', 'RandomSampler', 'SamplingResult', 'batched_nms', 'multiclass_nms', 'bbox2result', 'bbox2delta',
           'bbox_overlaps', 'delta2bbox', 'nms', 'CrossEntropyLoss', 'L1Loss', 'SmoothL1Loss', 'accuracy', 'Bottleneck',
           'ResNet', 'RPNHead', 'CascadeRCNN', 'FasterRCNN', 'TwoStageDetector', 'parse_losses', 'synthetic_batch']
