"""arfe_amd — MI355X-native (gfx950) AR-FPN / AR-RFF forward+backward behind the mmdetection plugin surface.

Public surface (names, ctor arguments and state_dict keys follow the reference):
    NECKS['WFPNDualSpatial' | 'FPN'], HEADS['MultiBBoxHead' | 'MultiRoIsBBoxHead' | 'StandardRoIHead'],
    ROI_EXTRACTORS['SingleRoIExtractor'], ops NonLocal2D / RoIAlign, build_neck / build_head / ...
All arithmetic runs in libarfe_hip.so (include/arfe_hip.h); there is no CPU or PyTorch fallback.
"""
from .registry import (ANCHOR_GENERATORS, BACKBONES, BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, DETECTORS, HEADS,
                       IOU_CALCULATORS, LOSSES, NECKS, ROI_EXTRACTORS, SHARED_HEADS, Registry, build, build_anchor_generator,
                       build_assigner, build_backbone, build_bbox_coder, build_detector, build_from_cfg, build_head,
                       build_loss, build_neck, build_roi_extractor, build_sampler)
from .cnn import ConvModule, constant_init, normal_init, xavier_init
from .non_local import NonLocal2D
from .necks import FPN, WFPNDualSpatial
from .roi_align import RoIAlign
from .roi_heads import (CascadeRoIHead, MultiBBoxHead, MultiRoIsBBoxHead, SingleRoIExtractor, StandardRoIHead,
                        get_adaptive_scale_rois)
from .config import Config
from .fp16 import auto_fp16, cast_tensor_type, force_fp32, patch_norm_fp32, wrap_fp16_model
from . import detector

__all__ = [
    'BACKBONES', 'DETECTORS', 'HEADS', 'LOSSES', 'NECKS', 'ROI_EXTRACTORS', 'SHARED_HEADS', 'Registry', 'build',
    'build_from_cfg', 'build_head', 'build_neck', 'build_roi_extractor', 'ConvModule', 'constant_init',
    'normal_init', 'xavier_init', 'NonLocal2D', 'WFPNDualSpatial', 'FPN', 'RoIAlign', 'MultiBBoxHead',
    'MultiRoIsBBoxHead', 'SingleRoIExtractor', 'StandardRoIHead', 'CascadeRoIHead', 'get_adaptive_scale_rois', 'Config', 'auto_fp16', 'force_fp32',
    'cast_tensor_type', 'patch_norm_fp32', 'wrap_fp16_model', 'detector', 'build_detector', 'build_backbone', 'build_loss',
    'build_anchor_generator', 'build_assigner', 'build_sampler', 'build_bbox_coder', 'ANCHOR_GENERATORS', 'BBOX_ASSIGNERS',
    'BBOX_SAMPLERS', 'BBOX_CODERS', 'IOU_CALCULATORS',
]
