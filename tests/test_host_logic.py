"""CPU: the plugin surface mirrors the reference (registry contract, ctor arguments, state_dict keys, init)."""
import os

import pytest
import torch

import arfe_amd
from arfe_amd import hip


def test_registry_contract():
    assert 'WFPNDualSpatial' in arfe_amd.NECKS
    for n in ('MultiBBoxHead', 'MultiRoIsBBoxHead', 'StandardRoIHead', 'BBoxHead'):
        assert n in arfe_amd.HEADS
    assert 'SingleRoIExtractor' in arfe_amd.ROI_EXTRACTORS
    with pytest.raises(KeyError):
        arfe_amd.build_neck(dict(type='NoSuchNeck'))
    with pytest.raises(KeyError):
        arfe_amd.build_neck(dict(in_channels=3))
    # a LIST of cfgs composes into nn.Sequential (builder.py:13-20)
    seq = arfe_amd.build_neck([dict(type='WFPNDualSpatial', in_channels=8, num_levels=5),
                               dict(type='WFPNDualSpatial', in_channels=8, num_levels=5, refine_level=3)])
    assert isinstance(seq, torch.nn.Sequential) and seq[1].refine_level == 3

    R = arfe_amd.Registry('x')

    @R.register_module
    class A:
        pass

    @R.register_module()
    class B:
        pass

    assert R.get('A') is A and R.get('B') is B
    with pytest.raises(KeyError):
        R.register_module(A)


def test_neck_state_dict_and_init():
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=256, num_levels=5))
    sd = neck.state_dict()
    assert sum(v.numel() for v in sd.values()) == 286218
    for i in range(5):
        assert sd[f'reduce_convs.{i}.conv.weight'].shape == (1, 256, 3, 3)
        assert sd[f'reduce_convs2.{i}.conv.bias'].shape == (1,)
    for n in ('g', 'theta', 'phi', 'conv_out'):
        assert sd[f'refine.{n}.conv.weight'].shape == (256, 256, 1, 1)
    # NonLocal2D's own init zeroes conv_out; the neck's init_weights() overrides it with xavier-uniform (§3.3)
    assert neck.refine.conv_out.conv.weight.abs().max() == 0
    neck.init_weights()
    w = neck.refine.conv_out.conv.weight
    bound = (6.0 / 512) ** 0.5
    assert 0 < w.abs().max() <= bound + 1e-6
    assert all(v.abs().max() == 0 for k, v in neck.state_dict().items() if k.endswith('bias'))
    with pytest.raises(AssertionError):
        neck([torch.zeros(1, 256, 4, 4)] * 4)      # len(inputs) must equal num_levels (wfpn_dual_spatial.py:100)


def test_head_state_dict_and_init():
    head = arfe_amd.build_head(dict(
        type='MultiRoIsBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
        bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
        reg_class_agnostic=False, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_bbox=dict(type='L1Loss', loss_weight=1.0)))
    sd = head.state_dict()
    assert sum(v.numel() for v in sd.values()) == 16076945
    assert sd['wh_conv.conv.weight'].shape == (256, 256, 3, 3)
    assert sd['shared_fcs.0.weight'].shape == (1024, 12544) and sd['shared_fcs.1.weight'].shape == (1024, 1024)
    assert sd['fc_cls.weight'].shape == (81, 1024) and sd['fc_reg.weight'].shape == (320, 1024)
    head.init_weights()
    assert abs(head.fc_cls.weight.std().item() - 0.01) < 2e-3 and abs(head.fc_reg.weight.std().item() - 0.001) < 2e-4
    with pytest.raises(NotImplementedError):
        arfe_amd.build_head(dict(type='MultiBBoxHead', num_shared_convs=1, num_shared_fcs=2))


def test_roi_head_builds_from_cfg_dicts():
    rh = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=256, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7,
                       num_classes=80)))
    assert rh.bbox_roi_extractor.num_inputs == 5 and rh.bbox_roi_extractor.finest_scale == 56
    assert len(rh.bbox_roi_extractor.roi_layers) == 5 and rh.bbox_roi_extractor.roi_layers[2].spatial_scale == 1 / 16


def test_product_path_refuses_cpu_tensors():
    """No CPU/PyTorch fallback: a CPU tensor is an error, never a silent slow path."""
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=4, num_levels=2, refine_level=0))
    with pytest.raises(hip.ArfeHipError):
        neck([torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 2, 2)])


def test_product_package_never_imports_the_oracle():
    import glob
    from conftest import ROOT
    for f in glob.glob(os.path.join(ROOT, 'arfe_amd', '**', '*.py'), recursive=True):
        src = open(f).read()
        assert 'oracle' not in src, f"{f} references the oracle"


def test_config_loader(tmp_path):
    base = tmp_path / 'base.py'
    base.write_text("model = dict(type='FasterRCNN', neck=dict(type='FPN', in_channels=[256, 512], out_channels=256),\n"
                    "             roi_head=dict(type='StandardRoIHead', bbox_head=dict(type='Shared2FCBBoxHead', n=1)))\n"
                    "lr = 0.02\n")
    child = tmp_path / 'child.py'
    child.write_text("_base_ = './base.py'\n"
                     "model = dict(neck=[dict(type='FPN', in_channels=[256, 512], out_channels=256, num_outs=5),\n"
                     "                   dict(type='WFPNDualSpatial', in_channels=256, num_levels=5)],\n"
                     "             roi_head=dict(bbox_head=dict(_delete_=True, type='MultiRoIsBBoxHead', in_channels=256)))\n")
    cfg = arfe_amd.Config.fromfile(str(child))
    assert cfg.lr == 0.02 and cfg.model.type == 'FasterRCNN'
    assert cfg.model.neck[1].type == 'WFPNDualSpatial'
    assert cfg.model.roi_head.bbox_head == dict(type='MultiRoIsBBoxHead', in_channels=256)   # _delete_ replaced it
    assert cfg.model.roi_head.type == 'StandardRoIHead'
    cfg.merge_from_dict({'model.neck': None, 'lr': 0.01})
    assert cfg.lr == 0.01 and cfg.model.neck is None


REF = '/root/reference/configs'


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference configs exist only in the build container')
def test_reference_configs_load_unchanged():
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        cfg = arfe_amd.Config.fromfile(os.path.join(REF, 'mytrain', 'faster_rcnn_r50_drfpn_drrff+fac_1x_coco.py'))
    assert any('stray' in str(x.message) for x in w)           # the flagship config's back-tick is flagged
    assert cfg.model.neck[1] == dict(type='WFPNDualSpatial', in_channels=256, num_levels=5)
    neck = arfe_amd.build_neck(cfg.model.neck[1])
    assert neck.refine_level == 2
    # the whole neck list of the flagship config builds: [FPN, WFPNDualSpatial] -> nn.Sequential (builder.py:14-18)
    assert cfg.model.neck[0]['type'] == 'FPN'
    fpn = arfe_amd.build_neck(cfg.model.neck[0])
    assert fpn.num_outs == 5 and len(fpn.lateral_convs) == 4 and len(fpn.fpn_convs) == 4
    assert set(dict(fpn.named_parameters())) == {f'{g}.{i}.conv.{p}' for g in ('lateral_convs', 'fpn_convs')
                                                 for i in range(4) for p in ('weight', 'bias')}
    both = arfe_amd.build_neck(cfg.model.neck)
    assert [type(m).__name__ for m in both] == ['FPN', 'WFPNDualSpatial']
    roi_head = arfe_amd.build_head(cfg.model.roi_head)
    assert type(roi_head.bbox_head).__name__ == 'MultiRoIsBBoxHead'
    assert roi_head.bbox_roi_extractor.featmap_strides == [4, 8, 16, 32, 64]
    for rel in ('faster_rcnn/faster_rcnn_r50_fpn_1x_coco.py', 'cascade_rcnn/cascade_rcnn_r50_fpn_1x_coco.py',
                'libra_rcnn/libra_faster_rcnn_r50_fpn_1x_coco.py'):
        c = arfe_amd.Config.fromfile(os.path.join(REF, rel))
        assert 'model' in c and c.model.type in ('FasterRCNN', 'CascadeRCNN')
    libra = arfe_amd.Config.fromfile(os.path.join(REF, 'libra_rcnn', 'libra_faster_rcnn_r50_fpn_1x_coco.py'))
    wf = [n for n in libra.model.neck if n['type'] == 'WFPNDualSpatial'][0]
    assert arfe_amd.build_neck(wf).refine_level == wf.get('refine_level', 2)


def test_fp16_decorators_and_wrap_mirror_the_reference():
    """mmdet/core/fp16: decorators.py:8-123, utils.py:7-23, hooks.py:86-127 (host logic, CPU tensors)."""
    import torch.nn as nn
    from arfe_amd import auto_fp16, cast_tensor_type, force_fp32, wrap_fp16_model

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.fp16_enabled = False
            self.lin = nn.Linear(2, 2)
            self.bn = nn.BatchNorm1d(2)

        @auto_fp16(apply_to=('x', ))
        def f(self, x, y):
            return x, y

        @auto_fp16(out_fp32=True)
        def g(self, x, y=None):
            return [x, dict(y=y)]

        @force_fp32(apply_to=('feats', ), out_fp16=True)
        def h(self, feats, rois):
            return feats[0] * 2, rois

    m = M()
    a, b = torch.ones(2), torch.ones(2)
    x, y = m.f(a, b)
    assert x.dtype == torch.float32 and y.dtype == torch.float32            # pass-through until enabled
    wrap_fp16_model(m)
    assert m.fp16_enabled and m.lin.weight.dtype == torch.float16
    assert m.bn.weight.dtype == torch.float32                               # patch_norm_fp32
    x, y = m.f(a, b)
    assert x.dtype == torch.float16 and y.dtype == torch.float32            # apply_to
    out = m.g(a, y=b)
    assert out[0].dtype == torch.float32 and out[1]['y'].dtype == torch.float32     # cast in, out_fp32 back
    o, r = m.h((a.half(), b.half()), b)
    assert o.dtype == torch.float16 and r.dtype == torch.float16            # out_fp16 converts every output tensor
    # like the reference's cast_tensor_type, src_type is not consulted: every tensor in the structure is converted
    c = cast_tensor_type({'i': torch.ones(2, dtype=torch.int64), 's': 'txt', 'n': 3}, torch.float, torch.half)
    assert c['i'].dtype == torch.float16 and c['s'] == 'txt' and c['n'] == 3

    class NotAModule:
        @auto_fp16()
        def f(self, x):
            return x
    with pytest.raises(TypeError):
        NotAModule().f(a)


def test_fp16_mode_mismatch_and_cpu_refusal():
    import arfe_amd
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=4, num_levels=5))
    arfe_amd.wrap_fp16_model(neck)
    xs = [torch.randn(1, 4, h, w) for h, w in ((8, 8), (4, 4), (2, 2), (1, 1), (1, 1))]
    with pytest.raises(RuntimeError, match='should be the same'):
        neck(xs)
    with pytest.raises(arfe_amd.hip.ArfeHipError):          # half CPU tensors: no CPU fallback for the casts either
        neck([x.half() for x in xs])


def test_cascade_roi_head_builds_per_stage_modules():
    """cascade_roi_head.py:16-60: one extractor/head config broadcast to num_stages, or one per stage; state_dict keys
    `bbox_head.{stage}.*` as in a reference Cascade R-CNN checkpoint."""
    ext = dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0), out_channels=8,
               featmap_strides=[4, 8, 16, 32, 64])
    head = dict(type='MultiRoIsBBoxHead', in_channels=8, conv_out_channels=8, fc_out_channels=16, roi_feat_size=7,
                num_classes=80)
    rh = arfe_amd.build_head(dict(type='CascadeRoIHead', num_stages=3, stage_loss_weights=[1, 0.5, 0.25],
                                  bbox_roi_extractor=ext, bbox_head=[dict(head), dict(head), dict(head)]))
    assert rh.num_stages == 3 and len(rh.bbox_roi_extractor) == 3 and len(rh.bbox_head) == 3
    assert rh.bbox_head[0] is not rh.bbox_head[1]
    keys = set(rh.state_dict())
    assert 'bbox_head.2.wh_conv.conv.weight' in keys and 'bbox_head.0.shared_fcs.1.bias' in keys
    assert 'bbox_head.1.fc_reg.weight' in keys and len(keys) == 3 * 14
    rh.init_weights()
    with pytest.raises(AssertionError):
        arfe_amd.build_head(dict(type='CascadeRoIHead', num_stages=2, stage_loss_weights=[1, 1], bbox_roi_extractor=ext,
                                 bbox_head=[dict(head)]))
    with pytest.raises(RuntimeError, match='train_cfg'):      # built without train_cfg: no assigners / samplers
        rh.forward_train([], [], [], [], [])


def test_bench_inputs_equal_the_checkers():
    """bench.py draws its synthetic pyramid / RoIs from arfe_amd.synth (the product path never imports oracle/); the CPU
    baseline leg and the tests draw theirs from the oracle's copies: same seeds, same tensors."""
    import torch
    from arfe_amd import synth
    from oracle import arfe_oracle as O
    assert synth.PYRAMID_1333x800 == O.PYRAMID_1333x800 and synth.STRIDES == O.STRIDES
    for a, b in zip(synth.synth_pyramid(1, 4, seed=7), O.synth_pyramid(1, 4, seed=7)):
        assert torch.equal(a, b)
    assert torch.equal(synth.synth_rois(2, 33, seed=5), O.synth_rois(2, 33, seed=5))


def test_bench_refuses_to_measure_fewer_gpus_than_it_claims():
    """`bench.py --gpus N` must never print a line for a different rank count.  Without a launcher it starts
    its own ranks (and says so when the node has too few GPUs for RCCL); with a launcher whose WORLD_SIZE disagrees with
    --gpus it stops before anything is measured."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'ARFE_DIST_BACKEND')}
    # launcher present, wrong world size
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], env=dict(env, WORLD_SIZE='1'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1'], env=dict(env, WORLD_SIZE='2'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    # no launcher: self-launch; on a node with fewer GPUs than ranks (this container has none) it refuses, loudly
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 2 and 'needs 2 GPUs' in r.stderr and '"metric"' not in r.stdout


def test_bench_cpu_share_respects_affinity_and_override(monkeypatch):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    n = bench.host_cpu_share()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    monkeypatch.setenv('ARFE_CPU_THREADS', '3')
    assert bench.host_cpu_share() == 3
