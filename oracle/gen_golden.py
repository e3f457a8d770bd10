"""TEST INFRASTRUCTURE (build container only). Generates tests/golden/*.npz by running the
UNMODIFIED reference files (loaded by path, oracle/ref_loader.py) and the compiled reference
RoIAlign (oracle/_ref) on seeded inputs.  Fixtures hold data only: inputs, parameters,
expected outputs and gradients.  Run:  python oracle/gen_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arfe_oracle as O            # noqa: E402  (only for the shared synthetic-input helpers)
from oracle.ref_loader import load_reference   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
torch.set_num_threads(8)


def npd(d):
    out = {}
    for k, v in d.items():
        if v is None:      # a parameter/level no RoI touched: autograd leaves .grad None == zero gradient
            src = d[k[1:]] if k[1:] in d else d[k.replace('dneck.', 'neck.').replace('dhead.', 'head.').replace('dp.', 'p.')]
            v = torch.zeros_like(src)
        out[k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    return out


def randomise(module, gen, bias_scale=0.1):
    """init_weights() zeroes every bias; give biases small random values so they are exercised."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=gen) * bias_scale)


def arfpn_case(ns, name, channels, sizes, batch, refine_level, seed, scale_w=1.0):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    neck = ns.WFPNDualSpatial(channels, len(sizes), refine_level=refine_level)
    neck.init_weights()
    randomise(neck, gen)
    if scale_w != 1.0:   # larger NonLocal weights -> a peaky softmax, exercises the online-softmax rescale
        with torch.no_grad():
            for n, p in neck.named_parameters():
                if n.startswith('refine') and n.endswith('weight'):
                    p.mul_(scale_w)
    xs = [torch.randn(batch, channels, h, w, generator=gen, requires_grad=True) for h, w in sizes]
    cots = [torch.randn(batch, channels, h, w, generator=gen) for h, w in sizes]
    outs = neck(xs)
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    d = {'refine_level': refine_level}
    for i in range(len(sizes)):
        d[f'x{i}'] = xs[i]
        d[f'cot{i}'] = cots[i]
        d[f'out{i}'] = outs[i]
        d[f'dx{i}'] = xs[i].grad
    for n, p in neck.named_parameters():
        d['p.' + n] = p
        d['dp.' + n] = p.grad
    np.savez_compressed(os.path.join(OUT, name), **npd(d))
    print(name, 'ok')


def checksum(t):
    t = t.detach().double()
    flat = t.flatten()
    idx = (torch.arange(64, dtype=torch.int64) * (flat.numel() - 1)) // 63
    return dict(sum=t.sum().item(), abssum=t.abs().sum().item(), samples=flat[idx].numpy(), idx=idx.numpy())


def arfpn_full(ns, name, refine_level):
    """Full BASELINE size (B=1, 256 ch, 1333x800 pyramid): checksums + strided samples only."""
    p = O.arfpn_params(256, 5, seed=11)
    neck = ns.WFPNDualSpatial(256, 5, refine_level=refine_level)
    neck.load_state_dict(p)
    xs = [x.requires_grad_(True) for x in O.synth_pyramid(1, 256, seed=1234)]
    cots = O.synth_pyramid(1, 256, seed=4321)
    outs = neck(xs)
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    d = {}
    for i in range(5):
        for k, v in checksum(outs[i]).items():
            d[f'out{i}.{k}'] = v
        for k, v in checksum(xs[i].grad).items():
            d[f'dx{i}.{k}'] = v
    for n, q in neck.named_parameters():
        for k, v in checksum(q.grad).items():
            d[f'dp.{n}.{k}'] = v
    np.savez_compressed(os.path.join(OUT, name), **d)
    print(name, 'ok')


def rois_cases(ns):
    gen = torch.Generator().manual_seed(5)
    r = O.synth_rois(2, 300, seed=99)
    special = torch.tensor([
        [0, 0., 0., 0., 0.],            # degenerate point at the origin
        [0, 0., 0., 15., 7.],           # at the border (x1/y1 clamp to 0.1)
        [1, 1343., 799., 1343., 799.],  # degenerate at the far corner
        [1, 10., 20., 10., 300.],       # zero width
        [0, 5., 5., 1300., 30.],        # extreme aspect
        [0, 100.5, 200.25, 163.75, 420.125],
    ])
    rois = torch.cat([special, r], 0)
    h, w = ns.get_adaptive_scale_rois(rois, 1)
    np.savez_compressed(os.path.join(OUT, 'rois_make.npz'), rois=rois.numpy(), h_rois=h.numpy(), w_rois=w.numpy())

    # level mapping: dense sweep of box sizes whose sqrt-area sits within +-64 ulp of every step
    class E:
        finest_scale = 56
    sides = []
    for thr in (112.0, 224.0, 448.0, 896.0):
        v = np.float32(thr)
        lo = v
        for _ in range(200):
            lo = np.nextafter(lo, np.float32(0))
        cur = lo
        for _ in range(400):
            sides.append(cur)
            cur = np.nextafter(cur, np.float32(1e9))
    sides = np.array(sides, dtype=np.float32)
    sq = np.stack([np.zeros_like(sides), np.zeros_like(sides), np.zeros_like(sides), sides, sides], 1)
    # non-square boxes with the same nominal area, random aspect
    asp = np.exp(np.random.RandomState(3).uniform(-1.5, 1.5, sides.shape)).astype(np.float32)
    ns_w = (sides * asp).astype(np.float32)
    ns_h = (sides / asp).astype(np.float32)
    rect = np.stack([np.zeros_like(sides), np.full_like(sides, 7.25), np.full_like(sides, 3.5),
                     (7.25 + ns_w).astype(np.float32), (3.5 + ns_h).astype(np.float32)], 1)
    allr = torch.from_numpy(np.concatenate([sq, rect, rois.numpy()], 0))
    lv = ns.SingleRoIExtractor.map_roi_levels(E, allr, 5)
    lv4 = ns.SingleRoIExtractor.map_roi_levels(E, allr, 4)
    np.savez_compressed(os.path.join(OUT, 'roi_levels.npz'), rois=allr.numpy(), levels5=lv.numpy(), levels4=lv4.numpy())
    print('rois ok', np.bincount(lv.numpy()))


def roialign_case(ns):
    gen = torch.Generator().manual_seed(21)
    C = 6
    sizes = ((40, 64), (20, 32), (10, 16), (5, 8), (3, 4))   # 160x256 image, strides 4..64
    feats = [torch.randn(2, C, h, w, generator=gen, requires_grad=True) for h, w in sizes]
    n = 40
    x1 = torch.rand(n, generator=gen) * 200
    y1 = torch.rand(n, generator=gen) * 120
    w = torch.exp(torch.rand(n, generator=gen) * 4.0 + 1.0)
    h = torch.exp(torch.rand(n, generator=gen) * 4.0 + 1.0)
    rois = torch.stack([(torch.arange(n) % 2).float(), x1, y1, x1 + w, y1 + h], 1)
    rois = torch.cat([rois, torch.tensor([[0, -20., -30., 40., 50.],     # partly outside (negative)
                                          [1, 200., 100., 400., 300.],  # partly outside (beyond)
                                          [0, 17., 23., 17., 23.],      # zero size -> grid 0 -> zeros
                                          [1, 0.1, 0.1, 255., 159.]])], 0)
    ext = ns.SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=0), C, [4, 8, 16, 32, 64])
    h_r, w_r = ns.get_adaptive_scale_rois(rois, 1)
    ori = ext(feats, rois)
    fw = ext(feats, w_r)
    fh = ext(feats, h_r)
    cat = torch.cat([ori, fw, fh], dim=1)
    cot = torch.randn(cat.shape, generator=gen)
    (cat * cot).sum().backward()
    d = dict(rois=rois, cat=cat, cot=cot)
    for i in range(5):
        d[f'feat{i}'] = feats[i]
        d[f'dfeat{i}'] = feats[i].grad if feats[i].grad is not None else torch.zeros_like(feats[i])
    # single-level, fixed sample_num variant straight from the compiled reference
    f0 = feats[1].detach()
    o2 = ns.roi_align_ext.forward_v2(f0, rois, 1.0 / 8, 7, 7, 2, True)
    d['single_lvl1_sn2'] = o2
    np.savez_compressed(os.path.join(OUT, 'roialign.npz'), **npd(d))
    print('roialign ok')


def head_case(ns, name, C, fc, R, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    head = ns.MultiRoIsBBoxHead(in_channels=C, conv_out_channels=C, fc_out_channels=fc, roi_feat_size=7,
                                num_classes=80)
    head.init_weights()
    randomise(head, gen)
    x = torch.randn(R, 3 * C, 7, 7, generator=gen, requires_grad=True)
    cls, reg = head(x)
    c1 = torch.randn(cls.shape, generator=gen)
    c2 = torch.randn(reg.shape, generator=gen)
    ((cls * c1).sum() + (reg * c2).sum()).backward()
    d = dict(x=x, cls=cls, reg=reg, cot_cls=c1, cot_reg=c2, dx=x.grad)
    for n, p in head.named_parameters():
        d['p.' + n] = p
        d['dp.' + n] = p.grad
    np.savez_compressed(os.path.join(OUT, name), **npd(d))
    print(name, 'ok')


def e2e_case(ns):
    """neck -> [ori | w | h] RoIAlign concat (the intended, disabled glue of standard_roi_head.py:138-155,
    restated in 6 lines here) -> MultiRoIsBBoxHead; grads w.r.t. the pyramid and every parameter."""
    gen = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    C = 8
    sizes = ((40, 64), (20, 32), (10, 16), (5, 8), (3, 4))
    neck = ns.WFPNDualSpatial(C, 5)
    neck.init_weights()
    randomise(neck, gen)
    head = ns.MultiRoIsBBoxHead(in_channels=C, conv_out_channels=C, fc_out_channels=24, roi_feat_size=7,
                                num_classes=80)
    head.init_weights()
    randomise(head, gen)
    ext = ns.SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=0), C, [4, 8, 16, 32, 64])
    xs = [torch.randn(2, C, h, w, generator=gen, requires_grad=True) for h, w in sizes]
    n = 24
    x1 = torch.rand(n, generator=gen) * 180
    y1 = torch.rand(n, generator=gen) * 100
    w = torch.exp(torch.rand(n, generator=gen) * 3.5 + 1.5)
    h = torch.exp(torch.rand(n, generator=gen) * 3.5 + 1.5)
    rois = torch.stack([(torch.arange(n) // 12).float(), x1, y1, x1 + w, y1 + h], 1)
    feats = neck(xs)
    lh_rois, lw_rois = ns.get_adaptive_scale_rois(rois, 1)
    bbox_feats = torch.cat([ext(feats, rois), ext(feats, lw_rois), ext(feats, lh_rois)], dim=1)
    cls, reg = head(bbox_feats)
    c1 = torch.randn(cls.shape, generator=gen)
    c2 = torch.randn(reg.shape, generator=gen)
    ((cls * c1).sum() + (reg * c2).sum()).backward()
    d = dict(rois=rois, cls=cls, reg=reg, cot_cls=c1, cot_reg=c2, bbox_feats=bbox_feats)
    for i in range(5):
        d[f'x{i}'] = xs[i]
        d[f'dx{i}'] = xs[i].grad
        d[f'feat{i}'] = feats[i]
    for n_, p in neck.named_parameters():
        d['neck.' + n_] = p
        d['dneck.' + n_] = p.grad
    for n_, p in head.named_parameters():
        d['head.' + n_] = p
        d['dhead.' + n_] = p.grad
    np.savez_compressed(os.path.join(OUT, 'e2e_small.npz'), **npd(d))
    print('e2e ok')


def fpn_case(ns, name, in_channels, out_channels, sizes, batch, num_outs, seed):
    """Base FPN (necks/fpn.py) forward + backward on seeded inputs; sizes may be ragged (odd halves)."""
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    neck = ns.FPN(list(in_channels), out_channels, num_outs)
    neck.init_weights()
    randomise(neck, gen)
    xs = [torch.randn(batch, c, h, w, generator=gen, requires_grad=True) for c, (h, w) in zip(in_channels, sizes)]
    outs = neck(xs)
    cots = [torch.randn(o.shape, generator=gen) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    d = {'num_outs': num_outs}
    for i in range(len(xs)):
        d[f'x{i}'] = xs[i]
        d[f'dx{i}'] = xs[i].grad
    for i in range(len(outs)):
        d[f'cot{i}'] = cots[i]
        d[f'out{i}'] = outs[i]
    for n, p in neck.named_parameters():
        d['p.' + n] = p
        d['dp.' + n] = p.grad
    np.savez_compressed(os.path.join(OUT, name), **npd(d))
    print('wrote', name)


def wrap_fp16(model):
    """wrap_fp16_model of mmdet/core/fp16/hooks.py:86-105 restated (that file imports mmcv.runner and cannot be loaded);
    there are no norm layers on the path, so patch_norm_fp32 has nothing to patch."""
    model.half()
    for m in model.modules():
        if hasattr(m, 'fp16_enabled'):
            m.fp16_enabled = True


def fp16_case(ns):
    """The path in the reference's fp16 mode, run by the reference itself in half on the CPU: FPN (@auto_fp16, fpn.py:166)
    -> WFPNDualSpatial -> SingleRoIExtractor (@force_fp32(feats), out_fp16, single_level.py:109) x 3 RoI sets ->
    MultiRoIsBBoxHead, forward and backward.  Parameters are stored as the half values the modules hold."""
    gen = torch.Generator().manual_seed(91)
    torch.manual_seed(91)
    C = 8
    in_ch = (4, 6, 8, 10)
    sizes = ((40, 64), (20, 32), (10, 16), (5, 8))
    fpn = ns.FPN(list(in_ch), C, 5)
    fpn.init_weights()
    randomise(fpn, gen)
    neck = ns.WFPNDualSpatial(C, 5)
    neck.init_weights()
    randomise(neck, gen)
    with torch.no_grad():                      # conv_out is zero-initialised: give the NonLocal branch a visible effect
        neck.refine.conv_out.conv.weight.copy_(torch.randn(C, C, 1, 1, generator=gen) * 0.3)
    head = ns.MultiRoIsBBoxHead(in_channels=C, conv_out_channels=C, fc_out_channels=24, roi_feat_size=7, num_classes=80)
    head.init_weights()
    randomise(head, gen)
    ext = ns.SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=0), C, [4, 8, 16, 32, 64])
    for m in (fpn, neck, head, ext):
        wrap_fp16(m)
    xs = [torch.randn(2, c, h, w, generator=gen, requires_grad=True) for c, (h, w) in zip(in_ch, sizes)]
    n = 24
    x1 = torch.rand(n, generator=gen) * 180
    y1 = torch.rand(n, generator=gen) * 100
    w = torch.exp(torch.rand(n, generator=gen) * 3.5 + 1.5)
    h = torch.exp(torch.rand(n, generator=gen) * 3.5 + 1.5)
    rois = torch.stack([(torch.arange(n) // 12).float(), x1, y1, x1 + w, y1 + h], 1)
    pyr = fpn(xs)
    feats = neck(pyr)
    lh_rois, lw_rois = ns.get_adaptive_scale_rois(rois, 1)
    bbox_feats = torch.cat([ext(feats, rois), ext(feats, lw_rois), ext(feats, lh_rois)], dim=1)
    cls, reg = head(bbox_feats)
    assert all(t.dtype == torch.float16 for t in (*pyr, *feats, bbox_feats, cls, reg))
    c1 = torch.randn(cls.shape, generator=gen).half()
    c2 = torch.randn(reg.shape, generator=gen).half()
    ((cls * c1).sum() + (reg * c2).sum()).backward()
    d = dict(rois=rois, cls=cls, reg=reg, cot_cls=c1, cot_reg=c2, bbox_feats=bbox_feats)
    for i in range(4):
        d[f'x{i}'] = xs[i]
        d[f'dx{i}'] = xs[i].grad
    for i in range(5):
        d[f'pyr{i}'] = pyr[i]
        d[f'feat{i}'] = feats[i]
    for tag, mod in (('fpn.', fpn), ('neck.', neck), ('head.', head)):
        for n_, p in mod.named_parameters():
            d[tag + n_] = p
            d['d' + tag + n_] = p.grad if p.grad is not None else torch.zeros_like(p)
    np.savez_compressed(os.path.join(OUT, 'fp16_e2e_small.npz'), **npd(d))
    print('fp16 e2e ok')


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = load_reference()
    assert ns.RoIAlign is not None, "run oracle/build_ref.py first"
    tiny = ((16, 24), (8, 12), (4, 6), (2, 3), (1, 2))
    ragged = ((40, 67), (20, 34), (10, 17), (5, 9), (3, 5))
    arfpn_case(ns, 'arfpn_tiny_r2.npz', 8, tiny, 2, 2, seed=1)
    arfpn_case(ns, 'arfpn_tiny_r3.npz', 8, tiny, 2, 3, seed=2)
    arfpn_case(ns, 'arfpn_ragged_r2.npz', 8, ragged, 1, 2, seed=3)
    arfpn_case(ns, 'arfpn_ragged_r3.npz', 8, ragged, 2, 3, seed=4)
    arfpn_case(ns, 'arfpn_peaky_r2.npz', 16, ragged, 1, 2, seed=5, scale_w=6.0)
    arfpn_full(ns, 'arfpn_full_r2_checksum.npz', 2)
    rois_cases(ns)
    roialign_case(ns)
    head_case(ns, 'rff_head_r8.npz', 16, 32, 8, seed=6)
    head_case(ns, 'rff_head_r40.npz', 8, 24, 40, seed=7)
    e2e_case(ns)
    fpn_cases(ns)
    fp16_case(ns)
    harness_case()


def _boxes(n, w, h, gen, lo=4.0, hi=120.0):
    wh = torch.exp(torch.rand(n, 2, generator=gen) * (np.log(hi) - np.log(lo)) + np.log(lo))
    xy = torch.rand(n, 2, generator=gen) * (torch.tensor([w, h], dtype=torch.float32) - wh).clamp(min=1.0)
    return torch.cat([xy, xy + wh], 1)


def head_train_case():
    """Fixtures for the training side of BBoxHead and the RandomSampler, from the reference's own classes: two images, an
    R-CNN assignment each, sampling under a fixed torch seed (both implementations draw torch.randperm the same way),
    targets, losses (class-specific and class-agnostic regression), per-class regression and the Cascade refinement."""
    import types as _t
    from oracle.ref_loader import load_reference_head_train
    R = load_reference_head_train()
    g = torch.Generator().manual_seed(77)
    nc = 6
    d = {}
    assigner = R.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
    sampler = R.RandomSampler(num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    results, rois = [], []
    for i in range(2):
        gts = _boxes(7 + i, 96, 64, g)
        labels = torch.randint(0, nc, (gts.shape[0], ), generator=g)
        jit = torch.randn(40, 4, generator=g) * 2.0
        props = torch.cat([_boxes(120, 96, 64, g), gts.repeat(6, 1)[:40] + jit])
        d[f'ht_gts{i}'], d[f'ht_labels{i}'], d[f'ht_props{i}'] = gts, labels, props
        a = assigner.assign(props, gts, None, labels)
        torch.manual_seed(1000 + i)
        r = sampler.sample(a, props, gts, labels)
        d[f'ht_pos_inds{i}'], d[f'ht_neg_inds{i}'] = r.pos_inds, r.neg_inds
        d[f'ht_pos_is_gt{i}'], d[f'ht_pos_assigned{i}'] = r.pos_is_gt, r.pos_assigned_gt_inds
        d[f'ht_pos_bboxes{i}'], d[f'ht_neg_bboxes{i}'] = r.pos_bboxes, r.neg_bboxes
        results.append(r)
        rois.append(torch.cat([torch.full((r.bboxes.shape[0], 1), float(i)), r.bboxes], 1))
    rois = torch.cat(rois)
    d['ht_rois'] = rois
    cfg = _t.SimpleNamespace(pos_weight=-1)
    coder = dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])
    for tag, agnostic in (('cs', False), ('ag', True)):
        head = R.BBoxHead(with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=2, num_classes=nc,
                          bbox_coder=coder, reg_class_agnostic=agnostic,
                          loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                          loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
        t = head.get_targets(results, None, None, cfg)
        if not agnostic:
            d['ht_t_labels'], d['ht_t_label_weights'], d['ht_t_bbox_targets'], d['ht_t_bbox_weights'] = t
        n = rois.shape[0]
        cls = torch.randn(n, nc + 1, generator=g)
        reg = torch.randn(n, 4 if agnostic else 4 * nc, generator=g) * 0.3
        d[f'ht_{tag}_cls'], d[f'ht_{tag}_reg'] = cls, reg
        losses = head.loss(cls, reg, rois, *t)
        d[f'ht_{tag}_loss_cls'], d[f'ht_{tag}_loss_bbox'], d[f'ht_{tag}_acc'] = losses['loss_cls'], losses['loss_bbox'], losses['acc']
        metas = [dict(img_shape=(64, 96, 3)), dict(img_shape=(64, 96, 3))]
        # Cascade: the regressor's class is the predicted one where the target is background (cascade_roi_head.py:262-266)
        roi_labels = torch.where(t[0] == nc, cls[:, :-1].argmax(1), t[0])
        d[f'ht_{tag}_roi_labels'] = roi_labels
        one = head.regress_by_class(rois[:9], roi_labels[:9], reg[:9], metas[0])
        d[f'ht_{tag}_regress'] = one
        refined = head.refine_bboxes(rois, roi_labels, reg, [r.pos_is_gt for r in results], metas)
        d[f'ht_{tag}_refined0'], d[f'ht_{tag}_refined1'] = refined
        # test-time decode without the NMS (cfg=None; the reference's multiclass_nms sits on its compiled NMS extension): plain,
        # rescaled by a float and by a per-coordinate factor, and with a list of stage scores (Cascade's ensemble)
        gb, gs = head.get_bboxes(rois, cls, reg, (64, 96, 3), 1.0)
        d[f'ht_{tag}_gb_boxes'], d[f'ht_{tag}_gb_scores'] = gb, gs
        d[f'ht_{tag}_gb_boxes_f'] = head.get_bboxes(rois, cls, reg, (64, 96, 3), 2.0, rescale=True)[0]
        d[f'ht_{tag}_gb_boxes_v'] = head.get_bboxes(rois, cls, reg, (64, 96, 3), np.array([1.5, 1.25, 1.5, 1.25], dtype=np.float32),
                                                    rescale=True)[0]
        d[f'ht_{tag}_gb_scores_list'] = head.get_bboxes(rois, [cls, cls * 0.5, -cls], reg, (64, 96, 3), 1.0)[1]
        d[f'ht_{tag}_gb_noreg'] = head.get_bboxes(rois, cls, None, (40, 50, 3), 1.0)[0]
    return d


def harness_case():
    """Fixtures of the detector harness from the reference's own box / anchor / loss code."""
    from oracle.ref_loader import load_reference_harness
    hs = load_reference_harness()
    g = torch.Generator().manual_seed(21)
    d = {}
    # anchors: the RPN generator of the AR configs on a 60 x 90 image padded to 64 x 96
    gen = hs.AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    d['anchor_sizes'] = torch.tensor(sizes)
    anchors = gen.grid_anchors(sizes, device='cpu')
    flags = gen.valid_flags(sizes, (60, 90, 3), device='cpu')
    d['anchors'] = torch.cat(anchors)
    d['anchor_valid'] = torch.cat(flags)
    d['anchor_inside'] = hs.anchor_inside_flags(torch.cat(anchors), torch.cat(flags), (60, 90), allowed_border=0)
    gen2 = hs.AnchorGenerator(strides=[8, 16], ratios=[1.0], octave_base_scale=4, scales_per_octave=3, scale_major=False)
    d['anchors_octave'] = torch.cat(gen2.grid_anchors([(3, 4), (2, 2)], device='cpu'))
    # coder
    props, gts = _boxes(300, 96, 64, g), _boxes(300, 96, 64, g)
    d['coder_props'], d['coder_gts'] = props, gts
    means, stds = (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)
    d['deltas'] = hs.bbox2delta(props, gts, means, stds)
    raw = torch.randn(300, 12, generator=g) * 0.7
    d['raw_deltas'] = raw
    d['decoded'] = hs.delta2bbox(props, raw, means, stds, max_shape=(64, 96))
    d['decoded_noclip'] = hs.delta2bbox(props, raw[:, :4] * 8.0, (0.1, 0., 0., 0.), (1., 1., 1., 1.))
    # overlaps
    b1, b2 = _boxes(40, 96, 64, g), _boxes(25, 96, 64, g)
    d['ov_b1'], d['ov_b2'] = b1, b2
    d['iou'] = hs.bbox_overlaps(b1, b2)
    d['iof'] = hs.bbox_overlaps(b1, b2, mode='iof')
    d['iou_aligned'] = hs.bbox_overlaps(b1[:25], b2, is_aligned=True)
    # assigner: the RPN and the R-CNN settings of the AR configs (+ duplicates of gt boxes among the candidates: ties)
    cand = torch.cat([_boxes(500, 96, 64, g), b2[:5], b2[:5] + 0.5])
    labels = torch.randint(0, 80, (25, ), generator=g)
    d['as_boxes'], d['as_gts'], d['as_labels'] = cand, b2, labels
    rpn = hs.MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1)
    r = rpn.assign(cand, b2, None, None)
    d['as_rpn_inds'], d['as_rpn_max'] = r.gt_inds, r.max_overlaps
    rcnn = hs.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
    r = rcnn.assign(cand, b2, None, labels)
    d['as_rcnn_inds'], d['as_rcnn_labels'] = r.gt_inds, r.labels
    one = hs.MaxIoUAssigner(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.4), min_pos_iou=0.2, gt_max_assign_all=False,
                            ignore_iof_thr=0.5)
    r = one.assign(cand, b2[:10], b2[10:14], labels[:10])
    d['as_one_inds'], d['as_one_labels'] = r.gt_inds, r.labels
    # losses: N = 200 >= 81 samples (the fork's softmax criterion indexes a length-N vector with class indices)
    pred, lab = torch.randn(200, 81, generator=g), torch.randint(0, 81, (200, ), generator=g)
    w = (torch.rand(200, generator=g) > 0.2).float()
    d['ce_pred'], d['ce_label'], d['ce_weight'] = pred, lab, w
    ce = hs.CrossEntropyLoss(use_sigmoid=False, loss_weight=1.0)
    d['ce_plain'] = ce(pred, lab)
    d['ce_weighted'] = ce(pred, lab, w, avg_factor=float(w.sum()))
    d['ce_none'] = ce(pred, lab, w, reduction_override='none')
    d['acc'] = hs.accuracy(pred, lab)
    bp, bl = torch.randn(500, 1, generator=g), torch.randint(0, 2, (500, ), generator=g)
    bw = (torch.rand(500, generator=g) > 0.3).float()
    d['bce_pred'], d['bce_label'], d['bce_weight'] = bp, bl, bw
    d['bce'] = hs.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)(bp, bl, bw, avg_factor=256.0)
    rp, rt, rw = torch.randn(64, 4, generator=g), torch.randn(64, 4, generator=g), (torch.rand(64, 4, generator=g) > 0.5).float()
    d['reg_pred'], d['reg_target'], d['reg_weight'] = rp, rt, rw
    d['l1'] = hs.L1Loss(loss_weight=1.0)(rp, rt, rw, avg_factor=512.0)
    d['smooth_l1'] = hs.SmoothL1Loss(beta=1.0 / 9.0, loss_weight=1.0)(rp, rt, rw, avg_factor=512.0)
    d.update(head_train_case())
    np.savez_compressed(os.path.join(OUT, 'harness.npz'), **npd(d))
    print('harness.npz', {k: tuple(v.shape) for k, v in d.items() if hasattr(v, 'shape')})


def fpn_cases(ns):
    fpn_case(ns, 'fpn_tiny.npz', (4, 6, 8, 10), 8, ((16, 24), (8, 12), (4, 6), (2, 3)), 2, 5, seed=11)
    fpn_case(ns, 'fpn_ragged.npz', (8, 16, 24, 32), 16, ((41, 67), (21, 34), (11, 17), (6, 9)), 1, 5, seed=12)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'fpn':     # only the base-FPN fixtures (keeps the others byte-identical)
        os.makedirs(OUT, exist_ok=True)
        fpn_cases(load_reference())
    elif len(sys.argv) > 1 and sys.argv[1] == 'harness':  # only the detector-harness fixture
        os.makedirs(OUT, exist_ok=True)
        harness_case()
    elif len(sys.argv) > 1 and sys.argv[1] == 'fp16':  # only the fp16-mode fixture
        os.makedirs(OUT, exist_ok=True)
        fp16_case(load_reference())
    else:
        main()
