"""Detector harness around the hot path (arfe_amd/detector): host logic in plain PyTorch.

CPU tests: every box / anchor / loss helper against tests/golden/harness.npz, generated from the REFERENCE's own files
(oracle/gen_golden.py harness; loaded by path, oracle/ref_loader.py:load_reference_harness), and against the known
answers in the reference's docstrings; the training side of BBoxHead (targets, losses, per-class regression, Cascade
refinement) and the RandomSampler's bookkeeping against fixtures from the reference's own classes run under fixed seeds
(oracle/gen_golden.py head_train_case); NMS against a plain greedy loop (the reference's NMS is a compiled extension:
unpinned beyond the greedy definition of ops/nms/src/cpu/nms_cpu.cpp:5-68); the detector builds from the reference's
config files.  GPU tests: Faster / Cascade R-CNN forward_train + backward end to end with the HIP hot path inside."""
import os

import pytest
import torch

import arfe_amd
from arfe_amd import detector as D
from conftest import load_golden, relerr

G = load_golden('harness.npz')


def test_anchor_generator_matches_reference():
    gen = D.AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    sizes = [tuple(int(v) for v in s) for s in G['anchor_sizes']]
    anchors = torch.cat(gen.grid_anchors(sizes, device='cpu'))
    assert torch.equal(anchors, G['anchors'])
    valid = torch.cat(gen.valid_flags(sizes, (60, 90, 3), device='cpu'))
    assert torch.equal(valid, G['anchor_valid'].bool())
    assert torch.equal(D.anchor_inside_flags(anchors, valid, (60, 90), allowed_border=0), G['anchor_inside'].bool())
    assert D.anchor_inside_flags(anchors, valid, (60, 90), allowed_border=-1) is valid
    assert gen.num_base_anchors == [3] * 5 and gen.num_levels == 5
    octave = D.AnchorGenerator(strides=[8, 16], ratios=[1.0], octave_base_scale=4, scales_per_octave=3, scale_major=False)
    assert torch.equal(torch.cat(octave.grid_anchors([(3, 4), (2, 2)], device='cpu')), G['anchors_octave'])
    # the reference's docstring example (anchor_generator.py:41-48)
    a = D.AnchorGenerator([16], [1.], [1.], [9]).grid_anchors([(2, 2)], device='cpu')[0]
    assert a.tolist() == [[-4.5, -4.5, 4.5, 4.5], [11.5, -4.5, 20.5, 4.5], [-4.5, 11.5, 4.5, 20.5], [11.5, 11.5, 20.5, 20.5]]
    with pytest.raises(AssertionError):
        D.AnchorGenerator([8], [1.0], scales=[8], octave_base_scale=4, scales_per_octave=3)


def test_delta_coder_matches_reference():
    means, stds = (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)
    assert torch.equal(D.bbox2delta(G['coder_props'], G['coder_gts'], means, stds), G['deltas'])
    assert torch.equal(D.delta2bbox(G['coder_props'], G['raw_deltas'], means, stds, max_shape=(64, 96)), G['decoded'])
    assert torch.equal(D.delta2bbox(G['coder_props'], G['raw_deltas'][:, :4] * 8.0, (0.1, 0., 0., 0.), (1., 1., 1., 1.)),
                       G['decoded_noclip'])
    coder = arfe_amd.build_bbox_coder(dict(type='DeltaXYWHBBoxCoder', target_means=means, target_stds=stds))
    assert torch.equal(coder.encode(G['coder_props'], G['coder_gts']), G['deltas'])
    back = coder.decode(G['coder_props'], G['deltas'])
    assert relerr(back, G['coder_gts']) <= 1e-5                       # encode -> decode round trip
    # the reference's docstring example (delta_xywh_bbox_coder.py:124-137)
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    want = torch.tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                         [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    assert (D.delta2bbox(rois, deltas, max_shape=(32, 32)) - want).abs().max() < 5e-5


def test_bbox_overlaps_matches_reference():
    assert torch.equal(D.bbox_overlaps(G['ov_b1'], G['ov_b2']), G['iou'])
    assert torch.equal(D.bbox_overlaps(G['ov_b1'], G['ov_b2'], mode='iof'), G['iof'])
    assert torch.equal(D.bbox_overlaps(G['ov_b1'][:25], G['ov_b2'], is_aligned=True), G['iou_aligned'])
    empty, one = torch.FloatTensor([]), torch.FloatTensor([[0, 0, 10, 9]])      # iou2d_calculator.py:75-81
    assert tuple(D.bbox_overlaps(empty, one).shape) == (0, 1)
    assert tuple(D.bbox_overlaps(one, empty).shape) == (1, 0)
    assert tuple(D.bbox_overlaps(empty, empty).shape) == (0, 0)
    b1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    assert torch.equal(D.bbox_overlaps(b1, b2), torch.tensor([[0.5, 0., 0.], [0., 0., 1.], [0., 0., 0.]]))


def test_max_iou_assigner_matches_reference():
    boxes, gts, labels = G['as_boxes'], G['as_gts'], G['as_labels']
    rpn = arfe_amd.build_assigner(dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                                       match_low_quality=True, ignore_iof_thr=-1))
    r = rpn.assign(boxes, gts, None, None)
    assert torch.equal(r.gt_inds, G['as_rpn_inds']) and torch.equal(r.max_overlaps, G['as_rpn_max']) and r.labels is None
    rcnn = D.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
    r = rcnn.assign(boxes, gts, None, labels)
    assert torch.equal(r.gt_inds, G['as_rcnn_inds']) and torch.equal(r.labels, G['as_rcnn_labels'])
    one = D.MaxIoUAssigner(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.4), min_pos_iou=0.2, gt_max_assign_all=False, ignore_iof_thr=0.5)
    r = one.assign(boxes, gts[:10], gts[10:14], labels[:10])
    assert torch.equal(r.gt_inds, G['as_one_inds']) and torch.equal(r.labels, G['as_one_labels'])
    # docstring example (max_iou_assigner.py:84-90) and the empty cases (:140-155)
    r = D.MaxIoUAssigner(0.5, 0.5).assign(torch.Tensor([[0, 0, 10, 10], [10, 10, 20, 20]]), torch.Tensor([[0, 0, 10, 9]]))
    assert r.gt_inds.tolist() == [1, 0]
    r = rcnn.assign(boxes, gts[:0], None, labels[:0])
    assert (r.gt_inds == 0).all() and (r.labels == -1).all() and r.num_gts == 0
    r = rcnn.assign(boxes[:0], gts, None, labels)
    assert r.gt_inds.numel() == 0


def test_losses_match_reference():
    ce = arfe_amd.build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    pred, lab, w = G['ce_pred'], G['ce_label'], G['ce_weight']
    assert abs(ce(pred, lab).item() - G['ce_plain']) <= 1e-6
    # the fork's positional-argument quirk (detector/losses.py header): weights enter as 1 - softmax(w)[argmax pred]
    assert abs(ce(pred, lab, w, avg_factor=float(w.sum())).item() - G['ce_weighted']) <= 1e-5
    assert torch.allclose(ce(pred, lab, w, reduction_override='none'), G['ce_none'], atol=1e-6)
    upstream = D.CrossEntropyLoss(fork_weight_quirk=False)
    want = (torch.nn.functional.cross_entropy(pred, lab, reduction='none') * w).sum() / w.sum()
    assert abs(upstream(pred, lab, w, avg_factor=float(w.sum())).item() - want.item()) <= 1e-6
    assert torch.equal(D.accuracy(pred, lab), G['acc'])
    bce = D.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)
    assert abs(bce(G['bce_pred'], G['bce_label'], G['bce_weight'], avg_factor=256.0).item() - G['bce']) <= 1e-6
    l1 = arfe_amd.build_loss(dict(type='L1Loss', loss_weight=1.0))
    sl1 = arfe_amd.build_loss(dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0))
    assert abs(l1(G['reg_pred'], G['reg_target'], G['reg_weight'], avg_factor=512.0).item() - G['l1']) <= 1e-6
    assert abs(sl1(G['reg_pred'], G['reg_target'], G['reg_weight'], avg_factor=512.0).item() - G['smooth_l1']) <= 1e-6
    # the documented example of weighted_loss (losses/utils.py:72-84)
    p, t, ww = torch.Tensor([0, 2, 3]), torch.Tensor([1, 1, 1]), torch.Tensor([1, 0, 1])
    from arfe_amd.detector.losses import l1_loss
    assert abs(l1_loss(p, t).item() - 4 / 3) < 1e-6 and l1_loss(p, t, ww).item() == 1.0
    assert l1_loss(p, t, reduction='none').tolist() == [1., 1., 2.] and l1_loss(p, t, ww, avg_factor=2).item() == 1.5
    with pytest.raises(ValueError):
        l1_loss(p, t, ww, reduction='sum', avg_factor=2)


def _greedy_nms(dets, thr):
    order = sorted(range(len(dets)), key=lambda i: -float(dets[i, 4]))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            if float(D.bbox_overlaps(dets[i:i + 1, :4], dets[j:j + 1, :4])) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def test_nms_is_the_greedy_algorithm():
    g = torch.Generator().manual_seed(3)
    for n, thr in ((1, 0.5), (60, 0.3), (300, 0.7)):
        xy = torch.rand(n, 2, generator=g) * 60
        wh = torch.rand(n, 2, generator=g) * 30 + 4
        dets = torch.cat([xy, xy + wh, torch.rand(n, 1, generator=g)], 1)
        kept, inds = D.nms(dets, thr)
        assert inds.tolist() == _greedy_nms(dets, thr) and torch.equal(kept, dets[inds])
    # a chain a > b > c where a suppresses b and b would suppress c: greedy keeps a and c
    chain = torch.tensor([[0., 0., 10., 10., 0.9], [4., 0., 14., 10., 0.8], [8., 0., 18., 10., 0.7]])
    assert D.nms(chain, 0.4)[1].tolist() == [0, 2]
    assert D.nms(torch.zeros(0, 5), 0.5)[1].numel() == 0
    # batched: identical boxes in different groups do not suppress each other
    boxes = torch.tensor([[0., 0., 10., 10.]] * 4)
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6])
    dets, keep = D.batched_nms(boxes, scores, torch.tensor([0, 0, 1, 1]), dict(type='nms', iou_thr=0.5))
    assert keep.tolist() == [0, 2] and torch.equal(dets[:, :4], boxes[:2]) and torch.equal(dets[:, 4], scores[[0, 2]])


def test_batched_nms_per_group_equals_the_single_shifted_pass():
    """nms_wrapper.py:119-160 runs one greedy pass over all groups after shifting each into its own coordinate range; the
    harness runs one pass per group on the same shifted coordinates - same kept set, same order (score ties included)."""
    g = torch.Generator().manual_seed(8)
    n = 700
    xy = torch.rand(n, 2, generator=g) * 80
    wh = torch.rand(n, 2, generator=g) * 40 + 4
    boxes = torch.cat([xy, xy + wh], 1)
    scores = (torch.rand(n, generator=g) * 20).round() / 20            # many exact ties
    groups = torch.randint(0, 5, (n, ), generator=g)
    dets, keep = D.batched_nms(boxes, scores, groups, dict(type='nms', iou_thr=0.5))
    shifted = boxes + (groups.to(boxes) * (boxes.max() + 1))[:, None]
    ref = _greedy_nms(torch.cat([shifted, scores[:, None]], 1), 0.5)
    assert keep.tolist() == ref
    assert torch.equal(dets[:, :4], boxes[keep]) and torch.equal(dets[:, 4], scores[keep])


def test_random_sampler_invariants():
    torch.manual_seed(0)
    boxes, gts, labels = G['as_boxes'], G['as_gts'], G['as_labels']
    assigner = D.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    sampler = arfe_amd.build_sampler(dict(type='RandomSampler', num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True))
    res = sampler.sample(assigner.assign(boxes, gts, None, labels), boxes, gts, labels)
    n_pos, n_neg = res.pos_inds.numel(), res.neg_inds.numel()
    assert n_pos <= 32 and n_pos + n_neg == 128 and res.bboxes.shape == (128, 4)
    assert res.pos_is_gt.sum() <= len(gts) and res.pos_is_gt.dtype == torch.uint8
    assert torch.equal(res.pos_gt_bboxes, gts[res.pos_assigned_gt_inds]) and torch.equal(res.pos_gt_labels, labels[res.pos_assigned_gt_inds])
    assert (D.bbox_overlaps(res.pos_bboxes, res.pos_gt_bboxes, is_aligned=True) >= 0.5).all()
    assert (D.bbox_overlaps(res.neg_bboxes, gts).max(dim=1)[0] < 0.5).all()
    with pytest.raises(ValueError):
        sampler.sample(assigner.assign(boxes, gts, None, labels), boxes, gts, None)
    rpn_sampler = D.RandomSampler(num=64, pos_fraction=0.5, neg_pos_ub=2, add_gt_as_proposals=False)
    res = rpn_sampler.sample(D.MaxIoUAssigner(0.7, 0.3, min_pos_iou=0.3).assign(boxes, gts), boxes, gts)
    assert res.neg_inds.numel() <= 2 * max(1, res.pos_inds.numel()) and res.pos_gt_labels is None


def _head_train_results():
    """the two images of the fixture through OUR assigner + sampler under the fixture's seeds"""
    assigner = D.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    sampler = arfe_amd.build_sampler(dict(type='RandomSampler', num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True))
    out = []
    for i in range(2):
        gts, labels, props = G[f'ht_gts{i}'], G[f'ht_labels{i}'], G[f'ht_props{i}']
        a = assigner.assign(props, gts, None, labels)
        torch.manual_seed(1000 + i)
        out.append(sampler.sample(a, props, gts, labels))
    return out


def test_random_sampler_matches_reference_under_a_fixed_seed():
    """base_sampler.py:35-78 + random_sampler.py:22-77: gt boxes join the candidates, positives / negatives drawn with
    torch.randperm on the candidates' device - same seed, same draw, same bookkeeping (indices, is-gt flags, assigned gts)."""
    for i, r in enumerate(_head_train_results()):
        assert torch.equal(r.pos_inds, G[f'ht_pos_inds{i}']) and torch.equal(r.neg_inds, G[f'ht_neg_inds{i}'])
        assert torch.equal(r.pos_is_gt, G[f'ht_pos_is_gt{i}']) and torch.equal(r.pos_assigned_gt_inds, G[f'ht_pos_assigned{i}'])
        assert torch.equal(r.pos_bboxes, G[f'ht_pos_bboxes{i}']) and torch.equal(r.neg_bboxes, G[f'ht_neg_bboxes{i}'])


@pytest.mark.parametrize('tag,agnostic', [('cs', False), ('ag', True)])
def test_bbox_head_targets_loss_refine_match_reference(tag, agnostic):
    """bbox_head.py:86-156 (targets), :158-205 (loss), :324-352 (regress_by_class), :245-322 (refine_bboxes) of the reference's
    own BBoxHead on the fixture's two images (oracle/gen_golden.py head_train_case): targets bit-exact, losses and boxes 1e-6."""
    nc = 6
    head = arfe_amd.build_head(dict(
        type='MultiRoIsBBoxHead', in_channels=16, conv_out_channels=16, fc_out_channels=8, roi_feat_size=7, num_classes=nc,
        reg_class_agnostic=agnostic,
        bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)))
    results = _head_train_results()
    rois = torch.cat([torch.cat([torch.full((r.bboxes.shape[0], 1), float(i)), r.bboxes], 1) for i, r in enumerate(results)])
    assert torch.equal(rois, G['ht_rois'])
    t = head.get_targets(results, None, None, dict(pos_weight=-1))
    for got, key in zip(t, ('ht_t_labels', 'ht_t_label_weights', 'ht_t_bbox_targets', 'ht_t_bbox_weights')):
        assert torch.equal(got, G[key]), key
    per_img = head.get_targets(results, None, None, dict(pos_weight=-1), concat=False)
    assert len(per_img[0]) == 2 and torch.equal(torch.cat(per_img[2]), G['ht_t_bbox_targets'])
    cls, reg = G[f'ht_{tag}_cls'], G[f'ht_{tag}_reg']
    losses = head.loss(cls, reg, rois, *t)
    assert abs(losses['loss_cls'].item() - float(G[f'ht_{tag}_loss_cls'])) <= 1e-6
    assert abs(losses['loss_bbox'].item() - float(G[f'ht_{tag}_loss_bbox'])) <= 1e-6
    assert torch.equal(losses['acc'], G[f'ht_{tag}_acc'])
    from arfe_amd.roi_heads import cascade_refine_labels
    roi_labels = cascade_refine_labels(head, t[0], cls)
    if agnostic:                    # the labels are not used by a class-agnostic regressor: passed through unchanged
        assert torch.equal(roi_labels, t[0])
    else:
        assert torch.equal(roi_labels, G[f'ht_{tag}_roi_labels'])
    metas = [dict(img_shape=(64, 96, 3)), dict(img_shape=(64, 96, 3))]
    one = head.regress_by_class(rois[:9], roi_labels[:9], reg[:9], metas[0])
    assert (one - G[f'ht_{tag}_regress']).abs().max().item() <= 1e-5
    refined = head.refine_bboxes(rois, roi_labels, reg, [r.pos_is_gt for r in results], metas)
    for i in range(2):
        want = G[f'ht_{tag}_refined{i}']
        assert refined[i].shape == want.shape and (refined[i] - want).abs().max().item() <= 1e-5, i


@pytest.mark.parametrize('tag,agnostic', [('cs', False), ('ag', True)])
def test_bbox_head_get_bboxes_matches_reference(tag, agnostic):
    """bbox_head.py:206-243 without the NMS (cfg=None): softmax scores, per-class decode + clip, rescale by a float or a
    per-coordinate factor, a list of stage scores averaged (Cascade's ensemble), and the no-regression branch."""
    import numpy as np
    nc = 6
    head = arfe_amd.build_head(dict(
        type='MultiRoIsBBoxHead', in_channels=16, conv_out_channels=16, fc_out_channels=8, roi_feat_size=7, num_classes=nc,
        reg_class_agnostic=agnostic,
        bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])))
    rois, cls, reg = G['ht_rois'], G[f'ht_{tag}_cls'], G[f'ht_{tag}_reg']
    boxes, scores = head.get_bboxes(rois, cls, reg, (64, 96, 3), 1.0)
    assert (boxes - G[f'ht_{tag}_gb_boxes']).abs().max().item() <= 1e-5 and (scores - G[f'ht_{tag}_gb_scores']).abs().max().item() <= 1e-6
    assert (head.get_bboxes(rois, cls, reg, (64, 96, 3), 2.0, rescale=True)[0] - G[f'ht_{tag}_gb_boxes_f']).abs().max().item() <= 1e-5
    sf = np.array([1.5, 1.25, 1.5, 1.25], dtype=np.float32)
    assert (head.get_bboxes(rois, cls, reg, (64, 96, 3), sf, rescale=True)[0] - G[f'ht_{tag}_gb_boxes_v']).abs().max().item() <= 1e-5
    avg = head.get_bboxes(rois, [cls, cls * 0.5, -cls], reg, (64, 96, 3), 1.0)[1]
    assert (avg - G[f'ht_{tag}_gb_scores_list']).abs().max().item() <= 1e-6
    assert (head.get_bboxes(rois, cls, None, (40, 50, 3), 1.0)[0] - G[f'ht_{tag}_gb_noreg']).abs().max().item() <= 1e-6
    # with a test cfg: the per-class NMS of multiclass_nms on those boxes / scores
    dets, labels = head.get_bboxes(rois, cls, reg, (64, 96, 3), 1.0, cfg=dict(score_thr=0.05, nms=dict(type='nms', iou_thr=0.5), max_per_img=20))
    want_d, want_l = D.multiclass_nms(boxes, scores, 0.05, dict(type='nms', iou_thr=0.5), 20)
    assert torch.equal(dets, want_d) and torch.equal(labels, want_l) and dets.shape[0] <= 20


def test_multiclass_nms_is_per_class_greedy_nms():
    """core/post_processing/bbox_nms.py:6-60 (the reference's own NMS is a compiled extension: checked against the greedy loop):
    (box, class) pairs above the threshold, background column ignored, greedy NMS inside each class, best max_num by score."""
    g = torch.Generator().manual_seed(12)
    n, nc = 120, 5
    xy = torch.rand(n, 2, generator=g) * 70
    wh = torch.rand(n, 2, generator=g) * 35 + 4
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.softmax(torch.randn(n, nc + 1, generator=g) * 2.0, 1)
    for per_class in (False, True):
        mb = boxes if not per_class else (boxes[:, None, :] + torch.arange(nc).view(1, nc, 1) * 0.5).reshape(n, nc * 4)
        dets, labels = D.multiclass_nms(mb, scores, 0.1, dict(type='nms', iou_thr=0.45), max_num=30)
        want = []
        for c in range(nc):
            bc = mb if not per_class else mb.view(n, nc, 4)[:, c]
            sel = torch.nonzero(scores[:, c] > 0.1, as_tuple=False).squeeze(1)
            dc = torch.cat([bc[sel], scores[sel, c:c + 1]], 1)
            want += [(float(dc[k, 4]), c, dc[k]) for k in _greedy_nms(dc, 0.45)]
        want.sort(key=lambda t: -t[0])
        want = want[:30]
        assert dets.shape[0] == len(want) and labels.tolist() == [w[1] for w in want]
        assert torch.allclose(dets, torch.stack([w[2] for w in want]))
    res = D.bbox2result(dets, labels, nc)
    assert len(res) == nc and sum(r.shape[0] for r in res) == dets.shape[0] and all(r.dtype.name == 'float32' for r in res)
    assert all(r.shape == (0, 5) for r in D.bbox2result(torch.zeros(0, 5), torch.zeros(0, dtype=torch.long), nc))
    e_d, e_l = D.multiclass_nms(boxes, scores * 0, 0.1, dict(type='nms', iou_thr=0.5))
    assert e_d.shape == (0, 5) and e_l.shape == (0, )


def test_cascade_refine_with_class_specific_regressor_and_background_targets():
    """with reg_class_agnostic=False the target label of a background RoI (== num_classes) must not reach
    regress_by_class's gather ([n, 4*num_classes] columns): background RoIs are refined by their best foreground class."""
    import arfe_amd
    from arfe_amd.roi_heads import cascade_refine_labels
    nc = 5
    head = arfe_amd.build_head(dict(
        type='MultiRoIsBBoxHead', in_channels=16, conv_out_channels=16, fc_out_channels=8, roi_feat_size=7, num_classes=nc,
        reg_class_agnostic=False,
        bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])))
    g = torch.Generator().manual_seed(0)
    n = 12
    rois = torch.cat([torch.zeros(n, 1), torch.rand(n, 2, generator=g) * 50, 60 + torch.rand(n, 2, generator=g) * 50], 1)
    labels = torch.tensor([0, 1, nc, nc, 4, 2, nc, 3, nc, 0, nc, 1])
    cls_score = torch.randn(n, nc + 1, generator=g)
    cls_score[:, nc] += 10.0                         # background is the arg-max everywhere: it must still not be chosen
    bbox_pred = torch.randn(n, 4 * nc, generator=g) * 0.1
    lab = cascade_refine_labels(head, labels, cls_score)
    assert int(lab.max()) < nc and torch.equal(lab[labels != nc], labels[labels != nc])
    assert torch.equal(lab[labels == nc], cls_score[labels == nc, :nc].argmax(1))
    out = head.refine_bboxes(rois, lab, bbox_pred, [torch.zeros(3, dtype=torch.uint8)], [dict(img_shape=(200, 200, 3))])
    assert out[0].shape == (n, 4) and torch.isfinite(out[0]).all()
    with pytest.raises((RuntimeError, IndexError)):   # what the raw target labels do (the hole in the reference at HEAD)
        head.refine_bboxes(rois, labels, bbox_pred, [torch.zeros(3, dtype=torch.uint8)], [dict(img_shape=(200, 200, 3))])
    head.reg_class_agnostic = True                   # class-agnostic heads (every AR cascade config): labels pass through
    assert cascade_refine_labels(head, labels, cls_score) is labels


def test_resnet_surface():
    net = arfe_amd.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                                       norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'))
    net.init_weights()
    keys = set(net.state_dict())
    for k in ('conv1.weight', 'bn1.running_mean', 'layer1.0.conv1.weight', 'layer1.0.downsample.0.weight',
              'layer1.0.downsample.1.weight', 'layer2.3.bn3.bias', 'layer3.5.conv2.weight', 'layer4.2.conv3.weight'):
        assert k in keys, k
    assert sum(p.numel() for p in net.parameters()) == 23508032          # torchvision / mmdet ResNet-50 without the fc layer
    net.train()
    assert not net.bn1.training and not net.layer1[0].bn1.training and not net.layer3[0].bn1.training   # norm_eval
    assert not net.conv1.weight.requires_grad and not net.layer1[2].conv3.weight.requires_grad and net.layer2[0].conv1.weight.requires_grad
    assert all((m.bn3.weight == 0).all() for m in net.modules() if isinstance(m, D.Bottleneck))        # zero_init_residual
    assert net.layer2[0].conv2.stride == (2, 2) and net.layer2[0].conv1.stride == (1, 1)                # style='pytorch'
    with torch.no_grad():
        outs = net(torch.randn(1, 3, 64, 96))
    assert [tuple(o.shape) for o in outs] == [(1, 256, 16, 24), (1, 512, 8, 12), (1, 1024, 4, 6), (1, 2048, 2, 3)]
    assert len(arfe_amd.build_backbone(dict(type='ResNet', depth=101)).layer3) == 23
    with pytest.raises(KeyError):
        D.ResNet(18)


REF = '/root/reference/configs'


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference configs exist only in the build container')
def test_detectors_build_from_the_reference_configs():
    """BASELINE configs #2-#5: the AR flagship config and the stock Faster / Cascade R-CNN configs build whole detectors."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cfg = arfe_amd.Config.fromfile(os.path.join(REF, 'mytrain', 'faster_rcnn_r50_drfpn_drrff+fac_1x_coco.py'))
        det = arfe_amd.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(det).__name__ == 'FasterRCNN' and [type(m).__name__ for m in det.neck] == ['FPN', 'WFPNDualSpatial']
    assert type(det.roi_head.bbox_head).__name__ == 'MultiRoIsBBoxHead' and det.rpn_head.num_anchors == 3
    assert det.roi_head.bbox_sampler.num == 512 and det.rpn_head.sampler.num == 256
    keys = set(det.state_dict())
    for k in ('backbone.layer4.2.conv3.weight', 'neck.0.lateral_convs.0.conv.weight', 'neck.1.refine.theta.conv.weight',
              'neck.1.reduce_convs2.4.conv.bias', 'rpn_head.rpn_conv.weight', 'rpn_head.rpn_cls.bias',
              'roi_head.bbox_head.wh_conv.conv.weight', 'roi_head.bbox_head.shared_fcs.0.weight', 'roi_head.bbox_head.fc_reg.bias'):
        assert k in keys, k
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        c = arfe_amd.Config.fromfile(os.path.join(REF, 'cascade_rcnn', 'cascade_rcnn_r50_fpn_1x_coco.py'))
    assert c.model.type == 'CascadeRCNN' and c.model.roi_head.num_stages == 3 and len(c.train_cfg.rcnn) == 3


def _small_detector(kind, dev, depth=50):
    """Faster / Cascade R-CNN R50 + [FPN, AR-FPN] + AR-RFF with the AR configs' settings at a reduced RoI / proposal budget."""
    rcnn = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False,
                              ignore_iof_thr=-1),
                sampler=dict(type='RandomSampler', num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
                pos_weight=-1, debug=False)
    train_cfg = dict(
        rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True,
                               ignore_iof_thr=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                 allowed_border=-1, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_across_levels=False, nms_pre=300, nms_post=200, max_num=200, nms_thr=0.7, min_bbox_size=0),
        rcnn=rcnn if kind == 'FasterRCNN' else [dict(rcnn, assigner=dict(rcnn['assigner'], pos_iou_thr=t, neg_iou_thr=t, min_pos_iou=t))
                                                for t in (0.5, 0.6, 0.7)])
    head = dict(type='MultiRoIsBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2]),
                reg_class_agnostic=False, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                loss_bbox=dict(type='L1Loss', loss_weight=1.0))
    ext = dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0), out_channels=256,
               featmap_strides=[4, 8, 16, 32, 64])
    if kind == 'FasterRCNN':
        roi_head = dict(type='StandardRoIHead', bbox_roi_extractor=ext, bbox_head=head)
    else:
        stds = ([0.1, 0.1, 0.2, 0.2], [0.05, 0.05, 0.1, 0.1], [0.033, 0.033, 0.067, 0.067])
        roi_head = dict(type='CascadeRoIHead', num_stages=3, stage_loss_weights=[1, 0.5, 0.25], bbox_roi_extractor=ext,
                        bbox_head=[dict(head, reg_class_agnostic=True, bbox_coder=dict(head['bbox_coder'], target_stds=s)) for s in stds])
    model = dict(type=kind, pretrained=None,
                 backbone=dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                               norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch'),
                 neck=[dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
                       dict(type='WFPNDualSpatial', in_channels=256, num_levels=5)],
                 rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256,
                               anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
                               bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                               loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                               loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
                 roi_head=roi_head)
    test_cfg = dict(rpn=dict(nms_across_levels=False, nms_pre=300, nms_post=200, max_num=200, nms_thr=0.7, min_bbox_size=0),
                    rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_thr=0.5), max_per_img=100))
    torch.manual_seed(0)
    return arfe_amd.build_detector(model, train_cfg=train_cfg, test_cfg=test_cfg).to(dev).train()


@pytest.mark.gpu
@pytest.mark.parametrize('kind,depth', [('FasterRCNN', 50), ('CascadeRCNN', 50), ('CascadeRCNN', 101)])
def test_detector_forward_train_end_to_end(kind, depth):
    """BASELINE configs #2/#3 (Faster R-CNN R50) and #5's architecture (Cascade R-CNN R101, 3 AR-RFF stages): synthetic
    COCO-shaped batch -> losses -> backward, with FPN, AR-FPN, the 3-set RoIAlign and the AR-RFF head on the HIP library."""
    dev = torch.device('cuda:0')
    det = _small_detector(kind, dev, depth)
    assert len(det.backbone.layer3) == (23 if depth == 101 else 6)
    img, metas, gt_bboxes, gt_labels = D.synthetic_batch(batch=2, height=256, width=384, max_gt=8, seed=1, device=dev)
    losses = det.forward_train(img, metas, gt_bboxes, gt_labels)
    want = {'loss_rpn_cls', 'loss_rpn_bbox'} | ({'loss_cls', 'acc', 'loss_bbox'} if kind == 'FasterRCNN' else
                                                 {f's{i}.{k}' for i in range(3) for k in ('loss_cls', 'acc', 'loss_bbox')})
    assert set(losses) == want
    loss, log = D.parse_losses(losses)
    assert torch.isfinite(loss) and all(v == v for v in log.values())
    loss.backward()
    torch.cuda.synchronize()
    named = dict(det.named_parameters())
    for k in ('neck.1.refine.theta.conv.weight', 'neck.1.reduce_convs.0.conv.weight', 'neck.0.fpn_convs.0.conv.weight',
              'rpn_head.rpn_conv.weight', 'backbone.layer2.0.downsample.0.weight', 'backbone.layer3.1.bn3.weight'):
        # (zero_init_residual: a block's bn3 scale starts at 0, so its own convs see zero gradient at step 0 - the
        #  shortcut convolution and the bn3 scale itself do not)
        assert named[k].grad is not None and torch.isfinite(named[k].grad).all() and named[k].grad.abs().sum() > 0, k
    head0 = 'roi_head.bbox_head.' if kind == 'FasterRCNN' else 'roi_head.bbox_head.0.'
    for k in ('wh_conv.conv.weight', 'final_conv.conv.weight', 'shared_fcs.0.weight', 'fc_cls.weight'):
        g = named[head0 + k].grad
        assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0, k
    assert named['backbone.conv1.weight'].grad is None                   # frozen stem
    # forward_dummy (two_stage.py:137-153): rpn outputs per level + (cls, reg) per stage
    with torch.no_grad():
        outs = det.forward_dummy(img[:1])
    assert len(outs) == 2 + 2 * (1 if kind == 'FasterRCNN' else 3) and outs[-1].shape[0] == 1000


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['FasterRCNN', 'CascadeRCNN'])
def test_detector_simple_test_end_to_end(kind):
    """two_stage.py:236-330 / base.py:100-135: one image -> RPN proposals (test cfg) -> AR-RFF head(s) on the HIP library ->
    decode -> per-class NMS -> per-class [n_c, 5] arrays; `forward(return_loss=False)` is the same call."""
    import numpy as np
    dev = torch.device('cuda:0')
    det = _small_detector(kind, dev, 50).eval()
    img, metas, _, _ = D.synthetic_batch(batch=1, height=256, width=384, max_gt=4, seed=2, device=dev)
    res = det.simple_test(img, metas)
    nc = (det.roi_head.bbox_head if kind == 'FasterRCNN' else det.roi_head.bbox_head[-1]).num_classes
    assert len(res) == nc and all(isinstance(r, np.ndarray) and r.ndim == 2 and r.shape[1] == 5 for r in res)
    n = sum(r.shape[0] for r in res)
    assert n <= det.roi_head.test_cfg['max_per_img']
    for r in res:
        if r.shape[0]:
            assert np.isfinite(r).all() and (r[:, 4] > det.roi_head.test_cfg['score_thr']).all() and (r[:, 4] <= 1.0).all()
            assert (r[:, 0] >= 0).all() and (r[:, 2] <= 384).all() and (r[:, 1] >= 0).all() and (r[:, 3] <= 256).all()
            assert (np.diff(r[:, 4]) <= 1e-7).all()                      # inside a class: descending score
    again = det(img=[img], img_metas=[metas], return_loss=False)
    assert len(again) == nc and all(np.array_equal(a, b) for a, b in zip(again, res))
    with pytest.raises(NotImplementedError):
        det(img=[img, img], img_metas=[metas, metas], return_loss=False)
    # proposals handed in (two_stage.py:324-327): the RPN is skipped
    props = [torch.tensor([[10., 20., 120., 140., 0.9], [50., 60., 300., 200., 0.8]], device=dev)]
    res2 = det.simple_test(img, metas, proposals=props)
    assert len(res2) == nc and sum(r.shape[0] for r in res2) <= 2 * nc
