"""TEST INFRASTRUCTURE — whole-path parity of a device run against the CPU oracle (used by tests/ and by bench.py's
`cpu_baseline` leg only; the product package never imports this).

One call compares one forward+backward of neck -> 3-RoI extractor -> AR-RFF head (reference:
mmdet/models/necks/wfpn_dual_spatial.py:99-137, roi_heads/standard_roi_head.py:138-155 [intended body],
roi_heads/bbox_heads/multirois_bbox_head.py:165-235) on the same inputs, parameters and cotangents:

  * forward outputs against the free-running oracle, max-norm (max|a-b| / max|b|);
  * the DISCRETE choices of the two forward passes (ReLU open/closed, pooling argmax; `arfe_oracle.Decisions`) are counted:
    `mask_flips` = how many differ, out of `decisions`;
  * every gradient against the oracle evaluated UNDER THE DEVICE'S choices, max-norm - all elements are comparable there,
    an element whose choice differs no longer carries its whole gradient on one side only;
  * for information, the same gradients against the free-running oracle in L2 (what the flips cost).
"""
import torch

from . import arfe_oracle as O

SKIP = ('refine.phi.conv.bias',)    # d/d(b_phi) is identically zero (softmax shift invariance): rounding noise on both sides


def _rel_max(a, b):
    if a is None:
        a = torch.zeros_like(b)
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _rel_l2(a, b):
    if a is None:
        a = torch.zeros_like(b)
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).norm().item() / max(b.norm().item(), 1e-300)


def decisions_from_tap(tap, refine_level=2):
    """arfe_amd.functional's decision tap (the tensors the device forward saved for its own backward) -> Decisions on the CPU."""
    am = tap.get('argmax')
    return O.Decisions(argmax=[am[i].cpu().long() for i in range(refine_level)] if (am is not None and refine_level > 0) else
                       ([] if refine_level == 0 else None),
                       gate_a=[t.cpu() > 0 for t in tap['gate_a']], gate_b=[t.cpu() > 0 for t in tap['gate_b']],
                       **{k: tap[k].cpu() > 0 for k in ('lw', 'lh', 'z2', 'f1', 'f2')})


def device_step(neck, roi_head, xs, rois, cot_cls, cot_reg, refine_level=2):
    """One forward+backward of the PRODUCT (registry-built `neck`, `roi_head` on the GPU; xs require grad) with the decision
    tap installed -> the dict `compare` takes.  Parameter gradients are read from .grad (cleared here first)."""
    from arfe_amd import functional as Fn
    for p in list(neck.parameters()) + list(roi_head.bbox_head.parameters()):
        p.grad = None
    for x in xs:
        x.grad = None
    tap = {}
    Fn.tap_decisions(tap)
    try:
        feats = neck(xs)
        res = roi_head._bbox_forward(feats, rois)
        torch.autograd.backward([res['cls_score'], res['bbox_pred']], [cot_cls, cot_reg])
        torch.cuda.synchronize()
    finally:
        Fn.tap_decisions(None)
    return dict(cls=res['cls_score'].detach().cpu(), reg=res['bbox_pred'].detach().cpu(),
                dxs=[x.grad.detach().cpu() if x.grad is not None else None for x in xs],
                dneck={k: (v.grad.detach().cpu() if v.grad is not None else None) for k, v in neck.named_parameters()},
                dhead={k: (v.grad.detach().cpu() if v.grad is not None else None)
                       for k, v in roi_head.bbox_head.named_parameters()},
                decisions=decisions_from_tap(tap, refine_level))


def oracle_step(xs, rois, pn, ph, cot_cls, cot_reg, refine_level=2, decisions=None, backward=True):
    """One oracle pass -> dict(cls, reg, dxs, dneck, dhead, decisions).  xs / pn / ph are plain CPU tensors (cloned here)."""
    xr = [x.detach().clone().requires_grad_(backward) for x in xs]
    pnr = {k: v.detach().clone().requires_grad_(backward) for k, v in pn.items()}
    phr = {k: v.detach().clone().requires_grad_(backward) for k, v in ph.items()}
    with torch.set_grad_enabled(backward):
        feats, parts_n = O.arfpn_forward(xr, pnr, refine_level, return_parts=True, decisions=decisions)
        cls, reg, parts_h = O.rff_head(O.roi_extract3(list(feats), rois), phr, return_parts=True, decisions=decisions)
        if backward:
            torch.autograd.backward([cls, reg], [cot_cls, cot_reg])
    out = dict(cls=cls.detach(), reg=reg.detach(), decisions=O.decisions_of(xr, parts_n, parts_h, refine_level))
    if backward:
        out.update(dxs=[x.grad for x in xr], dneck={k: v.grad for k, v in pnr.items()},
                   dhead={k: v.grad for k, v in phr.items()})
    return out


def compare(dev, xs, rois, pn, ph, cot_cls, cot_reg, refine_level=2, free=None):
    """dev: dict(cls, reg, dxs, dneck, dhead, decisions) of the device run (any device; moved to the CPU here).
    free: an `oracle_step` result of the free-running oracle on the same data if the caller already has one.
    -> report dict (JSON-serialisable)."""
    dz = dev['decisions']
    if free is None:
        free = oracle_step(xs, rois, pn, ph, cot_cls, cot_reg, refine_level, backward=True)
    under = oracle_step(xs, rois, pn, ph, cot_cls, cot_reg, refine_level, decisions=dz, backward=True)
    flips = free['decisions'].differing(dz)
    rep = {"max_rel_fwd": max(_rel_max(dev['cls'], free['cls']), _rel_max(dev['reg'], free['reg'])),
           "decisions": dz.count(), "mask_flips": int(sum(flips.values())),
           "mask_flips_by_site": {k: v for k, v in flips.items() if v}}
    worst, worst_name, l2_free, l2_name = 0.0, None, 0.0, None

    def one(name, got, ref_under, ref_free):
        nonlocal worst, worst_name, l2_free, l2_name
        if ref_under is None:           # the oracle left no gradient (a level no RoI touched): the device's must be zero
            e = 0.0 if got is None else float(got.abs().max())
            if e > worst:
                worst, worst_name = e, name
            return
        e = _rel_max(got, ref_under)
        if e > worst:
            worst, worst_name = e, name
        if ref_free is not None:
            l = _rel_l2(got, ref_free)
            if l > l2_free:
                l2_free, l2_name = l, name

    for i, g in enumerate(dev['dxs']):
        one(f'dx{i}', g, under['dxs'][i], free.get('dxs', [None] * 5)[i] if 'dxs' in free else None)
    for tag in ('dneck', 'dhead'):
        for k, g in dev[tag].items():
            if k in SKIP:
                continue
            one(f'{tag}.{k}', g, under[tag][k], free[tag][k] if tag in free else None)
    rep.update({"max_rel_grad_agreeing": worst, "worst_grad": worst_name,
                "l2_rel_grad_free_running_max": l2_free, "l2_worst_grad": l2_name,
                "definition": "fwd: max-norm vs the free-running CPU oracle; mask_flips: ReLU/argmax choices that differ "
                              "from it; grads: max-norm vs the oracle evaluated under the device's choices"})
    return rep
