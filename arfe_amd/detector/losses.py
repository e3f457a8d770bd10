"""Losses of the detector harness (reference mmdet/models/losses/{utils,cross_entropy_loss,smooth_l1_loss,accuracy}.py).

The fork's cross_entropy() differs from upstream mmdetection: its signature is
`cross_entropy(pred, label, multi_cls_pred=None, weight=None, ...)` (cross_entropy_loss.py:9) while
CrossEntropyLoss.forward still passes the sample weights as the THIRD POSITIONAL argument (:183-190), so for the softmax
criterion the per-sample weights arrive as `multi_cls_pred`.  The reference then weights sample i with
`1 - softmax(label_weights)[argmax_c pred[i, c]]` (:19-27): the weight vector is soft-maxed over the SAMPLES and indexed
with the predicted CLASS.  That is the behaviour of the reference at HEAD and it is reproduced here as is (flag
`fork_weight_quirk`, default True; with uniform weights it scales the loss by 1 - 1/N).  The sigmoid criterion (RPN)
has the upstream signature and is unaffected."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import LOSSES


def reduce_loss(loss, reduction):
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError(f'{reduction} is not a valid value for reduction')


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """element-wise weight, then mean / sum, or sum / avg_factor when an averaging factor is given (utils.py:29-55)."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, fork_weight_quirk=True):
    loss = F.cross_entropy(pred, label, reduction='none')
    if weight is not None:
        if fork_weight_quirk:
            over_samples = torch.softmax(weight, dim=-1)
            weight = 1.0 - over_samples[torch.argmax(pred, dim=-1).view(-1)]
        else:
            weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


def _expand_binary_labels(labels, label_weights, label_channels):
    """RPN only: class index (1-based foreground) -> one-hot over `label_channels` (cross_entropy_loss.py:106-120)."""
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    inds = torch.nonzero(labels >= 1, as_tuple=False).squeeze(1)
    if inds.numel() > 0:
        bin_labels[inds, labels[inds] - 1] = 1
    w = None if label_weights is None else label_weights.view(-1, 1).expand(label_weights.size(0), label_channels)
    return bin_labels, w


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None):
    if pred.dim() != label.dim():
        label, weight = _expand_binary_labels(label, weight, pred.size(-1))
    if weight is not None:
        weight = weight.float()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), weight, reduction='none')
    return weight_reduce_loss(loss, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, use_dis=False, use_multi_cls=False, reduction='mean',
                 loss_weight=1.0, fork_weight_quirk=True):
        super().__init__()
        if use_mask or use_dis or use_multi_cls:
            raise NotImplementedError("mask / distribution / multi-class criteria are the fork's ablations, not on the AR path")
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight
        self.fork_weight_quirk = fork_weight_quirk

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if self.use_sigmoid:
            loss = binary_cross_entropy(cls_score, label, weight, reduction=reduction, avg_factor=avg_factor)
        else:
            loss = cross_entropy(cls_score, label, weight, reduction=reduction, avg_factor=avg_factor,
                                 fork_weight_quirk=self.fork_weight_quirk)
        return self.loss_weight * loss


def smooth_l1_loss(pred, target, weight=None, beta=1.0, reduction='mean', avg_factor=None):
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


def l1_loss(pred, target, weight=None, reduction='mean', avg_factor=None):
    assert pred.size() == target.size() and target.numel() > 0
    return weight_reduce_loss(torch.abs(pred - target), weight, reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor)


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * l1_loss(pred, target, weight, reduction=reduction, avg_factor=avg_factor)


def accuracy(pred, target, topk=1):
    """top-k accuracy in percent (models/losses/accuracy.py:4-36)."""
    ks = (topk, ) if isinstance(topk, int) else tuple(topk)
    if pred.size(0) == 0:
        res = [pred.new_tensor(0.) for _ in ks]
    else:
        _, top = pred.topk(max(ks), dim=1)
        hit = top.t().eq(target.view(1, -1).expand_as(top.t()))
        res = [hit[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0)) for k in ks]
    return res[0] if isinstance(topk, int) else res
