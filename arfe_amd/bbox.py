"""bbox2roi (reference mmdet/core/bbox/transforms.py:41-60): per-image boxes -> [R,5] (batch_idx,x1,y1,x2,y2)."""
import torch


def bbox2roi(bbox_list):
    rois_list = []
    for img_id, bboxes in enumerate(bbox_list):
        if bboxes.size(0) > 0:
            img_inds = bboxes.new_full((bboxes.size(0), 1), img_id)
            rois = torch.cat([img_inds, bboxes[:, :4]], dim=-1)
        else:
            rois = bboxes.new_zeros((0, 5))
        rois_list.append(rois)
    return torch.cat(rois_list, 0)
