"""bbox2roi (reference mmdet/core/bbox/transforms.py:41-60): per-image boxes -> [R,5] (batch_idx,x1,y1,x2,y2)."""
import torch


def bbox2roi(bbox_list):
    """Concatenates the per-image box tensors ([n_i, >=4]) into one RoI tensor whose first column is the image index;
    an image without boxes contributes no rows."""
    parts = []
    for index, boxes in enumerate(bbox_list):
        xyxy = boxes[:, :4]
        batch_col = torch.full((xyxy.shape[0], 1), float(index), dtype=xyxy.dtype, device=xyxy.device)
        parts.append(torch.cat((batch_col, xyxy), dim=1))
    return torch.cat(parts, dim=0)
