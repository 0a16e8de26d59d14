"""Inference-side post-processing of the SPE detector: the tensor work of
reference engine_loc.py:99-124 (`decouple_output`, flip test-time augmentation) and engine_loc.py:150-174
(`postprocessors['bbox'](outputs, sizes, 300)` followed by per-class torchvision NMS at IoU 0.5).
"""
import torch

from . import kernels as K

_COMBINE = ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls", "aux_outputs")


def decouple_output(output, bs=2):
    """Merge the outputs of a [images ; horizontally flipped images] batch (engine_loc.py:99-124): flipped boxes get
    cx -> 1 - cx, image-level logits take the element-wise maximum, everything else is concatenated along the query
    axis, recursively for `aux_outputs`.  In place on the dict, like the reference; tensors are not aliased."""
    for k in _COMBINE:
        if k not in output:
            continue
        v = output[k]
        if k == "aux_outputs":
            for i, aux in enumerate(v):
                v[i] = decouple_output(aux, bs=bs)
            continue
        pre, pos = v[:bs], v[bs:2 * bs]
        if k == "pred_boxes":
            pos = pos.clone()
            pos[..., 0] = 1 - pos[..., 0]
        if k in ("x_logits", "x_cls_logits"):
            output[k] = torch.maximum(pre, pos)
            continue
        output[k] = torch.cat((pre, pos), dim=1)
    return output


@torch.no_grad()
def per_class_nms(results, iou_threshold=0.5):
    """results: list (one per image) of {'scores' [n], 'labels' [n], 'boxes' [n,4] xyxy} as returned by PostProcess.
    Returns the list the reference builds at engine_loc.py:154-174: for every predicted class in ascending order the
    NMS survivors in descending score order.  One sort + one HIP launch for the whole batch, no per-class host loop."""
    if not results:
        return results
    n = results[0]["scores"].numel()
    assert all(r["scores"].numel() == n for r in results), "PostProcess returns the same count for every image"
    scores = torch.stack([r["scores"] for r in results]).float()
    labels = torch.stack([r["labels"] for r in results]).long()
    boxes = torch.stack([r["boxes"] for r in results]).float()
    # order by (label asc, score desc): stable sort by score first, then stable sort by label
    o1 = torch.sort(scores, dim=1, descending=True, stable=True).indices
    o2 = torch.sort(labels.gather(1, o1), dim=1, stable=True).indices
    order = o1.gather(1, o2)
    sl = labels.gather(1, order).contiguous()
    sb = boxes.gather(1, order[..., None].expand(-1, -1, 4)).contiguous()
    ss = scores.gather(1, order)
    keep = K.nms_sorted(sb, sl, iou_threshold)
    out = []
    for i in range(len(results)):
        m = keep[i]
        out.append({"scores": ss[i][m], "labels": sl[i][m], "boxes": sb[i][m]})
    return out
