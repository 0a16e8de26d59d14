"""CAM -> pseudo-label boxes, the step before the criterion in the reference training loop (SURVEY.md section 8(f)
rank 1): `engine.get_pseudo_label_multi_boxes` (engine.py:356-398) with `cams_deit.resize_cam` / `get_multi_bboxes`
(cams_deit.py:9-13, 61-96).  The per-pixel work (bilinear resize to image size, min-max, quantise, threshold) runs on
the device for all (image, present class) maps at once; one device->host copy of the thresholded uint8 images follows
and the border following / box selection runs in native host code (csrc/cambox.hip) - the reference runs the whole
thing through NumPy + OpenCV per class on the host.  OpenCV is absent in this environment: the arithmetic is restated
from its published algorithms and pinned only by oracle/cam_oracle.py (same restatement in NumPy) and structural
checks against scipy.ndimage in the tests.
"""
import torch

from . import kernels as K


def _xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


@torch.no_grad()
def get_pseudo_label_multi_boxes(outputs, samples, targets, args):
    """Same signature and result as engine.py:356-398: list (one per image) of {'boxes': [n,4] normalised cxcywh,
    'labels': [n] class ids (1-based)} on the device of `samples.tensors`.

    Reproduced quirk: the reference passes `size = (H, W)` to `resize_cam`, which hands it to cv2.resize as
    (width, height) - the CAM is resized to W rows x H columns - and then divides the boxes by [W, H, W, H]."""
    tensors = samples.tensors if hasattr(samples, "tensors") else samples
    device = tensors.device
    cams = outputs["cams_cls"]
    B, Kc = cams.shape[0], cams.shape[1]
    H, W = tensors.shape[-2:]
    rows, cols = int(W), int(H)
    labels_host = [t["img_label"].detach().cpu().reshape(-1) for t in targets]        # the reference does the same
    sel = [(b, c) for b in range(B) for c in range(min(args.num_classes, Kc)) if labels_host[b][c] > 0]
    out = []
    if not sel:
        return [{"boxes": torch.zeros((0, 4), device=device), "labels": torch.zeros((0,), dtype=torch.long, device=device)}
                for _ in range(B)]
    bi = torch.tensor([s[0] for s in sel], device=cams.device)
    ci = torch.tensor([s[1] for s in sel], device=cams.device)
    maps = cams[bi, ci].float().contiguous()
    host = K.cam_prepare(maps, rows, cols, args.cam_thr).cpu()                       # one device->host copy
    per_img = [([], []) for _ in range(B)]
    for m, (b, c) in enumerate(sel):
        bx = K.cam_contour_boxes(host[m], args.multi_box_ratio).to(torch.int64)       # torch.tensor(list of ints)
        bx = _xyxy_to_cxcywh(bx)                                                      # integer arithmetic -> true division -> float
        per_img[b][0].append(bx)
        per_img[b][1].extend([c + 1] * bx.shape[0])
    scale = torch.tensor([W, H, W, H], dtype=torch.float32)
    for b in range(B):
        if per_img[b][0]:
            boxes = torch.cat(per_img[b][0], dim=0).float() / scale
            labels = torch.tensor(per_img[b][1], dtype=torch.long)
        else:                       # no class present: the reference would fail on torch.cat([]); return empty sets
            boxes, labels = torch.zeros((0, 4)), torch.zeros((0,), dtype=torch.long)
        out.append({"boxes": boxes.to(device), "labels": labels.to(device)})
    return out
