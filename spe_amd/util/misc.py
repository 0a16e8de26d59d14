"""Host-side helpers the hot path shares with its callers (mirror of reference util/misc.py:
NestedTensor 291-311, nested_tensor_from_tensor_list 314-336, inverse_sigmoid 477-481,
dist helpers 385-396).  Pure plumbing: no device arithmetic beyond padding copies."""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]):
    """Zero-pad a list of [C,H,W] images to the batch maximum; mask is True on padding."""
    if isinstance(tensor_list, Tensor) and tensor_list.ndim == 4:
        tensor_list = list(tensor_list.unbind(0))
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(img.shape[1] for img in tensor_list)
    w = max(img.shape[2] for img in tensor_list)
    b = len(tensor_list)
    dtype, device = tensor_list[0].dtype, tensor_list[0].device
    tensor = torch.zeros((b, c, h, w), dtype=dtype, device=device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(tensor, mask)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of 0-dim loss tensors for logging (reference util/misc.py:139-163, called at engine.py:147):
    keys sorted so every rank stacks the same order, ONE collective for the whole dict, / world when `average`."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return dict(zip(names, values))


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """precision@k in percent of [n, classes] scores against [n] labels (reference util/misc.py:439-455); the criterion's
    `class_error` is 100 - accuracy(...)[0] (inside SetCriterion the top-1 comes out of the focal kernel's pass instead)."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def host_to_device(values, dtype, device):
    """Small host list -> device tensor through pinned memory with an asynchronous copy.  `torch.tensor(list,
    device=cuda)` stages through pageable memory and blocks the host until every kernel queued before it has run."""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)
