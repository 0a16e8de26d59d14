"""Hungarian matcher (mirror of reference models/matcher.py:20-87, 151-152).

The pairwise cost C = w_bbox*L1 + w_class*focal + w_giou*(-GIoU) is one HIP launch for ALL
prediction sets (decoder layers) and images at once and only over same-image (query, target) pairs
(the reference builds the full [B*Q, sum M] matrix, cross-image blocks included, once per layer);
a single device->host copy follows and the assignment itself stays on the host (SciPy
linear_sum_assignment, as in the reference)."""
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from .. import kernels as K
from ..util.misc import host_to_device


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, match_ratio: int = 1):
        super().__init__()
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        self.match_ratio = match_ratio
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    def _costs(self, logits, boxes, targets):
        L, B, Q, _ = logits.shape
        dev = logits.device
        sizes = [int(len(t["boxes"])) for t in targets]
        toff = [0]
        for s in sizes:
            toff.append(toff[-1] + s)
        tgt_ids = torch.cat([t["labels"] for t in targets]).to(device=dev, dtype=torch.int32).contiguous()
        tgt_box = torch.cat([t["boxes"] for t in targets]).to(device=dev, dtype=torch.float32).contiguous()
        toff_t = host_to_device(toff, torch.int32, dev)
        cost, err = K.matcher_cost(logits.contiguous().float(), boxes.contiguous().float(), tgt_ids, tgt_box, toff_t,
                                   toff[-1], self.cost_class, self.cost_bbox, self.cost_giou)
        return cost, err, toff_t, sizes, toff

    def _check_degenerate_async(self, err):
        """box_ops.py:64-65 asserts non-degenerate boxes.  On the device path the flag of step t is inspected at
        step t+1 (pinned host copy + event), so the check costs no synchronisation."""
        prev = getattr(self, "_err_prev", None)
        if prev is not None:
            host, ev = prev
            ev.synchronize()            # a step old: already complete
            flag = int(host[0])
            assert not (flag & 1), "degenerate box (x1 < x0 or y1 < y0) in matcher inputs"
            if flag & 2:      # what scipy.optimize.linear_sum_assignment raises at reference matcher.py:86
                raise ValueError("matrix contains invalid numeric entries (non-finite matching cost)")
        host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
        host.copy_(err, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._err_prev = (host, ev)

    @torch.no_grad()
    def match_flat(self, logits, boxes, targets):
        """Device-side assignment for all layers and images at once (csrc/loss.hip: hungarian_kernel): ->
        (srow, gidx int64, lidx int32) device tensors of length L*sum(M_b), ordered (layer, image, query) - the
        flattened form SetCriterion consumes - or None when a problem does not fit the kernel (M_b > Q or Q > 1024;
        the caller then uses match_many)."""
        L, B, Q, _ = logits.shape
        sizes = [int(len(t["boxes"])) for t in targets]
        if sum(sizes) == 0 or max(sizes) > Q or Q > 1024 or not logits.is_cuda:
            return None
        cost, err, toff_t, sizes, toff = self._costs(logits, boxes, targets)
        res = K.hungarian(cost, toff_t, L, B, Q, toff[-1], err=err)
        self._check_degenerate_async(err)           # after the assignment: it raises bit 1 of the same flag word
        return res

    @torch.no_grad()
    def match_many(self, logits, boxes, targets):
        """logits [L,B,Q,Kc], boxes [L,B,Q,4]; -> list (len L) of per-image [(idx_i, idx_j)] int64 CPU tensors."""
        L, B, Q, _ = logits.shape
        dev = logits.device
        sizes = [int(len(t["boxes"])) for t in targets]
        total = sum(sizes)
        toff = [0]
        for s in sizes:
            toff.append(toff[-1] + s)
        empty = (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64))
        if total == 0:
            return [[empty for _ in range(B)] for _ in range(L)]
        tgt_ids = torch.cat([t["labels"] for t in targets]).to(device=dev, dtype=torch.int32).contiguous()
        tgt_box = torch.cat([t["boxes"] for t in targets]).to(device=dev, dtype=torch.float32).contiguous()
        toff_t = torch.tensor(toff, dtype=torch.int32, device=dev)
        cost, err = K.matcher_cost(logits.contiguous().float(), boxes.contiguous().float(), tgt_ids, tgt_box, toff_t,
                                   total, self.cost_class, self.cost_bbox, self.cost_giou)
        host = torch.cat([cost.view(-1), err.float()]).cpu()     # the one D2H copy (and sync) of the matcher
        assert host[-1].item() == 0, "degenerate box (x1 < x0 or y1 < y0) in matcher inputs"   # box_ops.py:64-65
        c = host[:-1].view(L, Q * total)
        res = []
        for l in range(L):
            per = []
            for b in range(B):
                if sizes[b] == 0:
                    per.append(empty)
                    continue
                blk = c[l, Q * toff[b]:Q * toff[b + 1]].view(Q, sizes[b])
                i, j = linear_sum_assignment(blk.numpy())
                per.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            res.append(per)
        return res

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Reference signature: outputs {'pred_logits' [B,Q,Kc], 'pred_boxes' [B,Q,4]} -> list of B (idx_i, idx_j)."""
        return self.match_many(outputs["pred_logits"][None], outputs["pred_boxes"][None], targets)[0]


def build_matcher(args):
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou,
                            match_ratio=args.hung_match_ratio)
