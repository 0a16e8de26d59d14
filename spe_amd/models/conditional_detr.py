"""ConditionalDETR_Refine + SetCriterion / SetCriterionRefine + post-processors + build()
(host-side mirror of reference models/conditional_detr.py: model 33-124, SetCriterion 190-494,
SetCriterionRefine 497-589, PostProcess 592-623, PostProcessRefine 641-677, build 733-802).

Same constructor/forward signatures, output dictionaries, loss keys and state_dict names, so
reference main.py / engine.py drive it unchanged.  Device arithmetic is libspe_hip.so:
all decoder layers' matching costs in one launch + one D2H copy (reference: 2*dec_layers round
trips), all layers' focal losses in one launch, all layers' matched-box L1/GIoU in one launch.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from .. import kernels as K
from ..util import box_ops
from ..util.misc import (NestedTensor, get_world_size, host_to_device, inverse_sigmoid, is_dist_avail_and_initialized,
                         nested_tensor_from_tensor_list)
from .cait_backbone import build_backbone
from .layers import MLP, Linear
from .matcher import build_matcher
from .transformer import build_transformer


class ConditionalDETR_Refine(nn.Module):
    """CaiT backbone + conditional-DETR decoder run once per sparse-proposal set (stage)."""

    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, num_refines=1, drloc=False):
        super().__init__()
        self.num_queries = num_queries
        self.num_refines = num_refines
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = nn.ModuleList([Linear(hidden_dim, num_classes) for _ in range(num_refines + 1)])
        self.bbox_embed = nn.ModuleList([MLP(hidden_dim, hidden_dim, 4, 3) for _ in range(num_refines + 1)])
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.queries_embed_refine = nn.ModuleList([nn.Embedding(num_queries, hidden_dim) for _ in range(num_refines)])
        self.backbone = backbone
        self.aux_loss = aux_loss
        bias_value = -math.log((1 - 0.01) / 0.01)
        for ce in self.class_embed:
            ce.bias.data = torch.ones(num_classes) * bias_value
        for be in self.bbox_embed:
            nn.init.constant_(be.layers[-1].weight.data, 0)
            nn.init.constant_(be.layers[-1].bias.data, 0)

    def forward(self, samples: NestedTensor):
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        _, _, H, W = samples.decompose()[0].size()
        ps = self.backbone[0].body.patch_size
        self.transformer.H, self.transformer.W = H // ps, W // ps
        features, pos = self.backbone(samples)
        src, mask = features["x_patch"].decompose()
        assert mask is not None
        Hs, references = self.transformer(src, mask, self.query_embed.weight, pos[-1],
                                          queries_embed_refine=self.queries_embed_refine)
        out = {}
        for r in range(self.num_refines + 1):
            hs = Hs[r]                                              # [L,B,Q,d]
            ref_before = inverse_sigmoid(references[r])             # [B,Q,2]
            tmp = self.bbox_embed[r](hs)                            # all layers in one GEMM chain
            tmp = torch.cat([tmp[..., :2] + ref_before, tmp[..., 2:]], dim=-1)
            outputs_coord = tmp.sigmoid()
            outputs_class = self.class_embed[r](hs)
            o = {"pred_logits": outputs_class[-1], "pred_boxes": outputs_coord[-1], **features}
            if self.aux_loss:
                o["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b}
                                    for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
            out[r] = o
        return out


# ------------------------------------------------------------------------------------------------
class _FocalLoss(Function):
    """sum over (b,q,c) of the weighted sigmoid focal loss / num_boxes, per prediction set."""

    @staticmethod
    def forward(ctx, logits, tclass, roww, alpha, gamma, num_boxes):
        loss, grad, amax = K.focal_loss(logits, tclass, roww, alpha, gamma)
        ctx.save_for_backward(grad)
        ctx.nb = num_boxes
        ctx.mark_non_differentiable(amax)
        return loss / num_boxes, amax

    @staticmethod
    def backward(ctx, dloss, _):
        (grad,) = ctx.saved_tensors
        return grad * (dloss / ctx.nb).view(-1, 1, 1), None, None, None, None, None


class _BoxLoss(Function):
    """[L,2] = (sum w*L1, sum w*(1-GIoU)) / num_boxes over the matched pairs of each prediction set."""

    @staticmethod
    def forward(ctx, pred_flat, srow, tbox, w, lidx, L, num_boxes):
        sums, g1, g2 = K.box_loss(pred_flat, srow, tbox, w, lidx, L)
        ctx.save_for_backward(srow, lidx, g1, g2)
        ctx.nb, ctx.shape = num_boxes, pred_flat.shape
        return sums / num_boxes

    @staticmethod
    def backward(ctx, d):
        srow, lidx, g1, g2 = ctx.saved_tensors
        d = (d / ctx.nb).contiguous()
        dpred = K.box_loss_bwd(srow, lidx, g1, g2, d[:, 0].contiguous(), d[:, 1].contiguous(), ctx.shape)
        return dpred, None, None, None, None, None, None


JITTER_KERNEL = True        # tests: False runs the elementwise composition on device tensors too


def jitter_targets(targets, ratio, jitter):
    """One-to-many targets (conditional_detr.py:409-431): each GT box -> `ratio` rows: up to ratio-1
    multiplicatively jittered copies (first of 1000 candidates with IoU>0.7), original box last; labels
    and scores repeated.  Vectorised over ALL boxes of the batch and over the ratio-1 picks (the reference loops over
    images, boxes and attempts): a dozen small launches per call whatever the number of images; RNG = torch's generator
    on the device."""
    # (the reference deep-copies the target dicts, conditional_detr.py:409: ten tiny device copies per call; nothing below writes INTO a tensor - every key
    # that changes is rebound to a new tensor - so new dicts over the same tensors are equivalent)
    out = [dict(t) for t in targets]
    counts = [t["boxes"].shape[0] for t in out]
    if sum(counts) > 0:
        box = torch.cat([t["boxes"] for t in out if t["boxes"].shape[0] > 0])
        M = box.shape[0]
        scale = torch.empty((M, 1000, 4), dtype=box.dtype, device=box.device).uniform_(1 - jitter, 1 + jitter)
        if JITTER_KERNEL and box.is_cuda and box.dtype == torch.float32:
            # candidates, IoU test and the first ratio - 1 kept ones per box in ONE launch (csrc/loss.hip: jitter_pick_kernel); same picks
            # as the elementwise composition below, which stays for host tensors (tests of the host logic)
            rep = K.jitter_pick(box.contiguous(), scale, ratio).reshape(M * ratio, 4)
            return _split_jittered(out, counts, rep, ratio)
        cand = scale * box[:, None, :]
        a = box_ops.box_cxcywh_to_xyxy(cand)
        b = box_ops.box_cxcywh_to_xyxy(box)[:, None, :]
        wh = (torch.min(a[..., 2:], b[..., 2:]) - torch.max(a[..., :2], b[..., :2])).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        area = lambda z: (z[..., 2] - z[..., 0]) * (z[..., 3] - z[..., 1])
        iou = inter / (area(a) + area(b) - inter)
        keep = iou > 0.7
        rank = keep.cumsum(1)
        rep = box[:, None, :].repeat(1, ratio, 1)
        if ratio > 1:
            picks = torch.arange(1, ratio, device=box.device)
            hit = keep[:, None, :] & (rank[:, None, :] == picks[None, :, None])          # [M, ratio-1, 1000]
            has = hit.any(2)
            idx = hit.to(torch.uint8).argmax(2)                                          # the (s+1)-th kept candidate
            got = cand[torch.arange(M, device=box.device)[:, None], idx]                 # [M, ratio-1, 4]
            rep[:, :ratio - 1] = torch.where(has[..., None], got, box[:, None, :])
        rep = rep.reshape(M * ratio, 4)
        return _split_jittered(out, counts, rep, ratio)
    return _split_jittered(out, counts, None, ratio)


def _split_jittered(out, counts, rep, ratio):
    """Hand the [sum(M) * ratio, 4] jittered boxes back to the per-image target dicts; labels / scores repeat `ratio` times."""
    off = 0
    for t, m in zip(out, counts):
        if m > 0 and rep is not None:
            t["boxes"] = rep[off * ratio:(off + m) * ratio]
            off += m
    for t in out:
        t["labels"] = t["labels"].unsqueeze(1).repeat(1, ratio).reshape(-1)
        if "scores" in t:
            t["scores"] = t["scores"].unsqueeze(1).repeat(1, ratio).reshape(-1)
    return out


# tests: take the cross-rank `num_boxes` branch (device scalar + all-reduce) even in a one-rank process group
FORCE_NUM_BOXES_ALLREDUCE = False


class SetCriterion(nn.Module):
    """Hungarian matching + focal / L1 / GIoU set loss (+ image-label BCE, cardinality, class_error)."""

    refine = False

    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses, gamma, box_jitter):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha
        self.gamma = gamma
        self.eos_coef = 0.1
        self.hung_match_ratio = getattr(matcher, "match_ratio", 1)
        self.box_jitter = box_jitter
        empty_weight = torch.ones(self.num_classes)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)

    def update_hung_match_ratio(self, ratio=5):
        assert hasattr(self.matcher, "match_ratio")
        self.matcher.match_ratio = ratio
        self.hung_match_ratio = ratio

    def loss_img_label(self, outputs, targets):
        """Multi-label image classification BCE on [B,K] logits (conditional_detr.py:225-235)."""
        logits, tokens_logits = outputs["x_logits"], outputs["x_cls_logits"]
        t = torch.stack([tt["img_label"] for tt in targets]).to(logits.device).float()
        return {"img_label_logits": F.binary_cross_entropy_with_logits(logits, t),
                "img_label_logits_tokens": F.binary_cross_entropy_with_logits(tokens_logits, t)}

    def forward(self, outputs, targets, targets_cp=None):
        """outputs: one stage of the model output; targets: list of dicts (boxes cxcywh, labels,
        img_label[, scores]).  `targets_cp` (extension, default None) injects the one-to-many targets
        instead of drawing the training jitter here."""
        for l in self.losses:
            assert l in ("labels", "boxes", "cardinality", "image_label"), f"do you really want to compute {l} loss?"
        outs = [outputs] + list(outputs.get("aux_outputs", []))
        suffix = [""] + [f"_{i}" for i in range(len(outs) - 1)]
        logits = torch.stack([o["pred_logits"] for o in outs]).float()          # [L,B,Q,Kc]
        boxes = torch.stack([o["pred_boxes"] for o in outs]).float()            # [L,B,Q,4]
        L, B, Q, Kc = logits.shape
        dev = logits.device
        if targets_cp is None:
            if self.training:
                targets_cp = jitter_targets(targets, self.hung_match_ratio, self.box_jitter)
            else:
                targets_cp = [dict(t) for t in targets]              # (no tensor is written in place: see jitter_targets)
        sizes = [int(len(t["labels"])) for t in targets_cp]
        # normaliser: a host number on one GPU; across ranks it stays a device scalar (no .item() stall per step)
        if is_dist_avail_and_initialized() and (get_world_size() > 1 or FORCE_NUM_BOXES_ALLREDUCE):
            nb = host_to_device([float(sum(sizes))], torch.float, dev)             # conditional_detr.py:436-440
            torch.distributed.all_reduce(nb)
            num_boxes = torch.clamp(nb / get_world_size(), min=1)[0]
        else:
            num_boxes = max(float(sum(sizes)), 1.0)

        toff = [0]
        for s in sizes:
            toff.append(toff[-1] + s)
        flat = self.matcher.match_flat(logits, boxes, targets_cp) if hasattr(self.matcher, "match_flat") else None
        if flat is not None:                   # assignment solved on the device: nothing comes back to the host
            srow, gidx, lidx = flat
        else:
            indices = self.matcher.match_many(logits, boxes, targets_cp)
            # ---- host: flatten the assignment into (prediction row, global target index, layer) triples
            srow, gidx, lidx = [], [], []
            for l in range(L):
                for b in range(B):
                    I, J = indices[l][b]
                    srow.append((l * B + b) * Q + I)
                    gidx.append(toff[b] + J)
                    lidx.append(torch.full_like(I, l))
            trip = torch.stack([torch.cat(srow), torch.cat(gidx), torch.cat(lidx)]).to(dev)      # one H2D copy
            srow, gidx, lidx = trip[0].contiguous(), trip[1], trip[2].to(torch.int32).contiguous()
        n_match = srow.numel()
        tgt_labels = torch.cat([t["labels"] for t in targets_cp]).to(dev)
        tgt_boxes = torch.cat([t["boxes"] for t in targets_cp]).to(dev).float()
        labels_o = tgt_labels[gidx]
        scores = torch.cat([t["scores"] for t in targets_cp]).to(dev).float() if self.refine else None
        losses = {}

        amax = None
        if "labels" in self.losses or "cardinality" in self.losses:
            tclass = torch.full((L * B * Q,), Kc, dtype=torch.int32, device=dev)
            tclass[srow] = labels_o.to(torch.int32)
            roww = None
            if self.refine:                                                     # conditional_detr.py:524-529
                avg = torch.stack([t["scores"].float().mean() for t in targets_cp]).to(dev)
                roww = avg.view(1, B, 1).expand(L, B, Q).reshape(-1).clone()
                roww[srow] = (scores[gidx] * 3).clamp(max=1.0)
            loss_ce, amax = _FocalLoss.apply(logits.view(L, B * Q, Kc), tclass.view(L, B * Q), None if roww is None else
                                             roww.view(L, B * Q), float(self.focal_alpha), float(self.gamma), 1.0)
            loss_ce = loss_ce / num_boxes
        if "boxes" in self.losses:
            w = scores[gidx].contiguous() if self.refine else None
            bl = _BoxLoss.apply(boxes.view(-1, 4), srow, tgt_boxes[gidx].contiguous(), w, lidx, L, 1.0) / num_boxes
        card_err = None
        if "cardinality" in self.losses:                                        # conditional_detr.py:286-298 (logging), all layers at once
            sizes_t = host_to_device(sizes, torch.float, dev)
            card = (amax.view(L, B, Q) != Kc - 1).sum(2).float()                # [L, B] predicted non-background counts
            card_err = (card - sizes_t[None, :]).abs().mean(1)                  # = F.l1_loss(card[l], sizes) per layer
        # one unbind per loss vector (its backward is ONE stack; indexing element by element costs a zero fill, a copy and an
        # accumulation per key in the backward - ~100 tiny launches per criterion call)
        ce_l = loss_ce.unbind(0) if "labels" in self.losses else None
        bl_f = bl.reshape(-1).unbind(0) if "boxes" in self.losses else None
        card_l = card_err.unbind(0) if card_err is not None else None
        for l in range(L):
            sfx = suffix[l]
            if "labels" in self.losses:
                losses["loss_ce" + sfx] = ce_l[l]
                if l == 0:                                                      # top-1 error on matched rows (logging)
                    m0 = (lidx == 0).float()          # sync-free masked mean (0 when nothing is matched)
                    hit = (amax.view(-1)[srow].long() == labels_o).float()
                    acc = (hit * m0).sum() / m0.sum().clamp(min=1) * 100
                    losses["class_error"] = 100 - acc
            if "boxes" in self.losses:
                losses["loss_bbox" + sfx] = bl_f[2 * l]
                losses["loss_giou" + sfx] = bl_f[2 * l + 1]
            if card_err is not None:
                losses["cardinality_error" + sfx] = card_l[l]
            if l == 0 and "image_label" in self.losses:
                losses.update(self.loss_img_label(outputs, targets_cp))
        return losses


class SetCriterionRefine(SetCriterion):
    """Stage >= 1 criterion: focal weights and box losses scaled by pseudo-label scores (504-560)."""

    refine = True


# ------------------------------------------------------------------------------------------------
class PostProcess(nn.Module):
    """Top-k over Q*Kc scores -> absolute xyxy boxes (conditional_detr.py:592-623)."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, keep_queries=100):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes) and target_sizes.shape[1] == 2
        prob = out_logits.sigmoid()
        topk_values, topk_indexes = torch.topk(prob.view(out_logits.shape[0], -1), keep_queries, dim=1)
        topk_boxes = torch.div(topk_indexes, out_logits.shape[2], rounding_mode="floor")
        labels = topk_indexes % out_logits.shape[2]
        boxes = box_ops.box_cxcywh_to_xyxy(out_bbox).clamp(min=0)
        boxes = torch.gather(boxes, 1, topk_boxes.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.unbind(1)
        boxes = boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(topk_values, labels, boxes)]


class PostProcessRefine(nn.Module):
    """Stage-k detections -> stage-(k+1) pseudo targets (conditional_detr.py:641-677): for every class
    present in an image's labels (ascending id) the best query's score and NORMALISED cxcywh box."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, targets=None):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes) and target_sizes.shape[1] == 2
        prob = out_logits.sigmoid()
        top_values, top_indexes = torch.max(prob, dim=1)                        # [B,Kc]
        res = []
        for b, t in enumerate(targets):
            # `labels_unique` (sorted unique class ids < Kc, e.g. attached by the data loader on the host) avoids the
            # device synchronisation of torch.unique / boolean indexing on device tensors
            lab = t.get("labels_unique")
            if lab is None:
                lab = torch.unique(t["labels"])                                 # sorted ascending
                lab = lab[lab < out_logits.shape[2]]
            lab = lab.to(out_logits.device)
            res.append({"scores": top_values[b, lab], "labels": lab, "boxes": out_bbox[b, top_indexes[b, lab]]})
        return res


class PostProcessRefineMulti(nn.Module):
    """Multi-box variant of PostProcessRefine (reference conditional_detr.py:680-715; built by no script, kept for API
    parity): for every class present in an image's labels ALL queries whose probability reaches half of the best
    query's, with their normalised cxcywh boxes; classes ascending, queries ascending within a class."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, targets=None):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes) and target_sizes.shape[1] == 2
        prob = out_logits.sigmoid()
        keep = prob >= 0.5 * prob.max(dim=1, keepdim=True)[0]                   # [B,Q,Kc]
        res = []
        for b, t in enumerate(targets):
            lab = torch.unique(t["labels"]).to(out_logits.device)
            lab = lab[lab < out_logits.shape[2]]
            cq = keep[b][:, lab].t().nonzero(as_tuple=False)                    # (class position, query), class-major
            cls, q = lab[cq[:, 0]], cq[:, 1]
            res.append({"scores": prob[b, q, cls], "labels": cls, "boxes": out_bbox[b, q]})
        return res


def build(args):
    num_classes = 21 if args.dataset_file != "coco" else 91
    if args.dataset_file == "coco_panoptic":
        num_classes = 250
    if getattr(args, "masks", False):
        raise NotImplementedError("segmentation heads (--masks) are outside the SPE hot path")
    device = torch.device(args.device)
    backbone = build_backbone(args)
    transformer = build_transformer(args)
    model = ConditionalDETR_Refine(backbone, transformer, num_classes=num_classes, num_queries=args.num_queries,
                                   aux_loss=args.aux_loss, num_refines=args.num_refines)
    matcher = build_matcher(args)
    matcher_refine = build_matcher(args)
    weight_dict = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef,
                   "img_label_logits": args.img_label_loss_coef, "img_label_logits_tokens": args.img_label_tokens_loss_coef}
    weight_dict["loss_giou"] = args.giou_loss_coef
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    losses = ["labels", "boxes", "cardinality", "image_label"]
    losses_refine = ["labels", "boxes", "cardinality"]
    criterion = SetCriterion(num_classes, matcher=matcher, weight_dict=weight_dict, focal_alpha=args.focal_alpha,
                             losses=losses, gamma=args.focal_gamma, box_jitter=args.box_jitter)
    criterion.to(device)
    postprocessors = {"bbox": PostProcess()}
    refine_postprocessors = {"bbox": PostProcessRefine()}
    criterion_refine = SetCriterionRefine(num_classes, matcher=matcher_refine, weight_dict=weight_dict,
                                          focal_alpha=args.focal_alpha, losses=losses_refine, gamma=args.focal_gamma,
                                          box_jitter=args.box_jitter)
    criterion_refine.to(device)
    return model, criterion, criterion_refine, postprocessors, refine_postprocessors
