"""Generate the golden vectors under tests/golden/ by running the REFERENCE (imported from
/root/reference through tools/ref_harness.py) on seeded inputs.  Run in the build container:

    python tools/gen_golden.py

The fixtures hold only data (weights, inputs, expected outputs); no reference source travels.
Golden-sensitivity traps are defeated by randomising LayerScale gammas,
talking-heads mixers, bbox_embed[-1] and class_embed before capture, and by capturing the
jittered `targets_cp` the train-mode criterion hands to its matcher.
"""
import contextlib
import copy
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def randomise(model, g):
    """O(1) values where the reference's init would make a bug invisible."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma_1") or n.endswith("gamma_2"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3 + 1.0)
            elif "proj_l.weight" in n or "proj_w.weight" in n:
                p.copy_(torch.eye(p.shape[0]) + 0.3 * torch.randn(p.shape, generator=g))
            elif "proj_l.bias" in n or "proj_w.bias" in n:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.startswith("bbox_embed") and ".layers.2." in n:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif n.startswith("class_embed"):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif n.endswith("pos_embed") or n.endswith("cls_token"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias") and p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))


def make_targets(g, K, sizes):
    ts = []
    for n in sizes:
        labels = torch.randint(1, K + 1, (n,), generator=g)
        c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(n, 2, generator=g) * 0.35 + 0.05
        il = torch.zeros(K, dtype=torch.int64)
        il[labels - 1] = 1
        ts.append({"boxes": torch.cat([c, wh], 1), "labels": labels, "img_label": il,
                   "orig_size": torch.tensor([64, 96])})
    return ts


def detach_out(out):
    r = {}
    for k, v in out.items():
        if k == "aux_outputs":
            r[k] = [{kk: vv.detach().clone() for kk, vv in a.items()} for a in v]
        elif k == "x_patch":
            r[k] = (v.tensors.detach().clone(), v.mask.clone())
        else:
            r[k] = v.detach().clone()
    return r


def e2e(name, backbone, layer_to_det, enc_layers, seed):
    from models import build_model
    import util.misc as um
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    args = rh.default_args(backbone=backbone, layer_to_det=layer_to_det, enc_layers=enc_layers, dec_layers=2,
                           num_queries=7, dim_feedforward=64, nheads=4, dropout=0.0, dataset_file="voc")
    with quiet():
        model, crit, crit_r, pp, rpp = build_model(args)
    randomise(model, g)
    imgs = [torch.randn(3, 64, 96, generator=g), torch.randn(3, 48, 80, generator=g)]
    samples = um.nested_tensor_from_tensor_list(imgs)
    targets = make_targets(g, 20, [3, 2])

    # ---- eval-mode forward + both criteria (deterministic, 1-to-1 matching)
    model.eval(); crit.eval(); crit_r.eval()
    with torch.no_grad():
        out = model(samples)
        l0 = crit(out[0], targets)
        orig = torch.stack([t["orig_size"] for t in targets])
        pr = rpp["bbox"](out[0], orig, targets)
        pseudo = []
        for t, r in zip(targets, pr):
            p = copy.deepcopy(t)
            p.update({"labels": r["labels"].clone(), "boxes": r["boxes"].clone(), "scores": r["scores"].clone()})
            pseudo.append(p)
        l1 = crit_r(out[1], pseudo)
        post = pp["bbox"](out[0], orig, 10)
    eval_blob = {"out0": detach_out(out[0]), "out1": detach_out(out[1]), "loss0": {k: v.clone() for k, v in l0.items()},
                 "loss1": {k: v.clone() for k, v in l1.items()}, "pseudo": pseudo, "postprocess": post}

    # ---- train-mode step: capture what the criterion hands to its matcher (jittered targets_cp) and
    # the assignment it gets back, then total loss -> backward -> all parameter grads
    model.train(); crit.train(); crit_r.train()
    captured = {}

    def wrap(c, tag):
        inner = c.matcher.forward
        calls = []

        def fwd(outputs, tg):
            res = inner(outputs, tg)
            calls.append((copy.deepcopy(tg), [(i.clone(), j.clone()) for i, j in res]))
            return res
        c.matcher.forward = fwd
        captured[tag] = calls
    wrap(crit, "crit")
    wrap(crit_r, "crit_r")
    model.zero_grad()
    out = model(samples)
    with torch.no_grad():
        pr = rpp["bbox"](out[0], orig, targets)
        pseudo_t = []
        for t, r in zip(targets, pr):
            p = copy.deepcopy(t)
            p.update({"labels": r["labels"].clone(), "boxes": r["boxes"].clone(), "scores": r["scores"].clone()})
            pseudo_t.append(p)
    lt0 = crit(out[0], targets)
    lt1 = crit_r(out[1], pseudo_t)
    wd = crit.weight_dict
    total = sum(lt0[k] * wd[k] for k in lt0 if k in wd) + sum(lt1[k] * wd[k] for k in lt1 if k in wd)
    total.backward()
    grads = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    train_blob = {"loss0": {k: v.detach().clone() for k, v in lt0.items()},
                  "loss1": {k: v.detach().clone() for k, v in lt1.items()},
                  "total": total.detach().clone(), "grads": grads,
                  "targets_cp0": captured["crit"][0][0], "targets_cp1": captured["crit_r"][0][0],
                  "indices0": [c[1] for c in captured["crit"]], "indices1": [c[1] for c in captured["crit_r"]],
                  "pseudo": pseudo_t, "weight_dict": dict(wd)}
    blob = {"args": vars(args), "state_dict": {k: v.clone() for k, v in model.state_dict().items()},
            "images": imgs, "tensors": samples.tensors.clone(), "mask": samples.mask.clone(), "targets": targets,
            "eval": eval_blob, "train": train_blob}
    torch.save(blob, os.path.join(OUT, name + ".pt"))
    print(name, "params", sum(v.numel() for v in blob["state_dict"].values()),
          "total loss", float(total), "bytes", os.path.getsize(os.path.join(OUT, name + ".pt")))


def ops_golden(seed=7):
    """Per-op vectors from the reference modules/functions (small shapes, fp32)."""
    from functools import partial
    import models.cait as rc
    import models.transformer as rt
    import models.attention as ra
    import models.matcher as rm
    import models.position_encoding as rp
    import models.conditional_detr as rd
    import util.box_ops as rb
    import util.misc as um
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    G = {}
    rn = lambda *s: torch.randn(*s, generator=g)

    # talking-heads attention + LayerScale block (cait.py:344-416)
    for N in (12, 35):
        blk = rc.LayerScale_Block(dim=32, num_heads=4, mlp_ratio=4, qkv_bias=True,
                                  norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=1e-5).eval()
        randomise(blk, g)
        x = rn(2, N, 32)
        with torch.no_grad():
            G[f"thattn_N{N}"] = {"sd": {k: v.clone() for k, v in blk.state_dict().items()}, "x": x,
                                 "attn_out": blk.attn(blk.norm1(x)), "block_out": blk(x)}
    # class-attention block with map (cait.py:91-139, 311-328)
    cab = rc.LayerScale_Block_CA_MultiClass(dim=32, num_heads=4, mlp_ratio=4, qkv_bias=True,
                                            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=1e-5,
                                            num_classes=5).eval()
    randomise(cab, g)
    x, cls = rn(2, 12, 32), rn(2, 6, 32)
    with torch.no_grad():
        o = cab(x, cls)
        G["ca_block"] = {"sd": {k: v.clone() for k, v in cab.state_dict().items()}, "x": x, "cls": cls, "out": o,
                         "map": cab.attn.get_attention_map().clone()}
    # bicubic pos-embed resize (cait.py:598-613)
    pe = rn(1, 24 * 24, 8)
    for hw in ((14, 14), (50, 83), (5, 7)):
        p4 = pe.transpose(1, 2).view(1, 8, 24, 24)
        G[f"posembed_{hw[0]}x{hw[1]}"] = {"pe": pe, "out": torch.nn.functional.interpolate(
            p4, size=hw, mode="bicubic", align_corners=False).flatten(2).transpose(1, 2)}
    # sine position embedding with a padded mask (position_encoding.py:37-57)
    mask = torch.zeros(2, 6, 9, dtype=torch.bool)
    mask[1, 4:, :] = True
    mask[1, :, 7:] = True
    pes = rp.PositionEmbeddingSine(16, normalize=True)
    G["pos_sine"] = {"mask": mask, "out": pes(um.NestedTensor(torch.zeros(2, 32, 6, 9), mask))}
    # query sine embedding incl. the /128 quirk (transformer.py:35-49)
    for d in (32, 192):
        pos = torch.rand(5, 2, 2, generator=g)
        G[f"sineembed_d{d}"] = {"pos": pos, "out": rt.gen_sineembed_for_position(pos, d)}
    # custom MHA with padding mask, q/k dim != v dim (attention.py:55-383)
    mha = ra.MultiheadAttention(64, 4, dropout=0.0, vdim=32).eval()
    randomise(mha, g)
    q, k, v = rn(5, 2, 64), rn(11, 2, 64), rn(11, 2, 32)
    kpm = torch.zeros(2, 11, dtype=torch.bool)
    kpm[1, 8:] = True
    with torch.no_grad():
        o, wts = mha(q, k, v, key_padding_mask=kpm)
    G["mha"] = {"sd": {k_: v_.clone() for k_, v_ in mha.state_dict().items()}, "q": q, "k": k, "v": v, "kpm": kpm,
                "out": o, "weights": wts}
    # encoder layer / decoder layer (transformer.py:253-427)
    enc = rt.TransformerEncoderLayer(32, 4, 64, 0.0).eval()
    randomise(enc, g)
    src, pos = rn(11, 2, 32), rn(11, 2, 32)
    with torch.no_grad():
        G["enc_layer"] = {"sd": {k_: v_.clone() for k_, v_ in enc.state_dict().items()}, "src": src, "pos": pos,
                          "kpm": kpm, "out": enc(src, src_key_padding_mask=kpm, pos=pos)}
    dec = rt.TransformerDecoderLayer(32, 4, 64, 0.0).eval()
    randomise(dec, g)
    tgt, qpos, qsine = rn(5, 2, 32), rn(5, 2, 32), rn(5, 2, 32)
    with torch.no_grad():
        o1 = dec(tgt, src, memory_key_padding_mask=kpm, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=True)
        o2 = dec(tgt, src, memory_key_padding_mask=kpm, pos=pos, query_pos=qpos, query_sine_embed=qsine, is_first=False)
    G["dec_layer"] = {"sd": {k_: v_.clone() for k_, v_ in dec.state_dict().items()}, "tgt": tgt, "memory": src,
                      "pos": pos, "kpm": kpm, "query_pos": qpos, "query_sine": qsine, "out_first": o1, "out_other": o2}
    # GIoU incl. identical and disjoint pairs (box_ops.py:49-74)
    a = torch.tensor([[0.1, 0.1, 0.4, 0.5], [0.5, 0.5, 0.9, 0.9], [0.2, 0.2, 0.3, 0.3]])
    b = torch.tensor([[0.1, 0.1, 0.4, 0.5], [0.0, 0.6, 0.2, 0.8], [0.25, 0.15, 0.6, 0.45], [0.7, 0.1, 0.95, 0.3]])
    G["giou"] = {"a": a, "b": b, "out": rb.generalized_box_iou(a, b), "iou": rb.box_iou(a, b)[0]}
    # matcher: cost + assignment, incl. M > Q and an image without targets (matcher.py:41-87)
    m = rm.HungarianMatcher(2, 5, 2, 5)
    for tag, Q, sizes in (("small", 6, [3, 0]), ("many", 4, [9, 2])):
        outs = {"pred_logits": rn(2, Q, 21) * 2, "pred_boxes": torch.cat([torch.rand(2, Q, 2, generator=g) * 0.6 + 0.2,
                                                                        torch.rand(2, Q, 2, generator=g) * 0.3 + 0.05], -1)}
        tg = make_targets(g, 20, sizes)
        # the cost matrices the reference hands to SciPy (matcher.py:82-86), captured at the call (no RNG is consumed)
        costs, lsa = [], rm.linear_sum_assignment
        def record(c):
            costs.append(torch.as_tensor(c).clone())
            return lsa(c)
        rm.linear_sum_assignment = record
        try:
            idx = m(outs, tg)
        finally:
            rm.linear_sum_assignment = lsa
        G[f"matcher_{tag}"] = {"outputs": outs, "targets": tg, "indices": idx, "cost": costs}
    # weighted focal, gamma 0.5 and 2 (conditional_detr.py:468-494)
    crit = rd.SetCriterion(21, m, {}, 0.25, ["labels"], 2.0, 0.1)
    x = rn(2, 6, 21) * 3
    t = (torch.rand(2, 6, 21, generator=g) > 0.9).float()
    w = torch.rand(2, 6, 21, generator=g)
    for gam in (0.5, 2.0):
        G[f"focal_g{gam}"] = {"x": x, "t": t, "w": w, "out": crit.weighted_sigmoid_focal_loss(x, t, 3.0, w, 0.25, gam)}
    torch.save(G, os.path.join(OUT, "ops.pt"))
    print("ops", list(G.keys()), "bytes", os.path.getsize(os.path.join(OUT, "ops.pt")))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    rh.install_shims()
    rh.register_tiny_backbones()
    ops_golden()
    e2e("e2e_single", "TSCAM_cait_tiny", layer_to_det=2, enc_layers=1, seed=101)
    e2e("e2e_two_branch", "TSCAM_cait_tiny_Two_Branch", layer_to_det=3, enc_layers=0, seed=202)
