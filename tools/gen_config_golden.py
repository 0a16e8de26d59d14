"""Golden vectors of the REFERENCE at BASELINE.json's own dimensions (tests/cfg_cases.py has the case table).

    python tools/gen_config_golden.py [case ...]        # build container only: imports /root/reference

For every case: the product's seeded + randomised state dict (rebuilt from the seed on any box) is loaded - strict=True -
into the reference's ConditionalDETR_Refine; one training iteration's forward, SetCriterion, PostProcessRefine pseudo
labels, SetCriterionRefine, weighted total and backward run in the reference (all drop rates 0, criteria in eval mode =
no random jitter); results go to tests/golden/cfg_<case>.pt as data only.  The same run also checks the oracle against
the reference and stores the measured errors (the cfg1 parity report SURVEY.md section 8(c) asks for) in
tests/golden/cfg_report.json.
"""
import contextlib
import copy
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cfg_cases as cc  # noqa: E402
import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FULL_MAX = 20000          # tensors up to this many elements are stored whole


def register_reference_backbones():
    rh.install_shims()
    from functools import partial
    import models.cait as rc
    from timm.models.registry import register_model
    from torch import nn
    for name, c in cc.ALL_CASES.items():
        if hasattr(rc, c["backbone"]):
            continue                                     # the reference's own factory (TSCAM_cait_XXS24)

        def fac(pretrained=False, _c=c, **kwargs):
            m = rc.TSCAM_cait(img_size=384, patch_size=16, embed_dim=_c["width"], depth=_c["depth"], num_heads=_c["heads"],
                              mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                              init_scale=_c["init_scale"], depth_token_only=2, **kwargs)
            return m, _c["width"]
        fac.__name__ = c["backbone"]
        register_model(fac)


def checkpoint_blocks(model):
    """Full-depth cases: run every backbone block of the REFERENCE under torch.utils.checkpoint and drop the per-block
    `attention_map` clone (models/cait.py:392; only blocks_token_only[0]'s map is read, :658) - harness-side memory
    management, the arithmetic and its order are the reference's."""
    from torch.utils.checkpoint import checkpoint
    body = model.backbone[0].body
    for blk in body.blocks:
        blk.forward = (lambda x, _f=blk.forward: checkpoint(_f, x, use_reentrant=False))
        blk.attn.register_forward_hook(lambda m, i, o: setattr(m, "attention_map", None))


def keep(t):
    t = t.detach()
    return t.clone() if t.numel() <= FULL_MAX else cc.sample(t)


def out_record(o):
    r = {}
    for k, v in o.items():
        if k == "aux_outputs":
            r[k] = [{kk: keep(vv) for kk, vv in a.items()} for a in v]
        elif k == "x_patch":
            r[k] = (cc.sample(v.tensors), v.mask.clone())
        else:
            r[k] = keep(v)
    return r


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run_case(name):
    from models import build_model as ref_build
    import util.misc as um
    from oracle import spe_oracle as O
    c = cc.ALL_CASES[name]
    full = name in cc.FULL_CASES
    args, (pmodel, *_), tensors, mask, targets = cc.build_case(name)
    sd = {k: v.detach().clone() for k, v in pmodel.state_dict().items()}
    del pmodel
    with contextlib.redirect_stdout(io.StringIO()):
        model, crit, crit_r, pp, rpp = ref_build(copy.deepcopy(args))
    model.load_state_dict(sd, strict=True)                   # identical keys and shapes: the boundary contract
    model.train(); crit.eval(); crit_r.eval()                # every drop rate is 0; eval criteria = no jitter
    if full:
        checkpoint_blocks(model)
    t0 = time.time()
    out = model(um.NestedTensor(tensors, mask))
    l0 = crit(out[0], targets)
    orig = torch.stack([t["orig_size"] for t in targets])
    with torch.no_grad():
        pr = rpp["bbox"](out[0], orig, targets)
        pseudo = []
        for t, r in zip(targets, pr):
            p = copy.deepcopy(t)
            p.update({"labels": r["labels"].clone(), "boxes": r["boxes"].clone(), "scores": r["scores"].clone()})
            pseudo.append(p)
    # conditioning of the case: the pseudo labels take an argmax over the queries - the runner-up must be clearly behind
    # (random decoders separate their queries weakly): the relative top-2 gap of every pseudo label is stored, and the
    # tests compare a pseudo box only where the gap is far above the precision under test
    prob = out[0]["pred_logits"].detach().sigmoid()
    margin, margins = 1.0, []
    for b, t in enumerate(targets):
        mb = []
        for c_ in torch.unique(t["labels"]).tolist():
            top = prob[b, :, c_].topk(2).values
            mb.append(float((top[0] - top[1]) / top[0]))
        margins.append(mb)
        margin = min(margin, min(mb))
    l1 = crit_r(out[1], pseudo)
    wd = crit.weight_dict
    total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
    total.backward()
    t_ref = time.time() - t0
    grads = {n: (cc.sample(p.grad) if p.grad is not None else None) for n, p in model.named_parameters()}
    blob = {"case": name, "dims": c, "out0": out_record(out[0]), "out1": out_record(out[1]),
            "loss0": {k: v.detach().clone() for k, v in l0.items()}, "loss1": {k: v.detach().clone() for k, v in l1.items()},
            "pseudo": pseudo, "pseudo_margins": margins, "total": total.detach().clone(), "grads": grads, "weight_dict": dict(wd),
            "sd_checksum": float(sum(v.double().abs().sum() for v in sd.values() if v.is_floating_point()))}
    path = os.path.join(OUT, f"cfg_{name}.pt")
    torch.save(blob, path)
    if full:                                                 # the oracle's eager autograd does not fit here at full depth
        rep = {"reference_seconds": round(t_ref, 2), "pseudo_label_top2_margin": margin, "fixture_bytes": os.path.getsize(path),
               "total_loss_reference": float(total.detach()), "blocks_checkpointed": True}
        print(name, json.dumps(rep))
        return rep

    # ---- oracle against the reference on the same case (the in-container parity report)
    t0 = time.time()
    sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    otot, oout, ol0, ol1 = O.total_loss(sdg, cc.oracle_cfg(name), tensors, mask, targets)
    otot.backward()
    t_or = time.time() - t0
    rep = {"pseudo_label_top2_margin": margin, "reference_seconds": round(t_ref, 2), "oracle_seconds": round(t_or, 2), "fixture_bytes": os.path.getsize(path),
           "total_loss_reference": float(total.detach()), "total_loss_oracle": float(otot.detach()),
           "total_loss_rel_err": abs(float(otot.detach()) - float(total.detach())) / abs(float(total.detach())), "outputs_rel_err": {}, "losses_abs_err": {}}
    for st in (0, 1):
        for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
            rep["outputs_rel_err"][f"{st}.{k}"] = rel(oout[st][k], out[st][k])
    rep["outputs_rel_err"]["x_patch"] = rel(oout[0]["x_patch"][0], out[0]["x_patch"].tensors)
    for tag, ol, rl in (("0", ol0, l0), ("1", ol1, l1)):
        for k, v in rl.items():
            rep["losses_abs_err"][f"{tag}.{k}"] = abs(float(ol[k]) - float(v))
    worst = ("", 0.0)
    n = 0
    for k, p in model.named_parameters():
        if p.grad is None or float(p.grad.abs().max()) < 1e-7:
            continue
        e = rel(sdg[k].grad, p.grad)
        n += 1
        if e > worst[1]:
            worst = (k, e)
    rep["grads_compared"] = n
    rep["worst_grad_rel_err"] = {"param": worst[0], "err": worst[1]}
    rep["slices"] = {"pred_logits": out[0]["pred_logits"].detach().flatten()[:8].tolist(),
                     "pred_boxes": out[0]["pred_boxes"].detach().flatten()[:8].tolist()}
    print(name, json.dumps(rep)[:600])
    return rep


def main():
    register_reference_backbones()
    names = sys.argv[1:] or list(cc.CASES)                   # the full-depth cases (cc.FULL_CASES) are named explicitly
    rpath = os.path.join(OUT, "cfg_report.json")
    report = json.load(open(rpath)) if os.path.exists(rpath) else {}
    for n in names:
        report[n] = run_case(n)
    with open(rpath, "w") as fh:
        json.dump(report, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
