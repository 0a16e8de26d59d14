"""Multi-step TRAINING trajectory of the REFERENCE (engine.py:88-165 + main.py:177-191) as golden data.

    python tools/gen_traj_golden.py [case ...]          # build container only: imports /root/reference

For every case (tests/cfg_cases.py): the product's seeded state dict is loaded strict=True into the reference's
ConditionalDETR_Refine, then STEPS iterations of the reference's own loop run there:

    model.train(); criterion.train(); criterion_refine.train()         # one-to-many jitter ON (5x, box_jitter 0.1)
    outputs = model(samples) ; loss_dict = criterion(outputs[0], targets) ; criterion_refine(outputs[1], pseudo)
    losses = sum(weight * loss) ; optimizer.zero_grad() ; losses.backward()
    clip_grad_norm_(model.parameters(), 0.1) ; optimizer.step()       # AdamW, 3 LR groups: 1e-4 / lr_backbone 1e-5 / lr_cls_head 5e-5

(the CAM pseudo-box step of engine.py:116 needs cv2 and is bypassed exactly as bench.py bypasses it: the case's targets are
fed directly; stage-1 targets come from PostProcessRefine as in engine.py:295-308).  The jitter draws from torch's global RNG,
so what the criteria hand to their matchers - `targets_cp` of both stages, per step - is captured and stored: the product
replays the SAME one-to-many targets.  Stored per step: total loss, every loss key, the pre-clip gradient norm, targets_cp0 /
targets_cp1 and the stage-1 pseudo labels; after the last step: (norm, 64 strided samples) of every parameter's UPDATE p_K - p_0.
Data only -> tests/golden/traj_<case>.pt.
"""
import contextlib
import copy
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cfg_cases as cc  # noqa: E402
import gen_config_golden as gcg  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
STEPS = 5
LR, LR_BACKBONE, LR_CLS_HEAD, WD, CLIP = 1e-4, 1e-5, 5e-5, 1e-4, 0.1


def param_groups(model):
    """reference main.py:177-187"""
    named = list(model.named_parameters())
    return [{"params": [p for n, p in named if "backbone" not in n and p.requires_grad]},
            {"params": [p for n, p in named if "backbone" in n and p.requires_grad and "blocks_token_only" not in n], "lr": LR_BACKBONE},
            {"params": [p for n, p in named if "backbone" in n and p.requires_grad and "blocks_token_only" in n], "lr": LR_CLS_HEAD}]


def run_case(name, nsteps=STEPS, tag="traj"):
    from models import build_model as ref_build
    import util.misc as um
    args, (pmodel, *_), tensors, mask, targets = cc.build_case(name)
    sd = {k: v.detach().clone() for k, v in pmodel.state_dict().items()}
    del pmodel
    with contextlib.redirect_stdout(io.StringIO()):
        model, crit, crit_r, pp, rpp = ref_build(copy.deepcopy(args))
    model.load_state_dict(sd, strict=True)
    model.train(); crit.train(); crit_r.train()
    captured = {}

    def wrap(c, tag):
        inner = c.matcher.forward
        calls = []

        def fwd(outputs, tg):
            res = inner(outputs, tg)
            calls.append(copy.deepcopy(tg))
            return res
        c.matcher.forward = fwd
        captured[tag] = calls
    wrap(crit, "crit")
    wrap(crit_r, "crit_r")
    opt = torch.optim.AdamW(param_groups(model), lr=LR, weight_decay=WD)
    p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    wd = crit.weight_dict
    orig = torch.stack([t["orig_size"] for t in targets])
    torch.manual_seed(cc.ALL_CASES[name]["seed"] + 77)        # the jitter stream (any seed: the draws are stored)
    steps = []
    for s in range(nsteps):
        captured["crit"].clear(); captured["crit_r"].clear()
        out = model(um.NestedTensor(tensors, mask))
        with torch.no_grad():
            pr = rpp["bbox"](out[0], orig, targets)
            pseudo = []
            for t, r in zip(targets, pr):
                p = copy.deepcopy(t)
                p.update({"labels": r["labels"].clone(), "boxes": r["boxes"].clone(), "scores": r["scores"].clone()})
                pseudo.append(p)
        l0 = crit(out[0], targets)
        l1 = crit_r(out[1], pseudo)
        total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
        opt.zero_grad()
        total.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP)
        opt.step()
        # the first matcher call of a criterion is the last decoder layer's (conditional_detr.py:433); every call of one
        # criterion forward receives the same targets_cp
        steps.append({"total": float(total.detach()), "grad_norm": float(gnorm),
                      "loss0": {k: float(v.detach()) for k, v in l0.items()}, "loss1": {k: float(v.detach()) for k, v in l1.items()},
                      "targets_cp0": captured["crit"][0], "targets_cp1": captured["crit_r"][0], "pseudo": pseudo})
        print(name, "step", s, "total", steps[-1]["total"], "grad norm", steps[-1]["grad_norm"], flush=True)
    upd = {n: cc.sample(p.detach() - p0[n]) for n, p in model.named_parameters()}
    blob = {"case": name, "steps": steps, "updates": upd, "weight_dict": dict(wd),
            "hyper": {"lr": LR, "lr_backbone": LR_BACKBONE, "lr_cls_head": LR_CLS_HEAD, "weight_decay": WD, "clip_max_norm": CLIP, "steps": nsteps},
            "sd_checksum": float(sum(v.double().abs().sum() for v in sd.values() if v.is_floating_point()))}
    path = os.path.join(OUT, f"{tag}_{name}.pt")
    torch.save(blob, path)
    print(name, "bytes", os.path.getsize(path))


def train_criterion_record(name):
    """TRAIN-mode criteria (5x one-to-many jitter, conditional_detr.py:410-431) of the reference on a FULL-DEPTH case: the
    forward runs without autograd (the backward of these cases is pinned by cfg_<case>.pt with eval-mode criteria), then both
    criteria in train mode with what they hand to their matchers captured -> tests/golden/cfg_<case>_train.pt."""
    from models import build_model as ref_build
    import util.misc as um
    args, (pmodel, *_), tensors, mask, targets = cc.build_case(name)
    sd = {k: v.detach().clone() for k, v in pmodel.state_dict().items()}
    del pmodel
    with contextlib.redirect_stdout(io.StringIO()):
        model, crit, crit_r, pp, rpp = ref_build(copy.deepcopy(args))
    model.load_state_dict(sd, strict=True)
    model.train(); crit.train(); crit_r.train()
    for blk in model.backbone[0].body.blocks:                # drop the per-block attention_map clone (cait.py:392): harness-side memory only
        blk.attn.register_forward_hook(lambda m, i, o: setattr(m, "attention_map", None))
    captured = {}
    for c, tag in ((crit, "crit"), (crit_r, "crit_r")):
        calls = []

        def fwd(outputs, tg, _inner=c.matcher.forward, _calls=calls):
            _calls.append(copy.deepcopy(tg))
            return _inner(outputs, tg)
        c.matcher.forward = fwd
        captured[tag] = calls
    torch.manual_seed(cc.ALL_CASES[name]["seed"] + 78)
    orig = torch.stack([t["orig_size"] for t in targets])
    with torch.no_grad():
        out = model(um.NestedTensor(tensors, mask))
        pr = rpp["bbox"](out[0], orig, targets)
        pseudo = []
        for t, r in zip(targets, pr):
            p = copy.deepcopy(t)
            p.update({"labels": r["labels"].clone(), "boxes": r["boxes"].clone(), "scores": r["scores"].clone()})
            pseudo.append(p)
        l0 = crit(out[0], targets)
        l1 = crit_r(out[1], pseudo)
    wd = crit.weight_dict
    total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
    blob = {"case": name, "loss0": {k: v.detach().clone() for k, v in l0.items()}, "loss1": {k: v.detach().clone() for k, v in l1.items()},
            "total": total.detach().clone(), "targets_cp0": captured["crit"][0], "targets_cp1": captured["crit_r"][0], "pseudo": pseudo,
            "weight_dict": dict(wd), "sd_checksum": float(sum(v.double().abs().sum() for v in sd.values() if v.is_floating_point()))}
    path = os.path.join(OUT, f"cfg_{name}_train.pt")
    torch.save(blob, path)
    print(name, "train-mode criteria: total", float(total), "targets per image", [len(t["labels"]) for t in blob["targets_cp0"]], "bytes", os.path.getsize(path))


if __name__ == "__main__":
    gcg.register_reference_backbones()
    for n in (sys.argv[1:] or ["cfg1", "cfg2_depth2"]):
        if n.endswith(":train"):
            train_criterion_record(n[:-6])
        elif ":" in n:                                     # <case>:<steps> -> tests/golden/traj<steps>_<case>.pt (round 5: 30 steps)
            c, k = n.split(":")
            run_case(c, int(k), f"traj{int(k)}")
        else:
            run_case(n)
