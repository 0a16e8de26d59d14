import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfg_cases as cc
from spe_amd import kernels as K
from spe_amd.util.misc import NestedTensor
from oracle import spe_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_enc3_small"
dev = torch.device("cuda:0")
blob = torch.load(os.path.join(ROOT, "tests", "golden", f"cfg_{name}.pt"), weights_only=False)
args, (model, crit, crit_r, pp, rpp), tensors, mask, targets = cc.build_case(name)
K.set_precision("bf16x3")
model.to(dev).train(); crit.to(dev).eval(); crit_r.to(dev).eval()
tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
out = model(NestedTensor(tensors.to(dev), mask.to(dev)))
orig = torch.stack([t["orig_size"] for t in tg])
with torch.no_grad():
    pr = rpp["bbox"](out[0], orig, tg)
for p, r in zip(pr, blob["pseudo"]):
    print("labels eq", torch.equal(p["labels"].cpu(), r["labels"]), "scores", (p["scores"].cpu() - r["scores"]).abs().max().item(), "boxes", (p["boxes"].cpu() - r["boxes"]).abs().max().item())
    print("  labels", r["labels"].tolist(), "scores", [round(x, 4) for x in r["scores"].tolist()])
pseudo = []
for t, r in zip(tg, pr):
    p = dict(t); p.update({"labels": r["labels"], "boxes": r["boxes"], "scores": r["scores"]}); pseudo.append(p)
l1 = crit_r(out[1], pseudo)
for k, v in blob["loss1"].items():
    print(f"{k:28s} ref {float(v):12.6f} got {float(l1[k]):12.6f}")
# oracle criterion on the PRODUCT's stage-1 outputs and pseudo labels (CPU)
o1 = {"pred_logits": out[1]["pred_logits"].detach().cpu(), "pred_boxes": out[1]["pred_boxes"].detach().cpu(),
      "aux_outputs": [{k: v.detach().cpu() for k, v in a.items()} for a in out[1]["aux_outputs"]],
      "x_logits": out[1]["x_logits"].detach().cpu(), "x_cls_logits": out[1]["x_cls_logits"].detach().cpu()}
pc = [{k: v.cpu() for k, v in p.items()} for p in pseudo]
ol = O.set_criterion(o1, pc, refine=True)
for k, v in ol.items():
    print(f"oracle-on-product-outputs {k:28s} {float(v):12.6f} got {float(l1[k]):12.6f}")
# matching comparison, last layer
lg = out[1]["pred_logits"].detach()[None]; bx = out[1]["pred_boxes"].detach()[None]
srow, gidx, lidx = crit_r.matcher.match_flat(lg, bx, pseudo)
ind = O.hungarian(o1, pc)
print("device", srow.tolist(), gidx.tolist())
print("oracle", [(i.tolist(), j.tolist()) for i, j in ind])
