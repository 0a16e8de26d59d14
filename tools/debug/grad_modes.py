"""Full-tensor parameter-gradient error of the bf16 mode against the bf16x3 mode (itself within 6e-3 of the reference) on a case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, cfg_cases as cc
from spe_amd import kernels as K
from spe_amd.util.misc import NestedTensor
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_depth2"
dev = torch.device("cuda:0")
blob = torch.load(os.path.join(ROOT, "tests", "golden", f"cfg_{name}.pt"), weights_only=False)
res = {}
for prec in ("bf16x3", "bf16"):
    args, (model, crit, crit_r, pp, rpp), tensors, mask, targets = cc.build_case(name)
    K.set_precision(prec)
    model.to(dev).train(); crit.to(dev).eval(); crit_r.to(dev).eval()
    tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    out = model(NestedTensor(tensors.to(dev), mask.to(dev)))
    l0 = crit(out[0], tg)
    pseudo = [{k: v.to(dev) for k, v in p.items()} for p in blob["pseudo"]]
    l1 = crit_r(out[1], pseudo)
    wd = blob["weight_dict"]
    total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
    total.backward()
    res[prec] = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
K.set_precision("bf16")
errs = []
gmax = max(float(g.norm()) for g in res["bf16x3"].values())
for n, g in res["bf16x3"].items():
    if float(g.norm()) < 1e-6 * gmax: continue
    errs.append((float((res["bf16"][n] - g).norm() / g.norm()), n, float(g.norm())))
errs.sort(reverse=True)
for e in errs[:12]: print("%.3e  %-60s |g| %.3e" % e)
print("median %.3e over %d" % (errs[len(errs) // 2][0], len(errs)))
