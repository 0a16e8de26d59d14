"""Entry-point launch counts of one cfg2 training step."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from spe_amd import kernels as K, lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor
dev = torch.device("cuda", 0); lib.load(); K.set_precision("bf16"); K.manual_seed(1)
args = bench.model_args(); torch.manual_seed(0)
model, crit, crit_r, pp, rpp = build_model(args)
model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
wd = crit.weight_dict
params = [p for p in model.parameters() if p.requires_grad]
reducer = GradAllReducer(params, flatten_params=True)
opt = FlatAdamW(params, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
img, mask, targets = bench.synth_batch(1234, dev); samples = NestedTensor(img, mask)
def step():
    reducer.reset(); out = model(samples); l0 = crit(out[0], targets)
    with torch.no_grad(): ps = bench.pseudo_labels(rpp, out[0], targets)
    l1 = crit_r(out[1], ps); total = bench.weighted_total(l0, l1, wd); total.backward(); reducer.finish(); opt.step()
for _ in range(2): step()
cnt = collections.Counter(); orig = lib.call
def spy(nm, *a): cnt[nm] += 1; return orig(nm, *a)
lib.call = spy; K.lib.call = spy
step(); torch.cuda.synchronize()
print("total spe launches", sum(cnt.values()))
for k, v in cnt.most_common(40): print(f"{v:5d} {k}")
