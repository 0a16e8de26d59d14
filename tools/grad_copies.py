"""Which parameter gradients still reach their all-reduce bucket through a copy (produced outside the bucket by a torch
op or summed by autograd) in one cfg2 training step.  Run on the GPU box."""
import os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spe_amd import kernels as K, lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor

dev = torch.device("cuda", 0)
lib.load(); K.set_precision("bf16"); K.manual_seed(1234)
args = bench.model_args()
torch.manual_seed(0)
model, crit, crit_r, pp, rpp = build_model(args)
model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
names = {p: n for n, p in model.named_parameters()}
params = [p for p in model.parameters() if p.requires_grad]
copied = []
_orig = GradAllReducer._on_grad
def _hook(self, p):
    if p.grad.data_ptr() != self._views[p].data_ptr():
        copied.append(names[p])
    return _orig(self, p)
GradAllReducer._on_grad = _hook
reducer = GradAllReducer(params, flatten_params=True)
opt = FlatAdamW(params, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
img, mask, targets = bench.synth_batch(1234, dev)
samples = NestedTensor(img, mask)
def step():
    reducer.reset()
    out = model(samples)
    l0 = crit(out[0], targets)
    with torch.no_grad():
        ps = bench.pseudo_labels(rpp, out[0], targets)
    l1 = crit_r(out[1], ps)
    bench.weighted_total(l0, l1, crit.weight_dict).backward()
    reducer.finish(); opt.step()
step(); copied.clear(); step()
print(len(copied), "of", len(params), "gradients copied into their bucket")
import re
c = Counter(re.sub(r"\.\d+\.", ".N.", n) for n in copied)
for k, v in c.most_common(60):
    print("%4d  %s" % (v, k))
