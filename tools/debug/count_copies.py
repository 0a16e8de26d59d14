import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from spe_amd import kernels as K, lib, dp
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util import misc
from spe_amd.util.misc import NestedTensor
dev = torch.device("cuda", 0)
lib.load(); K.set_precision("bf16"); K.manual_seed(1)
args = bench.model_args()
torch.manual_seed(0)
model, crit, crit_r, pp, rpp = build_model(args)
model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
wd = crit.weight_dict
named = dict((p, n) for n, p in model.named_parameters())
params = [p for p in model.parameters() if p.requires_grad]
reducer = GradAllReducer(params, flatten_params=True)
opt = FlatAdamW(params, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
img, mask, targets = bench.synth_batch(1234, dev)
samples = NestedTensor(img, mask)
cnt = collections.Counter(); outside = []
orig = GradAllReducer._on_grad
def og(self, p):
    if p.grad.data_ptr() != self._views[p].data_ptr():
        cnt["grad_copied_into_bucket"] += 1; outside.append(named[p])
    else:
        cnt["grad_direct"] += 1
    return orig(self, p)
GradAllReducer._on_grad = og
for h in reducer._hooks: h.remove()
reducer._hooks = [p.register_post_accumulate_grad_hook(reducer._on_grad) for p in reducer.params]
def step():
    reducer.reset(); out = model(samples); l0 = crit(out[0], targets)
    with torch.no_grad(): ps = bench.pseudo_labels(rpp, out[0], targets)
    l1 = crit_r(out[1], ps); total = bench.weighted_total(l0, l1, wd); total.backward(); reducer.finish(); opt.step()
for _ in range(2): step()
cnt.clear(); outside.clear()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(dict(cnt))
print("outside:", collections.Counter(n.split(".")[-1] + ("(bb)" if "backbone" in n else "(dec)") for n in outside).most_common(30))
ev = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or "Memcpy" in e.name or "memcpy" in e.name:
        ev[e.name[:60]] += 1
for k, v in ev.most_common(12): print(v, k)
