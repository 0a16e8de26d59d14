"""Call sites of the ATen ops that end in a device copy (copy_, clone, _to_copy, contiguous, cat, stack, index, fill) in one
cfg2 training step: a TorchDispatchMode records the innermost spe_amd/bench frame of every such op on CUDA tensors.
Run on the GPU box:  python tools/copy_sites.py [--ops copy_,clone]"""
import argparse, os, sys, traceback
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from spe_amd import kernels as K, lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor

ap = argparse.ArgumentParser()
ap.add_argument("--ops", default="")
ap.add_argument("--top", type=int, default=60)
a = ap.parse_args()
want = set(x for x in a.ops.split(",") if x)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sites = Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if want and name not in want:
            return out
        ts = [t for t in (list(args) + ([out] if isinstance(out, torch.Tensor) else [])) if isinstance(t, torch.Tensor)]
        if not any(t.is_cuda for t in ts):
            return out
        site = "autograd/other"
        for fr in reversed(traceback.extract_stack(limit=24)):
            if fr.filename.startswith(ROOT) and "tools/copy_sites" not in fr.filename:
                site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                break
        shp = tuple(ts[0].shape) if ts else ()
        sites[(name, site, shp)] += 1
        return out


dev = torch.device("cuda", 0)
lib.load(); K.manual_seed(1234)
args = bench.model_args()
torch.manual_seed(0)
model, crit, crit_r, pp, rpp = build_model(args)
model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
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


step(); step()
with Rec():
    step()
tot = Counter()
for (name, site, shp), n in sites.items():
    tot[name] += n
print("ATen ops on CUDA tensors in one step:", sum(tot.values()))
print(", ".join("%s %d" % kv for kv in tot.most_common(25)))
for (name, site, shp), n in sites.most_common(a.top):
    print("%4d  %-14s %-58s %s" % (n, name, site, shp))
