"""Call sites of the torch operators dispatched in one cfg2 training step (TorchDispatchMode + Python stack): where the
conversions / copies / small elementwise launches of the step come from.  GPU box:  python tools/op_sites.py [--ops a,b] [--top 60]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from spe_amd import kernels as K
from spe_amd import lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor


class Sites(TorchDispatchMode):
    def __init__(self, only):
        super().__init__()
        self.only = only
        self.sites = collections.Counter()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        self.ops[name] += 1
        if not self.only or name in self.only:
            fr = [f for f in traceback.extract_stack()[:-1] if ("spe_amd" in f.filename or "bench" in f.filename) and "op_sites" not in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:]))
            self.sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="_to_copy,copy_,clone,fill_,zero_,zeros,cat,add,add_,mul,sub,div,sum,empty_like,index,stack")
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib.load()
    K.manual_seed(1234)
    args = bench.model_args()
    torch.manual_seed(0)
    model, crit, crit_r, pp, rpp = build_model(args)
    model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
    wd = crit.weight_dict
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
        total = bench.weighted_total(l0, l1, wd)
        total.backward()
        reducer.finish()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    only = set(x for x in a.ops.split(",") if x)
    with Sites(only) as s:
        step()
        torch.cuda.synchronize()
    print("operators dispatched in one step:", sum(s.ops.values()))
    print("  ", ", ".join(f"{k} {v}" for k, v in s.ops.most_common(40)))
    for (name, where), n in s.sites.most_common(a.top):
        print(f"{n:5d}  {name:14s} {where}")


if __name__ == "__main__":
    main()
