"""Which torch (aten) operators still launch kernels in one cfg2 training step: count, device time and the input
shapes, from torch.profiler.  Run on the GPU box:  python tools/aten_report.py [--top 40]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

import bench
from spe_amd import kernels as K
from spe_amd import lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--stack", action="store_true")
    ap.add_argument("--copies", action="store_true", help="host-visible copy operators (H2D / D2D / D2H) with their call sites")
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=a.stack or a.copies) as prof:
        step()
        torch.cuda.synchronize()
    if a.copies:
        names = {}
        for e in prof.events():
            if "emcpy" in e.name or "emset" in e.name:
                names[e.name] = names.get(e.name, 0) + 1
        print("memcpy / memset events:", names)
        sites = {}
        for e in prof.key_averages(group_by_stack_n=8):
            if e.key in ("aten::_to_copy", "aten::copy_", "aten::item", "aten::_local_scalar_dense", "aten::tensor", "aten::lift_fresh",
                         "aten::zeros", "aten::zero_", "aten::fill_", "aten::clone", "aten::contiguous"):
                where = [s for s in e.stack if "spe_amd" in s or "bench" in s][:2]
                sites[(e.key, tuple(w[-100:] for w in where))] = sites.get((e.key, tuple(w[-100:] for w in where)), 0) + e.count
        for (k, w), n in sorted(sites.items(), key=lambda kv: -kv[1])[:a.top]:
            print(f"{n:5d}  {k:26s} {' <- '.join(w)}")
        return
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6 if a.stack else 0):
        dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if dt <= 0 or not e.key.startswith("aten::"):
            continue
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110], e.stack if a.stack else None))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"aten device time in one step: {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} calls")
    for dt, n, k, shp, st in rows[:a.top]:
        print(f"{dt / 1e3:8.3f} ms {n:5d}  {k:28s} {shp}")
        if st:
            for s in st[:6]:
                if "spe_amd" in s or "bench" in s:
                    print("            ", s[-110:])


if __name__ == "__main__":
    main()
