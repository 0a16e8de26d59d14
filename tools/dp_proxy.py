"""One-GPU proxy for "the gradient all-reduce runs beside the backward" (reference main.py:172).

An RCCL ring all-reduce keeps `channels` persistent workgroups resident (one CU slot each) for as long as a bucket is in flight
and moves the bucket through HBM a few times.  No second GPU exists on the box, so this tool launches spe_occupy (csrc/misc.hip:
nwg workgroups that copy through a 256 MB buffer, or idle) on a side stream for the duration of every backward and measures the
step: per-step time against the solo run, for nwg in {0, 8, 16, 32, 64}, with and without memory traffic.

    python tools/dp_proxy.py        # GPU box; writes gpurun_out/dp_proxy.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from spe_amd import kernels as K, lib  # noqa: E402
from spe_amd.dp import GradAllReducer  # noqa: E402
from spe_amd.optim import FlatAdamW  # noqa: E402
from spe_amd.models import build_model  # noqa: E402
from spe_amd.util.misc import NestedTensor  # noqa: E402

dev = torch.device("cuda", 0)
lib.load()
K.manual_seed(1234)
args = bench.model_args()
torch.manual_seed(0)
model, crit, crit_r, pp, rpp = build_model(args)
model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
reducer = GradAllReducer(params, flatten_params=True)
opt = FlatAdamW(params, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
img, mask, targets = bench.synth_batch(1234, dev)
samples = NestedTensor(img, mask)
wd = crit.weight_dict
side = torch.cuda.Stream()
buf = torch.zeros(64 << 20, device=dev)          # 256 MB


def step(nwg, traffic, micros):
    reducer.reset()
    out = model(samples)
    l0 = crit(out[0], targets)
    with torch.no_grad():
        ps = bench.pseudo_labels(rpp, out[0], targets)
    l1 = crit_r(out[1], ps)
    total = bench.weighted_total(l0, l1, wd)
    if nwg:         # the occupier starts when the backward starts (side stream ordered behind the forward)
        side.wait_stream(torch.cuda.current_stream())
        lib.call("spe_occupy", nwg, micros, buf.data_ptr() if traffic else None, buf.numel(), side.cuda_stream)
    total.backward()
    reducer.finish()
    opt.step()
    if nwg:
        torch.cuda.current_stream().wait_stream(side)


def run(nwg, traffic, micros, steps=8):
    for _ in range(2):
        step(nwg, traffic, micros)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(nwg, traffic, micros)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


solo = run(0, False, 0)
res = {"solo_ms_per_step": solo, "note": "occupier resident for ~36 ms of every step (the backward); proxy, not RCCL", "runs": []}
for reserve in (0, 32):          # kernels.set_cu_reserve: attention grids of 512 - reserve workgroups (what GradAllReducer sets for world > 1)
    K.set_cu_reserve(reserve)
    base = run(0, False, 0)
    print(f"cu_reserve {reserve}: no occupier {base:7.2f} ms/step ({100 * (base / solo - 1):+.1f} % vs solo grids)", flush=True)
    res["runs"].append({"cu_reserve": reserve, "workgroups": 0, "hbm_traffic": False, "ms_per_step": base, "slowdown": base / solo - 1.0})
    for traffic in (False, True):
        for nwg in (8, 16, 32):
            ms = run(nwg, traffic, 36000)
            res["runs"].append({"cu_reserve": reserve, "workgroups": nwg, "hbm_traffic": traffic, "ms_per_step": ms, "slowdown": ms / solo - 1.0})
            print(f"cu_reserve {reserve:2d} nwg {nwg:3d} traffic {traffic!s:5s}: {ms:7.2f} ms/step ({100 * (ms / solo - 1):+.1f} % vs solo {solo:.2f})", flush=True)
K.set_cu_reserve(0)
od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(od, exist_ok=True)
json.dump(res, open(os.path.join(od, "dp_proxy.json"), "w"), indent=1)
