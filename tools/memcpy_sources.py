"""Which ATen operators the device memcpy / copyBuffer / fill launches of one cfg2 step belong to (chrome trace of
torch.profiler: GPU events are matched to the CPU op with the same External id).  Run on the GPU box."""
import json, os, sys, tempfile
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from spe_amd import kernels as K, lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor

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


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
cpu = {}
for e in ev:
    if e.get("cat") == "cpu_op" and "args" in e and "External id" in e["args"]:
        cpu[e["args"]["External id"]] = e
cnt, dur = Counter(), Counter()
for e in ev:
    if e.get("cat") in ("gpu_memcpy", "gpu_memset") or (e.get("cat") == "kernel" and ("copyBuffer" in e["name"] or "fillBuffer" in e["name"])):
        x = e.get("args", {}).get("External id")
        op = cpu.get(x)
        key = (e["name"][:40], op["name"] if op else "?", str(op["args"].get("Input Dims", ""))[:70] if op else "")
        cnt[key] += 1
        dur[key] += e.get("dur", 0)
print("memcpy/memset-like GPU events in one step:", sum(cnt.values()), "= %.2f ms" % (sum(dur.values()) / 1e3))
for k, n in cnt.most_common(40):
    print("%4d %7.1f us  %-40s %-22s %s" % (n, dur[k], *k))
