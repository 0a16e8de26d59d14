"""cProfile of the host side of one training step (enqueue only): where the Python thread spends its ~38 ms."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
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
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
if "--cum" in sys.argv:
    st.sort_stats("cumtime").print_stats(70)
