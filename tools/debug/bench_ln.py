"""LayerNorm backward / forward at the backbone's shape: us per launch (SPE_LN_BWD_WGS = workgroup cap of the backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (R, C) in ((8300, 384), (400, 384), (6200, 384)):
    x = torch.randn(R, C, generator=g).to(dev); dy = torch.randn(R, C, generator=g).to(dev); gam = torch.randn(C, generator=g).to(dev)
    y, mean, rstd = K.layernorm_fwd(x, gam, torch.zeros(C, device=dev), 1e-6)[:3]
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    big = torch.empty(64 << 20, device=dev)
    def run(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(n):
            big.zero_()                     # evict: the step never finds these operands in L2 / MALL
            a.record(); K.layernorm_bwd(dy, x, gam, mean, rstd, dg, db); b.record()
            torch.cuda.synchronize(); tot += a.elapsed_time(b)
        return tot / n * 1e3
    run(5)
    print(f"R={R} C={C}: ln_bwd {run(40):.1f} us (cold operands), WGS cap {os.environ.get('SPE_LN_BWD_WGS', '256')}")
