"""Timing of the row-wise backward kernels at the backbone's shape [8300, 384] (LayerNorm backward with skip gradient, LayerScale residual
backward with bf16 output): used with a -DSPE_DBG_NORED build (tools/ab.py) to see what the fixed-order cross-workgroup sums cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
R, C = 8300, 384
g = torch.Generator().manual_seed(0)
x = torch.randn(R, C, generator=g).to(dev); dy = torch.randn(R, C, generator=g).to(dev); add = torch.randn(R, C, generator=g).to(dev)
gam = torch.rand(C, generator=g).to(dev); bet = torch.zeros(C, device=dev)
y, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-6)
def timeit(f, n=100):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
import inspect
print(os.environ.get("SPE_HIP_LIB", "default"))
print("layernorm_bwd       %6.1f us" % timeit(lambda: K.layernorm_bwd(dy, x, gam, mean, rstd, add=add)))
yb = torch.randn(R, C, generator=g).to(dev)
Rp = ((R + 63) // 64) * 64
print("layernorm_bwd + LS  %6.1f us" % timeit(lambda: K.layernorm_bwd(dy, x, gam, mean, rstd, add=add, ls=(yb, gam, None, None))))
print("lsres_bwd16         %6.1f us" % timeit(lambda: K.layerscale_residual_bwd16(dy, yb, gam, Rp, want_rowmajor=True, want_T=False)))
x16 = torch.empty(R, C, device=dev, dtype=torch.bfloat16); cs = torch.zeros(C, device=dev)
print("cvt_bf16 + colsum   %6.1f us" % timeit(lambda: K.cvt_bf16(dy, True, False, colsum_out=cs, out=x16)))
print("cvt_bf16            %6.1f us" % timeit(lambda: K.cvt_bf16(dy, True, False, out=x16)))
