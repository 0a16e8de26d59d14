"""north_star's decoder cross-attention GEMM [8300 x 384] x [384 x 4608] (fp16 operands, fp16 output) under the phase-ablation builds of tools/ab.py
(ONLY=gemm_nt2.hip: -DSPE_ABL_NOSTORE = main loop only, -DSPE_ABL_NOLOOP = one contraction step + the whole epilogue): the committed answer to
"main-loop-only time vs store-only time vs overlapped".  Run once per library: SPE_HIP_LIB=build_ab/<name>.so python tools/debug/cagemm_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, Kd = 8300, 4608, 384
x = torch.randn(M, Kd, generator=g).to(dev); W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
xh, Wh = K.cvt_f16(x), K.cvt_f16(W)
y16 = torch.empty((M, N), device=dev, dtype=torch.float16)
fn = lambda: K.gemm16(xh, Wh, y16, M, N, Kd, Kd, Kd, N, bias=b, act=0x300)
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    fn()
e.record(); torch.cuda.synchronize()
us = a.elapsed_time(e) / 50 * 1e3
print("%-10s %6.1f us" % (os.path.basename(os.environ.get("SPE_HIP_LIB", "product")), us))
