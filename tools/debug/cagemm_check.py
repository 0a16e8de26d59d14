"""The decoder's memory-side projection GEMM [8300 x 384] x [384 x 4608] (north_star's decoder cross-attention GEMM): accuracy against
fp64 and time of the operand / output format variants of spe_gemm_bf16nt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, Kd = 8300, 4608, 384
x = torch.randn(M, Kd, generator=g).to(dev); W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
ref = (x.double() @ W.double().t() + b.double())
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
x16, _ = K.cvt_bf16(x); xlo = torch.empty_like(x16); K.cvt_bf16(x, out=x16, ldo=Kd, out_lo=xlo)
W16 = torch.empty((N, Kd), device=dev, dtype=torch.bfloat16); Wlo = torch.empty_like(W16); K.cvt_bf16(W, out=W16, ldo=Kd, out_lo=Wlo)
xh, Wh = K.cvt_f16(x), K.cvt_f16(W)
y32 = torch.empty((M, N), device=dev); y16 = torch.empty((M, N), device=dev, dtype=torch.float16)
rel = lambda a: float((a.double() - ref).norm() / ref.norm())
for name, fn, out in (("split bf16 -> fp32", lambda: K.gemm16(x16, W16, y32, M, N, Kd, Kd, Kd, N, bias=b, Alo=xlo, Blo=Wlo), y32),
                      ("split bf16 -> fp16", lambda: K.gemm16(x16, W16, y16, M, N, Kd, Kd, Kd, N, bias=b, Alo=xlo, Blo=Wlo, act=0x200), y16),
                      ("single bf16 -> fp32", lambda: K.gemm16(x16, W16, y32, M, N, Kd, Kd, Kd, N, bias=b), y32),
                      ("single fp16 -> fp32", lambda: K.gemm16(xh, Wh, y32, M, N, Kd, Kd, Kd, N, bias=b, act=0x100), y32),
                      ("single fp16 -> fp16", lambda: K.gemm16(xh, Wh, y16, M, N, Kd, Kd, Kd, N, bias=b, act=0x300), y16)):
    us = t(fn)
    fl = 2.0 * M * N * Kd
    print(f"{name:22s} {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic ({fl / us / 1e6 / 2500 * 100:4.1f} % of 2.5 PF)  rel err vs fp64 {rel(out):.2e}")
print("cvt_f16 x %.1f us, W %.1f us" % (t(lambda: K.cvt_f16(x)), t(lambda: K.cvt_f16(W))))
