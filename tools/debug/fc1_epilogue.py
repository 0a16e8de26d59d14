"""What the fc1 + GELU epilogue of the split forward GEMM costs, output by output (8300 x 1536 x 384): pre-activation fp16 / fp32 / none,
gelu hi only / hi + lo, and the dh backward epilogue with / without its pieces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, Kd = 8300, 1536, 384
x = torch.randn(M, Kd, generator=g).to(dev); W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
xh = x.to(torch.bfloat16); xl = (x - xh.float()).to(torch.bfloat16); Wh = W.to(torch.bfloat16); Wl = (W - Wh.float()).to(torch.bfloat16)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
pre32 = torch.empty(M, N, device=dev); pre16 = torch.empty(M, N, device=dev, dtype=torch.float16)
h = torch.empty(M, N, device=dev, dtype=torch.bfloat16); hl = torch.empty_like(h)
lo = dict(Alo=xl, Blo=Wl)
for name, kw in (("pre fp16 + h hi + h lo (shipped)", dict(C2=pre16, out16=h, out16lo=hl)), ("pre fp32 + h hi + h lo", dict(C2=pre32, out16=h, out16lo=hl)),
                 ("h hi + h lo", dict(out16=h, out16lo=hl)), ("pre fp16 + h hi", dict(C2=pre16, out16=h)), ("h hi", dict(out16=h)), ("pre fp16 only", dict(C2=pre16))):
    print("fc1 split  %-36s %6.1f us" % (name, timeit(lambda: K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, bias=b, act=2, **kw, **lo))))
C = torch.empty(M, N, device=dev)
print("fc1 split  %-36s %6.1f us" % ("plain fp32 C (no GELU)", timeit(lambda: K.gemm16(xh, Wh, C, M, N, Kd, Kd, Kd, N, bias=b, **lo))))
# dh backward: A = dy16 [M, 384], B = W2^T [1536, 384]
o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); cs = torch.zeros(N, device=dev)
for name, kw in (("aux fp16 + bf16 out + colsum (shipped)", dict(out16=o16, colsum=cs, aux=pre16, act=2)), ("aux fp16 + bf16 out", dict(out16=o16, aux=pre16, act=2)),
                 ("bf16 out + colsum", dict(out16=o16, colsum=cs)), ("bf16 out", dict(out16=o16))):
    print("dh single  %-36s %6.1f us" % (name, timeit(lambda: K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, **kw))))
# proj / fc2 forward (split, residual epilogue): what the saved branch output y (C2) costs
for nm, (N2, K2) in (("proj", (384, 384)), ("fc2", (384, 1536))):
    x2 = torch.randn(M, K2, generator=g).to(dev); W2 = (torch.randn(N2, K2, generator=g) / K2 ** 0.5).to(dev); b2 = torch.randn(N2, generator=g).to(dev)
    xh2 = x2.to(torch.bfloat16); xl2 = (x2 - xh2.float()).to(torch.bfloat16); Wh2 = W2.to(torch.bfloat16); Wl2 = (W2 - Wh2.float()).to(torch.bfloat16)
    res = torch.randn(M, N2, generator=g).to(dev); gam = torch.rand(N2, generator=g).to(dev)
    C2_ = torch.empty(M, N2, device=dev); y32 = torch.empty(M, N2, device=dev); y16 = torch.empty(M, N2, device=dev, dtype=torch.float16)
    for name, kw in (("res + y fp32 (shipped)", dict(C2=y32)), ("res + y fp16", dict(C2=y16)), ("res, no y", dict())):
        print("%-4s split  %-36s %6.1f us" % (nm, name, timeit(lambda: K.gemm16_ex(xh2, Wh2, M, N2, K2, K2, K2, bias=b2, C=C2_, res=res, rgamma=gam, Alo=xl2, Blo=Wl2, **kw))))
    print("%-4s split  %-36s %6.1f us" % (nm, "plain fp32 C", timeit(lambda: K.gemm16(xh2, Wh2, C2_, M, N2, K2, K2, K2, N2, bias=b2, Alo=xl2, Blo=Wl2))))
