"""Ring variant of spe_gemm_bf16nt (SPE_GEMM16_RING): correctness against fp32 matmul of the same bf16 operands + timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator().manual_seed(0)
for (M, N, Kd) in ((8300, 384, 1536), (8300, 384, 1152), (8300, 1536, 384), (8300, 1152, 384), (8300, 384, 384), (8300, 384, 4608), (4099, 200, 640)):
    A = torch.randn(M, Kd, generator=g).to(dev).to(torch.bfloat16)
    B = torch.randn(N, Kd, generator=g).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.zeros(M, N, device=dev)
    K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias)
    ref = A.float() @ B.float().t() + bias
    err = float((C - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias))
    print(f"M={M} N={N} K={Kd}: rel err {err:.2e}  {t:7.1f} us  {2.0*M*N*Kd/t/1e6:6.0f} TF/s")
