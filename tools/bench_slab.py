import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
K.set_precision("bf16")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
R = 8300
for (N, Kd) in ((1536, 384), (384, 1536), (1152, 384), (384, 384)):
    dy = torch.randn(R, N, device=dev); x = torch.randn(R, Kd, device=dev)
    for sk in (2, 4, 8, 14, 24):
        ws = torch.empty(sk, N * Kd, device=dev)
        def f():
            K.gemm(dy, x, ws, N, Kd, R, N, Kd, Kd, True, False, splitk=-sk)
            return K.colsum(ws)
        def g():
            K.gemm(dy, x, ws, N, Kd, R, N, Kd, Kd, True, False, splitk=-sk)
        print(f"dW N={N} K={Kd} slab sk={sk:2d}: gemm {t(g):7.1f} us  gemm+reduce {t(f):7.1f} us")
