import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
R = 8300; Rp = (R + 63) // 64 * 64
for (M, N) in [(384, 384), (1536, 384), (384, 1536), (1152, 384)]:
    dy = torch.randn(R, M, device=dev).to(torch.bfloat16); x = torch.randn(R, N, device=dev).to(torch.bfloat16)
    dyT = torch.zeros(M, Rp, device=dev, dtype=torch.bfloat16); dyT[:, :R] = dy.t()
    xT = torch.zeros(N, Rp, device=dev, dtype=torch.bfloat16); xT[:, :R] = x.t()
    sk = min(K.auto_splitk(M, N, Rp, 1), Rp // 64)
    ws = torch.empty(max(sk, 1), M * N, device=dev)
    nt = t(lambda: K.gemm16(dyT, xT, ws, M, N, Rp, Rp, Rp, N, splitk=-sk if sk > 1 else 1))
    res = []
    for s2 in sorted({sk, max(1, sk // 2), min(sk * 2, Rp // 64)}):
        ws2 = torch.empty(max(s2, 1), M * N, device=dev)
        res.append((s2, t(lambda: K.gemm16_tn(dy, x, ws2, M, N, R, M, N, N, splitk=-s2 if s2 > 1 else 1))))
    print(f"dW {M}x{N} (R={R}): NT splitk {sk}: {nt:.1f} us ; TN " + ", ".join(f"sk{a}: {b:.1f} us" for a, b in res))
