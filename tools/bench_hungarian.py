"""cfg4 matcher stress: device Hungarian vs SciPy on the host, per criterion call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy.optimize import linear_sum_assignment
from spe_amd import kernels as K

dev = torch.device("cuda:0")
for (L, B, Q, M) in [(6, 2, 100, 7), (6, 2, 300, 35), (6, 2, 300, 100), (6, 2, 300, 300)]:
    g = torch.Generator().manual_seed(Q + M)
    total = B * M
    toff = torch.tensor([i * M for i in range(B + 1)], dtype=torch.int32, device=dev)
    cost = (torch.randn(L, Q * total, generator=g) * 3).to(dev)
    K.hungarian(cost, toff, L, B, Q, total)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        K.hungarian(cost, toff, L, B, Q, total)
    b.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ch = cost.cpu()
    for l in range(L):
        for bb in range(B):
            linear_sum_assignment(ch[l, Q * bb * M:Q * (bb + 1) * M].view(Q, M).numpy())
    t1 = time.perf_counter()
    print(f"L={L} B={B} Q={Q} M={M}: device {a.elapsed_time(b) / 5:.3f} ms ; host copy + SciPy {1e3 * (t1 - t0):.3f} ms")
