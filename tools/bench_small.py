"""ops.linear forward / backward at the decoder's shapes (a few hundred rows): us per call, csrc/linear_small.hip (default) against the
conversion + GEMM path (SPE_LINEAR_SMALL=0).  Run on the GPU box, once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K, ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("SPE_LINEAR_SMALL =", os.environ.get("SPE_LINEAR_SMALL", "1"), "precision", K.get_precision())
for (R, Kd, N, act) in ((400, 384, 384, 0), (400, 384, 2048, 1), (400, 2048, 384, 0), (400, 384, 96, 0), (1200, 384, 384, 0), (182, 384, 384, 0)):
    x = torch.randn(R, Kd, generator=g).to(dev).requires_grad_()
    W = (torch.randn(N, Kd, generator=g) * 0.05).to(dev).requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    go = torch.randn(R, N, generator=g).to(dev)
    xs = [torch.randn(R, Kd, generator=g).to(dev).requires_grad_() for _ in range(60)]
    it = iter(range(10 ** 9))

    def fwd():
        return ops.linear(xs[next(it) % 60], W, b, act)          # a fresh activation object every call: no cached bf16 copy

    y = fwd()
    tf = t_us(fwd)

    def fb():
        yy = ops.linear(xs[next(it) % 60], W, b, act)
        torch.autograd.grad(yy, (xs[(next(it) - 1) % 60] if False else W, b), go)

    def fb2():
        xi = xs[next(it) % 60]
        yy = ops.linear(xi, W, b, act)
        torch.autograd.grad(yy, (xi, W, b), go)

    tfb = t_us(fb2)
    print(f"R={R:5d} K={Kd:5d} N={N:5d} act={act}: forward {tf:6.1f} us, forward+backward {tfb:6.1f} us")
