"""bf16 NT GEMMs of the cfg2 hot path at their model shapes: time, TFLOP/s (algorithmic 2MNK), % of the 2.5 PF bf16 MFMA peak,
error against fp64, for the single-term (backward) and the split-operand (bf16s forward) products and the fused epilogues.

    python tools/bench_nt.py                 # GPU box
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from spe_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def case(name, M, N, Kd, mode):
    """mode: plain1 / plain3 (C fp32 + bias), gelu3 (fc1 forward: pre fp32 + gelu as hi/lo bf16), res3 (proj / fc2 forward:
    residual epilogue + y), dgelu1 (dh backward: gelu' from aux, bf16 out + column sums)."""
    x = torch.randn(M, Kd, generator=g).to(dev)
    W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    xh, xl = split(x)
    Wh, Wl = split(W)
    sp = mode.endswith("3")
    lo = dict(Alo=xl, Blo=Wl) if sp else {}
    xr, Wr = (x.double(), W.double()) if sp else (xh.double(), Wh.double())
    ref = xr @ Wr.t() + b.double()
    C = torch.empty(M, N, device=dev)
    out_bytes = 4.0 * M * N
    if mode.startswith("plain"):
        f = lambda: K.gemm16(xh, Wh, C, M, N, Kd, Kd, Kd, N, bias=b, **lo)
        f(); err = rel(C, ref)
    elif mode == "gelu3":
        pre = torch.empty(M, N, device=dev); h = torch.empty(M, N, device=dev, dtype=torch.bfloat16); hl = torch.empty_like(h)
        f = lambda: K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, bias=b, C2=pre, out16=h, out16lo=hl, act=2, **lo)
        f(); err = max(rel(pre, ref), rel(h.float() + hl.float(), torch.nn.functional.gelu(ref)))
        out_bytes = 8.0 * M * N
    elif mode == "res3":
        res = torch.randn(M, N, generator=g).to(dev); gam = torch.rand(N, generator=g).to(dev); y = torch.empty(M, N, device=dev)
        f = lambda: K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, bias=b, C=C, C2=y, res=res, rgamma=gam, **lo)
        f(); err = max(rel(y, ref), rel(C, res.double() + gam.double() * ref))
        out_bytes = 12.0 * M * N
    elif mode == "dgelu1":
        aux = torch.randn(M, N, generator=g).to(dev); o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cs = torch.zeros(N, device=dev)
        f = lambda: K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, out16=o16, colsum=cs, aux=aux, act=2)
        f()
        a64 = aux.double()
        d = 0.5 * (1 + torch.erf(a64 / 2 ** 0.5)) + a64 * torch.exp(-0.5 * a64 * a64) / (2 * 3.141592653589793) ** 0.5
        err = rel(o16.float(), (xh.double() @ Wh.double().t()) * d)
        out_bytes = 6.0 * M * N
    t = timeit(f)
    fl = 2.0 * M * N * Kd
    by = 2.0 * (M * Kd + N * Kd) * (2 if sp else 1) + out_bytes
    tv = timeit(lambda: torch.mm(xh, Wh.t()))
    print(f"{name:26s} {mode:7s} M={M:5d} N={N:5d} K={Kd:5d} {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s {100*fl/t/2.5e15:5.1f}% of 2.5PF "
          f"{by/t/1e9:6.0f} GB/s err {err:.1e} | torch.mm bf16->bf16 {tv*1e6:6.1f} us")


def main():
    R = 8300
    case("qkv fwd", R, 1152, 384, "plain3")
    case("fc1 fwd + gelu", R, 1536, 384, "gelu3")
    case("proj fwd + res", R, 384, 384, "res3")
    case("fc2 fwd + res", R, 384, 1536, "res3")
    case("decoder ca k+v (12 stacked)", R, 4608, 384, "plain3")
    case("decoder ca kpos (6 stacked)", R, 2304, 384, "plain3")
    case("decoder ca k+v 1-term", R, 4608, 384, "plain1")
    case("qkv dx", R, 384, 1152, "plain1")
    case("proj dx", R, 384, 384, "plain1")
    case("fc1 dx", R, 384, 1536, "plain1")
    case("fc2 dh * gelu'", R, 1536, 384, "dgelu1")
    case("stacked dx (K=4608)", R, 384, 4608, "plain1")
    case("cfg5 qkv fwd", 6200, 1152, 384, "plain3")
    case("ragged", 8211, 1144, 448, "plain3")
    case("ragged 1-term", 8211, 392, 320, "plain1")


if __name__ == "__main__":
    main()
