"""Micro-benchmark of spe_gemm_f32 on the shapes of the cfg2 hot path (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def run(name, M, N, Kd, tA, tB, batch=(1, 1), splitk=1):
    b0, b1 = batch
    nb = b0 * b1
    A = torch.randn(nb, (Kd * M), device=dev)
    B = torch.randn(nb, (Kd * N), device=dev)
    C = torch.zeros(nb, M * N, device=dev)
    lda = M if tA else Kd
    ldb = Kd if tB else N
    f = lambda: K.gemm(A, B, C, M, N, Kd, lda, ldb, N, tA, tB, batch0=b0, batch1=b1, sA=(b1 * Kd * M, Kd * M),
                       sB=(b1 * Kd * N, Kd * N), sC=(b1 * M * N, M * N), splitk=splitk)
    t = timeit(f)
    fl = 2.0 * M * N * Kd * nb
    by = 4.0 * nb * (M * Kd + N * Kd + M * N)
    print(f"{name:34s} M={M:5d} N={N:5d} K={Kd:5d} b={nb:3d} sk={splitk:2d} {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s")


def main():
  for prec in (["bf16", "bf16x3"] if len(sys.argv) < 2 else [sys.argv[1]]):
      K.set_precision(prec)
      print("precision", prec)
      R = 8300
      run("qkv fwd NT", R, 1152, 384, False, True)
      run("fc1 fwd NT", R, 1536, 384, False, True)
      run("fc2 fwd NT", R, 384, 1536, False, True)
      run("proj fwd NT", R, 384, 384, False, True)
      run("fc1 dx NN", R, 384, 1536, False, False)
      run("fc2 dx NN", R, 1536, 384, False, False)
      run("fc1 dW TN", 1536, 384, R, True, False, splitk=K.auto_splitk(1536, 384, R, 1))
      run("fc2 dW TN", 384, 1536, R, True, False, splitk=K.auto_splitk(384, 1536, R, 1))
      run("qkv dW TN", 1152, 384, R, True, False, splitk=K.auto_splitk(1152, 384, R, 1))
      run("QK^T NT batched", 4150, 4150, 48, False, True, batch=(2, 8))
      run("PV NN batched", 4150, 48, 4150, False, False, batch=(2, 8))
      run("dV TN batched", 4150, 48, 4150, True, False, batch=(2, 8))
      run("square 4096 NT", 4096, 4096, 4096, False, True)
      run("dec proj NT", 200, 384, 384, False, True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("bf16nt", "epi", "dw")):
    main()


def run16(name, M, N, Kd, splitk=1):
    """bf16-operand NT kernel on the same logical shapes (operands pre-converted)."""
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    B = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    C = torch.zeros(max(1, abs(splitk)), M * N, device=dev)
    f = lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, splitk=splitk)
    t = timeit(f)
    fl = 2.0 * M * N * Kd
    by = 2.0 * (M * Kd + N * Kd) + 4.0 * M * N * abs(splitk)
    # vendor library on the same operands (bf16 output, i.e. half the C traffic): a yardstick, not a code path
    Bt = B.t()
    tv = timeit(lambda: torch.mm(A, Bt))
    print(f"{name:34s} M={M:5d} N={N:5d} K={Kd:5d} sk={splitk:3d} {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s"
          f"   | torch.mm bf16->bf16 {tv*1e6:7.1f} us")


def main16():
    R = 8300
    run16("qkv fwd", R, 1152, 384)
    run16("fc1 fwd", R, 1536, 384)
    run16("fc2 fwd", R, 384, 1536)
    run16("proj fwd", R, 384, 384)
    run16("fc1 dx", R, 384, 1536)
    run16("fc2 dx", R, 1536, 384)
    run16("qkv dx", R, 384, 1152)
    run16("fc1 dW", 1536, 384, 8320, splitk=-15)
    run16("fc2 dW", 384, 1536, 8320, splitk=-15)
    run16("qkv dW", 1152, 384, 8320, splitk=-16)
    run16("proj dW", 384, 384, 8320, splitk=-16)
    x = torch.randn(R, 1536, device=dev)
    print("cvt 8300x1536 out+outT %.1f us" % (timeit(lambda: K.cvt_bf16(x, True, True)) * 1e6))
    print("cvt 8300x1536 out      %.1f us" % (timeit(lambda: K.cvt_bf16(x, True, False)) * 1e6))
    x = torch.randn(R, 384, device=dev)
    print("cvt 8300x384  out+outT %.1f us" % (timeit(lambda: K.cvt_bf16(x, True, True)) * 1e6))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "bf16nt":
    main16()


def main_epi():
    """Epilogue cost breakdown on the fc1 shape."""
    M, N, Kd = 8300, 1536, 384
    A = torch.randn(M, Kd, device=dev).to(torch.bfloat16)
    B = torch.randn(N, Kd, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev); aux = torch.randn(M, N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); o16T = torch.empty(N, 8320, device=dev, dtype=torch.bfloat16)
    cs = torch.zeros(N, device=dev)
    us = lambda f: timeit(f) * 1e6
    print("plain C                 %.1f us" % us(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N)))
    print("plain C + bias          %.1f us" % us(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias)))
    print("plain C + C2            %.1f us" % us(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, C2=C2)))
    print("plain C + C2 + relu     %.1f us" % us(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, C2=C2, act=1)))
    print("plain C + C2 + gelu     %.1f us" % us(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, C2=C2, act=2)))
    print("plain C + gelu          %.1f us" % us(lambda: K.gemm16(A, B, C, M, N, Kd, Kd, Kd, N, bias=bias, act=2)))
    print("ex o16                  %.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, out16=o16)))
    print("ex o16T                 %.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, out16T=o16T)))
    print("ex o16 + o16T           %.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, out16=o16, out16T=o16T)))
    print("ex o16 + o16T + C2      %.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, bias=bias, C2=C2, out16=o16, out16T=o16T)))
    print("ex o16 + o16T + C2+gelu %.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, bias=bias, C2=C2, out16=o16, out16T=o16T, act=2)))
    print("ex o16 + o16T + cs      %.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, out16=o16, out16T=o16T, colsum=cs)))
    print("ex o16 + o16T + cs+relu'%.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, out16=o16, out16T=o16T, colsum=cs, aux=aux, act=1)))
    print("ex o16 + o16T + cs+gelu'%.1f us" % us(lambda: K.gemm16_ex(A, B, M, N, Kd, Kd, Kd, out16=o16, out16T=o16T, colsum=cs, aux=aux, act=2)))
    x = torch.randn(M, N, device=dev)
    print("cvt out+outT+cs+gelu'   %.1f us" % us(lambda: K.cvt_bf16(x, True, True, colsum_out=cs, act_aux=aux, act=2)))
    print("cvt out+outT            %.1f us" % us(lambda: K.cvt_bf16(x, True, True)))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "epi":
    main_epi()


def main_dw():
    """Split-K factor of the fc dW GEMMs: 36 output tiles x splits against the 512 resident workgroup slots."""
    for M, N in ((1536, 384), (384, 1536), (1152, 384), (384, 384)):
        for sk in (8, 10, 12, 13, 14, 15, 16):
            run16("dW %dx%d" % (M, N), M, N, 8320, splitk=-sk)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "dw":
    main_dw()
