"""Run-to-run determinism (bitwise) of the atomic-free kernels with foreign kernels in between: contractions, GEMMs (NT, ring, TN),
flash MHA forward + backward through ops.attention."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K, ops
dev = torch.device("cuda:0")
K.set_precision("bf16")
g = torch.Generator().manual_seed(3)
xs = torch.randn(2048, 1024, device=dev)
def dirty(t):
    if t % 3 == 1: (xs @ xs.t()[:, :512]).sum()
    elif t % 3 == 2: torch.softmax(xs * (1 + t), dim=1)
def check(name, fn, n=150):
    ref = [r.clone() for r in fn()]
    bad = 0
    for t in range(n):
        dirty(t)
        for a, b in zip(fn(), ref):
            if not torch.equal(a, b): bad += 1; break
    print(f"{name:40s} mismatching runs: {bad} of {n}")
# contractions
B, H, N, dh = 2, 8, 1100, 48
PT = K.score_blocks(B, H, N, dev); PT.view(torch.int16).random_(0, 16000)
v = torch.randn(B, N, H, dh, generator=g).to(dev)
V16 = K.attn_pack16(v)
O = torch.empty(B, N, H * dh, device=dev)
check("attn_contract trans=0", lambda: [K.attn_contract(PT, V16, O.view(B, N, H, dh), False).clone()])
check("attn_contract trans=1", lambda: [K.attn_contract(PT, V16, O.view(B, N, H, dh), True).clone()])
# GEMMs
for (M, Nn, Kd) in ((8300, 1536, 384), (8300, 384, 1536), (400, 384, 384)):
    A = torch.randn(M, Kd, generator=g).to(dev).to(torch.bfloat16); Bm = torch.randn(Nn, Kd, generator=g).to(dev).to(torch.bfloat16)
    C = torch.empty(M, Nn, device=dev)
    check(f"gemm16 {M}x{Nn}x{Kd}", lambda A=A, Bm=Bm, C=C, M=M, Nn=Nn, Kd=Kd: (K.gemm16(A, Bm, C, M, Nn, Kd, Kd, Kd, Nn), [C.clone()])[1])
A = torch.randn(8300, 1536, generator=g).to(dev).to(torch.bfloat16); Bm = torch.randn(8300, 384, generator=g).to(dev).to(torch.bfloat16)
ws = torch.empty(14, 1536 * 384, device=dev)
check("gemm16_tn 1536x384x8300 sk14", lambda: (K.gemm16_tn(A, Bm, ws, 1536, 384, 8300, 1536, 384, 384, splitk=-14), [ws.clone()])[1])
# flash MHA (decoder cross-attention shape), forward + backward
Bq, Lq, Lk, Hh = 2, 200, 4150, 8
q0 = torch.randn(Bq, Lq, Hh, 96, generator=g).to(dev); k0 = torch.randn(Bq, Lk, Hh, 96, generator=g).to(dev); v0 = torch.randn(Bq, Lk, Hh, 48, generator=g).to(dev)
go = torch.randn(Bq, Lq, Hh * 48, generator=g).to(dev)
def mha():
    K.manual_seed(5)
    t = [x.clone().requires_grad_() for x in (q0, k0, v0)]
    o, _ = ops.attention(t[0], t[1], t[2], None, 96 ** -0.5, 0.1)
    return [o.detach()] + [x.detach() for x in torch.autograd.grad(o, t, go.view_as(o))]
check("flash MHA fwd+bwd (dropout 0.1)", mha, 60)
