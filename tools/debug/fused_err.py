"""Error of the fused talking-heads attention (current SPE_HIP_LIB) against fp64 at H=8, N=1100, dh=48."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K, ops
dev = torch.device("cuda:0")
K.set_precision("bf16")
H, N, dh, B = int(os.environ.get("HH", 8)), int(os.environ.get("N", 1100)), 48, 1
g = torch.Generator().manual_seed(5)
C = H * dh
qkv = (1.5 * torch.randn(B, N, 3 * C, generator=g)).to(dev).requires_grad_()
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
bl = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
bw = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
scale = dh ** -0.5
def ref(qkv, Wl, bl, Ww, bw):
    q, k, v = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    a = (q * scale) @ k.transpose(-2, -1)
    a = torch.nn.functional.linear(a.permute(0, 2, 3, 1), Wl, bl).permute(0, 3, 1, 2).softmax(-1)
    a = torch.nn.functional.linear(a.permute(0, 2, 3, 1), Ww, bw).permute(0, 3, 1, 2)
    return (a @ v).transpose(1, 2).reshape(B, N, C)
out = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, scale, 0.0, fused=True)
go = torch.randn(out.shape, generator=g).to(dev)
gr = torch.autograd.grad(out, (qkv, Wl, bl, Ww, bw), go)
dd = [t.detach().double().requires_grad_() for t in (qkv, Wl, bl, Ww, bw)]
r = ref(*dd); rg = torch.autograd.grad(r, dd, go.double())
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
print("out %.3e dqkv %.3e dWl %.3e dWw %.3e dbw %.3e dbl(abs/dWl) %.3e" % (rel(out, r), rel(gr[0], rg[0]), rel(gr[1], rg[1]), rel(gr[3], rg[3]), rel(gr[4], rg[4]), float(gr[2].abs().max() / gr[1].abs().max())))
