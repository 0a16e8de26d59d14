"""Stage-by-stage check of the fused talking-heads kernels against an fp64 restatement (bf16-rounded operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
H, N, dh, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = torch.Generator().manual_seed(1)
C = H * dh
qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g)).to(dev)
dO = torch.randn(B, N, C, generator=g).to(dev)
scale = dh ** -0.5
v5 = qkv.view(B, N, 3, H, dh)
q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
nt = (N + 15) // 16
spw0, _ = K.fused_plan(B, N, 0)
spw, nwg = K.fused_plan(B, N, 2)
print("nt", nt, "spw", spw, "nwg", nwg)
Qf, Kf = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16)])
Vf, dOf = K.attn_pack(v), K.attn_pack(dO.view(B, N, H, dh))
ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
M, IL = K.attn_merge(ws, B, H, N, spw0, 0)
bf = lambda t: t.to(torch.bfloat16).double()
qd, kd, vd, dOd = bf(q * scale * K.LOG2E) / K.LOG2E, bf(k), bf(v), bf(dO.view(B, N, H, dh))
S = torch.einsum("bqhd,bkhd->bhqk", qd, kd)
S1 = torch.einsum("gh,bhqk->bgqk", Wl.double(), S) + bl.double()[None, :, None, None]
P = S1.softmax(-1)
print("M finite", torch.isfinite(M).all().item(), "M err (log2 domain)", (M.double() - S1.max(-1).values * K.LOG2E).abs().max().item(),
      "IL rel", ((IL.double() - 1 / (S1 - S1.max(-1, keepdim=True).values).exp().sum(-1)).abs().max() / IL.abs().max()).item())
P1 = torch.einsum("gh,bhqk->bgqk", Ww.double(), P) + bw.double()[None, :, None, None]
PT = K.score_blocks(B, H, N, dev)
K.talking_fused(1, Qf, Kf, None, None, Wl, bl, Ww, bw, M, IL, None, None, None, PT, B, H, N, dh, 0.0, 0, 0)
# blocks [B,H,qt,kt,lane,4] -> dense [B,H,q,key]: lane l = (q = l&15, keys 4*(l>>4)+i)
blk = PT.double().view(B, H, nt, nt, 4, 16, 4)            # [.., kgrp, qlocal, i]
dense = blk.permute(0, 1, 2, 5, 3, 4, 6).reshape(B, H, nt * 16, nt * 16)
print("P'd rel", ((dense[:, :, :N, :N] - P1).norm() / P1.norm()).item())
