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
nt = (N + 15) // 16; ldq = nt * 16
spw0, _ = K.fused_plan(B, N, 0)
spw, nwg = K.fused_plan(B, N, 2)
print("nt", nt, "spw", spw, "nwg", nwg)
Qf, Kf, Vf, dOf = K.attn_pack(q, scale * K.LOG2E), K.attn_pack(k), K.attn_pack(v), K.attn_pack(dO.view(B, N, H, dh))
ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, ldq, 0.0, 0, 0)
M, IL = K.attn_merge(ws, B, H, N, spw0, 0)
# reference (bf16-rounded q,k like the kernel)
bf = lambda t: t.to(torch.bfloat16).double()
qd, kd, vd, dOd = bf(q * scale), bf(k), bf(v), bf(dO.view(B, N, H, dh))
S = torch.einsum("bqhd,bkhd->bhqk", qd, kd)
S1 = torch.einsum("gh,bhqk->bgqk", Wl.double(), S) + bl.double()[None, :, None, None]
P = S1.softmax(-1)
print("M err", (M.double() - S1.max(-1).values).abs().max().item(), "IL rel", ((IL.double() - 1 / (S1 - S1.max(-1, keepdim=True).values).exp().sum(-1)).abs().max() / IL.abs().max()).item())
P1 = torch.einsum("gh,bhqk->bgqk", Ww.double(), P) + bw.double()[None, :, None, None]
PT = torch.empty(B, H, ldq, ldq, device=dev, dtype=torch.bfloat16)
K.talking_fused(1, Qf, Kf, None, None, Wl, bl, Ww, bw, M, IL, None, None, None, PT, B, H, N, dh, ldq, 0.0, 0, 0)
print("PT rel", ((PT[:, :, :N, :N].double() - P1).norm() / P1.norm()).item())
dP1 = torch.einsum("bqhd,bkhd->bhqk", dOd, vd)
dP = torch.einsum("gh,bgqk->bhqk", Ww.double(), dP1)
Dref = (dP * P).sum(-1)
dS1 = P * (dP - Dref[..., None])
dSref = torch.einsum("gh,bgqk->bhqk", Wl.double(), dS1)
ws_w = torch.zeros(nwg, 2 * (H * H + H), device=dev)
ws.zero_()
K.talking_fused(2, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, None, ws, ws_w, None, B, H, N, dh, ldq, 0.0, 0, 0)
D, _ = K.attn_merge(ws, B, H, N, spw, 2)
print("D finite", torch.isfinite(D).all().item(), "D rel", ((D.double() - Dref).norm() / Dref.norm()).item())
dWw_ref = torch.einsum("bgqk,bhqk->gh", dP1, P)
gsum = ws_w.sum(0)
hh = H * H
print("dWw rel", ((gsum[hh + H:2 * hh + H].view(H, H).double() - dWw_ref).norm() / dWw_ref.norm()).item(),
      "dbw rel", ((gsum[2 * hh + H:].double() - dP1.sum((0, 2, 3))).norm() / dP1.sum((0, 2, 3)).norm()).item())
dST = torch.empty(B, H, ldq, ldq, device=dev, dtype=torch.bfloat16)
K.talking_fused(3, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, None, ws_w, dST, B, H, N, dh, ldq, 0.0, 0, 0)
got = dST[:, :, :N, :N].double()
print("dS finite", torch.isfinite(got).all().item(), "dS rel", ((got - dSref).norm() / dSref.norm()).item())
bad = (~torch.isfinite(got)) | ((got - dSref).abs() > 1e-2 * dSref.abs().max())
print("bad count", bad.sum().item(), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()[:10]
    print(idx.tolist())
gsum = ws_w.sum(0)
dWl_ref = torch.einsum("bgqk,bhqk->gh", dS1, S)
print("dWl rel", ((gsum[:hh].view(H, H).double() - dWl_ref).norm() / dWl_ref.norm()).item())
