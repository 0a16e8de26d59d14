"""Launch the key-major backward pass (spe_talking_bwdk_pass1) and pass 2 a few times at cfg2 shapes: the target of the rocprofv3 --pmc runs of
tools/debug/fused_pmc.sh (second argument)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
B, H, N, dh = int(os.environ.get("B", 2)), 8, int(os.environ.get("N", 4150)), 48
g = torch.Generator().manual_seed(1)
C = H * dh
qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g) / N).to(dev)
dO = torch.randn(B, N, C, generator=g).to(dev)
scale = dh ** -0.5
v5 = qkv.view(B, N, 3, H, dh)
q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
nt = (N + 15) // 16
Qf, Kf, Vf, K16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 32), (k, 1.0, 16)])
dO4 = dO.view(B, N, H, dh)
dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 32), (dO4, 1.0, 16)])
spw0, _ = K.fused_plan(B, N, 0)
ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
M, IL, c0 = K.attn_merge_rows(ws, bl, B, H, N, spw0)
dS = K.score_blocks(B, H, N, dev)
dq = torch.zeros(B, N, H, dh, device=dev)
dv = torch.zeros(B, N, H, dh, device=dev)
for _ in range(int(os.environ.get("REP", 3))):
    Drows, ws_w = K.talking_bwdk_pass1(Qf, dOf, dO16, Kf, Vf, Wl, Ww, bw, c0, None, dv, None, B, H, N, dh, 0.0)
    K.talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Drows, ws_w, dS, dq, None, scale, None, B, H, N, dh, 0.0)
torch.cuda.synchronize()
