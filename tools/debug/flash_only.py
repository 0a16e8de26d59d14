"""Launch the flash forward (and, when present, backward) kernels a few times at cfg2 shapes: the target of rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
B, H, N, dh = 2, 8, int(os.environ.get("N", 4150)), 48
g = torch.Generator().manual_seed(1)
C = H * dh
qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g)).to(dev)
scale = dh ** -0.5
v5 = qkv.view(B, N, 3, H, dh)
q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
nt = (N + 15) // 16
Qf, Kf, V16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16)])
spw0, _ = K.fused_plan(B, N, 0)
ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
M, IL = K.attn_merge(ws, B, H, N, spw0, 0)
c0 = K.flash_rows(M, IL, bl, B, H, N, 0)
dO = torch.randn(B, N, C, generator=g).to(dev)
dO16 = K.attn_pack_multi([(dO.view(B, N, H, dh), 1.0, 16)])[0]
dqkv = torch.zeros(B, N, 3 * C, device=dev)
for _ in range(int(os.environ.get("REP", 3))):
    K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, 0.0, 0, 0, True, True)
    K.talking_flash_dv(Qf, Kf, dO16, Wl, Ww, bw, c0, dqkv.view(B, N, 3, H, dh)[:, :, 2], 0.0, 0, 0)
torch.cuda.synchronize()
