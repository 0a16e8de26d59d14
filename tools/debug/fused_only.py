"""Launch backward passes 1 and 2 of the talking-heads attention (spe_talking_fused modes 2, 3) and, when the library has it, the
q-major flash backward (spe_talking_flash_bwd) a few times at cfg2 shapes: the target of the rocprofv3 --pmc runs of
tools/debug/fused_pmc.sh."""
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
Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g)).to(dev)
dO = torch.randn(B, N, C, generator=g).to(dev)
scale = dh ** -0.5
v5 = qkv.view(B, N, 3, H, dh)
q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
nt = (N + 15) // 16
spw0, _ = K.fused_plan(B, N, 0)
spw, nwg = K.fused_plan(B, N, 2)
Qf, Kf = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16)])
Vf, dOf = K.attn_pack(v), K.attn_pack(dO.view(B, N, H, dh))
ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
ws_w = torch.zeros(nwg, 2 * (H * H + H), device=dev)
dST = K.score_blocks(B, H, N, dev)
K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
M, IL = K.attn_merge(ws, B, H, N, spw0, 0)
D = torch.zeros(B, H, N, device=dev)
modes = [int(m) for m in os.environ.get("MODES", "2,3").split(",")]
for _ in range(int(os.environ.get("REP", 3))):
    for mode in modes:
        K.talking_fused(mode, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, ws, ws_w, dST if mode == 3 else None, B, H, N, dh, 0.0, 7, 3)
torch.cuda.synchronize()
