import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
H, N, dh, B = 4, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 48, 2
g = torch.Generator().manual_seed(1)
x = torch.randn(B, N, H, dh, generator=g).to(dev)
nt = (N + 15) // 16
rem = dh % 32; full = dh // 32 + (1 if rem > 16 else 0); tail = 1 if 0 < rem <= 16 else 0
ref = torch.zeros(B, H, nt, full * 512 + tail * 256)
xc = x.cpu().to(torch.bfloat16).float()
for b in range(B):
    for h in range(H):
        for t in range(nt):
            for st in range(full):
                for ln in range(64):
                    row = t * 16 + (ln & 15)
                    for i in range(8):
                        d = st * 32 + (ln >> 4) * 8 + i
                        if row < N and d < dh:
                            ref[b, h, t, st * 512 + ln * 8 + i] = xc[b, row, h, d]
            if tail:
                for ln in range(64):
                    row = t * 16 + (ln & 15)
                    for i in range(4):
                        d = full * 32 + (ln >> 4) * 4 + i
                        if row < N and d < dh:
                            ref[b, h, t, full * 512 + ln * 4 + i] = xc[b, row, h, d]
a = K.attn_pack(x, 1.0).float().cpu()
m = K.attn_pack_multi([(x, 1.0, 32)])[0].float().cpu()
print("pack single max err", (a - ref).abs().max().item(), "multi", (m - ref).abs().max().item())
