"""Only the PV / dV contraction launches at cfg2 (for counter collection): 10 launches each of trans = 0 / 1, and 10 linear reads of the
same 554 MB by an ATen reduction as a yardstick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
H, N, dh, B = 8, int(os.environ.get("N", "4150")), 48, 2
g = torch.Generator().manual_seed(1)
C = H * dh
v = torch.randn(B, N, H, dh, generator=g).to(dev)
PT = K.score_blocks(B, H, N, dev)
PT.view(torch.int16).random_(0, 1000)
O = torch.empty(B, N, C, device=dev)
V16 = K.attn_pack16(v)
for _ in range(10):
    K.attn_contract(PT, V16, O.view(B, N, H, dh), False)
for _ in range(10):
    K.attn_contract(PT, V16, O.view(B, N, H, dh), True)
flat = PT.view(torch.int16).view(-1).view(torch.float32) if PT.numel() % 2 == 0 else None
z = torch.zeros_like(PT.view(-1)[: 1 << 20])
for _ in range(10):
    PT.view(-1).float().sum() if False else torch.sum(PT.view(torch.int32).view(-1))
torch.cuda.synchronize()
