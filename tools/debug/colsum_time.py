"""Timing of spe_colsum_bf16_blocks at the qkv bias-gradient shape [8300, 1152] (SPE_COLSUM_RY: row slabs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
x = torch.randn(8300, 1152, device=dev).to(torch.bfloat16)
out = [torch.zeros(1152, device=dev)]
for _ in range(5):
    K.colsum_bf16_blocks(x, 1152, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    K.colsum_bf16_blocks(x, 1152, out)
e1.record(); torch.cuda.synchronize()
print("SPE_COLSUM_RY", os.environ.get("SPE_COLSUM_RY"), "%.2f us" % (e0.elapsed_time(e1) * 1e3 / 200))
