import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (64, 512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    s = t(lambda: y.copy_(x)); print(f"torch copy {mb} MB: {2*n*4/s/1e9:.0f} GB/s")
    s = t(lambda: K.add_rows(x, x)); print(f"spe add_rows {mb} MB (2 reads 1 write): {3*n*4/s/1e9:.0f} GB/s")
    s = t(lambda: x.sum()); print(f"torch sum (read only) {mb} MB: {n*4/s/1e9:.0f} GB/s")
    s = t(lambda: y.zero_()); print(f"torch zero (write only) {mb} MB: {n*4/s/1e9:.0f} GB/s")
x = torch.randn(8300, 384, device=dev); g = torch.ones(384, device=dev)
s = t(lambda: K.layernorm_fwd(x, g, g, 1e-6)); print(f"ln fwd 8300x384: {s*1e6:.1f} us {2*x.numel()*4/s/1e9:.0f} GB/s")
x = torch.randn(8300*8, 384, device=dev)
s = t(lambda: K.layernorm_fwd(x, g, g, 1e-6)); print(f"ln fwd 66400x384: {s*1e6:.1f} us {2*x.numel()*4/s/1e9:.0f} GB/s")
