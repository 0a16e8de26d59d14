"""Time the flash kernels alone at cfg2 shapes (ablation builds: SPE_HIP_LIB=build_ab/<name>.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "flash_only.py")).read().split("for _ in range")[0].split("import torch\nfrom spe_amd import kernels as K\n")[1])
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print(os.environ.get("SPE_HIP_LIB", "default"), "flash fwd + merge %.3f ms" % t(lambda: K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, 0.0, 0, 0, True, True)))
