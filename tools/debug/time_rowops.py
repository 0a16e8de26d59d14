"""Row-wise kernels at the backbone's activation size (8300 x 384) against a plain device copy of the same bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
R, C = 8300, 384
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
x = torch.randn(R, C, device=dev); dy = torch.randn(R, C, device=dev); g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-6)
dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
o = torch.empty_like(x)
print("copy 12.7 MB -> 12.7 MB      %.1f us" % t(lambda: o.copy_(x)))
print("layernorm_fwd               %.1f us" % t(lambda: K.layernorm_fwd(x, g, b, 1e-6)))
print("layernorm_fwd + y16         %.1f us" % t(lambda: K.layernorm_fwd(x, g, b, 1e-6, want16=True)))
print("layernorm_bwd               %.1f us" % t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg_out=dg, db_out=db)))
print("cvt_bf16 (row-major)        %.1f us" % t(lambda: K.cvt_bf16(x, True, False)))
print("cvt_bf16 + colsum           %.1f us" % t(lambda: K.cvt_bf16(x, True, False, colsum_out=db)))
x4 = torch.randn(R, 4 * C, device=dev)
print("cvt_bf16 8300x1536          %.1f us" % t(lambda: K.cvt_bf16(x4, True, False)))
# direct C-ABI calls on preallocated buffers (the Python wrappers above allocate their outputs: host-bound at ~10 us per call)
from spe_amd import lib
import ctypes
P = lambda t_: ctypes.c_void_p(t_.data_ptr()) if t_ is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
y = torch.empty_like(x); mean = torch.empty(R, device=dev); rstd = torch.empty(R, device=dev); y16 = torch.empty(R, C, device=dev, dtype=torch.bfloat16)
dx = torch.empty_like(x)
print("direct spe_layernorm_fwd          %.1f us" % t(lambda: lib.call("spe_layernorm_fwd", P(x), P(g), P(b), P(y), P(mean), P(rstd), R, C, 1e-6, None, st), 200))
print("direct spe_layernorm_fwd + y16    %.1f us" % t(lambda: lib.call("spe_layernorm_fwd", P(x), P(g), P(b), P(y), P(mean), P(rstd), R, C, 1e-6, P(y16), st), 200))
print("direct spe_layernorm_bwd          %.1f us" % t(lambda: lib.call("spe_layernorm_bwd", P(dy), P(x), P(g), P(mean), P(rstd), P(dx), P(dg), P(db), R, C, None, st), 200))
print("copy (200 calls)                  %.1f us" % t(lambda: o.copy_(x), 200))
