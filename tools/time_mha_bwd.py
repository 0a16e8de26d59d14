"""dq / dkv kernel time of the flash MHA backward against the number of query tiles (fixed cost vs per-iteration cost)."""
import os, sys
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
    return a.elapsed_time(b) / n * 1e3


B, H, Lk, dk, dv = 2, 8, 4150, 96, 48
for Lq in (16, 64, 200, 400, 800):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, Lq, H, dk, generator=g).to(dev); k = torch.randn(B, Lk, H, dk, generator=g).to(dev)
    v = torch.randn(B, Lk, H, dv, generator=g).to(dev); go = torch.randn(B, Lq, H, dv, generator=g).to(dev)
    sc = dk ** -0.5 * K.LOG2E
    Qf, Kf, V16, Q16, K16, Vf = K.attn_pack_multi([(q, sc, 322 + K.F16), (k, 1.0, 322 + K.F16), (v, 1.0, 16 + K.F16), (q, sc, 16), (k, 1.0, 16), (v, 1.0, 322)])
    nch = K.mha_plan(B, H, Lq, Lk)
    O, lse, keep = K.mha_fwd(Qf, Kf, V16, None, B, H, Lq, Lk, dk, dv, nch, 0.0, 1, 2)
    D = (go * O.view(B, Lq, H, dv)).sum(-1).permute(0, 2, 1).contiguous()
    dOf, dO16 = K.attn_pack_multi([(go, 1.0, 322), (go, 1.0, 16)])
    tot = t(lambda: K.mha_bwd(Qf, Kf, Vf, dOf, K16, Q16, dO16, None, lse, D, None, B, H, Lq, Lk, dk, dv, nch, dk ** -0.5, 0.0))
    print("Lq %4d (ntq %2d) nch %2d: bwd total %.1f us ; fwd %.1f us" % (Lq, (Lq + 15) // 16, nch, tot,
          t(lambda: K.mha_fwd(Qf, Kf, V16, None, B, H, Lq, Lk, dk, dv, nch, 0.0, 1, 2))))
