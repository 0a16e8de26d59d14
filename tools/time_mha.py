"""Time the flash MHA kernels against the materialising path (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K, ops
dev = torch.device("cuda:0")


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, (B, Lq, Lk, H, dk, dv) in {"encoder": (2, 4150, 4150, 8, 48, 48), "cross": (2, 200, 4150, 8, 96, 48), "self": (4, 100, 100, 8, 48, 48)}.items():
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, Lq, H, dk, generator=g).to(dev).requires_grad_()
    k = torch.randn(B, Lk, H, dk, generator=g).to(dev).requires_grad_()
    v = torch.randn(B, Lk, H, dv, generator=g).to(dev).requires_grad_()
    go = torch.randn(B, Lq, H * dv, generator=g).to(dev)
    for flash in (True, False):
        ops.FLASH_MHA = flash
        f = lambda: ops.attention(q, k, v, None, scale=dk ** -0.5, p_drop=0.1)[0]
        tf = t(f)
        o = f()
        tb = t(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
        print(f"{name:8s} flash={flash}: fwd {tf:.3f} ms  bwd {tb:.3f} ms")
    ops.FLASH_MHA = True
    sc = dk ** -0.5 * K.LOG2E
    Qf, Kf, V16, Q16, K16, Vf = K.attn_pack_multi([(q, sc, 322 + K.F16), (k, 1.0, 322 + K.F16), (v, 1.0, 16 + K.F16), (q, sc, 16), (k, 1.0, 16), (v, 1.0, 322)])
    nch = K.mha_plan(B, H, Lq, Lk)
    print("   nch", nch, "pack6 %.3f ms" % t(lambda: K.attn_pack_multi([(q, sc, 322 + K.F16), (k, 1.0, 322 + K.F16), (v, 1.0, 16 + K.F16), (q, sc, 16), (k, 1.0, 16), (v, 1.0, 322)])))
    print("   fwd+merge %.3f ms" % t(lambda: K.mha_fwd(Qf, Kf, V16, None, B, H, Lq, Lk, dk, dv, nch, 0.1, 1, 2)))
    O, lse, keep = K.mha_fwd(Qf, Kf, V16, None, B, H, Lq, Lk, dk, dv, nch, 0.1, 1, 2)
    dO4 = go.view(B, Lq, H, dv)
    D = (dO4 * O.view(B, Lq, H, dv)).sum(-1).permute(0, 2, 1).contiguous()
    dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 322), (dO4, 1.0, 16)])
    print("   bwd (dq + dkv) %.3f ms" % t(lambda: K.mha_bwd(Qf, Kf, Vf, dOf, K16, Q16, dO16, None, lse, D, keep, B, H, Lq, Lk, dk, dv, nch, dk ** -0.5, 0.1)))
    print("   bwd p=0        %.3f ms ; fwd p=0 %.3f ms" % (t(lambda: K.mha_bwd(Qf, Kf, Vf, dOf, K16, Q16, dO16, None, lse, D, None, B, H, Lq, Lk, dk, dv, nch, dk ** -0.5, 0.0)),
                                                      t(lambda: K.mha_fwd(Qf, Kf, V16, None, B, H, Lq, Lk, dk, dv, nch, 0.0, 1, 2))))
