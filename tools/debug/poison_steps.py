"""Which intermediate of the fused attention backward depends on uninitialised memory: the steps of ops._TalkingHeadsAttentionFused
replayed through spe_amd.kernels with the allocator's free blocks poisoned before every replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
K.set_precision("bf16")
H, N, dh, B, p = int(os.environ.get("HH", 4)), int(os.environ.get("N", 200)), 48, int(os.environ.get("B", 1)), float(os.environ.get("P", 0.05))
g = torch.Generator().manual_seed(5)
C = H * dh
qkv = (1.5 * torch.randn(B, N, 3 * C, generator=g)).to(dev)
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev)
bl = (0.1 * torch.randn(H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g)).to(dev)
dO = torch.randn(B, N, C, generator=g).to(dev)
scale = dh ** -0.5
nt = (N + 15) // 16
def poison(val):
    junk = [torch.full((1 << 26,), val, device=dev) for _ in range(4)]
    small = [torch.full((n,), val, device=dev) for n in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22) for _ in range(8)]
    del junk, small
def run():
    res = {}
    v5 = qkv.view(B, N, 3, H, dh); q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
    spw0, _ = K.fused_plan(B, N, 0)
    Qf, Kf, V16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32), (k, 1.0, 32), (v, 1.0, 16)])
    res.update(Qf=Qf, Kf=Kf, V16=V16)
    ws_stats = torch.empty((B * nt * 8 * H * 32,), device=dev, dtype=torch.float32)
    K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws_stats, None, None, B, H, N, dh, 0.0, 0, 0)
    M, IL = K.attn_merge(ws_stats, B, H, N, spw0, 0)
    res.update(M=M, IL=IL)
    Pd = K.score_blocks(B, H, N, dev)
    K.talking_fused(1, Qf, Kf, None, None, Wl, bl, Ww, bw, M, IL, None, None, None, Pd, B, H, N, dh, p, 7, 3)
    res.update(Pd=Pd.float())
    spw, nwg = K.fused_plan(B, N, 2)
    dO4 = dO.view(B, N, H, dh)
    Vf, dOf, dO16, K16, Q16 = K.attn_pack_multi([(v, 1.0, 32), (dO4, 1.0, 32), (dO4, 1.0, 16), (k, 1.0, 16), (q, 1.0, 16)])
    res.update(Vf=Vf, dOf=dOf, dO16=dO16, K16=K16, Q16=Q16)
    ws2 = torch.empty((B * nt * 8 * H * 32,), device=dev, dtype=torch.float32)
    ws_w = torch.empty((nwg, 2 * (H * H + H)), device=dev, dtype=torch.float32)
    K.talking_fused(2, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, None, ws2, ws_w, None, B, H, N, dh, p, 7, 3)
    res.update(ws_w_mode2=ws_w[:, H * H + H:].clone())
    D, _ = K.attn_merge(ws2, B, H, N, spw, 2)
    res.update(D=D)
    dS = K.score_blocks(B, H, N, dev)
    K.talking_fused(3, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, None, ws_w, dS, B, H, N, dh, p, 7, 3)
    res.update(ws_w_mode3=ws_w[:, :H * H + H].clone(), dS=dS.float())
    return {k_: (t.float().clone() if t.dtype != torch.float32 else t.clone()) for k_, t in res.items()}
ref = run()
for trial in range(int(os.environ.get("TRIALS", 30))):
    if os.environ.get("POISON", "1e30") != "none":
        poison(float(os.environ.get("POISON", "1e30")))
    r = run()
    for k_ in ref:
        a, b_ = r[k_], ref[k_]
        if not torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b_, nan=-7.0)):
            bad = (torch.nan_to_num(a, nan=-7.0) != torch.nan_to_num(b_, nan=-7.0))
            idx = bad.flatten().nonzero().flatten()
            print(f"trial {trial}: {k_} differs in {int(bad.sum())} of {bad.numel()} elements; first flat index {int(idx[0])}, shape {tuple(a.shape)}")
            break
print("done")
