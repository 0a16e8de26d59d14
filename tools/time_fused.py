import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spe_amd import kernels as K
dev = torch.device("cuda:0")
import os
H, N, dh, B = 8, int(os.environ.get("N", "4150")), int(os.environ.get("DH", "48")), int(os.environ.get("B", "2"))
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
PT = K.score_blocks(B, H, N, dev)
dST = K.score_blocks(B, H, N, dev)
O = torch.empty(B, N, C, device=dev)
K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
M, IL = K.attn_merge(ws, B, H, N, spw0, 0)
D = torch.zeros(B, H, N, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
PD = float(os.environ.get("PDROP", "0"))
f = lambda mode, out=None: K.talking_fused(mode, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, ws, ws_w, out, B, H, N, dh, PD if mode else 0.0, 7, 3)
print("pack q", t(lambda: K.attn_pack(q, scale * K.LOG2E)))
for mode, out in ((0, None), (1, PT), (2, None), (3, dST)):
    print("mode", mode, "%.3f ms" % t(lambda: f(mode, out)))
dq = torch.empty_like(qkv)
d5 = dq.view(B, N, 3, H, dh)
print("pack16 v %.3f ms" % t(lambda: K.attn_pack16(v)))
V16, dO16 = K.attn_pack16(v), K.attn_pack16(dO.view(B, N, H, dh))
print("PV  contract   %.3f ms" % t(lambda: K.attn_contract(PT, V16, O.view(B, N, H, dh), False)))
print("dV  contract^T %.3f ms" % t(lambda: K.attn_contract(PT, dO16, d5[:, :, 2], True)))
print("dQ  contract   %.3f ms" % t(lambda: K.attn_contract(dST, V16, d5[:, :, 0], False, alpha=scale)))
print("dK  contract^T %.3f ms" % t(lambda: K.attn_contract(dST, dO16, d5[:, :, 1], True, alpha=scale)))
mb = PT.numel() * 2 / 1e6
print("score tensor %.0f MB" % mb)
tp = t(lambda: (f(1, PT), K.attn_contract(PT, V16, O.view(B, N, H, dh), False)))
print("mode1 + PV contract pair %.3f ms" % tp)
tp = t(lambda: (f(3, dST), K.attn_contract(dST, V16, d5[:, :, 0], False, alpha=scale), K.attn_contract(dST, dO16, d5[:, :, 1], True, alpha=scale)))
print("mode3 + dQ + dK contract triple %.3f ms" % tp)
