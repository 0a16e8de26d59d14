"""The five attention kernels of one backbone block in isolation at cfg2 (B = 2, H = 8, N = 4150, dh = 48) or another shape, timed with HIP events on the
launch stream (kernel + its merges per entry point):

    python tools/debug/attn_time.py [B N H dh [p_drop]]                         # the shipped library
    SPE_HIP_LIB=build_ab/<name>.so python tools/debug/attn_time.py ...          # a -DSPE_ABLATE variant built by tools/ab.py

Reference: models/cait.py:377-389 and its autograd."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spe_amd import kernels as K  # noqa: E402

STAMPS = "--stamps" in sys.argv
a = [float(x) for x in sys.argv[1:] if not x.startswith("--")]
B, N, H, dh = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (2, 4150, 8, 48)
p_drop = a[4] if len(a) >= 5 else 0.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
C = H * dh
qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g) / N).to(dev)
dO = torch.randn(B, N, C, generator=g).to(dev)
scale = dh ** -0.5
v5 = qkv.view(B, N, 3, H, dh)
q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
Qf, Kf, V16, Vf, K16, Q16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16), (v, 1.0, 32), (k, 1.0, 16), (q, 1.0, 16)])
dO4 = dO.view(B, N, H, dh)
dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 32), (dO4, 1.0, 16)])
nt = (N + 15) // 16
spw0, _ = K.fused_plan(B, N)
ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
dqkv = torch.empty(B, N, 3 * C, device=dev, dtype=torch.bfloat16)
d5 = dqkv.view(B, N, 3, H, dh)
dS = K.score_blocks(B, H, N, dev)
state = {}


def stats():
    K.talking_stats(Qf, Kf, Wl, bl, ws, B, H, N, dh)
    state["c0"] = K.attn_merge_rows(ws, bl, B, H, N, spw0)[2]


def fwd():
    state["bits"] = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, state["c0"], B, H, N, dh, p_drop, 7, 3, True, True, want_bits=True)[3]


def bwdk():
    state["D"], state["ws_w"] = K.talking_bwdk_pass1(Qf, dOf, dO16, Kf, Vf, Wl, Ww, bw, state["c0"], state["bits"], None, d5[:, :, 2], B, H, N, dh, p_drop)


def bwdq():
    K.talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, state["c0"], state["D"], state["ws_w"], dS, None, d5[:, :, 0], scale, state["bits"], B, H, N, dh, p_drop)


def dk():
    K.attn_contract(dS, Q16, None, True, alpha=scale, out16=d5[:, :, 1])


steps = [("statistics + merge", stats), ("flash forward + merge", fwd), ("key-major backward (D, dWw, dbw, dV) + merges", bwdk),
         ("query-major backward (dS, dWl, dbl, dQ) + merge", bwdq), ("dK contraction", dk)]
for _, f in steps:
    f()
torch.cuda.synchronize()
REP = 10
tot = 0.0
print(f"library {K.lib.LIBPATH}; B={B} N={N} H={H} dh={dh} p_drop={p_drop}")
for name, f in steps:
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(REP)]
    for a_, b_ in e:
        a_.record(); f(); b_.record()
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in e)
    tot += ms[REP // 2]
    print(f"  {name:52s} median {ms[REP // 2] * 1e3:8.1f} us   min {ms[0] * 1e3:8.1f} us")
print(f"  {'sum of medians':52s}        {tot * 1e3:8.1f} us")
if STAMPS:          # key-major kernel: region sums in row 0 of the weight-gradient workspace
    bwdk(); torch.cuda.synchronize()
    st = state["ws_w"].view(-1)[:14].view(torch.int64).tolist()
    n = max(st[6], 1)
    names = ["admission (vmcnt + barrier + next tile loads of waves 0, 1)", "D flush of the previous q-tile", "region 1: S + fp32 mix of heads 0-3 | exp2 (+ tile loads of waves 2, 3)",
             "region 2: S + mix of heads 4-7 | D terms, P' mix", "region 3: dP' of heads 0-3 | outer product", "region 4: dP' of heads 4-7 | dV product"]
    print(f"  key-major kernel, s_memtime ticks per pipelined step of workgroup 0 / wave 0 ({n} steps):")
    for k in range(6):
        print(f"    {names[k]:72s} {st[k] / n:9.1f}")
    print(f"    {'total':72s} {sum(st[:6]) / n:9.1f}")
if STAMPS:          # a -DFLB_DBG_STAMP build (tools/ab.py) leaves the region sums of workgroup 0 / wave 0 in the first bytes of dS
    bwdq(); torch.cuda.synchronize()
    st = dS.view(-1)[:24].view(torch.int64).tolist()
    n = max(st[5], 1)
    names = ["admission (vmcnt + barrier + next tile loads)", "region 1: S + fp32 mix of heads 0-3 | exp2", "region 2: S + mix of heads 4-7 | dS' = P (dP - D), dbl, packs",
             "region 3: dP' + dP mix of heads 0-3 | outer product, dS mix", "region 4: dP' + dP mix of heads 4-7 | dS store, dQ product"]
    print(f"  query-major kernel, s_memtime ticks per pipelined step of workgroup 0 / wave 0 ({n} steps; 100 MHz ticks x 10 ns or core cycles: compare with the total):")
    for k in range(5):
        print(f"    {names[k]:64s} {st[k] / n:9.1f}")
    print(f"    {'total':64s} {sum(st[:5]) / n:9.1f}")
