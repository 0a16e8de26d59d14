"""q-major flash-skeleton backward passes (spe_talking_bwdq_pass1 / _pass2) against the round-3 kernels they replace (spe_talking_fused
modes 2 / 3 + the dQ contraction) on the same fragments, statistics and dropout flags; then both timed.
  python tools/debug/bwdq_check.py            (cfg2 tokens; N, B, H, DH, PDROP from the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K

dev = torch.device("cuda:0")


def run(B, H, N, dh, p_drop, time_it=False):
    g = torch.Generator().manual_seed(1)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g) / N).to(dev)
    dO = torch.randn(B, N, C, generator=g).to(dev)
    scale = dh ** -0.5
    v5 = qkv.view(B, N, 3, H, dh)
    q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
    nt = (N + 15) // 16
    Qf, Kf, V16, Vf, K16, Q16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16),
                                                   (v, 1.0, 32), (k, 1.0, 16), (q, 1.0, 16)])
    dO4 = dO.view(B, N, H, dh)
    dOf, dO16 = K.attn_pack_multi([(dO4, 1.0, 32), (dO4, 1.0, 16)])
    spw0, _ = K.fused_plan(B, N, 0)
    spw, nwg = K.fused_plan(B, N, 2)
    ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
    K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
    M, IL, c0 = K.attn_merge_rows(ws, bl, B, H, N, spw0)
    seed, off = 7, 3
    bits = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, seed, off, want_bits=True)[3] if p_drop > 0 else None
    nw = 2 * (H * H + H)

    def old():
        ws_w = torch.zeros(nwg, nw, device=dev)
        K.talking_fused(2, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, None, ws, ws_w, None, B, H, N, dh, p_drop, seed, off, keepbits=bits)
        D, _ = K.attn_merge(ws, B, H, N, spw, 2)
        dS = K.score_blocks(B, H, N, dev)
        K.talking_fused(3, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, None, ws_w, dS, B, H, N, dh, p_drop, seed, off, keepbits=bits)
        dq = torch.zeros(B, N, H, dh, device=dev)
        K.attn_contract(dS, K16, dq, False, alpha=scale)
        return D, dS, dq, ws_w.sum(0)

    def new():
        Drows, ws_w = K.talking_bwdq_pass1(Qf, dOf, Kf, Vf, Wl, Ww, c0, bits, B, H, N, dh, p_drop)
        dS = K.score_blocks(B, H, N, dev)
        dq = torch.zeros(B, N, H, dh, device=dev)
        K.talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Drows, ws_w, dS, dq, None, scale, bits, B, H, N, dh, p_drop)
        return Drows[:, :N].permute(0, 2, 1).contiguous(), dS, dq, ws_w.sum(0)

    Do, dSo, dqo, wo = old()
    Dn, dSn, dqn, wn = new()
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    hh = H * H
    names = [("D", Dn, Do), ("dS", dSn.float(), dSo.float()), ("dq", dqn, dqo), ("dWl", wn[:hh], wo[:hh]), ("dbl(abs)", None, None),
             ("dWw", wn[hh + H:2 * hh + H], wo[hh + H:2 * hh + H]), ("dbw", wn[2 * hh + H:], wo[2 * hh + H:])]
    out = {}
    for nm, a, b in names:
        if a is None:
            out[nm] = (wn[hh:hh + H].abs().max().item(), wo[hh:hh + H].abs().max().item())
        else:
            out[nm] = rel(a, b)
    fin = all(torch.isfinite(t).all().item() for t in (Dn, dqn, wn, dSn.float()))
    print(f"B={B} H={H} N={N} dh={dh} p={p_drop}: finite={fin} " + " ".join(f"{k}={v if isinstance(v, tuple) else round(v, 6)}" for k, v in out.items()), flush=True)
    # run-to-run determinism of the new path
    Dn2, dSn2, dqn2, wn2 = new()
    torch.cuda.synchronize()
    print("   bitwise rerun:", torch.equal(Dn, Dn2), torch.equal(dSn.view(torch.int16), dSn2.view(torch.int16)), torch.equal(dqn, dqn2), torch.equal(wn, wn2), flush=True)
    if time_it:
        def t(fn, n=5):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        ws_w = torch.zeros(nwg, nw, device=dev)
        D = Do
        dS = K.score_blocks(B, H, N, dev)
        dq = torch.zeros(B, N, H, dh, device=dev)
        print("   old pass1 %.3f ms" % t(lambda: K.talking_fused(2, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, None, ws, ws_w, None, B, H, N, dh, p_drop, seed, off, keepbits=bits)))
        print("   old pass2 %.3f ms" % t(lambda: K.talking_fused(3, Qf, Kf, Vf, dOf, Wl, bl, Ww, bw, M, IL, D, None, ws_w, dS, B, H, N, dh, p_drop, seed, off, keepbits=bits)))
        print("   old dQ    %.3f ms" % t(lambda: K.attn_contract(dS, K16, dq, False, alpha=scale)))
        Drows, ws_w2 = K.talking_bwdq_pass1(Qf, dOf, Kf, Vf, Wl, Ww, c0, bits, B, H, N, dh, p_drop)
        print("   new pass1 %.3f ms (with its merge)" % t(lambda: K.talking_bwdq_pass1(Qf, dOf, Kf, Vf, Wl, Ww, c0, bits, B, H, N, dh, p_drop)))
        print("   new pass2 %.3f ms (with dQ and its merge)" % t(lambda: K.talking_bwdq_pass2(Qf, dOf, Kf, Vf, K16, Wl, Ww, c0, Drows, ws_w2, dS, dq, None, scale, bits, B, H, N, dh, p_drop)), flush=True)


if __name__ == "__main__":
    if os.environ.get("QUICK"):
        run(2, 8, int(os.environ.get("N", 4150)), 48, float(os.environ.get("PDROP", 0)), time_it=True)
        sys.exit(0)
    run(1, 8, 100, 48, 0.0)
    run(2, 4, 196, 48, 0.0)
    run(2, 8, 1100, 48, 0.1)
    run(1, 8, 520, 64, 0.0) if K.bwdq_supported(8, 64) else None
    run(1, 4, 300, 32, 0.05)
    run(2, 8, 400, 16, 0.0)
    run(2, 8, 4150, 48, 0.0, time_it=True)
    run(2, 8, 4150, 48, 0.05, time_it=True)
