"""Developer check of the flash-style talking-heads kernels against the materialising fused path (same operands, same
statistics): forward O, and - once built - the backward gradients.  Prints max / norm-relative differences and timings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spe_amd import kernels as K

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def case(B, H, N, dh, p_drop=0.0, time=False):
    g = torch.Generator().manual_seed(1)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g)).to(dev)
    scale = dh ** -0.5
    v5 = qkv.view(B, N, 3, H, dh)
    q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
    nt = (N + 15) // 16
    Qf, Kf, V16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16)])
    spw0, _ = K.fused_plan(B, N, 0)
    ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
    K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
    M, IL = K.attn_merge(ws, B, H, N, spw0, 0)
    seed, off = 7, 3
    # reference: materialising write pass + streaming contraction
    Pd = K.score_blocks(B, H, N, dev, torch.float16)
    K.talking_fused(1, Qf, Kf, None, None, Wl, bl, Ww, bw, M, IL, None, None, None, Pd, B, H, N, dh, p_drop, seed, off)
    Oref = torch.empty(B, N, C, device=dev)
    K.attn_contract(Pd, V16, Oref.view(B, N, H, dh), False, alpha=1.0 / K.PD_SCALE)
    c0 = K.flash_rows(M, IL, bl, B, H, N, 0)
    O, O16, O16lo = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, seed, off, True, True)
    # dV = P'd^T dO: streaming contraction of the stored P'd against the flash dV pass
    dO = torch.randn(B, N, C, generator=g).to(dev)
    dO16 = K.attn_pack_multi([(dO.view(B, N, H, dh), 1.0, 16)])[0]
    dref = torch.zeros(B, N, 3 * C, device=dev); dnew = torch.zeros(B, N, 3 * C, device=dev)
    K.attn_contract(Pd, dO16, dref.view(B, N, 3, H, dh)[:, :, 2], True, alpha=1.0 / K.PD_SCALE)
    K.talking_flash_dv(Qf, Kf, dO16, Wl, Ww, bw, c0, dnew.view(B, N, 3, H, dh)[:, :, 2], p_drop, seed, off)
    torch.cuda.synchronize()
    errv = (dnew - dref).norm() / dref.norm()
    print(f"      dV rel {errv.item():.3e} max {float((dnew - dref).abs().max()):.3e} (|dV| max {float(dref.abs().max()):.2f})")
    err = (O - Oref).norm() / Oref.norm()
    e16 = (O16.float().view_as(O) + O16lo.float().view_as(O) - O).abs().max() / O.abs().max()
    print(f"B={B} H={H} N={N} dh={dh} p={p_drop}: O rel {err.item():.3e} max {float((O - Oref).abs().max()):.3e}  hi+lo residual {e16.item():.2e}"
          f"  finite {bool(torch.isfinite(O).all())}")
    if time:
        t_old = timeit(lambda: (K.talking_fused(1, Qf, Kf, None, None, Wl, bl, Ww, bw, M, IL, None, None, None, Pd, B, H, N, dh, p_drop, seed, off),
                                K.attn_contract(Pd, V16, Oref.view(B, N, H, dh), False, alpha=1.0 / K.PD_SCALE)))
        t_new = timeit(lambda: K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, seed, off, True, True))
        t_rows = timeit(lambda: K.flash_rows(M, IL, bl, B, H, N, 0))
        t_dvo = timeit(lambda: K.attn_contract(Pd, dO16, dref.view(B, N, 3, H, dh)[:, :, 2], True, alpha=1.0 / K.PD_SCALE))
        t_dvn = timeit(lambda: K.talking_flash_dv(Qf, Kf, dO16, Wl, Ww, bw, c0, dnew.view(B, N, 3, H, dh)[:, :, 2], p_drop, seed, off))
        print(f"   dV contraction {t_dvo:.3f} ms ; flash dV (+ merge) {t_dvn:.3f} ms")
        print(f"   write pass + PV contraction {t_old:.3f} ms ; flash forward (+ merge) {t_new:.3f} ms ; row constants {t_rows:.3f} ms")
    return max(err.item(), errv.item() / 100)


if __name__ == "__main__":
    worst = 0.0
    for (B, H, N, dh) in () if os.environ.get("QUICK") else ((1, 4, 12, 8), (2, 4, 35, 8), (2, 4, 196, 48), (1, 8, 130, 48), (2, 8, 1100, 48), (1, 4, 300, 32), (1, 4, 257, 64), (1, 4, 77, 24)):
        worst = max(worst, case(B, H, N, dh))
        worst = max(worst, case(B, H, N, dh, p_drop=0.1))
    case(2, 8, 4150, 48, time=True)
    case(1, 8, 6200, 48, time=True)
    case(2, 8, 4150, 48, p_drop=0.05, time=True)
    print("worst", worst)
