"""Round-5 GPU tests: the query-major backward passes of the talking-heads attention on the flash skeleton (csrc/attn_flash_bwd.hip,
reference: the autograd of models/cait.py:377-389) against the round-3 kernels they replace, through the C-ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _inputs(B, H, N, dh, p_drop, dev):
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(1)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g) / N).to(dev)
    dO = torch.randn(B, N, C, generator=g).to(dev)
    scale = dh ** -0.5
    v5 = qkv.view(B, N, 3, H, dh)
    q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
    Qf, Kf, V16, Vf, K16 = K.attn_pack_multi([(q, scale * K.LOG2E, 32 + K.F16), (k, 1.0, 32 + K.F16), (v, 1.0, 16 + K.F16), (v, 1.0, 32), (k, 1.0, 16)])
    dOf = K.attn_pack_multi([(dO.view(B, N, H, dh), 1.0, 32)])[0]
    nt = (N + 15) // 16
    spw0, _ = K.fused_plan(B, N, 0)
    ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
    K.talking_fused(0, Qf, Kf, None, None, Wl, bl, Ww, bw, None, None, None, ws, None, None, B, H, N, dh, 0.0, 0, 0)
    M, IL, c0 = K.attn_merge_rows(ws, bl, B, H, N, spw0)
    seed, off = 7, 3
    bits = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p_drop, seed, off, want_bits=True)[3] if p_drop > 0 else None
    return dict(Qf=Qf, Kf=Kf, Vf=Vf, K16=K16, dOf=dOf, Wl=Wl, bl=bl, Ww=Ww, bw=bw, M=M, IL=IL, c0=c0, bits=bits, seed=seed, off=off, scale=scale, ws=ws)


@pytest.mark.parametrize("B,H,N,dh,p_drop", [(1, 8, 100, 48, 0.0), (2, 4, 196, 48, 0.0), (2, 8, 1100, 48, 0.1), (1, 4, 300, 32, 0.05),
                                               (2, 8, 400, 16, 0.0), (1, 8, 2070, 48, 0.0)])
def test_flash_skeleton_backward_passes_match_round3_kernels(dev, B, H, N, dh, p_drop):
    """spe_talking_bwdq_pass1 / _pass2 (one wave per SIMD, Q / dO fragments and the dQ accumulators in AccVGPRs, dQ accumulated in pass 2)
    against spe_talking_fused modes 2 / 3 + the dQ contraction on the same fragments, statistics and dropout flags: D, dS blocks, dQ, dWl,
    dWw, dbw agree to rounding (the two paths sum in different orders and round S once more / once less), dbl is noise around its exact
    value 0 in both; ragged N, both head counts, every head-dim decomposition; and the new path is bitwise reproducible run to run."""
    from spe_amd import kernels as K
    if not K.bwdq_supported(H, dh):
        pytest.skip("shape not on the flash-skeleton backward")
    x = _inputs(B, H, N, dh, p_drop, dev)
    spw, nwg = K.fused_plan(B, N, 2)
    nw = 2 * (H * H + H)

    def old():
        ws_w = torch.zeros(nwg, nw, device=dev)
        K.talking_fused(2, x["Qf"], x["Kf"], x["Vf"], x["dOf"], x["Wl"], x["bl"], x["Ww"], x["bw"], x["M"], x["IL"], None, x["ws"], ws_w, None, B, H, N, dh,
                        p_drop, x["seed"], x["off"], keepbits=x["bits"])
        D, _ = K.attn_merge(x["ws"], B, H, N, spw, 2)
        dS = K.score_blocks(B, H, N, dev)
        K.talking_fused(3, x["Qf"], x["Kf"], x["Vf"], x["dOf"], x["Wl"], x["bl"], x["Ww"], x["bw"], x["M"], x["IL"], D, None, ws_w, dS, B, H, N, dh,
                        p_drop, x["seed"], x["off"], keepbits=x["bits"])
        dq = torch.zeros(B, N, H, dh, device=dev)
        K.attn_contract(dS, x["K16"], dq, False, alpha=x["scale"])
        return D, dS, dq, ws_w.sum(0)

    def new(hybrid_D=None):
        if hybrid_D is None:
            Drows, ws_w = K.talking_bwdq_pass1(x["Qf"], x["dOf"], x["Kf"], x["Vf"], x["Wl"], x["Ww"], x["c0"], x["bits"], B, H, N, dh, p_drop)
        else:       # the default composition: pass 1 on the round-3 kernel, its D transposed to rows
            Drows = K.flash_rows(hybrid_D, None, None, B, H, N, 1)
            ws_w = torch.zeros(4 * K.bwdq_plan(B, N)[1], nw, device=dev)
        dS = K.score_blocks(B, H, N, dev)
        dq = torch.zeros(B, N, H, dh, device=dev)
        dq16 = torch.zeros(B, N, H, dh, device=dev, dtype=torch.bfloat16)
        K.talking_bwdq_pass2(x["Qf"], x["dOf"], x["Kf"], x["Vf"], x["K16"], x["Wl"], x["Ww"], x["c0"], Drows, ws_w, dS, dq, dq16, x["scale"], x["bits"],
                             B, H, N, dh, p_drop)
        return Drows[:, :N].permute(0, 2, 1).contiguous(), dS, dq, ws_w.sum(0), dq16

    Do, dSo, dqo, wo = old()
    Dn, dSn, dqn, wn, dq16 = new()
    Dh, dSh, dqh, wh, _ = new(hybrid_D=Do)
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    hh = H * H
    assert all(torch.isfinite(t).all() for t in (Dn, dqn, wn, dSn.float()))
    assert rel(Dn, Do) <= 1e-4
    for dS_, dq_, w_ in ((dSn, dqn, wn), (dSh, dqh, wh)):
        assert rel(dS_.float(), dSo.float()) <= 5e-4          # bf16 blocks: a few elements round the other way
        assert rel(dq_, dqo) <= 5e-4
        assert rel(w_[:hh], wo[:hh]) <= 1e-3                  # dWl
        assert w_[hh:hh + H].abs().max() <= 2e-2 * wo[:hh].abs().max()      # dbl: exact value 0
    assert rel(wn[hh + H:2 * hh + H], wo[hh + H:2 * hh + H]) <= 1e-3          # dWw
    assert rel(wn[2 * hh + H:], wo[2 * hh + H:]) <= 1e-3                      # dbw
    assert torch.equal(dq16.float(), dqn.to(torch.bfloat16).float())          # the bf16 copy is the rounded fp32 result
    # rows >= N of the padded D rows are zero (pass 2 reads them for the padded queries of the last tile)
    Drows, _ = K.talking_bwdq_pass1(x["Qf"], x["dOf"], x["Kf"], x["Vf"], x["Wl"], x["Ww"], x["c0"], x["bits"], B, H, N, dh, p_drop)
    assert (Drows[:, N:] == 0).all()
    Dn2, dSn2, dqn2, wn2, _ = new()
    torch.cuda.synchronize()
    assert torch.equal(Dn, Dn2) and torch.equal(dSn.view(torch.int16), dSn2.view(torch.int16)) and torch.equal(dqn, dqn2) and torch.equal(wn, wn2)


def test_flash_skeleton_backward_in_the_attention_node(dev):
    """The attention autograd node with the flash-skeleton pass 2 (default), with both flash-skeleton passes and with the round-3 passes:
    the same gradients to rounding, with attention dropout on (the flags come from the flash forward in all three)."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16s")
    g = torch.Generator().manual_seed(11)
    B, H, N, dh = 2, 8, 700, 48
    C = H * dh
    qkv0 = torch.randn(B, N, 3 * C, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.2 * torch.randn(H, H, generator=g)).to(dev).requires_grad_(True)
    Ww = (torch.eye(H) + 0.2 * torch.randn(H, H, generator=g)).to(dev).requires_grad_(True)
    bl = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
    bw = (0.1 * torch.randn(H, generator=g) / N).to(dev).requires_grad_(True)
    w = torch.randn(B, N, C, generator=g).to(dev)
    res = {}
    saved = (ops.BWDQ, ops.BWDQ_MODE)
    try:
        for mode in (0, 1, 2):
            ops.BWDQ, ops.BWDQ_MODE = mode != 0, mode
            K.manual_seed(31)
            qkv = qkv0.clone().requires_grad_(True)
            for t in (Wl, Ww, bl, bw):
                t.grad = None
            O = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, dh ** -0.5, 0.1)
            (O * w).sum().backward()
            res[mode] = (O.detach().clone(), qkv.grad.clone(), Wl.grad.clone(), Ww.grad.clone(), bw.grad.clone())
    finally:
        ops.BWDQ, ops.BWDQ_MODE = saved
    for mode in (1, 2):
        assert torch.equal(res[mode][0], res[0][0])
        for a, b in zip(res[mode][1:], res[0][1:]):
            assert (a - b).norm() <= 1e-3 * b.norm(), (mode, float((a - b).norm() / b.norm()))
