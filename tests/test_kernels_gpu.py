"""Kernel-level parity (GPU): every libspe_hip.so kernel against a plain fp32/fp64 torch
restatement of the same arithmetic, on seeded asymmetric inputs.  Tolerances are stated per mode:
`bf16x3` (3-term split) must sit at fp32 round-off; `bf16` is bounded by bf16 operand rounding
(2^-9 per operand -> ~3e-3 relative per contraction)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


TOL = {"bf16": 6e-3, "bf16x3": 2e-5}


@pytest.fixture(params=["bf16x3", "bf16"])
def prec(request, dev):
    from spe_amd import kernels as K
    K.set_precision(request.param)
    yield request.param
    K.set_precision("bf16")


@pytest.mark.parametrize("M,N,K_", [(128, 128, 32), (200, 91, 48), (300, 384, 384), (77, 130, 4150), (1, 4, 384), (513, 257, 100)])
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_gemm_layouts(dev, prec, M, N, K_, layout):
    from spe_amd import kernels as K
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K_)
    A = torch.randn(M, K_, generator=g).to(dev)
    B = torch.randn(K_, N, generator=g).to(dev)
    ref = (A.double() @ B.double())
    C = torch.full((M, N), float("nan"), device=dev)
    if layout == "NT":
        Bt = B.t().contiguous()
        K.gemm(A, Bt, C, M, N, K_, K_, K_, N, False, True)
    elif layout == "NN":
        K.gemm(A, B, C, M, N, K_, K_, N, N, False, False)
    else:
        At = A.t().contiguous()
        K.gemm(At, B, C, M, N, K_, M, N, N, True, False)
    torch.cuda.synchronize()
    assert torch.isfinite(C).all()
    assert rel(C, ref) < TOL[prec], (layout, rel(C, ref))


def test_gemm_epilogue_splitk_batched(dev, prec):
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    x = torch.randn(333, 96, generator=g).to(dev)
    W = torch.randn(200, 96, generator=g).to(dev)
    b = torch.randn(200, generator=g).to(dev)
    y, pre, _ = K.linear_fwd(x, W, b, act=2, want_pre=True)
    refpre = x.double() @ W.double().t() + b.double()
    assert rel(pre, refpre) < TOL[prec]
    assert rel(y, torch.nn.functional.gelu(refpre)) < TOL[prec]
    y1 = K.linear_fwd(x, W, b, act=1)[0]
    assert rel(y1, torch.relu(refpre)) < TOL[prec]
    # split-K (TN, long contraction)
    dy = torch.randn(5000, 64, generator=g).to(dev)
    xx = torch.randn(5000, 96, generator=g).to(dev)
    dW = torch.zeros(64, 96, device=dev)
    K.gemm(dy, xx, dW, 64, 96, 5000, 64, 96, 96, True, False, splitk=8)
    assert rel(dW, dy.double().t() @ xx.double()) < TOL[prec]
    slabs = torch.full((11, 64 * 96), float("nan"), device=dev)           # slab split-K: no atomics, summed afterwards
    K.gemm(dy, xx, slabs, 64, 96, 5000, 64, 96, 96, True, False, splitk=-11)
    assert rel(slabs.sum(0).view(64, 96), dy.double().t() @ xx.double()) < TOL[prec]
    # two-level batch with strides (the attention addressing): qkv [B,N,3,H,dh]
    B_, N_, H_, dh = 2, 37, 4, 24
    qkv = torch.randn(B_, N_, 3, H_, dh, generator=g).to(dev)
    ld = K.pad4(N_)
    S = torch.zeros(B_, H_, N_, ld, device=dev)
    C3 = 3 * H_ * dh
    K.gemm(qkv[:, :, 0], qkv[:, :, 1], S, N_, N_, dh, C3, C3, ld, False, True, batch0=B_, batch1=H_,
           sA=(N_ * C3, dh), sB=(N_ * C3, dh), sC=(H_ * N_ * ld, N_ * ld), alpha=0.5)
    ref = 0.5 * torch.einsum("bqhd,bkhd->bhqk", qkv[:, :, 0].double(), qkv[:, :, 1].double())
    assert rel(S[..., :N_], ref) < TOL[prec]
    assert (S[..., N_:] == 0).all()


def test_linear_bwd(dev, prec):
    from spe_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 50, 96, generator=g).to(dev).requires_grad_()
    W = (torch.randn(130, 96, generator=g) * 0.1).to(dev).requires_grad_()
    b = torch.randn(130, generator=g).to(dev).requires_grad_()
    for act, f in [(0, lambda t: t), (1, torch.relu), (2, torch.nn.functional.gelu)]:
        y = ops.linear(x, W, b, act)
        go = torch.randn(y.shape, generator=g).to(dev)
        gx, gW, gb = torch.autograd.grad(y, (x, W, b), go)
        xd, Wd, bd = (t.detach().double().requires_grad_() for t in (x, W, b))
        yr = f(xd @ Wd.t() + bd)
        rx, rW, rb = torch.autograd.grad(yr, (xd, Wd, bd), go.double())
        assert rel(y, yr) < TOL[prec]
        # bf16 operand rounding flips the ReLU mask of pre-activations within ~3e-3 of zero
        gt = 6e-2 if (act == 1 and prec == "bf16") else TOL[prec]
        assert rel(gx, rx) < gt and rel(gW, rW) < gt and rel(gb, rb) < max(gt, 1e-5), (act, rel(gx, rx), rel(gW, rW), rel(gb, rb))


@pytest.mark.parametrize("R,C", [(7, 192), (1000, 384), (33, 1024)])
def test_layernorm(dev, R, C):
    from spe_amd import ops
    g = torch.Generator().manual_seed(R)
    x = (torch.randn(R, C, generator=g) * 3 + 1).to(dev).requires_grad_()
    w = torch.randn(C, generator=g).to(dev).requires_grad_()
    b = torch.randn(C, generator=g).to(dev).requires_grad_()
    y = ops.layer_norm(x, w, b, 1e-6)
    go = torch.randn(R, C, generator=g).to(dev)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    yr = torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-6)
    rx, rw, rb = torch.autograd.grad(yr, (xd, wd, bd), go.double())
    assert rel(y, yr) < 1e-5 and rel(gx, rx) < 1e-5 and rel(gw, rw) < 1e-5 and rel(gb, rb) < 1e-5


def test_softmax_mask_and_bwd(dev):
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    B, H, Nq, Nk = 2, 3, 5, 203
    ld = K.pad4(Nk)
    S = torch.zeros(B, H, Nq, ld, device=dev)
    S[..., :Nk] = (torch.randn(B, H, Nq, Nk, generator=g) * 4).to(dev)
    mask = torch.zeros(B, Nk, dtype=torch.bool)
    mask[1, 150:] = True
    mask[0, ::7] = True
    Sd = S[..., :Nk].double().masked_fill(mask.to(dev)[:, None, None, :], float("-inf")).requires_grad_()
    Pr = Sd.softmax(-1)
    P, Pd = K.softmax_fwd(S.clone(), mask.to(torch.uint8).to(dev), B, H, Nq, Nk, ld, 0.0, 0, 0)
    assert Pd is None and rel(P[..., :Nk], Pr) < 1e-5
    go = torch.zeros(B, H, Nq, ld, device=dev)
    go[..., :Nk] = torch.randn(B, H, Nq, Nk, generator=g).to(dev)
    (rs,) = torch.autograd.grad(Pr, Sd, go[..., :Nk].double())
    dS = K.softmax_bwd(go.clone(), P, B, H, Nq, Nk, ld, 0.0, 0, 0)
    assert rel(dS[..., :Nk], rs) < 1e-5
    # dropout: forward mask == backward mask, keep-rate ~ 1-p, scale 1/(1-p)
    P2, Pd2 = K.softmax_fwd(S.clone(), None, B, H, Nq, Nk, ld, 0.25, 1234, 7)
    ratio = Pd2[..., :Nk] / P2[..., :Nk]
    kept = ratio > 0
    assert abs(kept.float().mean().item() - 0.75) < 0.03
    assert torch.allclose(ratio[kept], torch.full_like(ratio[kept], 1 / 0.75), rtol=1e-5)
    ones = torch.zeros(B, H, Nq, ld, device=dev); ones[..., :Nk] = 1
    # d(sum Pd)/dS through the same mask
    Pdd = (P2[..., :Nk].double() * ratio.double())
    Sd2 = S[..., :Nk].double().requires_grad_()
    (r2,) = torch.autograd.grad((Sd2.softmax(-1) * ratio.double()).sum(), Sd2)
    d2 = K.softmax_bwd(ones.clone(), P2, B, H, Nq, Nk, ld, 0.25, 1234, 7)
    assert (d2[..., :Nk].double() - r2).abs().max().item() < 1e-5


def _talking_ref(qkv, Wl, bl, Ww, bw, H, scale, keep=None):
    B, N, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = torch.nn.functional.linear(attn.permute(0, 2, 3, 1), Wl, bl).permute(0, 3, 1, 2)
    attn = attn.softmax(-1)
    attn = torch.nn.functional.linear(attn.permute(0, 2, 3, 1), Ww, bw).permute(0, 3, 1, 2)
    if keep is not None:
        attn = attn * keep
    return (attn @ v).transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("H,N,dh", [(4, 50, 48), (8, 131, 48), (6, 40, 32)])
def test_talking_heads_attention(dev, prec, H, N, dh):
    from spe_amd import ops
    g = torch.Generator().manual_seed(H * N)
    B, C = 2, H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev).requires_grad_()
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
    bl = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
    bw = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
    scale = dh ** -0.5
    out = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, scale, 0.0)
    go = torch.randn(out.shape, generator=g).to(dev)
    grads = torch.autograd.grad(out, (qkv, Wl, bl, Ww, bw), go)
    dd = [t.detach().double().requires_grad_() for t in (qkv, Wl, bl, Ww, bw)]
    ref = _talking_ref(*dd, H, scale)
    rg = torch.autograd.grad(ref, dd, go.double())
    assert rel(out, ref) < TOL[prec]
    for a, b, nm in zip(grads, rg, ["qkv", "Wl", "bl", "Ww", "bw"]):
        if nm == "bl":   # softmax is shift invariant: the exact gradient is 0
            assert a.abs().max().item() < 1e-3 * grads[1].abs().max().item()
        else:
            assert rel(a, b) < 2 * TOL[prec], (nm, rel(a, b))


def test_talking_heads_dropout_consistency(dev):
    """With attn_drop > 0 the backward must regenerate the forward mask: check d(out)/d(Ww) against
    a reference that uses the mask recovered from the forward (Pd / P')."""
    from spe_amd import kernels as K
    K.set_precision("bf16x3")
    g = torch.Generator().manual_seed(9)
    B, H, N = 1, 4, 33
    ld = K.pad4(N)
    S = torch.zeros(B, H, N, ld, device=dev); S[..., :N] = torch.randn(B, H, N, N, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.2 * torch.randn(H, H, generator=g)).to(dev); bl = torch.zeros(H, device=dev)
    Ww = (torch.eye(H) + 0.2 * torch.randn(H, H, generator=g)).to(dev); bw = (0.05 * torch.randn(H, generator=g)).to(dev)
    S0 = S.clone()
    P, Pd = K.talking_fwd(S, Wl, bl, Ww, bw, B, H, N, N, ld, 0.3, 99, 5)
    Pm = torch.nn.functional.linear(P[..., :N].permute(0, 2, 3, 1), Ww, bw).permute(0, 3, 1, 2)
    ratio = Pd[..., :N] / Pm
    keep = ratio.abs() > 1e-6
    assert abs(keep.float().mean().item() - 0.7) < 0.05
    go = torch.zeros(B, H, N, ld, device=dev); go[..., :N] = torch.randn(B, H, N, N, generator=g).to(dev)
    Sd = S0[..., :N].double().requires_grad_()
    Wld, Wwd = Wl.double().requires_grad_(), Ww.double().requires_grad_()
    a = torch.nn.functional.linear(Sd.permute(0, 2, 3, 1), Wld, bl.double()).permute(0, 3, 1, 2).softmax(-1)
    a = torch.nn.functional.linear(a.permute(0, 2, 3, 1), Wwd, bw.double()).permute(0, 3, 1, 2) * (keep.double() / 0.7)
    rS, rWl, rWw = torch.autograd.grad(a, (Sd, Wld, Wwd), go[..., :N].double())
    dS, dWl, dbl, dWw, dbw = K.talking_bwd(go.clone(), P, S0, Wl, Ww, B, H, N, N, ld, 0.3, 99, 5)
    assert rel(dS[..., :N], rS) < 1e-4 and rel(dWl, rWl) < 1e-4 and rel(dWw, rWw) < 1e-4
    K.set_precision("bf16")


@pytest.mark.parametrize("Lq,Lk,H,dk,dv", [(10, 77, 4, 48, 24), (100, 2100, 8, 96, 48)])     # second: split-K slab path (long Lk)
def test_attention_generic(dev, prec, Lq, Lk, H, dk, dv):
    from spe_amd import ops
    g = torch.Generator().manual_seed(21)
    B = 2
    q = torch.randn(B, Lq, H, dk, generator=g).to(dev).requires_grad_()
    k = torch.randn(B, Lk, H, dk, generator=g).to(dev).requires_grad_()
    v = torch.randn(B, Lk, H, dv, generator=g).to(dev).requires_grad_()
    mask = torch.zeros(B, Lk, dtype=torch.bool); mask[1, 60:] = True
    out, pmap = ops.attention(q, k, v, mask.to(dev), scale=dk ** -0.5, p_drop=0.0, need_map=True)
    go = torch.randn(out.shape, generator=g).to(dev)
    gq, gk, gv = torch.autograd.grad(out, (q, k, v), go)
    qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qd * dk ** -0.5, kd).masked_fill(mask.to(dev)[:, None, None], float("-inf"))
    p = s.softmax(-1)
    ref = torch.einsum("bhqk,bkhd->bqhd", p, vd).reshape(B, Lq, H * dv)
    rq, rk, rv = torch.autograd.grad(ref, (qd, kd, vd), go.double())
    assert rel(out, ref) < TOL[prec] and rel(pmap, p) < TOL[prec]
    assert rel(gq, rq) < 2 * TOL[prec] and rel(gk, rk) < 2 * TOL[prec] and rel(gv, rv) < 2 * TOL[prec]


def test_elementwise(dev):
    from spe_amd import kernels as K, ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 50, 192, generator=g).to(dev).requires_grad_()
    y = torch.randn(2, 50, 192, generator=g).to(dev).requires_grad_()
    gam = torch.randn(192, generator=g).to(dev).requires_grad_()
    ss = torch.tensor([0.0, 1.25], device=dev)
    for s in (None, ss):
        out = ops.layerscale_residual(x, y, gam, s)
        go = torch.randn(out.shape, generator=g).to(dev)
        gx, gy, gg = torch.autograd.grad(out, (x, y, gam), go)
        xd, yd, gd = (t.detach().double().requires_grad_() for t in (x, y, gam))
        sc = 1.0 if s is None else s.double()[:, None, None]
        ref = xd + sc * gd * yd
        rx, ry, rg = torch.autograd.grad(ref, (xd, yd, gd), go.double())
        assert rel(out, ref) < 1e-6 and rel(gx, rx) < 1e-6 and rel(gy, ry) < 1e-6 and rel(gg, rg) < 1e-5
    h = torch.randn(1000, 64, generator=g).to(dev)
    d = torch.randn(1000, 64, generator=g).to(dev)
    hd = h.double().requires_grad_()
    (r,) = torch.autograd.grad(torch.nn.functional.gelu(hd), hd, d.double())
    assert rel(K.act_bwd(d, h, 2), r) < 1e-5
    assert rel(K.colsum(h), h.double().sum(0)) < 1e-5
    dr = K.dropout(h, 0.1, 42, 1)
    kept = dr != 0
    assert abs(kept.float().mean().item() - 0.9) < 0.02
    assert torch.equal(K.dropout(h, 0.1, 42, 1), dr) and not torch.equal(K.dropout(h, 0.1, 42, 2), dr)
    img = torch.randn(2, 3, 64, 96, generator=g).to(dev)
    W = torch.randn(32, 3, 16, 16, generator=g).to(dev).requires_grad_()
    b = torch.randn(32, generator=g).to(dev).requires_grad_()
    K.set_precision("bf16x3")
    pe = ops.patch_embed(img, W, b, 16)
    ref = torch.nn.functional.conv2d(img.double(), W.double(), b.double(), stride=16).flatten(2).transpose(1, 2)
    assert rel(pe, ref) < 2e-5
    K.set_precision("bf16")
    t = torch.randn(50, 192, generator=g).to(dev).requires_grad_()
    o = ops.add_rows(x, t)
    assert rel(o, x + t) < 1e-7
    (gt,) = torch.autograd.grad(o, t, torch.ones_like(o))
    assert rel(gt, torch.full_like(t, 2.0)) < 1e-6
    # bicubic grid resize (+ adjoint) vs F.interpolate
    for (gh, gw, hh, ww) in ((24, 24, 14, 14), (50, 84, 50, 83), (5, 7, 9, 4)):
        pe = torch.randn(1, gh * gw, 32, generator=g).to(dev).requires_grad_()
        out = ops.bicubic_grid(pe, gh, gw, hh, ww)
        ped = pe.detach().double().requires_grad_()
        ref = torch.nn.functional.interpolate(ped.transpose(1, 2).reshape(1, 32, gh, gw), size=(hh, ww), mode="bicubic",
                                              align_corners=False).flatten(2).transpose(1, 2)
        go = torch.randn(out.shape, generator=g).to(dev)
        (gp,) = torch.autograd.grad(out, pe, go)
        (rp,) = torch.autograd.grad(ref, ped, go.double())
        assert rel(out, ref) < 1e-5 and rel(gp, rp) < 1e-5, (gh, gw, hh, ww, rel(out, ref), rel(gp, rp))


def _giou(a, b):
    def xyxy(x):
        cx, cy, w, h = x.unbind(-1)
        return torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)
    a, b = xyxy(a), xyxy(b)
    a1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); a2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[:, :2]); rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0); inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2 - inter; iou = inter / union
    lt = torch.min(a[:, None, :2], b[:, :2]); rb = torch.max(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0); area = wh[..., 0] * wh[..., 1]
    return iou - (area - union) / area


def _rand_boxes(n, g):
    c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n, 2, generator=g) * 0.35 + 0.05
    return torch.cat([c, wh], 1)


def test_matcher_cost_and_losses(dev):
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(17)
    L, B, Q, Kc = 3, 2, 20, 21
    logits = (torch.randn(L, B, Q, Kc, generator=g) * 2).to(dev)
    boxes = torch.stack([torch.stack([_rand_boxes(Q, g) for _ in range(B)]) for _ in range(L)]).to(dev)
    sizes = [5, 0] if False else [5, 9]
    tgt_ids = torch.randint(1, Kc, (sum(sizes),), generator=g)
    tgt_boxes = _rand_boxes(sum(sizes), g)
    toff = torch.tensor([0, sizes[0], sum(sizes)], dtype=torch.int32)
    cost, err = K.matcher_cost(logits, boxes, tgt_ids.int().to(dev), tgt_boxes.to(dev), toff.to(dev), sum(sizes), 2.0, 5.0, 2.0)
    assert err.item() == 0
    for l in range(L):
        for b in range(B):
            p = logits[l, b].double().sigmoid().cpu()
            ids = tgt_ids[toff[b]:toff[b + 1]]
            tb = tgt_boxes[toff[b]:toff[b + 1]].double()
            neg = 0.75 * p ** 2 * (-(1 - p + 1e-8).log()); pos = 0.25 * (1 - p) ** 2 * (-(p + 1e-8).log())
            ref = 5.0 * torch.cdist(boxes[l, b].double().cpu(), tb, p=1) + 2.0 * (pos[:, ids] - neg[:, ids]) \
                - 2.0 * _giou(boxes[l, b].double().cpu(), tb)
            got = cost[l, Q * toff[b]:Q * toff[b + 1]].view(Q, sizes[b])
            assert (got.double().cpu() - ref).abs().max().item() < 2e-5
    # focal loss + gradient
    tcls = torch.randint(0, Kc + 1, (L, B * Q), generator=g)
    roww = torch.rand(L, B * Q, generator=g)
    for gamma in (2.0, 0.5):
        for rw in (None, roww):
            lg = logits.view(L, B * Q, Kc)
            loss, grad, amax = K.focal_loss(lg, tcls.int().to(dev), None if rw is None else rw.to(dev), 0.25, gamma)
            x = lg.double().cpu().requires_grad_()
            t = torch.nn.functional.one_hot(tcls, Kc + 1)[..., :Kc].double()
            prob = x.sigmoid()
            ce = torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="none")
            pt = (prob * t + (1 - prob) * (1 - t)).clamp(1e-5, 1 - 1e-5)
            w = 1.0 if rw is None else rw.double()[..., None]
            le = (0.25 * t + 0.75 * (1 - t)) * w * ce * (1 - pt) ** gamma
            ref = le.sum((1, 2))
            (rg,) = torch.autograd.grad(ref.sum(), x)
            assert rel(loss, ref) < 1e-5 and rel(grad, rg) < 1e-4
            assert torch.equal(amax.cpu().long(), lg.cpu().argmax(-1))
    # box losses + gradient
    n = 40
    pb = boxes.reshape(-1, 4)
    srow = torch.randint(0, pb.shape[0], (n,), generator=g)
    tb = _rand_boxes(n, g)
    w = torch.rand(n, generator=g)
    lidx = (srow // (B * Q)).int()
    sums, g1, g2 = K.box_loss(pb, srow.to(dev), tb.to(dev), w.to(dev), lidx.to(dev), L)
    s = pb.double().cpu()[srow].requires_grad_()
    l1 = ((s - tb.double()).abs().sum(1) * w.double())
    gi = (1 - torch.diag(_giou(s, tb.double()))) * w.double()
    (r1,) = torch.autograd.grad(l1.sum(), s, retain_graph=True)
    (r2,) = torch.autograd.grad(gi.sum(), s)
    ref_sums = torch.zeros(L, 2, dtype=torch.double)
    ref_sums.index_add_(0, lidx.long(), torch.stack([l1, gi], 1).detach())
    assert rel(sums, ref_sums) < 1e-5 and rel(g1, r1) < 1e-6 and rel(g2, r2) < 1e-4
    c1 = torch.tensor([1.0, 2.0, 3.0], device=dev); c2 = torch.tensor([0.5, 0.25, 2.0], device=dev)
    dp = K.box_loss_bwd(srow.to(dev), lidx.to(dev), g1, g2, c1, c2, pb.shape)
    refd = torch.zeros(pb.shape, dtype=torch.double)
    refd.index_add_(0, srow, c1.cpu().double()[lidx.long()][:, None] * r1 + c2.cpu().double()[lidx.long()][:, None] * r2)
    assert rel(dp, refd) < 1e-4


@pytest.mark.parametrize("H,N,dh,B", [(4, 50, 8, 2), (8, 131, 48, 2), (4, 200, 48, 1), (8, 330, 48, 1), (8, 1100, 48, 2), (4, 1031, 32, 1), (4, 100, 64, 1), (8, 70, 40, 1), (4, 90, 20, 2)])
def test_talking_heads_attention_fused(dev, H, N, dh, B):
    """Fused score kernels (bf16 operands, bf16 P'd/dS storage) vs the fp64 restatement; tail tiles, several
    segments per workgroup and several workgroups per q-tile are all exercised by these shapes."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(H * N + 1)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev).requires_grad_()
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
    bl = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
    bw = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
    scale = dh ** -0.5
    out = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, scale, 0.0, fused=True)
    go = torch.randn(out.shape, generator=g).to(dev)
    grads = torch.autograd.grad(out, (qkv, Wl, bl, Ww, bw), go)
    dd = [t.detach().double().requires_grad_() for t in (qkv, Wl, bl, Ww, bw)]
    ref = _talking_ref(*dd, H, scale)
    rg = torch.autograd.grad(ref, dd, go.double())
    assert torch.isfinite(out).all()
    assert rel(out, ref) < 1e-2, rel(out, ref)
    for a, b, nm in zip(grads, rg, ["qkv", "Wl", "bl", "Ww", "bw"]):
        assert torch.isfinite(a).all(), nm
        if nm == "bl":
            assert a.abs().max().item() < 2e-2 * grads[1].abs().max().item()
        else:
            assert rel(a, b) < 2e-2, (nm, rel(a, b))
    # and against the materialised bf16 path (same operand rounding): tighter
    out2 = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, scale, 0.0, fused=False)
    g2 = torch.autograd.grad(out2, (qkv, Wl, Ww), go)
    assert rel(out, out2) < 8e-3
    assert rel(grads[0], g2[0]) < 1.5e-2 and rel(grads[1], g2[1]) < 1.5e-2 and rel(grads[3], g2[2]) < 1.5e-2


@pytest.mark.parametrize("R,C", [(15, 8192), (16, 589824), (8300, 384), (8300, 1536), (777, 91), (3, 100), (200, 2048)])
def test_colsum_shapes(dev, R, C):
    """wide (split-K slabs), tall (bias gradients) and unaligned fallbacks; out accumulates."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(R * 7 + C)
    x = torch.randn(R, C, generator=g).to(dev)
    ref = x.double().sum(0)
    out = K.colsum(x)
    assert rel(out, ref) < 2e-6
    K.colsum(x, out=out)                       # += semantics
    assert rel(out, 2 * ref) < 2e-6


def test_grad_buffer_direct_placement(dev):
    """Parameter gradients written straight into the GradAllReducer buckets (kernels.grad_buffer) equal the plain
    autograd gradients - including a weight used twice, a torch-op parameter and an unused parameter."""
    from spe_amd import kernels as K
    from spe_amd import ops
    from spe_amd.dp import GradAllReducer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(3)
            self.w1 = torch.nn.Parameter(torch.randn(64, 32, generator=g) * 0.2)
            self.b1 = torch.nn.Parameter(torch.randn(64, generator=g) * 0.1)
            self.w2 = torch.nn.Parameter(torch.randn(32, 64, generator=g) * 0.2)      # used twice
            self.lg = torch.nn.Parameter(torch.rand(32, generator=g) + 0.5)
            self.lb = torch.nn.Parameter(torch.randn(32, generator=g) * 0.1)
            self.gam = torch.nn.Parameter(torch.rand(32, generator=g))
            self.pos = torch.nn.Parameter(torch.randn(1, 50, 32, generator=g) * 0.1)  # plain torch add
            self.unused = torch.nn.Parameter(torch.ones(5))

        def forward(self, x):
            x = x + self.pos
            h = ops.linear(ops.layer_norm(x, self.lg, self.lb, 1e-6), self.w1, self.b1, ops.ACT_GELU)
            y = ops.linear(h, self.w2)
            x = ops.layerscale_residual(x, y, self.gam)
            h2 = ops.linear(x, self.w1, self.b1, ops.ACT_RELU)
            return ops.linear(h2, self.w2)

    K.set_precision("bf16x3")
    try:
        net = Net().to(dev)
        x = torch.randn(4, 50, 32, generator=torch.Generator().manual_seed(5)).to(dev)
        go = torch.randn(4, 50, 32, generator=torch.Generator().manual_seed(6)).to(dev)
        named = dict(net.named_parameters())
        ref = torch.autograd.grad(net(x), [p for n, p in named.items() if n != "unused"], go)
        ref = dict(zip([n for n in named if n != "unused"], ref))
        red = GradAllReducer(list(net.parameters()), bucket_bytes=4096)
        for it in range(2):                      # second step: buckets re-armed
            red.reset()
            net(x).backward(go)
            red.finish()
            for n, p in named.items():
                assert p.grad is not None and p.grad.data_ptr() == p._spe_grad_buf.data_ptr(), n
                if n == "unused":
                    assert float(p.grad.abs().sum()) == 0.0
                else:
                    assert rel(p.grad, ref[n]) < 1e-5, (n, it, rel(p.grad, ref[n]))
        red.remove()
    finally:
        K.set_precision("bf16")


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (300, 384, 384), (8300, 1536, 384), (8300, 384, 1536), (77, 136, 4152), (65, 72, 8), (1000, 64, 200),
                                    (384, 384, 448), (400, 2048, 384), (91, 40, 392), (200, 384, 512)])
def test_gemm_bf16nt(dev, M, N, K_):
    """bf16-operand NT GEMM: exact products of the bf16-rounded operands (fp32 accumulate) + epilogues + slabs."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(M + N + K_)
    A = torch.randn(M, K_, generator=g).to(dev)
    B = torch.randn(N, K_, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    A16, A16T = K.cvt_bf16(A, True, True)
    B16, _ = K.cvt_bf16(B)
    assert torch.equal(A16, A.to(torch.bfloat16)) and torch.equal(B16, B.to(torch.bfloat16))
    Rp = A16T.shape[1]
    assert Rp % 64 == 0 and torch.equal(A16T[:, :M], A.to(torch.bfloat16).t()) and float(A16T[:, M:].float().abs().sum()) == 0.0
    ref = A16.double() @ B16.double().t()
    C = torch.full((M, N), float("nan"), device=dev)
    K.gemm16(A16, B16, C, M, N, K_, K_, K_, N)
    assert rel(C, ref) < 2e-6
    C2 = torch.full((M, N), float("nan"), device=dev)
    K.gemm16(A16, B16, C, M, N, K_, K_, K_, N, bias=bias, C2=C2, act=2, alpha=0.5)
    pre = 0.5 * ref + bias.double()
    assert rel(C2, pre) < 2e-6 and rel(C, torch.nn.functional.gelu(pre)) < 2e-6
    if K_ >= 256:
        sk = min(4, (K_ + 63) // 64)
        ws = torch.full((sk, M * N), float("nan"), device=dev)
        K.gemm16(A16, B16, ws, M, N, K_, K_, K_, N, splitk=-sk)
        assert rel(ws.sum(0).view(M, N), ref) < 2e-6


@pytest.mark.parametrize("M,N,K_", [(300, 384, 384), (8300, 1536, 384), (130, 72, 64), (77, 200, 136), (1000, 64, 256)])
@pytest.mark.parametrize("mode", ["gelu_fwd", "gelu_bwd", "relu_fwd", "relu_bwd", "plain"])
def test_gemm_bf16nt_extended_epilogue(dev, M, N, K_, mode):
    """spe_gemm_bf16nt_ex: fp32 / pre-activation / bf16 / transposed-bf16 / column-sum outputs against the plain kernel
    followed by spe_cvt_bf16 (bit-identical values), for full, ragged and single-tile shapes."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(M + N + K_)
    A16 = torch.randn(M, K_, generator=g).to(dev).to(torch.bfloat16)
    B16 = (torch.randn(N, K_, generator=g) * 0.2).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev) if mode.endswith("fwd") else None
    aux = torch.randn(M, N, generator=g).to(dev) if mode.endswith("bwd") else None
    act = {"g": 2, "r": 1, "p": 0}[mode[0]]
    Rp = ((M + 63) // 64) * 64
    # reference: plain kernel -> fp32, then the conversion kernel (which also applies act' from aux and sums columns)
    C = torch.empty(M, N, device=dev); C2 = torch.empty(M, N, device=dev) if bias is not None else None
    K.gemm16(A16, B16, C, M, N, K_, K_, K_, N, bias=bias, C2=C2, act=act if aux is None else 0)
    cs_ref = torch.zeros(N, device=dev)
    r16, r16T = K.cvt_bf16(C, True, True, ldt=Rp, colsum_out=cs_ref, act_aux=aux, act=act if aux is not None else 0)
    # extended epilogue, all outputs at once
    Cx = torch.full((M, N), float("nan"), device=dev); C2x = torch.full((M, N), float("nan"), device=dev) if bias is not None else None
    o16 = torch.full((M, N), float("nan"), device=dev).to(torch.bfloat16)
    o16T = torch.full((N, Rp), float("nan"), device=dev).to(torch.bfloat16)
    cs = torch.zeros(N, device=dev)
    K.gemm16_ex(A16, B16, M, N, K_, K_, K_, bias=bias, C=Cx, C2=C2x, out16=o16, out16T=o16T, colsum=cs, aux=aux, act=act)
    if aux is None:
        assert torch.equal(Cx, C)
    if C2 is not None:
        assert torch.equal(C2x, C2)
    assert torch.equal(o16, r16)
    assert torch.equal(o16T, r16T)                       # includes the zero padding columns M..Rp-1
    assert torch.equal(Cx.to(torch.bfloat16), o16)
    assert rel(cs, cs_ref) < 1e-5
    # outputs are individually optional
    o16b = torch.empty_like(o16)
    K.gemm16_ex(A16, B16, M, N, K_, K_, K_, bias=bias, out16=o16b, aux=aux, act=act)
    assert torch.equal(o16b, r16)


def test_fused_mlp_matches_two_linears(dev):
    """ops.mlp_gelu (one autograd node, bf16 intermediates emitted by the GEMM epilogues) against linear(gelu) + linear on
    the same bf16-copy GEMMs: identical forward, identical weight/bias/input gradients up to summation order."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(5)
    R, C, Hd = 1000, 384, 1536
    x = torch.randn(2, R // 2, C, generator=g).to(dev).requires_grad_()
    W1 = (torch.randn(Hd, C, generator=g) * 0.05).to(dev).requires_grad_(); b1 = (torch.randn(Hd, generator=g) * 0.1).to(dev).requires_grad_()
    W2 = (torch.randn(C, Hd, generator=g) * 0.05).to(dev).requires_grad_(); b2 = (torch.randn(C, generator=g) * 0.1).to(dev).requires_grad_()
    go = torch.randn(2, R // 2, C, generator=g).to(dev)
    y0 = ops.linear(ops.linear(x, W1, b1, ops.ACT_GELU), W2, b2)
    g0 = torch.autograd.grad(y0, (x, W1, b1, W2, b2), go)
    old = K.MLP_PRE_F16
    try:
        # pre-activation saved in fp32: the same arithmetic as the two-node path; saved in fp16 (the default): gelu'(.) is taken of a
        # value rounded to 11 bits - the gradients through it move by < 1e-3, an order below their bf16 operand rounding
        for pre16, tol in ((False, 2e-6), (True, 1e-3)):
            K.MLP_PRE_F16 = pre16
            y = ops.mlp_gelu(x, W1, b1, W2, b2)
            assert y.grad_fn.name().startswith("_MlpGelu")
            gr = torch.autograd.grad(y, (x, W1, b1, W2, b2), go)
            assert torch.equal(y, y0)
            for a, b, nm in zip(gr, g0, ["x", "W1", "b1", "W2", "b2"]):
                assert rel(a, b) < (2e-6 if nm in ("W2", "b2") else tol), (pre16, nm, rel(a, b))
    finally:
        K.MLP_PRE_F16 = old
    # and against fp64 on the bf16-rounded operands of the first GEMM only (sanity of the whole chain)
    xd, W1d, W2d = x.detach().double(), W1.detach().double(), W2.detach().double()
    ref = torch.nn.functional.gelu(xd @ W1d.t() + b1.detach().double()) @ W2d.t() + b2.detach().double()
    assert rel(y, ref) < 1e-2
    # small inputs fall back to the two-node path
    xs = torch.randn(1, 7, C, generator=g).to(dev)
    assert torch.allclose(ops.mlp_gelu(xs, W1, b1, W2, b2), ops.linear(ops.linear(xs, W1, b1, ops.ACT_GELU), W2, b2))


@pytest.mark.parametrize("R,C", [(1000, 384), (8300, 384), (130, 64), (77, 1024)])
def test_layerscale_residual_bwd16(dev, R, C):
    """spe_layerscale_residual_bwd16 against layerscale_residual_bwd + spe_cvt_bf16: identical bf16 copies (incl. the zero
    padding of the transpose), column sums within summation order."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(R + C)
    dout = torch.randn(R, C, generator=g).to(dev); y = torch.randn(R, C, generator=g).to(dev)
    gamma = (torch.randn(C, generator=g) * 0.3).to(dev)
    Rp = ((R + 63) // 64) * 64
    dy, dg_ref = K.layerscale_residual_bwd(dout, y, gamma, None, R)
    db_ref = torch.zeros(C, device=dev)
    r16, r16T = K.cvt_bf16(dy, True, True, ldt=Rp, colsum_out=db_ref)
    dy16, dy16T, db, dg = K.layerscale_residual_bwd16(dout, y, gamma, Rp)
    assert torch.equal(dy16, r16) and torch.equal(dy16T, r16T)
    assert rel(db, db_ref) < 1e-5 and rel(dg, dg_ref) < 1e-5


def test_fused_mlp_residual_matches_composite(dev):
    """ops.mlp_gelu_residual (fc2 epilogue applies x + gamma * y; backward emits gamma * dout as bf16 operands) against
    layerscale_residual(x, mlp_gelu(...)), including gradient placement into all-reduce bucket views."""
    from spe_amd import kernels as K, ops
    from spe_amd.dp import GradAllReducer
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(6)
    R, C, Hd = 1100, 384, 1536
    mk = lambda *s, sc=1.0: torch.nn.Parameter((torch.randn(*s, generator=g) * sc).to(dev))
    W1, b1, W2, b2, gamma = mk(Hd, C, sc=0.05), mk(Hd, sc=0.1), mk(C, Hd, sc=0.05), mk(C, sc=0.1), mk(C, sc=0.5)
    params = [W1, b1, W2, b2, gamma]
    xn = torch.randn(2, R // 2, C, generator=g).to(dev).requires_grad_()
    xr = torch.randn(2, R // 2, C, generator=g).to(dev).requires_grad_()
    go = torch.randn(2, R // 2, C, generator=g).to(dev)
    saved16 = K.MLP_PRE_F16
    K.MLP_PRE_F16 = False                  # fp32 saves: the fused node must reproduce the composite to summation order
    try:
        out = ops.mlp_gelu_residual(xn, W1, b1, W2, b2, xr, gamma)
        assert out.grad_fn.name().startswith("_MlpGeluRes")
        gr = torch.autograd.grad(out, [xn, xr] + params, go)
        ref = ops.layerscale_residual(xr, ops.mlp_gelu(xn, W1, b1, W2, b2), gamma)
        g0 = torch.autograd.grad(ref, [xn, xr] + params, go)
        assert rel(out, ref) < 1e-6
        for a, b, nm in zip(gr, g0, ["xn", "xres", "W1", "b1", "W2", "b2", "gamma"]):
            assert rel(a, b) < 5e-6, (nm, rel(a, b))
        # the default: pre-activation and branch output saved as fp16 (they only enter gelu'(.) and the gamma gradient) - same
        # output bit for bit, gradients within 1e-3 of the fp32-save ones
        K.MLP_PRE_F16 = True
        out16 = ops.mlp_gelu_residual(xn, W1, b1, W2, b2, xr, gamma)
        assert torch.equal(out16, out)
        for a, b, nm in zip(torch.autograd.grad(out16, [xn, xr] + params, go), g0, ["xn", "xres", "W1", "b1", "W2", "b2", "gamma"]):
            assert rel(a, b) < 1e-3, (nm, rel(a, b))
        K.MLP_PRE_F16 = False
        # with a reducer: every parameter gradient lands in its bucket view without a copy
        red = GradAllReducer(params, flatten_params=False)
        red.reset()
        ops.mlp_gelu_residual(xn, W1, b1, W2, b2, xr, gamma).backward(go)
        red.finish()
        for p, b in zip(params, g0[2:]):
            assert p.grad.data_ptr() == red._views[p].data_ptr() and rel(p.grad, b) < 5e-6
        # no gradient wanted (inference): same output, nothing saved
        with torch.no_grad():
            oi = ops.mlp_gelu_residual(xn.detach(), W1.detach(), b1.detach(), W2.detach(), b2.detach(), xr.detach(), gamma.detach())
        assert torch.equal(oi, out.detach())
        # DropPath scale present: still ONE node (round 4: the per-sample scale rides on the fc2 epilogue) with the composite's result
        ss = torch.tensor([1.25, 0.0], device=dev)
        o2 = ops.mlp_gelu_residual(xn, W1, b1, W2, b2, xr, gamma, ss)
        assert o2.grad_fn.name().startswith("_MlpGeluRes")
        oc = ops.layerscale_residual(xr, ops.mlp_gelu(xn, W1, b1, W2, b2), gamma, ss)
        assert rel(o2, oc) < 1e-6
        gf = torch.autograd.grad(o2, [xn, W1, W2, gamma], go)
        gc = torch.autograd.grad(oc, [xn, W1, W2, gamma], go)
        for a_, b_ in zip(gf, gc):
            assert rel(a_, b_) < 2e-3
    finally:
        K.MLP_PRE_F16 = saved16


@pytest.mark.parametrize("B,H,N,dh", [(2, 8, 4150, 48), (1, 8, 8200, 48), (2, 4, 300, 32), (3, 8, 2100, 64)])
def test_attn_contract_blocked_scores(dev, B, H, N, dh):
    """Streaming contractions of a blocked bf16 score tensor against a dense fp64 product, both orientations; the first two
    shapes put a few groups of output tiles past the last full round of workgroups (cut into quarters of the contraction
    range and combined by the last arriver), launched repeatedly to check that the counters are left clean."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(B * N + dh)
    torch.manual_seed(B * N + dh)
    nt = (N + 15) // 16
    T = K.score_blocks(B, H, N, dev)
    T.normal_(0.0, 0.5)
    x = torch.randn(B, N, H, dh, generator=g).to(dev)
    X16 = K.attn_pack16(x)
    # dense [B,H,q,key] from the blocks: lane l of block (qt,kt) = query qt*16+(l&15), keys kt*16+4*(l>>4)+i
    blk = T.view(B, H, nt, nt, 4, 16, 4)
    dense = blk.permute(0, 1, 2, 5, 3, 4, 6).reshape(B, H, nt * 16, nt * 16)[:, :, :N, :N]
    xb = x.to(torch.bfloat16)
    for trans in (False, True):
        ref = torch.empty(B, N, H, dh, device=dev)
        for b in range(B):                                  # per batch element: bounded fp32 temporaries
            d = dense[b].float()
            ref[b] = torch.einsum("hkq,khd->qhd" if trans else "hqk,khd->qhd", d, xb[b].float()) * 0.37
        for rep in range(3):
            out = torch.full((B, N, H, dh), float("nan"), device=dev)
            K.attn_contract(T, X16, out, trans, alpha=0.37)
            assert rel(out, ref) < 2e-5, (trans, rep, rel(out, ref))
        # a different shape in between must not disturb the shared workspace
        T2 = K.score_blocks(1, H, 100, dev); T2.zero_()
        o2 = torch.full((1, 100, H, dh), float("nan"), device=dev)
        K.attn_contract(T2, K.attn_pack16(x[:1, :100].contiguous()), o2, trans)
        assert float(o2.abs().max()) == 0.0


def test_fused_linear_residual_matches_composite(dev):
    """ops.linear_residual (projection + LayerScale residual as one node) against layerscale_residual(x, linear(...))."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(9)
    R, C = 1100, 384
    mk = lambda *s, sc=1.0: torch.nn.Parameter((torch.randn(*s, generator=g) * sc).to(dev))
    W, b, gamma = mk(C, C, sc=0.05), mk(C, sc=0.1), mk(C, sc=0.5)
    xa = torch.randn(2, R // 2, C, generator=g).to(dev).requires_grad_()
    xr = torch.randn(2, R // 2, C, generator=g).to(dev).requires_grad_()
    go = torch.randn(2, R // 2, C, generator=g).to(dev)
    out = ops.linear_residual(xa, W, b, xr, gamma)
    assert out.grad_fn.name().startswith("_LinearRes")
    gr = torch.autograd.grad(out, [xa, xr, W, b, gamma], go)
    ref = ops.layerscale_residual(xr, ops.linear(xa, W, b), gamma)
    g0 = torch.autograd.grad(ref, [xa, xr, W, b, gamma], go)
    assert rel(out, ref) < 1e-6
    for a, c, nm in zip(gr, g0, ["x", "xres", "W", "b", "gamma"]):
        # the branch output is saved as fp16 for the gamma gradient (the only place it is used): 1e-3 there, summation order elsewhere
        assert rel(a, c) < (1e-3 if (nm == "gamma" and K.MLP_PRE_F16) else 5e-6), (nm, rel(a, c))
    with torch.no_grad():
        assert torch.equal(ops.linear_residual(xa.detach(), W.detach(), b.detach(), xr.detach(), gamma.detach()), out.detach())
    # DropPath scale present: still one node (round 4: the per-sample scale rides on the GEMM epilogue), same result as the composite
    ss = torch.tensor([1.25, 0.0], device=dev)
    o2 = ops.linear_residual(xa, W, b, xr, gamma, ss)
    assert o2.grad_fn.name().startswith("_LinearRes")
    oc = ops.layerscale_residual(xr, ops.linear(xa, W, b), gamma, ss)
    assert rel(o2, oc) < 1e-6
    go2 = torch.randn_like(oc)
    for a_, b_ in zip(torch.autograd.grad(o2, [xa, W, gamma], go2), torch.autograd.grad(oc, [xa, W, gamma], go2)):
        assert rel(a_, b_) < 2e-3


def test_layer_norm_skip_sums_both_gradients(dev):
    """ops.layer_norm_skip returns (LN(x), x); the gradient over the skip result is added inside the LayerNorm backward kernel."""
    from spe_amd import ops
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 50, 384, generator=g).to(dev).requires_grad_()
    w = (1 + 0.1 * torch.randn(384, generator=g)).to(dev).requires_grad_(); b = (0.1 * torch.randn(384, generator=g)).to(dev).requires_grad_()
    u = torch.randn(3, 50, 384, generator=g).to(dev); v = torch.randn(3, 50, 384, generator=g).to(dev)
    y, xs = ops.layer_norm_skip(x, w, b, 1e-6)
    assert torch.equal(xs, x) and torch.equal(y, ops.layer_norm(x, w, b, 1e-6))
    gr = torch.autograd.grad((y * u).sum() + (xs * v).sum(), (x, w, b))
    xd = x.detach().double().requires_grad_(); wd = w.detach().double().requires_grad_(); bd = b.detach().double().requires_grad_()
    ref = torch.nn.functional.layer_norm(xd, (384,), wd, bd, 1e-6)
    g0 = torch.autograd.grad((ref * u.double()).sum() + (xd * v.double()).sum(), (xd, wd, bd))
    for a, c in zip(gr, g0):
        assert rel(a, c) < 1e-5
    # only one of the two results used
    y, xs = ops.layer_norm_skip(x, w, b, 1e-6)
    assert rel(torch.autograd.grad((xs * v).sum(), x)[0], v) < 1e-7
    y, xs = ops.layer_norm_skip(x, w, b, 1e-6)
    assert rel(torch.autograd.grad((y * u).sum(), x)[0], torch.autograd.grad((ops.layer_norm(x, w, b, 1e-6) * u).sum(), x)[0]) < 1e-6


def test_linear_bf16_path_matches_fp32_operand_path(dev):
    """ops.linear on the bf16-copy GEMMs == the fp32-operand kernel (same roundings), fwd and bwd, in the bf16 and bf16s modes."""
    from spe_amd import kernels as K
    from spe_amd import ops
    g = torch.Generator().manual_seed(11)
    R, Kd, N = 4150, 384, 1152
    x = torch.randn(2, R // 2, Kd, generator=g).to(dev).requires_grad_()
    W = (torch.randn(N, Kd, generator=g) * 0.05).to(dev).requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    go = torch.randn(2, R // 2, N, generator=g).to(dev)
    old = K.LINEAR16
    # bf16: identical roundings on both paths.  bf16s: the forward runs on (hi, lo) pairs on both paths - the bf16-copy kernel reads
    # the lo part its producer wrote, the fp32-operand kernel splits while staging - and agrees to fp32 round-off of a K = 384 sum
    for prec, tol in (("bf16", 1e-5), ("bf16s", 5e-5)):
        K.set_precision(prec)
        res = {}
        try:
            for mode in (True, False):
                K.LINEAR16 = mode
                y = ops.linear(x, W, b, ops.ACT_GELU)
                res[mode] = (y,) + torch.autograd.grad(y, (x, W, b), go)
        finally:
            K.LINEAR16 = old
        for a, c in zip(res[True], res[False]):
            assert rel(a, c) < tol, (prec, rel(a, c))
    # and the split forward is fp32-grade: against fp64
    yd = torch.nn.functional.gelu(x.detach().double() @ W.detach().double().t() + b.detach().double())
    assert rel(res[True][0], yd) < 2e-5, rel(res[True][0], yd)


@pytest.mark.parametrize("L,Q,sizes", [(1, 100, [7, 7]), (7, 100, [1, 20, 0, 64]), (2, 300, [93, 5]), (3, 16, [16, 15]), (1, 1000, [100])])
def test_hungarian_device_vs_scipy(dev, L, Q, sizes):
    """hungarian_kernel == scipy.optimize.linear_sum_assignment on every (layer, image) block, same pair order."""
    from scipy.optimize import linear_sum_assignment
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(Q + sum(sizes) + L)
    B, total = len(sizes), sum(sizes)
    toff = [0]
    for s in sizes:
        toff.append(toff[-1] + s)
    cost = (torch.randn(L, Q * total, generator=g) * 3).to(dev)
    toff_t = torch.tensor(toff, dtype=torch.int32, device=dev)
    srow, gidx, lidx = K.hungarian(cost, toff_t, L, B, Q, total)
    ch = cost.cpu()
    es, eg, el = [], [], []
    for l in range(L):
        for b in range(B):
            if sizes[b] == 0:
                continue
            blk = ch[l, Q * toff[b]:Q * toff[b + 1]].view(Q, sizes[b]).numpy()
            i, j = linear_sum_assignment(blk)
            es.append(torch.as_tensor(i) + (l * B + b) * Q)
            eg.append(torch.as_tensor(j) + toff[b])
            el.append(torch.full((len(i),), l))
    assert torch.equal(srow.cpu(), torch.cat(es)) and torch.equal(gidx.cpu(), torch.cat(eg))
    assert torch.equal(lidx.cpu().long(), torch.cat(el))


def test_criterion_device_matching_equals_host_matching(dev):
    """SetCriterion with the device-side assignment == the host (SciPy) fallback path, loss by loss."""
    import argparse
    from spe_amd.models.conditional_detr import SetCriterion
    from spe_amd.models.matcher import HungarianMatcher
    g = torch.Generator().manual_seed(9)
    L, B, Q, Kc = 3, 2, 50, 21
    outs = []
    for l in range(L):
        outs.append({"pred_logits": torch.randn(B, Q, Kc, generator=g).to(dev),
                     "pred_boxes": (torch.rand(B, Q, 4, generator=g) * 0.4 + 0.2).to(dev)})
    outputs = dict(outs[0]); outputs["aux_outputs"] = outs[1:]
    targets = []
    for b in range(B):
        n = 5 + b
        lab = torch.randint(0, Kc, (n,), generator=g)
        il = torch.zeros(Kc, dtype=torch.int64); il[lab] = 1
        targets.append({"labels": lab.to(dev), "boxes": (torch.rand(n, 4, generator=g) * 0.4 + 0.2).to(dev), "img_label": il.to(dev)})
    crit = SetCriterion(Kc, HungarianMatcher(2.0, 5.0, 2.0), {"loss_ce": 2.0}, 0.25, ["labels", "boxes", "cardinality"], 2.0, 0.1).to(dev).eval()
    a = crit(outputs, targets)
    orig = HungarianMatcher.match_flat
    try:
        HungarianMatcher.match_flat = lambda self, *args: None
        b_ = crit(outputs, targets)
    finally:
        HungarianMatcher.match_flat = orig
    assert set(a) == set(b_)
    for k in a:
        assert torch.allclose(a[k], b_[k], rtol=1e-6, atol=1e-7), (k, a[k], b_[k])


def test_flat_adamw_matches_torch(dev):
    """FlatAdamW (+ fused global-norm clip) == clip_grad_norm_ + torch.optim.AdamW with two parameter groups, 3 steps,
    LR change in between; state_dict round trip."""
    from spe_amd.dp import GradAllReducer
    from spe_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(21)
    shapes = [(64, 32), (64,), (7, 5, 3), (1,), (300, 17), (33,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, generator=torch.Generator().manual_seed(i)).to(dev)) for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()
    groups = lambda ps: [{"params": ps[:3], "lr": 1e-2, "weight_decay": 1e-2}, {"params": ps[3:], "lr": 3e-3, "weight_decay": 0.0}]
    ref = torch.optim.AdamW(groups(pa), betas=(0.9, 0.999), eps=1e-8)
    red = GradAllReducer(pb, bucket_bytes=4096, flatten_params=True)
    opt = FlatAdamW(groups(pb), red, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.1)
    for it in range(3):
        red.reset()
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(dev) * (10.0 if it == 1 else 0.01)      # clipped and unclipped steps
            x.grad = gr.clone()
            y.grad = red._views[y]; y.grad.copy_(gr)
        red.finish()
        tn = torch.nn.utils.clip_grad_norm_(pa, 0.1)
        ref.step(); opt.step()
        for x, y in zip(pa, pb):
            assert rel(y, x) < 2e-6, (it, rel(y, x))
            assert rel(y.grad, x.grad) < 2e-6
        if it == 0:
            for o in (ref, opt):
                o.param_groups[1]["lr"] = 1e-3
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    mbuf = opt._buckets[red._bucket_of[pb[0]]]["m"]           # the state entries are still views of the flat moments
    assert mbuf.data_ptr() <= opt.state[pb[0]]["exp_avg"].data_ptr() < mbuf.data_ptr() + mbuf.numel() * 4
    for x, y in zip(pa, pb):
        assert rel(opt.state[y]["exp_avg"], ref.state[x]["exp_avg"]) < 2e-6
        # v accumulates (clip * g)^2: twice the relative rounding difference of the two global-norm reductions
        assert rel(opt.state[y]["exp_avg_sq"], ref.state[x]["exp_avg_sq"]) < 5e-5
    # resume into the REFERENCE's optimizer (main.py:189-191, 224-232): the exported `step` entries are independent tensors - a shared one would be
    # bumped once per parameter by torch.optim.AdamW's foreach step
    steps = [st["step"] for st in sd["state"].values()]
    assert len({id(s) for s in steps}) == len(steps) and all(float(s) == 3.0 for s in steps)
    pc = mk()
    fresh = torch.optim.AdamW(groups(pc), betas=(0.9, 0.999), eps=1e-8)
    fresh.load_state_dict(sd)
    for p in pc:
        p.grad = torch.zeros_like(p)
    fresh.step()
    assert all(float(fresh.state[p]["step"]) == 4.0 for p in pc), [float(fresh.state[p]["step"]) for p in pc]


def test_per_class_nms_and_flip_merge_vs_oracle(dev):
    """spe_amd.infer (HIP NMS kernel, one launch per batch) == the oracle's restatement of engine_loc.py:99-124,150-174."""
    from oracle import spe_oracle as O
    from spe_amd import infer
    g = torch.Generator().manual_seed(33)
    results = []
    for i in range(3):
        n = 300
        c = torch.rand(n, 2, generator=g) * 600 + 100
        wh = torch.rand(n, 2, generator=g) * 200 + 20
        # clusters of near-duplicates so that suppression actually happens
        c[50:150] = c[:100].clone() + torch.randn(100, 2, generator=g) * 4
        wh[50:150] = wh[:100].clone() * (1 + 0.05 * torch.randn(100, 2, generator=g))
        boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
        labels = torch.randint(0, 6, (n,), generator=g)
        labels[50:150] = labels[:100].clone()
        scores = torch.rand(n, generator=g)
        results.append({"scores": scores, "labels": labels, "boxes": boxes})
    got = infer.per_class_nms([{k: v.to(dev) for k, v in r.items()} for r in results], 0.5)
    for r, o in zip(results, got):
        ref = O.per_class_nms(r, 0.5)
        assert 0 < ref["scores"].numel() < 300
        assert torch.equal(o["labels"].cpu(), ref["labels"]) and torch.equal(o["scores"].cpu(), ref["scores"])
        assert torch.equal(o["boxes"].cpu(), ref["boxes"])
    # flip test-time-augmentation merge
    bs, Q, Kc = 2, 7, 5
    mk = lambda *s: torch.randn(*s, generator=g)
    outp = {"pred_logits": mk(2 * bs, Q, Kc), "pred_boxes": torch.rand(2 * bs, Q, 4, generator=g), "x_logits": mk(2 * bs, Kc),
            "x_cls_logits": mk(2 * bs, Kc), "cams_cls": mk(2 * bs, Kc, 3, 4),
            "aux_outputs": [{"pred_logits": mk(2 * bs, Q, Kc), "pred_boxes": torch.rand(2 * bs, Q, 4, generator=g)}]}
    ref = O.decouple_output(outp, bs)
    cp = {k: ([dict((kk, vv.to(dev)) for kk, vv in a.items()) for a in v] if k == "aux_outputs" else v.to(dev)) for k, v in outp.items()}
    got = infer.decouple_output(cp, bs)
    for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
        assert torch.equal(got[k].cpu(), ref[k]), k
    assert torch.equal(got["aux_outputs"][0]["pred_boxes"].cpu(), ref["aux_outputs"][0]["pred_boxes"])


def test_cam_to_boxes_vs_oracle(dev):
    """device resize/normalise/quantise/threshold == the NumPy restatement (pixels that sit on a
    uint8 truncation boundary may differ by rounding: <= 0.05 % tolerated), and the whole driver
    (spe_amd.camboxes.get_pseudo_label_multi_boxes) == the reference's loop run on the oracle functions."""
    import argparse
    import numpy as np
    from oracle import cam_oracle as CO
    from spe_amd import camboxes, kernels as K
    g = torch.Generator().manual_seed(12)
    B, Kc, h, w, H, W = 2, 5, 9, 13, 144, 208
    cams = torch.zeros(B, Kc, h, w)
    for b in range(B):
        for c in range(Kc):
            for _ in range(2):                                          # two bumps per map
                cy, cx = torch.rand(2, generator=g) * torch.tensor([h - 1.0, w - 1.0])
                yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
                cams[b, c] += torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * (1.0 + 2 * torch.rand(1, generator=g)) ** 2))
    cams += 0.05 * torch.randn(cams.shape, generator=g)
    rows, cols = W, H                                                   # the reference's (H, W) -> dsize quirk
    got = K.cam_prepare(cams.view(-1, h, w).to(dev), rows, cols, 0.2).cpu().numpy()
    for m in range(B * Kc):
        ref = CO.threshold_image(cams.view(-1, h, w)[m].numpy(), rows, cols, 0.2)
        assert (got[m] != ref).mean() <= 5e-4, m
    labels = torch.zeros(B, Kc, dtype=torch.int64); labels[0, [0, 3]] = 1; labels[1, [1, 2, 4]] = 1
    targets = [{"img_label": labels[b].to(dev)} for b in range(B)]
    args = argparse.Namespace(num_classes=Kc, cam_thr=0.2, multi_box_ratio=0.5)
    samples = torch.zeros(B, 3, H, W, device=dev)
    res = camboxes.get_pseudo_label_multi_boxes({"cams_cls": cams.to(dev)}, samples, targets, args)
    for b in range(B):
        eb, el = [], []
        for c in range(Kc):
            if labels[b, c] > 0:
                img = got[b * Kc + c]                                   # same thresholded image: isolates the box logic
                bx = torch.tensor(CO.multi_bboxes_from_image(img, 0.5))
                x0, y0, x1, y1 = bx[..., 0], bx[..., 1], bx[..., 2], bx[..., 3]
                eb.append(torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], -1))
                el += [c + 1] * bx.shape[0]
        eb = torch.cat(eb).float() / torch.tensor([W, H, W, H], dtype=torch.float32)
        assert torch.equal(res[b]["labels"].cpu(), torch.tensor(el)) and torch.allclose(res[b]["boxes"].cpu(), eb)


def test_weight_cache_follows_flat_optimizer(dev):
    """The bf16 weight copies cached for the Linear GEMMs are refreshed after FlatAdamW (which writes parameters through
    raw pointers, without bumping Tensor._version) as well as after a torch optimiser step."""
    from spe_amd import kernels as K
    from spe_amd import ops
    from spe_amd.dp import GradAllReducer
    from spe_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(8)
    W = torch.nn.Parameter((torch.randn(64, 32, generator=g) * 0.1).to(dev))
    b = torch.nn.Parameter(torch.zeros(64, device=dev))
    x = torch.randn(300, 32, generator=g).to(dev)                    # >= LINEAR16_MIN_ROWS: bf16-copy path
    red = GradAllReducer([W, b], flatten_params=True)
    opt = FlatAdamW([W, b], red, lr=0.05, weight_decay=0.0)
    K.set_precision("bf16")                                         # the reference below rounds the operands to bf16 once
    ref = lambda: x.to(torch.bfloat16).double() @ W.detach().to(torch.bfloat16).double().t() + b.detach().double()
    # bf16s: the cached LOW parts of the split weight must follow the optimiser too (reference: the exact product)
    K.set_precision("bf16s")
    for it in range(2):
        red.reset()
        y = ops.linear(x, W, b)
        assert rel(y, x.double() @ W.detach().double().t() + b.detach().double()) < 2e-5, it
        y.square().sum().backward()
        red.finish()
        opt.step()
    K.set_precision("bf16")
    for it in range(3):
        red.reset()
        y = ops.linear(x, W, b)
        assert rel(y, ref()) < 1e-5, it                              # uses the CURRENT weights
        y.square().sum().backward()
        red.finish()
        opt.step()
    sgd = torch.optim.SGD([W, b], lr=0.1)
    red.reset()
    y = ops.linear(x, W, b)
    y.sum().backward()
    red.finish()
    sgd.step()
    assert rel(ops.linear(x, W, b), ref()) < 1e-5


def test_weight_copies_batched_refresh(dev):
    """After weights_changed() the first lookup re-converts EVERY cached weight copy in one spe_cvt_bf16_multi launch:
    bit-identical to the single-matrix conversion, for shapes with partial 64x64 tiles, and dead weights are dropped."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(21)
    shapes = [(384, 384), (1536, 384), (384, 1536), (72, 200), (8, 8), (1000, 24), (65, 129)]
    Ws = [torch.randn(s, generator=g).to(dev) for s in shapes]
    for W in Ws:
        K.weight16(W)
    dead = torch.randn(40, 40, generator=g).to(dev)
    K.weight16(dead)
    del dead
    with torch.no_grad():
        for W in Ws:
            W.view(-1).copy_(torch.randn(W.numel(), generator=g).to(dev) * 3)          # raw update (copy_ bumps _version too)
    K.weights_changed()
    first = K.weight16(Ws[3])
    # all copies are current now (no further conversion launches needed) and equal the single-matrix kernel's output
    key = lambda W: (W.data_ptr(), W.shape[0], W.shape[1])
    for W in Ws:
        ent = K._W16[key(W)]
        assert ent[2] == K._W16_EPOCH
        W16, W16T = K.cvt_bf16(W, True, True, ldt=W.shape[0])
        assert torch.equal(ent[3], W16) and torch.equal(ent[4], W16T), W.shape
        assert torch.equal(W16.float(), W.to(torch.bfloat16).float()) and torch.equal(W16T, W16.t())
    assert first[0] is K._W16[key(Ws[3])][3]
    assert K._W16_TABLE[2] == len(K._W16)                                                # the dead weight left the table
    # a 2-D view of a 4-D filter (the patch embedding) is cached under its address: no new entry per call
    conv = torch.randn(64, 3, 4, 4, generator=g).to(dev)
    a16 = K.weight16(conv.view(64, -1))[0]
    assert K.weight16(conv.view(64, -1))[0] is a16
    # an in-place torch update of one weight (version bump, same epoch) converts just that one
    with torch.no_grad():
        Ws[0].mul_(0.5)
    assert torch.equal(K.weight16(Ws[0])[0].float(), Ws[0].to(torch.bfloat16).float())


def test_patch_embed_large(dev):
    """Patch embedding with >= LINEAR16_MIN_ROWS patches: the dW-only backward of the bf16-copy Linear path."""
    from spe_amd import kernels as K
    from spe_amd import ops
    g = torch.Generator().manual_seed(4)
    img = torch.randn(2, 3, 192, 256, generator=g).to(dev)
    W = (torch.randn(48, 3, 16, 16, generator=g) * 0.05).to(dev).requires_grad_()
    b = torch.randn(48, generator=g).to(dev).requires_grad_()
    y = ops.patch_embed(img, W, b, 16)
    go = torch.randn(y.shape, generator=g).to(dev)
    gW, gb = torch.autograd.grad(y, (W, b), go)
    Wd, bd = W.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    ref = torch.nn.functional.conv2d(img.double(), Wd, bd, stride=16).flatten(2).transpose(1, 2)
    rW, rb = torch.autograd.grad(ref, (Wd, bd), go.double())
    assert rel(y, ref) < 6e-3 and rel(gW, rW) < 1.2e-2 and rel(gb, rb) < 1e-5


def test_position_embedding_sine_vs_reference_golden(dev):
    """pos_sine_kernel through the module vs the vector captured from the reference (padded mask) and a large unpadded grid
    vs the oracle."""
    import os
    from types import SimpleNamespace
    from oracle import spe_oracle as O
    from spe_amd.models.position_encoding import PositionEmbeddingSine
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops.pt"), weights_only=False)["pos_sine"]
    pe = PositionEmbeddingSine(16, normalize=True).to(dev)
    out = pe(SimpleNamespace(tensors=None, mask=g["mask"].to(dev)))
    assert out.shape == g["out"].shape and rel(out, g["out"]) < 2e-6
    mask = torch.zeros(2, 50, 83, dtype=torch.bool); mask[1, 40:, :] = True; mask[1, :, 70:] = True
    pe = PositionEmbeddingSine(192, normalize=True).to(dev)
    assert rel(pe(SimpleNamespace(tensors=None, mask=mask.to(dev))), O.position_embedding_sine(mask, 192)) < 2e-6


@pytest.mark.parametrize("Lq,Lk,H,dk,dv", [(10, 77, 4, 48, 24), (200, 2100, 8, 96, 48), (333, 333, 4, 48, 48), (7, 20, 4, 8, 8), (50, 1000, 2, 16, 64)])
def test_attention_flash(dev, Lq, Lk, H, dk, dv):
    """mha_flash kernels (no score tensor) vs an fp64 restatement, with a key-padding mask; and, with dropout, vs the
    materialising path run with the same Philox (seed, offset) - identical masks, so the results agree."""
    from spe_amd import kernels as K
    from spe_amd import ops
    g = torch.Generator().manual_seed(Lq + Lk)
    B = 2
    q = torch.randn(B, Lq, H, dk, generator=g).to(dev).requires_grad_()
    k = torch.randn(B, Lk, H, dk, generator=g).to(dev).requires_grad_()
    v = torch.randn(B, Lk, H, dv, generator=g).to(dev).requires_grad_()
    mask = torch.zeros(B, Lk, dtype=torch.bool); mask[1, (3 * Lk) // 4:] = True
    scale = dk ** -0.5
    assert ops.FLASH_MHA
    old_min = ops.FLASH_MIN_KEYS
    ops.FLASH_MIN_KEYS = 1                      # force the flash kernels for the small shapes too
    out, pmap = ops.attention(q, k, v, mask.to(dev), scale=scale, p_drop=0.0, need_map=False)
    assert pmap is None and out.grad_fn.__class__.__name__.startswith("_AttentionFlash")
    go = torch.randn(out.shape, generator=g).to(dev)
    gq, gk, gv = torch.autograd.grad(out, (q, k, v), go)
    qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", qd * scale, kd).masked_fill(mask.to(dev)[:, None, None], float("-inf"))
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vd).reshape(B, Lq, H * dv)
    rq, rk, rv = torch.autograd.grad(ref, (qd, kd, vd), go.double())
    assert rel(out, ref) < TOL["bf16"]
    assert rel(gq, rq) < 2 * TOL["bf16"] and rel(gk, rk) < 2 * TOL["bf16"] and rel(gv, rv) < 2 * TOL["bf16"]
    # dropout: same (seed, offset) stream in both implementations
    res = {}
    old = ops.FLASH_MHA
    try:
        for flash in (True, False):
            ops.FLASH_MHA = flash
            K.manual_seed(77)
            o, _ = ops.attention(q, k, v, mask.to(dev), scale=scale, p_drop=0.1, need_map=False)
            res[flash] = (o,) + torch.autograd.grad(o, (q, k, v), go)
    finally:
        ops.FLASH_MHA = old
        ops.FLASH_MIN_KEYS = old_min
    for a_, b_ in zip(res[True], res[False]):
        assert rel(a_, b_) < 2 * TOL["bf16"]


def _split16(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,Kd", [(2048, 384, 384), (8300, 1152, 384), (4150, 1536, 384), (8211, 1144, 448), (2300, 392, 320), (6200, 64, 128)])
@pytest.mark.parametrize("split", [False, True])
def test_gemm_nt2_plain(dev, M, N, Kd, split):
    """csrc/gemm_nt2.hip (128-row LDS-DMA kernels, >= 2048 rows, K % 64 == 0) through spe_gemm_bf16nt: single-term and split
    operands, ragged M / N, bias + ReLU + pre-activation copy, against fp64 on the operands the kernel sees."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(M + N + Kd)
    x = torch.randn(M, Kd, generator=g).to(dev); W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    xh, xl = _split16(x); Wh, Wl = _split16(W)
    ref = (x.double() @ W.double().t() if split else xh.double() @ Wh.double().t()) + b.double()
    C = torch.full((M, N), float("nan"), device=dev); C2 = torch.full((M, N), float("nan"), device=dev)
    K.gemm16(xh, Wh, C, M, N, Kd, Kd, Kd, N, bias=b, C2=C2, act=1, **(dict(Alo=xl, Blo=Wl) if split else {}))
    tol = 2e-5 if split else 1e-6
    assert rel(C2, ref) < tol and rel(C, ref.clamp(min=0)) < tol, (rel(C2, ref), rel(C, ref.clamp(min=0)))


@pytest.mark.parametrize("M,N,Kd", [(8300, 1536, 384), (2100, 384, 1536), (4150, 384, 384)])
def test_gemm_nt2_fused_epilogues(dev, M, N, Kd):
    """The fused epilogues on the gemm_nt2 kernels: fc1 forward (pre-activation + GELU as a (hi, lo) bf16 pair, split operands),
    projection + LayerScale residual (split operands), dh backward (gelu'(aux), bf16 result + column sums, single-term)."""
    from spe_amd import kernels as K
    g = torch.Generator().manual_seed(M * 3 + N)
    x = torch.randn(M, Kd, generator=g).to(dev); W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    xh, xl = _split16(x); Wh, Wl = _split16(W)
    ref = x.double() @ W.double().t() + b.double()
    pre = torch.empty(M, N, device=dev); h = torch.empty(M, N, device=dev, dtype=torch.bfloat16); hl = torch.empty_like(h)
    K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, bias=b, C2=pre, out16=h, out16lo=hl, act=2, Alo=xl, Blo=Wl)
    assert rel(pre, ref) < 2e-5 and rel(h.float() + hl.float(), torch.nn.functional.gelu(ref)) < 3e-5
    assert torch.equal(h, torch.nn.functional.gelu(pre).to(torch.bfloat16)) or rel(h.float(), torch.nn.functional.gelu(ref)) < 3e-3
    res = torch.randn(M, N, generator=g).to(dev); gam = torch.rand(N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev); y = torch.empty(M, N, device=dev)
    K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, bias=b, C=out, C2=y, res=res, rgamma=gam, Alo=xl, Blo=Wl)
    assert rel(y, ref) < 2e-5 and rel(out, res.double() + gam.double() * ref) < 2e-5
    aux = torch.randn(M, N, generator=g).to(dev); o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); cs = torch.zeros(N, device=dev)
    K.gemm16_ex(xh, Wh, M, N, Kd, Kd, Kd, out16=o16, colsum=cs, aux=aux, act=2)
    a64 = aux.double()
    d = 0.5 * (1 + torch.erf(a64 / 2 ** 0.5)) + a64 * torch.exp(-0.5 * a64 * a64) / (2 * 3.141592653589793) ** 0.5
    v = (xh.double() @ Wh.double().t()) * d
    assert rel(o16.float(), v) < 4e-3 and rel(cs, v.sum(0)) < 1e-4


@pytest.mark.parametrize("R,Kd,N,act", [(400, 384, 384, 0), (400, 384, 2048, 1), (400, 2048, 384, 0), (400, 384, 96, 2), (1200, 384, 384, 0),
                                        (182, 384, 384, 1), (130, 40, 24, 0), (2047, 200, 136, 2), (600, 776, 392, 0)])
@pytest.mark.parametrize("prec", ["bf16s", "bf16"])
def test_linear_small_row_kernels(dev, R, Kd, N, act, prec):
    """csrc/linear_small.hip (one launch each way for 128 <= rows < 2048: decoder / encoder / head Linears, reference
    models/transformer.py:206-250, 355-427): forward (split operands in bf16s), input / weight / bias gradient against fp64 - both
    tile configurations, ragged rows / columns / contraction tails, fused ReLU / GELU; the bias gradient must be bitwise reproducible."""
    from spe_amd import kernels as K
    from spe_amd import ops
    assert K._lin_small_ok(R, N, Kd), "shape must take the small-row path"
    g = torch.Generator().manual_seed(R + Kd + N)
    x = torch.randn(R, Kd, generator=g).to(dev).requires_grad_()
    W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev).requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    go = torch.randn(R, N, generator=g).to(dev)
    K.set_precision(prec)
    counts_before = K.lib.count_launches(True)
    try:
        y = ops.linear(x, W, b, act)
        gx, gW, gb = torch.autograd.grad(y, (x, W, b), go)
        counts = K.lib.count_launches(False)
    finally:
        if counts_before is not None:
            K.lib.count_launches(True)
    assert counts.get("spe_linear_small_fwd", 0) == 1 and counts.get("spe_linear_small_bwd", 0) == 1, counts
    xd, Wd, bd = x.detach().double().requires_grad_(), W.detach().double().requires_grad_(), b.detach().double().requires_grad_()
    pre = xd @ Wd.t() + bd
    yd = {0: pre, 1: torch.relu(pre), 2: torch.nn.functional.gelu(pre)}[act]
    rx, rW, rb = torch.autograd.grad(yd, (xd, Wd, bd), go.double())
    ftol = 2e-5 if prec == "bf16s" else 8e-3
    assert rel(y, yd) < ftol, rel(y, yd)
    # backward products on single bf16 operands in both modes; in bf16 the forward's single-term pre-activation flips the ReLU mask
    # of the ~1 % of elements next to zero against fp64
    btol = 6e-2 if (prec == "bf16" and act == 1) else 8e-3
    for a, r in ((gx, rx), (gW, rW), (gb, rb)):
        assert rel(a, r) < btol, rel(a, r)
    for _ in range(5):
        y2 = ops.linear(x, W, b, act)
        g2 = torch.autograd.grad(y2, (x, W, b), go)
        assert torch.equal(y2, y) and all(torch.equal(a, c) for a, c in zip(g2, (gx, gW, gb)))
