"""GPU tests added in round 2 (VERDICT r1 items 1, 2, 6, 7 and the ADVICE findings):

  * the reference's own e2e goldens through the BENCHMARK kernel set (bf16-copy GEMMs, fused MLP / projection nodes,
    flash MHA forced on for the tiny fixtures);
  * cfg4: matcher cost + device Hungarian at Q = 300, Kc in {91, 80}, M in {35, 100, 300}, 6 layers x 2 criteria,
    against fp64 + SciPy;
  * cfg2 / cfg5: fused talking-heads attention at N = 4150 (B = 2) and N = 6200 against the fp64 restatement on the device;
  * fused attention with attn_drop > 0 (the script's drop_attn_rate 0.05): mask recovered from the blocked P'd output;
  * data-parallel path under RCCL in a one-rank group: product model through GradAllReducer + FlatAdamW with the cross-rank
    num_boxes branch of SetCriterion forced;
  * Hungarian kernel on non-finite costs; squared-norm partials on sizes that are not multiples of 4.
"""
import argparse
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_dev(targets, dev):
    return [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]


def build(blob, dev):
    from spe_amd.models import build_model
    args = argparse.Namespace(**blob["args"])
    args.device = "cuda"
    model, crit, crit_r, pp, rpp = build_model(args)
    model.load_state_dict(blob["state_dict"], strict=True)
    return model.to(dev), crit.to(dev), crit_r.to(dev), pp, rpp


class bench_kernel_set:
    """Route even the tiny fixtures (48 token rows, 24 keys) through the kernels that carry the benchmark's time."""

    def __enter__(self):
        from spe_amd import kernels as K, ops
        self.old = (K.LINEAR16_MIN_ROWS, ops.FLASH_MIN_KEYS)
        K.set_precision("bf16")
        K.LINEAR16_MIN_ROWS, ops.FLASH_MIN_KEYS = 16, 1
        return self

    def __exit__(self, *a):
        from spe_amd import kernels as K, ops
        K.LINEAR16_MIN_ROWS, ops.FLASH_MIN_KEYS = self.old
        K.set_precision("bf16")


@pytest.mark.parametrize("name", ["e2e_single", "e2e_two_branch"])
def test_reference_goldens_through_benchmark_kernel_set(dev, name):
    """gemm_bf16nt* (all Linears), _MlpGeluRes, _LinearRes, lsres_bwd16, cvt_bf16*, flash MHA fwd/bwd and the fused
    talking-heads kernels against the REFERENCE's outputs, losses and every parameter gradient."""
    from spe_amd import kernels as K, ops
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    tr, ev = blob["train"], blob["eval"]
    seen = set()
    orig_call = K.lib.call

    def spy(nm, *a):
        seen.add(nm)
        return orig_call(nm, *a)
    with bench_kernel_set():
        K.lib.call = spy
        try:
            model, crit, crit_r, pp, rpp = build(blob, dev)
            model.train(); crit.train(); crit_r.train()
            out = model(NestedTensor(blob["tensors"].to(dev), blob["mask"].to(dev)))
            worst_o = 0.0
            for st, gold in ((0, ev["out0"]), (1, ev["out1"])):           # dropout 0: train-mode outputs = eval-mode outputs
                for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
                    worst_o = max(worst_o, rel(out[st][k], gold[k]))
                worst_o = max(worst_o, rel(out[st]["x_patch"].tensors, gold["x_patch"][0]))
            l0 = crit(out[0], to_dev(blob["targets"], dev), targets_cp=to_dev(tr["targets_cp0"], dev))
            l1 = crit_r(out[1], to_dev(tr["pseudo"], dev), targets_cp=to_dev(tr["targets_cp1"], dev))
            wd = tr["weight_dict"]
            worst_l = 0.0
            for l, ref in ((l0, tr["loss0"]), (l1, tr["loss1"])):
                for k, v in ref.items():
                    if k in wd:
                        worst_l = max(worst_l, abs(float(l[k].detach()) - float(v)) / max(1e-2, abs(float(v))))
            total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
            total.backward()
        finally:
            K.lib.call = orig_call
    worst_g, n = ("", 0.0), 0
    for k, p in model.named_parameters():
        gref = tr["grads"][k]
        if gref is None or float(gref.abs().max()) < 1e-7:
            continue
        r = rel(p.grad, gref)
        n += 1
        if r > worst_g[1]:
            worst_g = (k, r)
    te = abs(float(total.detach()) - float(tr["total"])) / abs(float(tr["total"]))
    print(f"[{name} bench kernel set] outputs {worst_o:.3e}, losses {worst_l:.3e}, total {te:.3e}, worst grad {worst_g} over {n}")
    # the kernels that carry the benchmark's GEMM / attention time really ran
    for nm in ("spe_gemm_bf16nt", "spe_gemm_bf16nt_ex", "spe_cvt_bf16", "spe_layerscale_residual_bwd16", "spe_mha_fwd", "spe_mha_bwd",
               "spe_talking_stats", "spe_talking_flash_fwd", "spe_talking_bwdk_pass1", "spe_talking_bwdq_pass2", "spe_attn_contract"):
        assert nm in seen, f"{nm} was not launched: the fixture did not reach the benchmark kernel set"
    assert worst_o < 3e-2 and worst_l < 2e-2 and te < 2e-2, (worst_o, worst_l, te)
    assert worst_g[1] < 2e-1 and n > 100, worst_g


# ---------------------------------------------------------------------------------------------------------------- cfg4
def _giou(a, b):
    ax0, ay0, ax1, ay1 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx0, by0, bx1, by1 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = (torch.min(ax1[:, None], bx1) - torch.max(ax0[:, None], bx0)).clamp(min=0)
    ih = (torch.min(ay1[:, None], by1) - torch.max(ay0[:, None], by0)).clamp(min=0)
    inter = iw * ih
    union = ((ax1 - ax0) * (ay1 - ay0))[:, None] + (bx1 - bx0) * (by1 - by0) - inter
    cw = (torch.max(ax1[:, None], bx1) - torch.min(ax0[:, None], bx0)).clamp(min=0)
    ch = (torch.max(ay1[:, None], by1) - torch.min(ay0[:, None], by0)).clamp(min=0)
    area = cw * ch
    return inter / union - (area - union) / area


def _rand_boxes(n, g):
    c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(n, 2, generator=g) * 0.35 + 0.05
    return torch.cat([c, wh], 1)


@pytest.mark.parametrize("Kc", [91, 80])
@pytest.mark.parametrize("M", [35, 100, 300])
def test_cfg4_matcher_stress(dev, Kc, M):
    """BASELINE.json configs[3]: Q = 300, COCO-scale class counts, M targets per image (7 / 20 / 60 ground-truth boxes x
    hung_match_ratio 5), 6 decoder layers, for both criteria of an iteration: the cost kernel against fp64
    (reference models/matcher.py:62-83) and the device assignment against SciPy (matcher.py:86), pair for pair."""
    from scipy.optimize import linear_sum_assignment
    from spe_amd import kernels as K
    L, B, Q = 6, 2, 300
    for crit_id in (0, 1):
        g = torch.Generator().manual_seed(Kc * 1000 + M + crit_id)
        logits = (torch.randn(L, B, Q, Kc, generator=g) * 2).to(dev)
        boxes = torch.stack([torch.stack([_rand_boxes(Q, g) for _ in range(B)]) for _ in range(L)]).to(dev)
        sizes = [M, M - 5 * crit_id]
        total = sum(sizes)
        toff = [0, sizes[0], total]
        tgt_ids = torch.randint(1, Kc, (total,), generator=g)
        tgt_boxes = _rand_boxes(total, g)
        toff_t = torch.tensor(toff, dtype=torch.int32, device=dev)
        cost, err = K.matcher_cost(logits, boxes, tgt_ids.int().to(dev), tgt_boxes.to(dev), toff_t, total, 2.0, 5.0, 2.0)
        srow, gidx, lidx = K.hungarian(cost, toff_t, L, B, Q, total, err=err)
        assert err.item() == 0
        ch = cost.cpu()
        es, eg = [], []
        worst, mean, nblk = 0.0, 0.0, 0
        for l in range(L):
            for b in range(B):
                p = logits[l, b].double().sigmoid().cpu()
                ids = tgt_ids[toff[b]:toff[b + 1]]
                tb = tgt_boxes[toff[b]:toff[b + 1]].double()
                neg = 0.75 * p ** 2 * (-(1 - p + 1e-8).log())
                pos = 0.25 * (1 - p) ** 2 * (-(p + 1e-8).log())
                pb = boxes[l, b].double().cpu()
                ref = 5.0 * torch.cdist(pb, tb, p=1) + 2.0 * (pos[:, ids] - neg[:, ids]) - 2.0 * _giou(pb, tb)
                blk = ch[l, Q * toff[b]:Q * toff[b + 1]].view(Q, sizes[b])
                worst = max(worst, float((blk.double() - ref).abs().max()))
                mean += float((blk.double() - ref).abs().mean()); nblk += 1
                i, j = linear_sum_assignment(blk.numpy())
                es.append(torch.as_tensor(i) + (l * B + b) * Q)
                eg.append(torch.as_tensor(j) + toff[b])
        # fp32 like the reference (matcher.py:70-73): at |logit| ~ 8 the reference's own `1 - out_prob` cancels to ~3e-4
        # relative, which the log turns into ~5e-4 absolute on the class cost - the worst element is bounded by that, the mean
        # error shows there is nothing systematic
        assert worst < 2e-3 and mean / nblk < 2e-5, (worst, mean / nblk)
        assert torch.equal(srow.cpu(), torch.cat(es)) and torch.equal(gidx.cpu(), torch.cat(eg))


def test_hungarian_non_finite_costs_do_not_hang(dev):
    """NaN / inf costs (diverged logits) must raise the flag and return, not index LDS out of bounds or spin."""
    from spe_amd import kernels as K
    L, B, Q, sizes = 2, 2, 40, [5, 7]
    total = sum(sizes)
    toff_t = torch.tensor([0, 5, 12], dtype=torch.int32, device=dev)
    for bad in (float("nan"), float("inf"), float("-inf")):
        cost = torch.randn(L, Q * total, device=dev)
        cost[1, Q * 5:] = bad                                   # layer 1, image 1: every cost non-finite
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        srow, gidx, lidx = K.hungarian(cost, toff_t, L, B, Q, total, err=err)
        torch.cuda.synchronize()
        assert err.item() & 2
        assert int(srow.min()) >= 0 and int(srow.max()) < L * B * Q and int(gidx.max()) < total
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    K.hungarian(torch.randn(L, Q * total, device=dev), toff_t, L, B, Q, total, err=err)
    assert err.item() == 0


@pytest.mark.parametrize("n", [1001, 4097, 64, 3, 1 << 20])
def test_sqnorm_partials_any_size(dev, n):
    from spe_amd import kernels as K
    g = torch.randn(n + 4, device=dev)[:n] if n % 4 else torch.randn(n, device=dev)
    g = g.contiguous()
    part = torch.zeros(256, device=dev)
    K.sqnorm_partials(g, part)
    assert abs(float(part.double().sum()) - float(g.double().pow(2).sum())) <= 1e-5 * float(g.double().pow(2).sum())


# -------------------------------------------------------------------------------------------------- cfg2 / cfg5 attention
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


@pytest.mark.parametrize("B,N", [(2, 4150), (1, 6200)])
def test_fused_attention_at_config_token_counts(dev, B, N):
    """The fused attention node (statistics pass, flash forward, the two backward kernels, dK contraction) at the token counts of cfg2 (2 x 4150) and cfg5 (1 x 6200), H = 8,
    dh = 48, against the fp64 restatement of reference models/cait.py:377-389 evaluated on the device."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16")
    H, dh = 8, 48
    g = torch.Generator().manual_seed(N)
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
    errs = {}
    ref_out, ref_g = [], None
    acc = None
    for b in range(B):                                     # per image: bounds the fp64 N x N temporaries (~2.5 GB each)
        dd = [qkv[b:b + 1].detach().double().requires_grad_()] + [t.detach().double().requires_grad_() for t in (Wl, bl, Ww, bw)]
        ref = _talking_ref(*dd, H, scale)
        rg = torch.autograd.grad(ref, dd, go[b:b + 1].double())
        ref_out.append(ref.detach())
        acc = [rg[0]] if acc is None else acc + [rg[0]]
        ref_g = list(rg[1:]) if ref_g is None else [x + y for x, y in zip(ref_g, rg[1:])]
        del ref, rg, dd
    errs["out"] = rel(out, torch.cat(ref_out))
    errs["dqkv"] = rel(grads[0], torch.cat(acc))
    for nm, a, b_ in zip(("dWl", "dbl", "dWw", "dbw"), grads[1:], ref_g):
        errs[nm] = float(a.abs().max() / grads[1].abs().max()) if nm == "dbl" else rel(a, b_)
    print(f"[fused attention B={B} N={N}] " + ", ".join(f"{k} {v:.3e}" for k, v in errs.items()))
    assert torch.isfinite(out).all() and all(torch.isfinite(t).all() for t in grads)
    # forward (fp16 operands, fp32 statistics): north_star's 1e-3 on unit-variance random inputs (measured 3.4e-4 at 2 x 4150);
    # backward (single bf16 operands, bf16 dS / P'd blocks): ~2.5x the measured 3.0e-3 .. 4.0e-3
    assert errs["out"] < 1e-3, errs
    assert errs["dqkv"] < 1e-2 and errs["dWl"] < 1e-2 and errs["dWw"] < 1e-2 and errs["dbw"] < 1e-2, errs
    assert errs["dbl"] < 1e-4, errs          # softmax is shift invariant: the exact gradient is 0


def _dense_from_blocks(T, N):
    """[B,H,nt,nt,64,4] blocked scores -> dense [B,H,N,N]: lane l of block (qt,kt) = query qt*16+(l&15), keys
    kt*16+4*(l>>4)+i (csrc/attn_contract.hip)."""
    B, H, nt = T.shape[:3]
    return T.view(B, H, nt, nt, 4, 16, 4).permute(0, 1, 2, 5, 3, 4, 6).reshape(B, H, nt * 16, nt * 16)[:, :, :N, :N]


@pytest.mark.parametrize("H,N,dh,B,p", [(8, 131, 48, 2, 0.3), (4, 200, 48, 1, 0.05), (8, 1100, 48, 1, 0.05)])
def test_fused_attention_with_dropout(dev, H, N, dh, B, p):
    """Attention dropout (reference scripts/run_voc0712.py: drop_attn_rate 0.05; models/cait.py:387 drops the post-softmax mixed
    probabilities): the keep mask is decoded from the flags the flash forward stores (same seed and offset the autograd node draws), its
    rate is checked, and the output and all gradients are compared with the fp64 restatement using THAT mask - so the forward must apply it
    and both backward kernels must load it."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(H * N + 7)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev).requires_grad_()
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev).requires_grad_()
    bl = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_()
    bw = (0.2 + 0.05 * torch.randn(H, generator=g)).to(dev).requires_grad_()      # P' stays away from 0: the mask is recoverable
    scale = dh ** -0.5
    K.manual_seed(4242)
    seed, off = 4242, 1                                     # what next_rng() hands to the node below
    out = ops.talking_heads_attention(qkv, Wl, bl, Ww, bw, H, scale, p, fused=True)
    go = torch.randn(out.shape, generator=g).to(dev)
    grads = torch.autograd.grad(out, (qkv, Wl, bl, Ww, bw), go)
    # --- the mask: the keep flags an identical flash forward stores (same fragments, statistics, seed and offset as the node's)
    with torch.no_grad():
        from test_round6_gpu import _keep_from_bits
        v5 = qkv.detach().view(B, N, 3, H, dh)
        Qf, Kf, V16 = K.attn_pack_multi([(v5[:, :, 0], scale * K.LOG2E, 32 + K.F16), (v5[:, :, 1], 1.0, 32 + K.F16), (v5[:, :, 2], 1.0, 16 + K.F16)])
        nt = (N + 15) // 16
        spw0, _ = K.fused_plan(B, N)
        ws = torch.empty((B * nt * 8 * H * 32,), device=dev)
        args = [t.detach().contiguous() for t in (Wl, bl, Ww, bw)]
        K.talking_stats(Qf, Kf, args[0], args[1], ws, B, H, N, dh)
        _, _, c0 = K.attn_merge_rows(ws, args[1], B, H, N, spw0)
        bits = K.talking_flash_fwd(Qf, Kf, V16, args[0], args[2], args[3], c0, B, H, N, dh, p, seed, off, want_bits=True)[3]
        keepm = _keep_from_bits(bits, H, N)
        rate = 1.0 - float(keepm.float().mean())
        assert abs(rate - p) < 0.02 + 3.0 * (p * (1 - p) / keepm.numel()) ** 0.5, (rate, p)
    dd = [t.detach().double().requires_grad_() for t in (qkv, Wl, bl, Ww, bw)]
    ref = _talking_ref(*dd, H, scale, keep=keepm.double() / (1.0 - p))
    rg = torch.autograd.grad(ref, dd, go.double())
    errs = {"out": rel(out, ref)}
    for nm, a, b_ in zip(("dqkv", "dWl", "dbl", "dWw", "dbw"), grads, rg):
        errs[nm] = float(a.abs().max() / grads[1].abs().max()) if nm == "dbl" else rel(a, b_)
    print(f"[fused attention dropout p={p} H={H} N={N}] rate {rate:.4f}, " + ", ".join(f"{k} {v:.3e}" for k, v in errs.items()))
    assert all(v < 2e-2 for v in errs.values()), errs


@pytest.mark.parametrize("H,N,B,p", [(4, 200, 1, 0.05), (4, 1100, 1, 0.05), (8, 331, 2, 0.05), (8, 1100, 1, 0.0)])
def test_fused_attention_run_to_run_determinism(dev, H, N, B, p):
    """Forward + backward of the fused talking-heads attention repeated with other kernels in between and the allocator's free
    blocks poisoned: every output and gradient must be BITWISE identical from run to run (no atomics on this path).  Guards
    against reads of memory no kernel wrote and against MFMA dependency hazards: a 16x16x16 MFMA accumulating onto the result of
    the 16x16x32 MFMA issued right before it gave run-to-run different weight gradients on gfx950 (tools/debug/race_mode2.py)."""
    from spe_amd import kernels as K, ops
    dh = 48
    g_ = torch.Generator().manual_seed(5)
    C = H * dh
    qkv0 = (1.5 * torch.randn(B, N, 3 * C, generator=g_)).to(dev)
    Wl0 = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g_)).to(dev); Ww0 = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g_)).to(dev)
    bl0 = (0.1 * torch.randn(H, generator=g_)).to(dev); bw0 = (0.1 * torch.randn(H, generator=g_)).to(dev)
    go = torch.randn(B, N, C, generator=g_).to(dev)
    xs = torch.randn(2048, 1024, device=dev)

    def run():
        K.manual_seed(77)
        t = [x.clone().requires_grad_() for x in (qkv0, Wl0, bl0, Ww0, bw0)]
        out = ops.talking_heads_attention(t[0], t[1], t[2], t[3], t[4], H, dh ** -0.5, p, fused=True)
        return [out.detach()] + [x.detach() for x in torch.autograd.grad(out, t, go)]

    K.set_precision("bf16")
    ref = run()
    for trial in range(10):
        junk = [torch.full((1 << 24,), float("nan") if trial % 2 else 1e30, device=dev) for _ in range(4)]
        small = [torch.full((n,), 1e30, device=dev) for n in (1 << 10, 1 << 14, 1 << 18, 1 << 22) for _ in range(4)]
        del junk, small
        if trial % 3:
            (xs @ xs.t()[:, :512]).sum()
        r = run()
        for nm, a, b_ in zip(("out", "dqkv", "dWl", "dbl", "dWw", "dbw"), r, ref):
            assert torch.isfinite(a).all() and torch.equal(a, b_), (trial, nm, float((a - b_).abs().max()))


def test_atomic_free_kernels_run_to_run_determinism(dev):
    """Contractions, bf16 GEMMs (register pipeline, LDS-DMA ring, TN with split slabs) and the flash MHA forward + backward have
    no atomics: repeated launches with foreign kernels in between must agree bitwise (tools/debug/race_others.py, shortened)."""
    from spe_amd import kernels as K, ops
    K.set_precision("bf16")
    g_ = torch.Generator().manual_seed(3)
    xs = torch.randn(2048, 1024, device=dev)

    def check(name, fn, n=25):
        ref = [r.clone() for r in fn()]
        for t in range(n):
            if t % 3 == 1:
                (xs @ xs.t()[:, :512]).sum()
            elif t % 3 == 2:
                torch.softmax(xs * (1 + t), dim=1)
            for a, b_ in zip(fn(), ref):
                assert torch.equal(a, b_), (name, t)

    B, H, N, dh = 2, 8, 1100, 48
    PT = K.score_blocks(B, H, N, dev); PT.view(torch.int16).random_(0, 16000)
    V16 = K.attn_pack16(torch.randn(B, N, H, dh, generator=g_).to(dev))
    O = torch.empty(B, N, H * dh, device=dev)
    check("contract", lambda: [K.attn_contract(PT, V16, O.view(B, N, H, dh), False).clone()])
    check("contract^T", lambda: [K.attn_contract(PT, V16, O.view(B, N, H, dh), True).clone()])
    for (M, Nn, Kd) in ((8300, 1536, 384), (8300, 384, 1536), (400, 384, 384)):
        A = torch.randn(M, Kd, generator=g_).to(dev).to(torch.bfloat16); Bm = torch.randn(Nn, Kd, generator=g_).to(dev).to(torch.bfloat16)
        C = torch.empty(M, Nn, device=dev)
        check(f"gemm16 {M}x{Nn}x{Kd}", lambda A=A, Bm=Bm, C=C, M=M, Nn=Nn, Kd=Kd: (K.gemm16(A, Bm, C, M, Nn, Kd, Kd, Kd, Nn), [C.clone()])[1])
    A = torch.randn(8300, 1536, generator=g_).to(dev).to(torch.bfloat16); Bm = torch.randn(8300, 384, generator=g_).to(dev).to(torch.bfloat16)
    ws = torch.empty(14, 1536 * 384, device=dev)
    check("gemm16_tn", lambda: (K.gemm16_tn(A, Bm, ws, 1536, 384, 8300, 1536, 384, 384, splitk=-14), [ws.clone()])[1])
    q0 = torch.randn(2, 200, 8, 96, generator=g_).to(dev); k0 = torch.randn(2, 4150, 8, 96, generator=g_).to(dev); v0 = torch.randn(2, 4150, 8, 48, generator=g_).to(dev)
    go = torch.randn(2, 200, 8 * 48, generator=g_).to(dev)

    def mha():
        K.manual_seed(5)
        t = [x.clone().requires_grad_() for x in (q0, k0, v0)]
        o, _ = ops.attention(t[0], t[1], t[2], None, 96 ** -0.5, 0.1)
        return [o.detach()] + [x.detach() for x in torch.autograd.grad(o, t, go.view_as(o))]
    check("flash MHA", mha, 12)


# ------------------------------------------------------------------------------------------ DP path under RCCL, one rank
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dp_path_rccl_world1_product_model(dev):
    """One product-model training step through GradAllReducer (collectives issued through RCCL in a one-rank group, fp32
    and bf16 wire formats) + FlatAdamW, with the cross-rank `num_boxes` branch of SetCriterion (device scalar +
    all-reduce, reference conditional_detr.py:436-440) forced - against the plain single-process path: same losses, same
    updated parameters."""
    import torch.distributed as dist
    from spe_amd import kernels as K
    from spe_amd.dp import GradAllReducer
    from spe_amd.models import conditional_detr as cd
    from spe_amd.optim import FlatAdamW
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    tr = blob["train"]

    def run(mode):
        model, crit, crit_r, pp, rpp = build(blob, dev)
        model.train(); crit.train(); crit_r.train()
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        groups = [{"params": [p for n, p in named if "backbone" not in n], "lr": 2e-3},
                  {"params": [p for n, p in named if "backbone" in n], "lr": 5e-4}]
        red = GradAllReducer([p for _, p in named], bucket_bytes=1 << 16, flatten_params=True, always_reduce=(mode != "plain"),
                             wire_dtype=torch.bfloat16 if mode.endswith("_bf16") else None, comm=comm_obj if mode.startswith("spe_comm") else None)
        assert red.collective == (mode != "plain")
        opt = FlatAdamW(groups, red, weight_decay=1e-2, max_grad_norm=0.1)
        cd.FORCE_NUM_BOXES_ALLREDUCE = mode != "plain"
        samples = NestedTensor(blob["tensors"].to(dev), blob["mask"].to(dev))
        losses = []
        try:
            for it in range(3):
                opt.zero_grad()                                 # the reference loop's call (engine.py:161) re-arms the buckets
                out = model(samples)
                l0 = crit(out[0], to_dev(blob["targets"], dev), targets_cp=to_dev(tr["targets_cp0"], dev))
                l1 = crit_r(out[1], to_dev(tr["pseudo"], dev), targets_cp=to_dev(tr["targets_cp1"], dev))
                wd = tr["weight_dict"]
                total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
                total.backward()
                red.finish()
                opt.step()
                losses.append(float(total.detach()))
        finally:
            cd.FORCE_NUM_BOXES_ALLREDUCE = False
            red.remove()
        return losses, {n: p.detach().clone() for n, p in named}

    K.set_precision("bf16x3")
    created = False
    comm_obj = None
    try:
        la, pa = run("plain")
        # the C-ABI collective layer (libspe_comm.so, include/spe_comm.h): its own communicator, no torch.distributed
        from spe_amd.comm import RcclComm
        comm_obj = RcclComm(rank=0, world=1)
        t = torch.arange(1000, device=dev, dtype=torch.float32)
        comm_obj.all_reduce(t); comm_obj.broadcast(t, 0)
        tb = torch.ones(64, device=dev, dtype=torch.bfloat16)
        comm_obj.all_reduce(tb)
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32)) and float(tb.float().sum()) == 64.0
        ld, pd = run("spe_comm")
        for x, y in zip(la, ld):
            assert abs(x - y) <= 1e-5 * abs(x), (la, ld)
        # ... and with bf16 gradients on its wire (one conversion pass each way around spe_comm_allreduce), through the optimizer steps
        le, pe = run("spe_comm_bf16")
        for x, y in zip(la, le):
            assert abs(x - y) <= 2e-2 * abs(x), (la, le)
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
            created = True
        lb, pb = run("rccl")
        lc, pc = run("rccl_bf16")
    finally:
        K.set_precision("bf16")
        if comm_obj is not None:
            comm_obj.destroy()
        if created:
            dist.destroy_process_group()
    print("plain", la, "rccl", lb, "rccl bf16 wire", lc)
    assert abs(la[0] - float(tr["total"])) <= 2e-4 * abs(float(tr["total"]))       # and it is the reference's loss
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-5 * abs(x), (la, lb)
    errs = sorted(((rel(pb[n], pa[n]), n) for n in pa), reverse=True)
    assert errs[len(errs) // 10][0] < 1e-5, errs[:3]
    for x, y in zip(la, lc):                                    # bf16 gradients on the wire: same trajectory within bf16 rounding
        assert abs(x - y) <= 2e-2 * abs(x), (la, lc)


def _dp2_worker(rank, world, port, out):
    """One of two processes on the SAME GPU (gloo moves the device buckets through the host; RCCL refuses two ranks on one
    device): product model, per-rank data, GradAllReducer + FlatAdamW."""
    import argparse as _ap
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spe_amd import kernels as K
        from spe_amd.dp import GradAllReducer
        from spe_amd.optim import FlatAdamW
        from spe_amd.util.misc import NestedTensor
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        K.set_precision("bf16x3")
        blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
        tr = blob["train"]
        torch.manual_seed(1000 + rank)                      # the reference seeds with seed + rank before build_model
        model, crit, crit_r, pp, rpp = build(blob, dev)
        if rank == 1:                                       # make the replicas differ before the reducer broadcasts rank 0's
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.add_(0.01)
        model.train(); crit.train(); crit_r.train()
        named = [(n, p_) for n, p_ in model.named_parameters() if p_.requires_grad]
        red = GradAllReducer([p_ for _, p_ in named], bucket_bytes=1 << 16, flatten_params=True)
        assert red.collective
        opt = FlatAdamW([{"params": [p_ for _, p_ in named], "lr": 1e-3}], red, weight_decay=1e-2, max_grad_norm=0.1)
        g_ = torch.Generator().manual_seed(50 + rank)       # different images and one target less on rank 1
        img = blob["tensors"] + (0.3 * torch.randn(blob["tensors"].shape, generator=g_) if rank else 0.0)
        samples = NestedTensor(img.to(dev), blob["mask"].to(dev))
        tg, cp0, ps, cp1 = (to_dev(x, dev) for x in (blob["targets"], tr["targets_cp0"], tr["pseudo"], tr["targets_cp1"]))
        wd = tr["weight_dict"]
        if rank == 1:                                       # one target less on rank 1: the ranks' num_boxes differ
            for lst in (cp0, cp1):
                n0 = lst[0]["labels"].shape[0]
                for k in ("boxes", "labels", "scores"):
                    if k in lst[0] and lst[0][k].shape[0] == n0:
                        lst[0][k] = lst[0][k][:-1]

        def loss_of():
            o = model(samples)
            l0 = crit(o[0], tg, targets_cp=cp0)
            l1 = crit_r(o[1], ps, targets_cp=cp1)
            return sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)

        for it in range(2):
            # local gradient of this rank (no bucket hooks: torch.autograd.grad), averaged over the ranks by hand
            ps_ = [p_ for _, p_ in named]
            gl = torch.autograd.grad(loss_of(), ps_, allow_unused=True)
            flat = torch.cat([(g if g is not None else torch.zeros_like(p_)).flatten() for g, p_ in zip(gl, ps_)])
            dist.all_reduce(flat)
            flat /= world
            opt.zero_grad()
            loss_of().backward()
            red.finish()
            got = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).flatten() for p_ in ps_]) * red.grad_scale()
            err = float((got - flat).norm() / flat.norm())
            if err >= 1e-5:                                  # name the worst parameters
                offs, worst = 0, []
                for n_, p_ in named:
                    k = p_.numel()
                    worst.append((float((got[offs:offs + k] - flat[offs:offs + k]).norm()), n_, float(flat[offs:offs + k].norm())))
                    offs += k
                worst.sort(reverse=True)
                raise AssertionError(("bucket gradients != mean of the ranks' gradients", rank, it, err, worst[:4]))
            opt.step()
            mine = torch.cat([p_.detach().flatten() for p_ in ps_])
            both = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert torch.equal(both[0], both[1]), ("replicas diverged", it)
        red.remove()
        out[rank] = True
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_dp_world2_product_model_two_processes_one_gpu(dev):
    """world_size 2 with the PRODUCT model on the GPU: two processes share the device and exchange through gloo.  Checks what a
    one-rank group cannot: rank 0's weights reach rank 1 at construction, the reduced buckets (x the 1/world the fused AdamW
    folds in) equal the mean of the two ranks' local gradients - with `num_boxes` all-reduced between different target
    counts inside both criteria (conditional_detr.py:436-440) - and the replicas stay bit-identical through optimiser steps."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp2_worker, args=(world, port, out), nprocs=world, join=True)
    assert out.get(0) and out.get(1)


def test_torch_ddp_wrapper_world1(dev):
    """The reference's own distributed path, unchanged: `DistributedDataParallel(model, find_unused_parameters=True)`
    (main.py:171-173) around the product model in a one-rank RCCL group, both criteria, torch AdamW - same loss and the same
    parameter gradients as the unwrapped model ("works as is")."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from spe_amd import kernels as K
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    tr = blob["train"]

    def run(wrap):
        model, crit, crit_r, pp, rpp = build(blob, dev)
        model.train(); crit.train(); crit_r.train()
        net = DDP(model, device_ids=[dev.index], find_unused_parameters=True) if wrap else model
        samples = NestedTensor(blob["tensors"].to(dev), blob["mask"].to(dev))
        out = net(samples)
        l0 = crit(out[0], to_dev(blob["targets"], dev), targets_cp=to_dev(tr["targets_cp0"], dev))
        l1 = crit_r(out[1], to_dev(tr["pseudo"], dev), targets_cp=to_dev(tr["targets_cp1"], dev))
        wd = tr["weight_dict"]
        total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
        total.backward()
        return float(total.detach()), {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}

    K.set_precision("bf16x3")
    created = False
    try:
        la, ga = run(False)
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
            created = True
        lb, gb = run(True)
    finally:
        K.set_precision("bf16")
        if created:
            dist.destroy_process_group()
    assert abs(la - float(tr["total"])) <= 2e-4 * abs(float(tr["total"]))
    assert abs(la - lb) <= 1e-6 * abs(la), (la, lb)
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n
        if ga[n] is not None:
            assert rel(gb[n], ga[n]) < 1e-5, (n, rel(gb[n], ga[n]))


@pytest.mark.parametrize("R,C,p", [(400, 384, 0.1), (8300, 384, 0.0), (77, 64, 0.3)])
def test_res_drop_layer_norm(dev, R, C, p):
    """norm(x + dropout(z)) in one kernel each way (reference models/transformer.py:384-386 etc.) against dropout + add +
    LayerNorm composed from the separate ops (same Philox stream -> same mask) and, for p = 0, against fp64."""
    from spe_amd import kernels as K, ops
    g_ = torch.Generator().manual_seed(R + C)
    x = torch.randn(2, R // 2 if R % 2 == 0 else R, C, generator=g_).to(dev)
    if R % 2:
        x = x[:1]
    z = torch.randn(x.shape, generator=g_).to(dev)
    w = (1 + 0.2 * torch.randn(C, generator=g_)).to(dev); b = (0.1 * torch.randn(C, generator=g_)).to(dev)
    go = torch.randn(x.shape, generator=g_).to(dev)
    res = {}
    for fused in (True, False):
        xs, zs, ws, bs = (t.clone().requires_grad_() for t in (x, z, w, b))
        K.manual_seed(99)
        if fused:
            y = ops.res_drop_layer_norm(xs, zs, ws, bs, 1e-5, p, True)
        else:
            y = ops.layer_norm(ops.add(xs, ops.dropout(zs, p, True)), ws, bs, 1e-5)
        res[fused] = (y,) + torch.autograd.grad(y, (xs, zs, ws, bs), go)
    for a, c in zip(res[True], res[False]):
        assert rel(a, c) < 1e-5, rel(a, c)
    if p == 0.0:
        xd, zd, wd_, bd = (t.double().requires_grad_() for t in (x, z, w, b))
        yd = torch.nn.functional.layer_norm(xd + zd, (C,), wd_, bd, 1e-5)
        gd = torch.autograd.grad(yd, (xd, zd, wd_, bd), go.double())
        for a, c in zip(res[True], (yd,) + gd):
            assert rel(a, c) < 1e-5
    else:
        dz, dx = res[True][2], res[True][1]
        dropped = (dz == 0) & (dx != 0)
        assert abs(float(dropped.float().mean()) - p) < 0.02


@pytest.mark.parametrize("M,N,R,sk", [(384, 384, 8300, 14), (1536, 384, 8300, 14), (384, 1536, 8300, 1), (1152, 384, 8300, 8), (72, 136, 400, 1),
                                      (64, 64, 77, 1), (384, 768, 4241, 4), (8, 8, 5, 1)])
def test_gemm_bf16tn(dev, M, N, R, sk):
    """spe_gemm_bf16tn (weight gradient on row-major operands through LDS transpose reads) against an fp64 product of the
    same bf16-rounded values; asymmetric random operands (a transposed or mis-tiled read cannot pass)."""
    from spe_amd import kernels as K
    g_ = torch.Generator().manual_seed(M + N + R)
    A = torch.randn(R, M, generator=g_).to(dev).to(torch.bfloat16)
    Bm = (torch.randn(R, N, generator=g_) + 0.3).to(dev).to(torch.bfloat16)
    ref = 0.7 * (A.double().t() @ Bm.double())
    if sk > 1:
        ws = torch.full((sk, M * N), float("nan"), device=dev)
        K.gemm16_tn(A, Bm, ws, M, N, R, M, N, N, alpha=0.7, splitk=-sk)
        C = ws.sum(0).view(M, N)
    else:
        C = torch.full((M, N), float("nan"), device=dev)
        K.gemm16_tn(A, Bm, C, M, N, R, M, N, N, alpha=0.7)
    assert torch.isfinite(C).all()
    assert rel(C, ref) < 2e-6, rel(C, ref)


@pytest.mark.parametrize("case", ["one_empty", "all_empty", "more_targets_than_queries", "more_targets_refine"])
def test_criterion_edge_cases_match_oracle(dev, case):
    """SetCriterion / SetCriterionRefine on ragged target lists against the oracle (= the reference's formulas with SciPy's
    assignment): an image without boxes (all its queries are background, `num_boxes` clamps at 1), a batch without any box, and
    more targets than queries (the assignment is then Q pairs per image - what the 5x one-to-many jitter of training produces
    for crowded images).  Every loss key and the gradients w.r.t. logits and boxes, all decoder layers."""
    import copy
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import spe_oracle as O
    from spe_amd.models import build_model
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    args = argparse.Namespace(**blob["args"]); args.device = "cuda"
    _, crit, crit_r, _, _ = build_model(args)
    refine = case == "more_targets_refine"
    cr = (crit_r if refine else crit).to(dev).eval()
    g_ = torch.Generator().manual_seed(11)
    B, Q, Kc, Kimg, L = 2, 10, 21, 20, 3
    counts = {"one_empty": [0, 4], "all_empty": [0, 0], "more_targets_than_queries": [14, 3], "more_targets_refine": [13, 11]}[case]
    def stage():
        return {"pred_logits": torch.randn(B, Q, Kc, generator=g_), "pred_boxes": torch.rand(B, Q, 4, generator=g_) * 0.5 + 0.2}
    outs = stage(); outs["aux_outputs"] = [stage() for _ in range(L - 1)]
    outs["x_logits"] = torch.randn(B, Kimg, generator=g_); outs["x_cls_logits"] = torch.randn(B, Kimg, generator=g_)
    targets = []
    for m in counts:
        t = {"labels": torch.randint(1, Kc - 1, (m,), generator=g_), "boxes": torch.rand(m, 4, generator=g_) * 0.4 + 0.2,
             "img_label": (torch.rand(Kimg, generator=g_) > 0.7).long(), "orig_size": torch.tensor([64, 64])}
        if refine:
            t["scores"] = torch.rand(m, generator=g_) * 0.9 + 0.05
        targets.append(t)
    def leaves(o, dev_):
        c = {k: (v.clone().to(dev_).requires_grad_() if torch.is_tensor(v) else v) for k, v in o.items() if k != "aux_outputs"}
        c["aux_outputs"] = [{k: v.clone().to(dev_).requires_grad_() for k, v in a.items()} for a in o["aux_outputs"]]
        return c
    o_ref, o_dev = leaves(outs, "cpu"), leaves(outs, dev)
    ref = O.set_criterion(o_ref, copy.deepcopy(targets), refine=refine)
    got = cr(o_dev, to_dev(targets, dev))
    assert set(got.keys()) == set(ref.keys()), (sorted(got.keys()), sorted(ref.keys()))
    for k in ref:
        assert torch.isfinite(got[k]).all(), k
        gv, rv = float(got[k].detach()), float(ref[k].detach())
        assert abs(gv - rv) <= 2e-5 * max(1.0, abs(rv)), (k, gv, rv)
    wd = cr.weight_dict
    sum(ref[k] * wd[k] for k in ref if k in wd).backward()
    sum(got[k] * wd[k] for k in got if k in wd).backward()
    for a_, b_ in [(o_dev, o_ref)] + list(zip(o_dev["aux_outputs"], o_ref["aux_outputs"])):
        for k in ("pred_logits", "pred_boxes"):
            gr = b_[k].grad if b_[k].grad is not None else torch.zeros_like(b_[k])
            gd = a_[k].grad if a_[k].grad is not None else torch.zeros_like(a_[k])
            assert (gd.cpu() - gr).abs().max() <= 1e-5 * max(1.0, float(gr.abs().max())), (k, float((gd.cpu() - gr).abs().max()))


@pytest.mark.parametrize("M,N,Kd", [(2100, 384, 1536), (8300, 200, 1024), (2048, 512, 1152), (2051, 64, 4608)])
def test_gemm_bf16nt_ring(dev, M, N, Kd):
    """The LDS-DMA ring variant of spe_gemm_bf16nt / _ex (128x64 tiles, taken for M >= 2048, K >= 1024, N <= 512: fc2 forward,
    fc1 / qkv input gradients) against an fp64 product of the same bf16 values: plain with bias, and the extended epilogue with the
    LayerScale residual, the pre-residual copy and the bf16 copy.  Ragged M / N exercise the clamped rows of the DMA."""
    from spe_amd import kernels as K
    g_ = torch.Generator().manual_seed(M + N + Kd)
    A = torch.randn(M, Kd, generator=g_).to(dev).to(torch.bfloat16)
    Bm = (torch.randn(N, Kd, generator=g_) + 0.2).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g_).to(dev)
    ref = A.double() @ Bm.double().t() + bias.double()
    C = torch.full((M, N), float("nan"), device=dev)
    K.gemm16(A, Bm, C, M, N, Kd, Kd, Kd, N, bias=bias)
    assert torch.isfinite(C).all()
    assert rel(C, ref) < 2e-6, rel(C, ref)
    res = torch.randn(M, N, generator=g_).to(dev)
    gam = torch.randn(N, generator=g_).to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    y = torch.full((M, N), float("nan"), device=dev)
    o16 = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    K.gemm16_ex(A, Bm, M, N, Kd, Kd, Kd, bias=bias, C=out, C2=y, out16=o16, res=res, rgamma=gam)
    assert rel(y, ref) < 2e-6, rel(y, ref)
    assert rel(out, res.double() + gam.double() * ref) < 2e-6
    assert rel(o16.float(), ref) < 4e-3


@pytest.mark.parametrize("tag", ["small", "many"])
def test_matcher_cost_matches_reference_golden(dev, tag):
    """spe_matcher_cost against the cost matrices the REFERENCE handed to SciPy (tests/golden/ops.pt, captured at the
    reference's linear_sum_assignment call by tools/gen_golden.py): 2 images, M > Q and an image without targets; and the
    device Hungarian on them gives the reference's pairs."""
    from spe_amd import kernels as K
    g_ = torch.load(os.path.join(GOLD, "ops.pt"), weights_only=False)[f"matcher_{tag}"]
    logits, boxes = g_["outputs"]["pred_logits"].to(dev)[None], g_["outputs"]["pred_boxes"].to(dev)[None]      # L = 1
    B, Q = logits.shape[1], logits.shape[2]
    sizes = [int(len(t["labels"])) for t in g_["targets"]]
    toff = [0]
    for s_ in sizes:
        toff.append(toff[-1] + s_)
    tgt_ids = torch.cat([t["labels"] for t in g_["targets"]]).int().to(dev)
    tgt_boxes = torch.cat([t["boxes"] for t in g_["targets"]]).float().to(dev)
    toff_t = torch.tensor(toff, dtype=torch.int32, device=dev)
    cost, err = K.matcher_cost(logits, boxes, tgt_ids, tgt_boxes, toff_t, toff[-1], 2.0, 5.0, 2.0)
    assert int(err.item()) == 0
    for b in range(B):
        if sizes[b] == 0:
            continue
        got = cost[0, Q * toff[b]:Q * toff[b + 1]].view(Q, sizes[b]).cpu()
        ref = g_["cost"][b].float()
        assert (got - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max())), (b, float((got - ref).abs().max()))
    from spe_amd.models.matcher import HungarianMatcher
    m = HungarianMatcher(2, 5, 2, 5)
    tg = to_dev(g_["targets"], dev)
    per = m.match_many(logits, boxes, tg)[0]                  # host SciPy on the device cost (the path taken when M > Q)
    for b in range(B):
        assert torch.equal(per[b][0], g_["indices"][b][0]) and torch.equal(per[b][1], g_["indices"][b][1]), b
    flat = m.match_flat(logits, boxes, tg)                    # device Hungarian (None when an image has more targets than queries)
    assert (flat is None) == (max(sizes) > Q)
    if flat is not None:
        srow, gidx = flat[0].cpu(), flat[1].cpu()
        for b in range(B):
            ri, rj = g_["indices"][b]
            sel = (srow >= b * Q) & (srow < (b + 1) * Q)
            pairs = sorted(zip((srow[sel] - b * Q).tolist(), (gidx[sel] - toff[b]).tolist()))
            assert pairs == sorted(zip(ri.tolist(), rj.tolist())), (b, pairs)


def test_cam_prepare_known_answers(dev):
    """Hand-derived cases of the CAM -> thresholded image stage (reference cams_deit.resize_cam + get_multi_bboxes,
    engine.py:356-398) from the documented semantics of the cv2 calls it makes: `cv2.resize(..., INTER_LINEAR)` samples at
    src = (dst + 0.5) * (src_size / dst_size) - 0.5 with replicated borders; min-max normalisation; `np.uint8(255 * cam)`
    truncates; `cv2.threshold(img, int(thr * img.max()), 255, THRESH_TOZERO)` keeps values STRICTLY above the threshold.
      1 x 2 map [0, 1] -> 1 x 4: samples at -0.25, 0.25, 0.75, 1.25 -> [0, .25, .75, 1] -> uint8 [0, 63, 191, 255];
      thr 0.5 -> level int(127.5) = 127 -> [0, 0, 191, 255]; thr 0.2 -> level 51 -> [0, 63, 191, 255];
      2 x 2 map [[0, 1], [1, 2]] -> 4 x 4 is the separable product: corner 0, then .25 steps -> after normalisation (max 2)
      row 0 = [0, .125, .375, .5], and the image is symmetric."""
    from spe_amd import kernels as K
    m = torch.tensor([[[0.0, 1.0]]], device=dev)
    assert K.cam_prepare(m, 1, 4, 0.5).cpu().view(-1).tolist() == [0, 0, 191, 255]
    assert K.cam_prepare(m, 1, 4, 0.2).cpu().view(-1).tolist() == [0, 63, 191, 255]
    m2 = torch.tensor([[[0.0, 1.0], [1.0, 2.0]]], device=dev)
    got = K.cam_prepare(m2, 4, 4, 0.0).cpu().view(4, 4)
    u = [0.0, 0.25, 0.75, 1.0]
    exp = torch.tensor([[int(255 * ((u[i] + u[j]) / 2.0)) for j in range(4)] for i in range(4)], dtype=torch.uint8)
    assert torch.equal(got, exp), (got, exp)
    assert torch.equal(got, got.t())
    # a constant offset and scale of the map change nothing (min-max normalisation)
    assert torch.equal(K.cam_prepare(3.0 * m2 - 7.0, 4, 4, 0.0).cpu().view(4, 4), exp)


def test_nms_known_answers(dev):
    """Hand-derived cases of greedy per-class NMS as torchvision documents it (`torchvision.ops.nms`: "iteratively removes
    lower scoring boxes which have an IoU greater than iou_threshold with another (higher scoring) box"; the reference calls it
    per predicted class at engine_loc.py:154-174): the threshold is strict (IoU == 0.5 survives), suppression is by KEPT boxes
    only (a box removed earlier suppresses nobody), classes do not interact, survivors come back per class in score order."""
    from spe_amd import infer
    boxes = torch.tensor([
        [0., 0., 10., 10.],      # 0  class 1  s .90  kept
        [0., 0., 10., 5.],       # 1  class 1  s .80  IoU with 0 = 50/100 = 0.5 exactly -> kept (not > 0.5)
        [1., 1., 10., 10.],      # 2  class 1  s .70  IoU with 0 = 81/100 -> removed
        [0., 0., 10., 10.],      # 3  class 2  s .60  same box as 0, other class -> kept
        [20., 0., 30., 10.],     # 4  class 2  s .95  kept
        [24., 0., 34., 10.],     # 5  class 2  s .85  IoU with 4 = 60/140 = 0.43 -> kept
        [28., 0., 38., 10.],     # 6  class 2  s .75  IoU with 5 = 0.43, with 4 = 20/180 -> kept
        [21., 0., 31., 10.],     # 7  class 2  s .50  IoU with 4 = 90/110 = 0.82 -> removed
        [22.5, 0., 32.5, 10.],   # 8  class 2  s .40  IoU with 4 = 75/125 = 0.6 -> removed; 7 (removed) would not have mattered
    ])
    labels = torch.tensor([1, 1, 1, 2, 2, 2, 2, 2, 2])
    scores = torch.tensor([.90, .80, .70, .60, .95, .85, .75, .50, .40])
    out = infer.per_class_nms([{"scores": scores.to(dev), "labels": labels.to(dev), "boxes": boxes.to(dev)}], 0.5)[0]
    assert out["labels"].tolist() == [1, 1, 2, 2, 2, 2]
    assert torch.allclose(out["scores"].cpu(), torch.tensor([.90, .80, .95, .85, .75, .60]))
    assert torch.equal(out["boxes"].cpu(), boxes[[0, 1, 4, 5, 6, 3]])


# ---------------------------------------------------------------------------------------------------------------------
# round 4: flash-style talking-heads forward (csrc/attn_flash.hip) - no N x N tensor in HBM
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,N,dh,p", [(1, 4, 12, 8, 0.0), (2, 4, 35, 8, 0.1), (2, 4, 196, 48, 0.0), (1, 8, 130, 48, 0.1),
                                        (2, 8, 1100, 48, 0.0), (1, 4, 300, 32, 0.05), (1, 4, 257, 64, 0.0), (1, 4, 77, 24, 0.1)])
def test_flash_forward_vs_fp64(dev, B, H, N, dh, p):
    """spe_talking_stats + spe_attn_merge_rows + spe_talking_flash_fwd against the fp64 restatement of reference models/cait.py:377-389 evaluated on
    the operands as the kernels see them (q scale log2 e, k, v rounded to fp16), with the dropout mask decoded from the stored keep flags: what
    remains is the fp16 rounding of P and P' in front of the matrix instructions.  Ragged N, both head counts, every head-dim decomposition (tail
    only, full + tail, full only); bitwise reproducible; the bf16 (hi, lo) copy of O is exact to fp32 rounding."""
    from spe_amd import kernels as K
    from test_round6_gpu import _keep_from_bits
    g = torch.Generator().manual_seed(17 * N + H)
    C = H * dh
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dev)
    Wl = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bl = (0.1 * torch.randn(H, generator=g)).to(dev)
    Ww = (torch.eye(H) + 0.3 * torch.randn(H, H, generator=g)).to(dev); bw = (0.1 * torch.randn(H, generator=g)).to(dev)
    scale = dh ** -0.5
    v5 = qkv.view(B, N, 3, H, dh)
    Qf, Kf, V16 = K.attn_pack_multi([(v5[:, :, 0], scale * K.LOG2E, 32 + K.F16), (v5[:, :, 1], 1.0, 32 + K.F16), (v5[:, :, 2], 1.0, 16 + K.F16)])
    nt = (N + 15) // 16
    spw0, _ = K.fused_plan(B, N)
    ws = torch.zeros(B * nt * 8 * H * 32, device=dev)
    K.talking_stats(Qf, Kf, Wl, bl, ws, B, H, N, dh)
    M, IL, c0 = K.attn_merge_rows(ws, bl, B, H, N, spw0)
    O, O16, O16lo, bits = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p, 11, 5, True, True, want_bits=True)
    O2, _, _ = K.talking_flash_fwd(Qf, Kf, V16, Wl, Ww, bw, c0, B, H, N, dh, p, 11, 5)
    assert torch.isfinite(O).all()
    assert torch.equal(O, O2)                                       # fixed summation order: bitwise reproducible
    qh = (v5[:, :, 0].permute(0, 2, 1, 3) * (scale * K.LOG2E)).half().double() / K.LOG2E
    kh = v5[:, :, 1].permute(0, 2, 1, 3).half().double()
    vh = v5[:, :, 2].permute(0, 2, 1, 3).half().double()
    S = qh @ kh.transpose(-2, -1)                                    # [B,H,N,N]
    Sp = torch.einsum("gh,bhqk->bgqk", Wl.double(), S) + bl.double()[None, :, None, None]
    # the statistics the kernels use: log2-domain row max and 1 / row sum
    assert rel(M, (Sp * K.LOG2E).amax(-1)) < 1e-6 and rel(IL, 1.0 / torch.exp2(Sp * K.LOG2E - (Sp * K.LOG2E).amax(-1, keepdim=True)).sum(-1)) < 1e-5
    P = Sp.softmax(-1)
    Pp = torch.einsum("gh,bhqk->bgqk", Ww.double(), P) + bw.double()[None, :, None, None]
    if p > 0:
        Pp = Pp * _keep_from_bits(bits, H, N).double() / (1.0 - p)
    Oref = (Pp @ vh).permute(0, 2, 1, 3).reshape(B, N, C)
    err = rel(O, Oref)
    print(f"[flash forward vs fp64 B={B} H={H} N={N} dh={dh} p={p}] {err:.2e}")
    assert err < 6e-4, err
    assert float((O16.float().view_as(O) + O16lo.float().view_as(O) - O).abs().max()) <= 2e-5 * float(O.abs().max())
