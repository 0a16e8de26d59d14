"""Error budget of the forward pass: which operand rounding contributes what to pred_logits / losses.

    python tools/error_budget.py [case ...] [--variants a,b,..] [--fwd-only]        # GPU box

Variants `bf16s` (the benchmark mode as shipped), `bf16s_noflash` (its decoder cross attention on the materialised split
kernels instead of the flash kernels) and `bf16` (single bf16 operands everywhere) run the product as is.  Every other variant
starts from `ref` = the bf16s POLICY evaluated with the library's most exact kernels (every forward product on 3-term split
operands through the fp32-operand GEMM, fp32 materialised attention; backward products on single bf16 operands) and degrades
ONE class of forward operands by rounding it through the candidate storage format before the (otherwise exact) product -
emulation in this tool only, no product code involved.  The case's reference fixture (tests/golden/cfg_*.pt) gives the error:

  qk_bf16      q*scale, k rounded to bf16 before Q K^T                    (what the bf16 fused kernels do)
  pd_bf16      P'd and v rounded to bf16 before P'd V                      (bf16 score blocks + bf16 V fragments)
  pd_fp16      P'd * 2^10 and v rounded to fp16                            (fp16 score blocks, same bytes as bf16)
  pd_fp16_vx   P'd * 2^10 rounded to fp16, v exact                          (fp16 blocks against split V)
  mixw_bf16    softmax output rounded to bf16 before the Ww head mix       (the matrix-pipe bf16 mix)
  lin_bf16     x, W of every Linear with >= 128 rows rounded to bf16       (single-term GEMMs)
  bblin_bf16 / declin_bf16   the same for the backbone's / the decoder's (incl. memory-side projections) Linears only
  lin_fp16     x, W of every Linear with >= 128 rows rounded to fp16
  mlp_fp16 / qkv_fp16 / proj_fp16   the same for ONE family of backbone Linears (fc1 + fc2, qkv, the output projection; round 5)
  a+b          several at once, e.g. qk_bf16+pd_bf16+mixw_bf16 = the bf16 fused attention forward
Results: one JSON line per (case, variant) -> stdout and gpurun_out/error_budget.jsonl.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfg_cases as cc  # noqa: E402
import test_config_golden as tg  # noqa: E402
from spe_amd import kernels as K  # noqa: E402
from spe_amd import ops  # noqa: E402
from spe_amd.util.misc import NestedTensor  # noqa: E402

EMU = set()


def rnd(t, dt):
    return t.to(dt).to(torch.float32)


class _EmuTalking(torch.autograd.Function):
    """ops._TalkingHeadsAttention.forward with optional operand roundings; backward = the product's."""

    @staticmethod
    def forward(ctx, qkv, Wl, bl, Ww, bw, H, scale, p_drop):
        B, N, C3 = qkv.shape
        C = C3 // 3
        dh = C // H
        qkv = qkv.contiguous()
        src = qkv
        if "qk_bf16" in EMU:
            src = qkv.clone()
            s5 = src.view(B, N, 3, H, dh)
            s5[:, :, 0] = rnd(s5[:, :, 0] * scale, torch.bfloat16) / scale
            s5[:, :, 1] = rnd(s5[:, :, 1], torch.bfloat16)
        v5 = src.view(B, N, 3, H, dh)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        ld = K.pad4(N)
        S = torch.empty((B, H, N, ld), device=qkv.device, dtype=torch.float32)
        sq = (N * C3, dh)
        sS = (H * N * ld, N * ld)
        K.gemm(q, k, S, N, N, dh, C3, C3, ld, False, True, batch0=B, batch1=H, sA=sq, sB=sq, sC=sS, alpha=scale)
        if "mixw_bf16" in EMU:
            # softmax(Wl S + bl) rounded to bf16, then Ww mix: evaluated with torch on the materialised tensor
            Sm = torch.einsum("gh,bhqk->bgqk", Wl, S[..., :N]) + bl.view(1, H, 1, 1)
            P = torch.softmax(Sm, -1)
            del Sm
            Pd = torch.einsum("gh,bhqk->bgqk", Ww, rnd(P, torch.bfloat16)) + bw.view(1, H, 1, 1)
            Pf = torch.zeros_like(S); Pf[..., :N] = P
            Pdf = torch.zeros_like(S); Pdf[..., :N] = Pd
            P, Pd = Pf, Pdf
        else:
            P, Pd = K.talking_fwd(S, Wl, bl, Ww, bw, B, H, N, N, ld, 0.0, 0, 0)
        Pd_use, v_use, alpha = Pd, v, 1.0
        if "pd_bf16" in EMU:
            Pd_use = rnd(Pd, torch.bfloat16)
            vs = qkv.clone(); vs.view(B, N, 3, H, dh)[:, :, 2] = rnd(v, torch.bfloat16); v_use = vs.view(B, N, 3, H, dh)[:, :, 2]
        if "pd_fp16" in EMU or "pd_fp16_vx" in EMU:
            Pd_use = rnd(Pd * 1024.0, torch.float16)
            alpha = 1.0 / 1024.0
            if "pd_fp16" in EMU:
                vs = qkv.clone(); vs.view(B, N, 3, H, dh)[:, :, 2] = rnd(v, torch.float16); v_use = vs.view(B, N, 3, H, dh)[:, :, 2]
        O = torch.empty((B, N, C), device=qkv.device, dtype=torch.float32)
        K.gemm(Pd_use, v_use, O, N, dh, N, ld, C3, C, False, False, batch0=B, batch1=H, sA=sS, sB=sq, sC=(N * C, dh), alpha=alpha)
        ctx.meta = (B, N, C, H, dh, ld, scale, 0.0, 0, 0)
        ctx.save_for_backward(qkv, P, Pd, Wl, Ww)
        return O

    backward = staticmethod(ops._TalkingHeadsAttention.backward)


_orig_tha = ops.talking_heads_attention
_orig_linear_fwd = K.linear_fwd


def _tha(qkv, Wl, bl, Ww, bw, num_heads, scale, p_drop=0.0, fused=None):
    if EMU & {"qk_bf16", "pd_bf16", "pd_fp16", "pd_fp16_vx", "mixw_bf16"}:
        return _EmuTalking.apply(qkv, Wl, bl, Ww, bw, num_heads, scale, p_drop)
    return _orig_tha(qkv, Wl, bl, Ww, bw, num_heads, scale, p_drop, False if REF[0] else fused)


IN_DEC = [False]      # inside model.transformer (the decoder and its memory-side projections)
REF = [False]         # `ref` settings active: most exact kernels under the bf16s policy
PRODUCT = ("bf16s", "bf16s_noflash", "bf16")


def _linear_fwd(x2, W, b, act=0, want_pre=False, save_for_dw=True, src=None):
    if x2.shape[0] >= 128 and not K._in_bwd():
        if "lin_bf16" in EMU or ("bblin_bf16" in EMU and not IN_DEC[0]) or ("declin_bf16" in EMU and IN_DEC[0]):
            y, pre, _ = _orig_linear_fwd(rnd(x2, torch.bfloat16), rnd(W, torch.bfloat16), b, act, want_pre, save_for_dw, None)
            return y, pre, x2
        n_out, n_in = W.shape
        fam = None if IN_DEC[0] else ("mlp" if (n_out == 4 * n_in or n_in == 4 * n_out) else ("qkv" if n_out == 3 * n_in else ("proj" if n_out == n_in else None)))
        if "lin_fp16" in EMU or (fam is not None and (fam + "_fp16") in EMU):
            y, pre, _ = _orig_linear_fwd(rnd(x2, torch.float16), rnd(W, torch.float16), b, act, want_pre, save_for_dw, None)
            return y, pre, x2
        if fam is not None and (fam + "_xfp16") in EMU:          # the activation alone in fp16 (the weight keeps its split)
            y, pre, _ = _orig_linear_fwd(rnd(x2, torch.float16), W, b, act, want_pre, save_for_dw, None)
            return y, pre, x2
    return _orig_linear_fwd(x2, W, b, act, want_pre, save_for_dw, src)


ops.talking_heads_attention = _tha
K.linear_fwd = _linear_fwd


def run(name, variant, fwd_only, dev):
    blob = torch.load(os.path.join(tg.GOLD, f"cfg_{name}.pt"), weights_only=False)
    args, (model, crit, crit_r, pp, rpp), tensors, mask, targets = cc.build_case(name)
    EMU.clear()
    K.set_precision("bf16" if variant == "bf16" else "bf16s")
    REF[0] = variant not in PRODUCT
    saved = (K.LINEAR16, ops.FLASH_MHA)
    if REF[0]:
        K.LINEAR16, ops.FLASH_MHA = False, False
        if variant != "ref":
            EMU.update(variant.split("+"))
    elif variant == "bf16s_noflash":
        ops.FLASH_MHA = False
    model.to(dev).train(); crit.to(dev).eval(); crit_r.to(dev).eval()
    _tf = model.transformer.forward

    def _tf_wrapped(*a, **k):
        IN_DEC[0] = True
        try:
            return _tf(*a, **k)
        finally:
            IN_DEC[0] = False
    model.transformer.forward = _tf_wrapped
    tgd = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    with torch.set_grad_enabled(not fwd_only):
        out = model(NestedTensor(tensors.to(dev), mask.to(dev)))
        l0 = crit(out[0], tgd)
        pseudo = [{k: v.to(dev) for k, v in p.items()} for p in blob["pseudo"]]
        l1 = crit_r(out[1], pseudo)
        wd = blob["weight_dict"]
        total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
        if not fwd_only:
            total.backward()
    torch.cuda.synchronize()
    oe = tg.compare_outputs(out, blob)
    le = tg.compare_losses(l0, l1, blob, skip_logging=True)
    wl = {k: v for k, v in le.items() if k.split(".", 1)[1] in wd}
    rec = {"case": name, "variant": variant, "pred_logits": oe["0.pred_logits"], "pred_boxes": oe["0.pred_boxes"],
           "x_patch": oe["0.x_patch"], "worst_output": max(oe.items(), key=lambda kv: kv[1]),
           "worst_weighted_loss": max(wl.items(), key=lambda kv: kv[1]), "worst_loss": max(le.items(), key=lambda kv: kv[1]),
           "total_loss": abs(float(total.detach()) - float(blob["total"])) / abs(float(blob["total"])),
           "peak_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if not fwd_only:
        ge = tg.compare_grads([(k, p.grad) for k, p in model.named_parameters()], blob)
        gs = sorted(ge.values())
        rec.update(median_grad=gs[len(gs) // 2], p90_grad=gs[(9 * len(gs)) // 10], worst_grad=max(ge.items(), key=lambda kv: kv[1]))
    K.set_precision("bf16")
    K.LINEAR16, ops.FLASH_MHA = saved
    EMU.clear()
    del model, out, l0, l1, total
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["cfg1", "cfg2_depth2"])
    ap.add_argument("--variants", default="bf16s,bf16s_noflash,bf16,ref")
    ap.add_argument("--fwd-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    od = os.path.join(ROOT, "gpurun_out")
    os.makedirs(od, exist_ok=True)
    with open(os.path.join(od, "error_budget.jsonl"), "a") as fh:
        for name in a.cases:
            for v in a.variants.split(","):
                try:
                    rec = run(name, v, a.fwd_only, dev)
                except Exception as e:                      # keep the sweep going (e.g. out of memory at full depth)
                    rec = {"case": name, "variant": v, "error": repr(e)[:300]}
                    torch.cuda.empty_cache()
                line = json.dumps(rec)
                print(line, flush=True)
                fh.write(line + "\n"); fh.flush()


if __name__ == "__main__":
    main()
