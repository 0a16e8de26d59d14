"""Parity at BASELINE.json's own dimensions, against the REFERENCE (fixtures tests/golden/cfg_*.pt written by
tools/gen_config_golden.py from the imported reference; case table and seeded weights: tests/cfg_cases.py).

  CPU  (-m "not gpu"): the oracle against the reference at cfg1 (real TSCAM_cait_XXS24, 24 blocks) and at cfg2 token
                       counts - the oracle is pinned at real dimensions, not only on the tiny e2e fixtures.
  GPU  (-m gpu)      : the product against the same reference results, in every precision mode:
        bf16s  - THE BENCHMARK MODE (bench.py's default): forward products on split bf16 operands (Linear GEMMs) / fp16
                 operands (fused talking-heads attention, flash MHA), backward products on single bf16 operands.  Asserted at
                 north_star's 1e-3 on EVERY output and EVERY loss key (measured <= 2.2e-4), at the depth-2 cases and at the
                 FULL-DEPTH cases cfg2_full (24 blocks, batch 2, N = 4150) and cfg5_full (36 blocks, N = 6200) - this is the
                 test that pins the benchmarked kernel set (fused attention, split bf16-copy GEMMs with fused epilogues,
                 flash MHA) to the reference at the real depth, batch and token counts.
        bf16x3 - 3-term split everywhere, fp32 materialised attention: 1e-3 asserted (measured <= 8e-5).
        bf16   - single bf16 operands everywhere (round 2's headline): bounded at ~2x its measured errors, NOT within
                 north_star's 1e-3 (kept as the "what the split buys" comparison).
  The measured errors are printed and written to gpurun_out/parity_*.json (tools/parity_summary.py -> profiles/parity_r03.json,
  quoted by bench.py and DESIGN.md).
"""
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cfg_cases as cc  # noqa: E402

GOLD = os.path.join(HERE, "golden")
LOGGING_ONLY = ("class_error", "cardinality_error")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def err_of(t, ref):
    """ref: full tensor or (norm, samples)."""
    return cc.sample_err(t.detach().cpu(), ref) if isinstance(ref, tuple) else rel(t, ref)


def compare_outputs(out, blob):
    errs = {}
    for st, key in ((0, "out0"), (1, "out1")):
        g = blob[key]
        for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
            errs[f"{st}.{k}"] = err_of(out[st][k], g[k])
        xp = out[st]["x_patch"]
        xt, xm = (xp.tensors, xp.mask) if hasattr(xp, "tensors") else xp
        errs[f"{st}.x_patch"] = err_of(xt, g["x_patch"][0])
        assert torch.equal(xm.cpu(), g["x_patch"][1])
        for i, (a, b) in enumerate(zip(out[st]["aux_outputs"], g["aux_outputs"])):
            errs[f"{st}.aux{i}.logits"] = err_of(a["pred_logits"], b["pred_logits"])
            errs[f"{st}.aux{i}.boxes"] = err_of(a["pred_boxes"], b["pred_boxes"])
    return errs


def compare_losses(l0, l1, blob, skip_logging):
    errs = {}
    for tag, l, ref in (("0", l0, blob["loss0"]), ("1", l1, blob["loss1"])):
        assert set(l) == set(ref), set(l) ^ set(ref)
        for k, v in ref.items():
            if skip_logging and k.startswith(LOGGING_ONLY):
                continue
            err = abs(float(l[k].detach()) - float(v))
            if k.startswith("cardinality_error"):
                # logging only, DISCRETE: |#(argmax != last class) - #targets| averaged over the images (reference models/conditional_detr.py:
                # 286-298).  One query whose two largest logits are tied to ~1e-4 may flip its argmax under a 1e-4 output error: that moves
                # this statistic by 1 / B; more than one flipped query per image is an error
                nb = len(blob["pseudo"])
                err = 0.0 if err <= 1.0 / nb + 1e-6 else err
            # north_star: "within 1e-3 REL".  Truly relative for every loss key (floor 1e-2: a key that is exactly 0 in the reference - an empty
            # image's box loss - is bounded at 1e-5 absolute); the two logging-only statistics are percentages / counts (O(1) ... O(100)) built from
            # discrete argmax decisions and keep the max(1, |v|) scale
            floor = 1.0 if k.startswith(LOGGING_ONLY) else 1e-2
            errs[f"{tag}.{k}"] = err / max(floor, abs(float(v)))
    return errs


def box_key_report(l1, blob):
    """{key: (reference value, relative error)} of the stage-1 box / GIoU keys - the smallest loss values of a step (0.1 ... 0.2 at cfg2_full),
    where an absolute and a relative 1e-3 differ most; printed and recorded with every parity case."""
    out = {}
    for k, v in blob["loss1"].items():
        if k.startswith(("loss_bbox", "loss_giou")):
            out[k] = (float(v), abs(float(l1[k].detach()) - float(v)) / max(1e-2, abs(float(v))))
    return out


def compare_grads(named_grads, blob, norm_errs=None):
    """-> errors per parameter (max of the norm's and the 64 samples' relative error); analytically-zero gradients are checked
    for smallness only.  norm_errs (dict, optional): filled with the relative error of the FULL tensor's norm alone."""
    errs = {}
    gmax = max(r[0] for r in blob["grads"].values() if r is not None)
    for k, g in named_grads:
        ref = blob["grads"][k]
        if ref is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert g is not None, k
        if ref[0] < 1e-6 * gmax:            # softmax shift invariance etc.: exact gradient ~0, only rounding noise remains
            continue
        errs[k] = cc.sample_err(g.detach().cpu(), ref)
        if norm_errs is not None:
            norm_errs[k] = abs(float(g.detach().double().norm()) - ref[0]) / ref[0]
    return errs


# -------------------------------------------------------------------------------------------------- CPU: oracle
@pytest.mark.parametrize("name", ["cfg1", "cfg2_enc3_small", "cfg2_depth2", "cfg2_enc3_depth2", "script_voc"])
def test_oracle_matches_reference_at_config_dims(name):
    from oracle import spe_oracle as O
    blob = torch.load(os.path.join(GOLD, f"cfg_{name}.pt"), weights_only=False)
    args, (model, *_), tensors, mask, targets = cc.build_case(name)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    chk = float(sum(v.detach().double().abs().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(chk - blob["sd_checksum"]) <= 1e-9 * blob["sd_checksum"], "seeded weights differ from the ones the fixture was made with"
    tot, out, l0, l1 = O.total_loss(sd, cc.oracle_cfg(name), tensors, mask, targets)
    tot.backward()
    oe = compare_outputs(out, blob)
    le = compare_losses(l0, l1, blob, skip_logging=False)
    ge = compare_grads([(k, v.grad) for k, v in sd.items() if k in blob["grads"]], blob)
    print(f"[oracle {name}] worst output {max(oe.values()):.2e}, loss {max(le.values()):.2e}, grad {max(ge.values()):.2e} over {len(ge)} parameters")
    assert max(oe.values()) < 1e-5, max(oe.items(), key=lambda kv: kv[1])
    assert max(le.values()) < 1e-5, max(le.items(), key=lambda kv: kv[1])
    assert abs(float(tot.detach()) - float(blob["total"])) <= 1e-5 * abs(float(blob["total"]))
    # fp32 CPU sums depend on the host's thread count / vector width: all but a few cancellation-prone gradients (proj_l / proj_w of the
    # first blocks) agree to 2e-4 everywhere; those few have been seen at 5e-4 on another host
    gsorted = sorted(ge.values())
    assert gsorted[int(0.95 * (len(gsorted) - 1))] < 2e-4 and gsorted[-1] < 1e-3 and len(ge) > 100, max(ge.items(), key=lambda kv: kv[1])
    for p, r in zip(O.postprocess_refine(out[0], targets), blob["pseudo"]):
        assert torch.equal(p["labels"], r["labels"])


# -------------------------------------------------------------------------------------------------- GPU: product
# norm-relative tolerances: (every output, every loss key, total loss, median / 90th-percentile parameter-gradient error over
# the 64 stored samples per tensor, worst error of a full-tensor gradient NORM).
#   bf16s / bf16x3: north_star's 1e-3 on outputs and losses (measured: <= 2.2e-4 / <= 8e-5).  bf16s gradients carry the
#   backward's single-bf16 operand rounding: measured median 3.5-5.5e-3, p90 <= 1.4e-2 (asserted at ~2.5x that).
#   bf16: ~2x its measured errors (outputs 6e-3, loss keys 7e-3, total 1.3e-3, median 3.7e-2, p90 5.2e-2).
# The 64 strided samples of a tensor whose entries are mostly tiny read worse than its norm-relative error over all elements
# (a few decoder matrices whose gradient is a small difference of large terms: up to 0.17 sampled in bf16s), so the WORST
# tensor is bounded on the error of its full norm and the sampled errors carry the median / p90 assertions.
TOL = {"bf16s": (1e-3, 1e-3, 1e-3, 1.5e-2, 4e-2, 5e-2), "bf16x3": (1e-3, 1e-3, 1e-3, 1e-2, 2e-2, 1e-2),
       "bf16": (1.2e-2, 1.5e-2, 3e-3, 8e-2, 1.2e-1, 6e-1)}
GPU_CASES = [(n, p) for n in cc.ALL_CASES for p in ("bf16s", "bf16x3", "bf16") if not (n in cc.FULL_CASES and p == "bf16x3" and n != "cfg2_full")]


@pytest.mark.gpu
@pytest.mark.parametrize("name,prec", GPU_CASES)
def test_product_matches_reference_at_config_dims(dev, name, prec):
    from spe_amd import kernels as K
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, f"cfg_{name}.pt"), weights_only=False)
    args, (model, crit, crit_r, pp, rpp), tensors, mask, targets = cc.build_case(name)
    K.set_precision(prec)
    try:
        model.to(dev).train(); crit.to(dev).eval(); crit_r.to(dev).eval()      # all drop rates 0; eval criteria = no jitter
        tg = [{k: v.to(dev) for k, v in t.items()} for t in targets]
        out = model(NestedTensor(tensors.to(dev), mask.to(dev)))
        l0 = crit(out[0], tg)
        orig = torch.stack([t["orig_size"] for t in tg])
        with torch.no_grad():
            pr = rpp["bbox"](out[0], orig, tg)
        # stage-1 targets: the reference's pseudo labels (detached inputs of criterion_refine, engine.py:122-130).  Our own
        # PostProcessRefine output is compared with them below; feeding the reference's keeps an argmax-over-queries flip
        # between near-tied queries (gaps stored in the fixture) from masquerading as a criterion error.
        pseudo = [{k: v.to(dev) for k, v in p.items()} for p in blob["pseudo"]]
        l1 = crit_r(out[1], pseudo)
        wd = blob["weight_dict"]
        total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
        total.backward()
        torch.cuda.synchronize()
        oe = compare_outputs(out, blob)
        le = compare_losses(l0, l1, blob, skip_logging=(prec == "bf16"))
        ne = {}
        ge = compare_grads([(k, p.grad) for k, p in model.named_parameters()], blob, ne)
        te = abs(float(total.detach()) - float(blob["total"])) / abs(float(blob["total"]))
        wo, wl, wg = (max(d.items(), key=lambda kv: kv[1]) for d in (oe, le, ge))
        gs = sorted(ge.values())
        wn = max(ne.items(), key=lambda kv: kv[1])
        rec = {"case": name, "precision": prec, "worst_grad_norm_err": wn, "tokens": int(out[0]["x_patch"].tensors.shape[2] * out[0]["x_patch"].tensors.shape[3]),
               "worst_output": wo, "pred_logits": oe["0.pred_logits"], "pred_boxes": oe["0.pred_boxes"], "x_patch": oe["0.x_patch"],
               "worst_loss": wl, "loss_metric": "|err| / max(|ref|, 1e-2) per key", "stage1_box_keys": box_key_report(l1, blob), "total_loss_rel_err": te, "worst_grad": wg, "median_grad": gs[len(gs) // 2], "p90_grad": gs[(9 * len(gs)) // 10], "grads_compared": len(ge)}
        print(f"[{name} {prec}] " + json.dumps(rec))
        od = os.path.join(os.path.dirname(HERE), "gpurun_out")
        if os.path.isdir(od):
            with open(os.path.join(od, f"parity_{name}_{prec}.json"), "w") as fh:
                json.dump(rec, fh)
        to, tl, tt, tgm, tg90, tgn = TOL[prec]
        assert wo[1] < to, wo
        assert wl[1] < tl and te < tt, (wl, te)
        assert gs[len(gs) // 2] < tgm and len(ge) > 100, gs[len(gs) // 2]
        assert gs[(9 * len(gs)) // 10] < tg90, gs[(9 * len(gs)) // 10]
        assert wn[1] < tgn, wn
        for p, r, mg in zip(pr, blob["pseudo"], blob["pseudo_margins"]):
            assert torch.equal(p["labels"].cpu(), r["labels"])
            assert rel(p["scores"], r["scores"]) < to
            clear = torch.tensor(mg) > 20 * to                  # classes whose best query is not nearly tied with the runner-up
            if clear.any():
                assert rel(p["boxes"].cpu()[clear], r["boxes"][clear]) < to
    finally:
        K.set_precision("bf16s")


# -------------------------------------------------------------------------------------------------- GPU: training trajectory
def _to_dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, dict):
        return {k: _to_dev(v, dev) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_to_dev(v, dev) for v in x]
    return x


# per-step bounds on |loss - reference loss| / reference loss: (step 0, later steps, median update error).  Step 0 is a pure
# forward of identical weights: north_star's 1e-3.  From step 1 on the loss also carries the parameter updates the product's OWN
# backward produced, and these cases move fast (cfg2_depth2's loss falls 101 -> 52 -> 39 -> 31 -> 27).  Measured (profiles/
# r04_traj.json): bf16x3 <= 7.4e-5 at every step; bf16s (single-bf16 backward operands) <= 7.3e-4 through step 3 and 3.9e-3 at
# step 4 of cfg2_depth2, 6.1e-4 at cfg1 - asserted at 1e-2 (~2.5x).  The 5-step parameter UPDATE is an Adam quantity: elements whose
# gradient is at rounding-noise level move by +-lr whatever their sign says (the reference's own k.bias updates are such noise:
# softmax is shift invariant, the exact gradient is 0), so it is bounded on the MEDIAN over parameters: measured 4.5e-5 / 1.5e-2
# (bf16x3) and 2.2e-2 / 5.7e-2 (bf16s).
TRAJ_TOL = {"bf16x3": (1e-3, 1e-3, 4e-2), "bf16s": (1e-3, 1e-2, 1.5e-1)}


def _replay_trajectory(dev, name, prec, blob, tag="traj"):
    """Replay the recorded optimizer steps of a trajectory fixture on the product (GradAllReducer + FlatAdamW on the flat buckets, the
    jittered targets_cp of both criteria injected from the fixture) -> record with the per-step loss / gradient-norm errors and the
    parameter-update errors; asserts every weighted loss key at step 0 (identical weights) at north_star's 1e-3."""
    from spe_amd import kernels as K
    from spe_amd.dp import GradAllReducer
    from spe_amd.optim import FlatAdamW
    from spe_amd.util.misc import NestedTensor
    hy = blob["hyper"]
    args, (model, crit, crit_r, pp, rpp), tensors, mask, targets = cc.build_case(name)
    chk = float(sum(v.detach().double().abs().sum() for v in model.state_dict().values() if v.is_floating_point()))
    assert abs(chk - blob["sd_checksum"]) <= 1e-9 * blob["sd_checksum"]
    K.set_precision(prec)
    red = None
    try:
        model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
        p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
        named = list(model.named_parameters())
        params = [p for _, p in named if p.requires_grad]
        red = GradAllReducer(params, flatten_params=True)
        groups = [{"params": [p for n, p in named if "backbone" not in n and p.requires_grad]},
                  {"params": [p for n, p in named if "backbone" in n and p.requires_grad and "blocks_token_only" not in n], "lr": hy["lr_backbone"]},
                  {"params": [p for n, p in named if "backbone" in n and p.requires_grad and "blocks_token_only" in n], "lr": hy["lr_cls_head"]}]
        opt = FlatAdamW(groups, red, lr=hy["lr"], weight_decay=hy["weight_decay"], max_grad_norm=hy["clip_max_norm"])
        wd = blob["weight_dict"]
        tg = _to_dev(targets, dev)
        samples = NestedTensor(tensors.to(dev), mask.to(dev))
        errs, gerrs, losses = [], [], []
        for s, st in enumerate(blob["steps"]):
            opt.zero_grad()
            out = model(samples)
            l0 = crit(out[0], tg, targets_cp=_to_dev(st["targets_cp0"], dev))
            l1 = crit_r(out[1], _to_dev(st["pseudo"], dev), targets_cp=_to_dev(st["targets_cp1"], dev))
            total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
            total.backward()
            red.finish()
            gn = float(torch.sqrt(sum((b["flat"].double() ** 2).sum() for b in red.buckets)))
            opt.step()
            losses.append(float(total.detach()))
            errs.append(abs(losses[-1] - st["total"]) / abs(st["total"]))
            gerrs.append(abs(gn - st["grad_norm"]) / st["grad_norm"])
            if s == 0:          # identical weights: every loss key of both criteria within north_star's bound
                for ld, ref in ((l0, st["loss0"]), (l1, st["loss1"])):
                    for k, v in ref.items():
                        if k in wd:
                            assert abs(float(ld[k]) - v) <= 1e-3 * max(abs(v), 1e-2), (k, float(ld[k]), v)
        ue = {}
        for n, p in model.named_parameters():
            ref = blob["updates"][n]
            if ref[0] > 0:
                ue[n] = cc.sample_err((p.detach() - p0[n]).cpu(), ref)
        us = sorted(ue.values())
        rec = {"case": name, "precision": prec, "loss_rel_err_per_step": errs, "grad_norm_rel_err_per_step": gerrs,
               "update_err_median": us[len(us) // 2], "update_err_p90": us[(9 * len(us)) // 10], "update_err_worst": max(ue.items(), key=lambda kv: kv[1]),
               "reference_losses": [st["total"] for st in blob["steps"]], "product_losses": losses, "updates_compared": len(us)}
        print(f"[{tag} {name} {prec}] " + json.dumps(rec))
        od = os.path.join(os.path.dirname(HERE), "gpurun_out")
        if os.path.isdir(od):
            with open(os.path.join(od, f"{tag}_{name}_{prec}.json"), "w") as fh:
                json.dump(rec, fh)
        return rec
    finally:
        if red is not None:
            red.remove()
        K.set_precision("bf16s")


@pytest.mark.gpu
@pytest.mark.parametrize("name,prec", [(n, p) for n in ("cfg1", "cfg2_depth2") for p in ("bf16s", "bf16x3")])
def test_training_trajectory_matches_reference(dev, name, prec):
    """FIVE optimizer steps of the reference's loop (engine.py:88-165: train-mode criteria with the 5x one-to-many jitter,
    clip_grad_norm_ 0.1, AdamW with main.py:177-187's three LR groups) recorded in the build container by
    tools/gen_traj_golden.py, replayed by the product: GradAllReducer + FlatAdamW on the flat buckets, the jittered targets_cp of
    both criteria injected from the fixture.  Asserted: every step's total loss and pre-clip gradient norm, and the norm-relative
    error of every parameter's 5-step UPDATE (median over parameters)."""
    blob = torch.load(os.path.join(GOLD, f"traj_{name}.pt"), weights_only=False)
    rec = _replay_trajectory(dev, name, prec, blob)
    errs = rec["loss_rel_err_per_step"]
    t0, t1, tu = TRAJ_TOL[prec]
    assert errs[0] < t0, errs
    assert max(errs[1:]) < t1, errs
    assert rec["update_err_median"] < tu and rec["updates_compared"] > 100, (rec["update_err_median"], rec["updates_compared"])


@pytest.mark.gpu
def test_long_training_trajectory_stays_near_reference(dev):
    """THIRTY optimizer steps of the reference's loop at cfg2_depth2 (N = 4150 tokens; tests/golden/traj30_cfg2_depth2.pt, tools/gen_traj_golden.py
    cfg2_depth2:30), replayed by the product with the recorded one-to-many targets, in the 3-term parity mode (bf16x3: ~1e-5 per step) and in
    the benchmark mode (bf16s: single-bf16 backward feeding AdamW).  What the fixture shows (round 5, profiles/r05_traj30.json): this
    trajectory - one repeated batch, gradient norm 9400 clipped to 0.1, loss 101 -> 16 - is CHAOTIC from its sixth step on: the parity
    mode, which follows the reference to 1e-4 through step 4, departs from it by 4-5 % at steps 6-11 as well (any other summation order
    does), so a per-step bound below that is not a property of the arithmetic.  Asserted: step 0 at north_star's 1e-3; the steps before the
    departure within 1e-3 (bf16x3) / 1e-2 (bf16s); every later step within 15 % in both modes; the benchmark mode drifts no further than
    three times the parity mode's worst step; both curves end near the reference's (final loss within 15 %, a sixth of the first)."""
    path = os.path.join(GOLD, "traj30_cfg2_depth2.pt")
    blob = torch.load(path, weights_only=False)
    assert len(blob["steps"]) == 30
    recs = {prec: _replay_trajectory(dev, "cfg2_depth2", prec, blob, tag="traj30") for prec in ("bf16x3", "bf16s")}
    worst = {}
    for prec, rec in recs.items():
        errs = rec["loss_rel_err_per_step"]
        assert errs[0] < 1e-3, (prec, errs)
        assert max(errs[:5]) < (1e-2 if prec == "bf16s" else 1e-3), (prec, errs[:5])
        assert max(errs) < 0.15, (prec, max(errs), errs.index(max(errs)), errs)
        assert rec["product_losses"][-1] < rec["product_losses"][0] / 5.0
        worst[prec] = max(errs)
    assert worst["bf16s"] <= 3.0 * max(worst["bf16x3"], 2e-2), worst


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["bf16s", "bf16x3"])
def test_train_mode_criteria_at_full_depth(dev, prec):
    """cfg2_full (24 blocks, batch 2, N = 4150, padded second image) with BOTH criteria in TRAIN mode - the 5x one-to-many jitter
    and Hungarian matching of reference models/conditional_detr.py:399-466 on 35 / 25 targets per image - against the reference's
    record (tools/gen_traj_golden.py cfg2_full:train: forward without autograd, the jittered targets_cp of both stages captured at
    its matcher calls and injected here): every loss key of both criteria and the weighted total within north_star's 1e-3."""
    from spe_amd import kernels as K
    from spe_amd.util.misc import NestedTensor
    name = "cfg2_full"
    blob = torch.load(os.path.join(GOLD, f"cfg_{name}_train.pt"), weights_only=False)
    args, (model, crit, crit_r, pp, rpp), tensors, mask, targets = cc.build_case(name)
    chk = float(sum(v.detach().double().abs().sum() for v in model.state_dict().values() if v.is_floating_point()))
    assert abs(chk - blob["sd_checksum"]) <= 1e-9 * blob["sd_checksum"]
    K.set_precision(prec)
    try:
        model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
        with torch.no_grad():
            out = model(NestedTensor(tensors.to(dev), mask.to(dev)))
            l0 = crit(out[0], _to_dev(targets, dev), targets_cp=_to_dev(blob["targets_cp0"], dev))
            l1 = crit_r(out[1], _to_dev(blob["pseudo"], dev), targets_cp=_to_dev(blob["targets_cp1"], dev))
        wd = blob["weight_dict"]
        total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
        errs = {}
        for tag, ld, ref in (("0", l0, blob["loss0"]), ("1", l1, blob["loss1"])):
            for k, v in ref.items():
                if k in wd:
                    errs[f"{tag}.{k}"] = abs(float(ld[k]) - float(v)) / max(abs(float(v)), 1e-2)
        te = abs(float(total) - float(blob["total"])) / abs(float(blob["total"]))
        worst = max(errs.items(), key=lambda kv: kv[1])
        print(f"[train-mode criteria {name} {prec}] total {te:.2e}, worst key {worst}, targets per image {[len(t['labels']) for t in blob['targets_cp0']]}")
        assert worst[1] < 1e-3 and te < 1e-3, (worst, te)
    finally:
        K.set_precision("bf16s")
