"""End-to-end parity (GPU): the product (spe_amd.models.build_model, HIP kernels) against the golden
vectors captured from the REFERENCE (tests/golden/e2e_*.pt) and against the oracle on the same inputs.

Tolerances (norm-relative):
  bf16x3 mode (3-term bf16 split, ~fp32): 1e-3 is north_star's bound; we assert 2e-4.
  bf16 mode (single bf16 MFMA pass, the benchmark mode): bounded by bf16 operand rounding through the
  block stack - asserted at 3e-2 on outputs, 2e-2 on every loss; measured values are printed.
"""
import argparse
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def to_dev(targets, dev):
    return [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]


def build(blob, dev):
    from spe_amd.models import build_model
    args = argparse.Namespace(**blob["args"])
    args.device = "cuda"
    model, crit, crit_r, pp, rpp = build_model(args)
    model.load_state_dict(blob["state_dict"], strict=True)
    return model.to(dev), crit.to(dev), crit_r.to(dev), pp, rpp


# bf16s = the benchmark mode (forward on split / fp16 operands): north_star's 1e-3
OUT_TOL = {"bf16x3": 2e-4, "bf16s": 1e-3, "bf16": 3e-2}
LOSS_TOL = {"bf16x3": 2e-4, "bf16s": 1e-3, "bf16": 2e-2}
# bf16 mode: scores/probabilities and their gradients are materialised in bf16 and every contraction rounds its
# operands to bf16; on the tiny fixture models the worst parameter (a small decoder gradient) sits at 0.11-0.14
GRAD_TOL = {"bf16x3": 2e-3, "bf16s": 1e-1, "bf16": 2e-1}


@pytest.mark.parametrize("prec", ["bf16x3", "bf16s", "bf16"])
@pytest.mark.parametrize("name", ["e2e_single", "e2e_two_branch"])
def test_forward_and_losses_match_reference(dev, name, prec):
    from spe_amd import kernels as K
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    K.set_precision(prec)
    try:
        model, crit, crit_r, pp, rpp = build(blob, dev)
        model.eval(); crit.eval(); crit_r.eval()
        ev = blob["eval"]
        with torch.no_grad():
            out = model(NestedTensor(blob["tensors"].to(dev), blob["mask"].to(dev)))
            worst = 0.0
            for st, gold in ((0, ev["out0"]), (1, ev["out1"])):
                for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
                    r = rel(out[st][k], gold[k]); worst = max(worst, r)
                    assert r < OUT_TOL[prec], (st, k, r)
                r = rel(out[st]["x_patch"].tensors, gold["x_patch"][0]); worst = max(worst, r)
                assert r < OUT_TOL[prec], ("x_patch", r)
                assert torch.equal(out[st]["x_patch"].mask.cpu(), gold["x_patch"][1])
                for a, b in zip(out[st]["aux_outputs"], gold["aux_outputs"]):
                    assert rel(a["pred_logits"], b["pred_logits"]) < OUT_TOL[prec]
                    assert rel(a["pred_boxes"], b["pred_boxes"]) < OUT_TOL[prec]
            print(f"[{name} {prec}] worst output rel err {worst:.3e}")
            tg = to_dev(blob["targets"], dev)
            l0 = crit(out[0], tg)
            assert set(l0) == set(ev["loss0"]), set(l0) ^ set(ev["loss0"])
            ps = to_dev(ev["pseudo"], dev)
            l1 = crit_r(out[1], ps)
            assert set(l1) == set(ev["loss1"])
            logging_only = ("class_error", "cardinality_error")
            for l, ref in ((l0, ev["loss0"]), (l1, ev["loss1"])):
                for k, v in ref.items():
                    tol = LOSS_TOL[prec] * max(1.0 if k.startswith(logging_only) else 1e-2, abs(float(v)))     # loss keys: truly relative
                    if k.startswith(logging_only) and prec == "bf16":
                        continue      # argmax-derived counters may flip under bf16 rounding
                    assert abs(float(l[k]) - float(v)) <= tol, (k, float(l[k]), float(v))
            # stage-0 -> stage-1 pseudo labels and top-k post-processing
            orig = torch.stack([t["orig_size"] for t in tg])
            pr = rpp["bbox"](out[0], orig, tg)
            for p, r in zip(pr, ev["pseudo"]):
                assert torch.equal(p["labels"].cpu(), r["labels"])
                assert rel(p["scores"], r["scores"]) < OUT_TOL[prec] and rel(p["boxes"], r["boxes"]) < OUT_TOL[prec]
            if prec in ("bf16x3", "bf16s"):
                # PostProcess top-k (reference models/conditional_detr.py:592-623) in the exact AND in the benchmark mode: the
                # sorted scores must agree everywhere; labels / boxes rank by rank (exact mode) or detection by detection (below)
                post = pp["bbox"](out[0], orig, 10)
                for p, r in zip(post, ev["postprocess"]):
                    assert rel(p["scores"], r["scores"]) < OUT_TOL[prec]
                    if prec == "bf16x3":
                        assert torch.equal(p["labels"].cpu(), r["labels"])
                        assert rel(p["boxes"], r["boxes"]) < OUT_TOL[prec]
                        continue
                    # benchmark mode: the fixture's top scores are nearly tied (gaps down to 1e-4), so neighbours may swap - match
                    # every reference detection whose score clears the k-th one to the product's detection of the same class with
                    # the nearest box, whatever its rank
                    sc, kth = r["scores"].double(), float(r["scores"][-1])
                    pb, pl, ps = p["boxes"].double().cpu(), p["labels"].cpu(), p["scores"].double().cpu()
                    matched = 0
                    for i in range(len(sc)):
                        if float(sc[i]) - kth < 4 * OUT_TOL[prec]:
                            continue            # could legitimately drop out of the top k
                        same = (pl == r["labels"][i]).nonzero().flatten()
                        assert len(same) > 0, ("class missing from the top k", i, int(r["labels"][i]))
                        dist = (pb[same] - r["boxes"][i].double()).norm(dim=1)
                        j = same[int(dist.argmin())]
                        assert float(dist.min()) <= OUT_TOL[prec] * float(r["boxes"][i].double().norm()), (i, float(dist.min()))
                        assert abs(float(ps[j]) - float(sc[i])) <= OUT_TOL[prec] * float(sc[i])
                        matched += 1
                    assert matched >= 3, "fixture leaves nothing to check"
    finally:
        K.set_precision("bf16")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16s", "bf16"])
@pytest.mark.parametrize("name", ["e2e_single", "e2e_two_branch"])
def test_train_step_grads_match_reference(dev, name, prec):
    """Train-mode criteria on the reference's captured one-to-many targets -> total loss -> backward
    through every HIP backward kernel -> all parameter gradients vs the reference's."""
    from spe_amd import kernels as K
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    K.set_precision(prec)
    try:
        model, crit, crit_r, pp, rpp = build(blob, dev)
        model.train(); crit.train(); crit_r.train()
        tr = blob["train"]
        out = model(NestedTensor(blob["tensors"].to(dev), blob["mask"].to(dev)))
        l0 = crit(out[0], to_dev(blob["targets"], dev), targets_cp=to_dev(tr["targets_cp0"], dev))
        l1 = crit_r(out[1], to_dev(tr["pseudo"], dev), targets_cp=to_dev(tr["targets_cp1"], dev))
        wd = tr["weight_dict"]
        for l, ref in ((l0, tr["loss0"]), (l1, tr["loss1"])):
            for k, v in ref.items():
                if k in wd:
                    assert abs(float(l[k].detach()) - float(v)) <= LOSS_TOL[prec] * max(1e-2, abs(float(v))), (k, float(l[k].detach()), float(v))
        total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
        assert abs(float(total.detach()) - float(tr["total"])) <= LOSS_TOL[prec] * abs(float(tr["total"]))
        total.backward()
        worst, n = ("", 0.0), 0
        for k, p in model.named_parameters():
            gref = tr["grads"][k]
            if gref is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            assert p.grad is not None, k
            if float(gref.abs().max()) < 1e-7:
                # analytically zero (softmax shift invariance): only rounding noise may remain
                assert float(p.grad.abs().max()) < (1e-4 if prec == "bf16x3" else 5e-3), k
                continue
            r = rel(p.grad, gref)
            if r > worst[1]:
                worst = (k, r)
            assert r < GRAD_TOL[prec], (k, r)
            n += 1
        print(f"[{name} {prec}] total {float(total.detach()):.6f} vs ref {float(tr['total']):.6f}; worst grad rel err {worst}")
        assert n > 100
    finally:
        K.set_precision("bf16")


def test_product_matches_oracle_on_fresh_inputs(dev):
    """Same seeded weights/inputs through the oracle (CPU fp32) and the HIP path, padded batch, dropout 0."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import spe_oracle as O
    from spe_amd import kernels as K
    from spe_amd.models import build_model
    from spe_amd.util.misc import nested_tensor_from_tensor_list
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    args = argparse.Namespace(**blob["args"])
    args.device = "cuda"; args.enc_layers = 2; args.dec_layers = 3; args.num_queries = 9
    torch.manual_seed(5)
    model, crit, crit_r, pp, rpp = build_model(args)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "gamma_" in n:
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif "proj_l.weight" in n or "proj_w.weight" in n:
                p.copy_(torch.eye(p.shape[0]) + 0.3 * torch.randn(p.shape, generator=g))
            elif n.startswith("bbox_embed") or n.startswith("class_embed"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    imgs = [torch.randn(3, 80, 112, generator=g), torch.randn(3, 64, 96, generator=g), torch.randn(3, 80, 48, generator=g)]
    nt = nested_tensor_from_tensor_list(imgs)
    cfg = O.make_cfg(embed_dim=32, depth=3, num_heads=4, num_cls_tokens=20, layer_to_det=args.layer_to_det, two_branch=False,
                     pos_grid=(50, 84), nheads=4, enc_layers=2, dec_layers=3, dim_feedforward=64, num_queries=9,
                     num_refines=1, num_det_classes=21, aux_loss=True)
    with torch.no_grad():
        ref = O.model_forward(sd, cfg, nt.tensors, nt.mask)
    K.set_precision("bf16x3")
    try:
        model = model.to(dev).eval()
        with torch.no_grad():
            out = model(nt.to(dev))
        for st in (0, 1):
            for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
                assert rel(out[st][k], ref[st][k]) < 2e-4, (st, k, rel(out[st][k], ref[st][k]))
    finally:
        K.set_precision("bf16")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16s", "bf16"])
def test_training_trajectory_flat_stack_equals_torch_stack(dev, prec):
    """Five optimisation steps of the tiny fixture model, twice: (a) plain autograd gradients + clip_grad_norm_ +
    torch.optim.AdamW; (b) GradAllReducer (gradients written into the buckets, flat parameters) + FlatAdamW.  Same
    kernels, same inputs -> same loss trajectory and parameters.  Run with the bf16-copy Linear path forced on in bf16
    mode, so a stale cached weight copy, a misplaced gradient or a wrong parameter group shows up as divergence."""
    from spe_amd import kernels as K
    from spe_amd.dp import GradAllReducer
    from spe_amd.optim import FlatAdamW
    from spe_amd.util.misc import NestedTensor
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    tr = blob["train"]
    old_rows = K.LINEAR16_MIN_ROWS
    K.set_precision(prec)
    K.LINEAR16_MIN_ROWS = 16
    try:
        def run(flat):
            model, crit, crit_r, pp, rpp = build(blob, dev)
            model.train(); crit.train(); crit_r.train()
            named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
            groups = [{"params": [p for n, p in named if "backbone" not in n], "lr": 2e-3},
                      {"params": [p for n, p in named if "backbone" in n], "lr": 5e-4}]
            params = [p for _, p in named]
            if flat:
                red = GradAllReducer(params, bucket_bytes=1 << 16, flatten_params=True)
                opt = FlatAdamW(groups, red, weight_decay=1e-2, max_grad_norm=0.1)
            else:
                opt = torch.optim.AdamW(groups, weight_decay=1e-2)
            samples = NestedTensor(blob["tensors"].to(dev), blob["mask"].to(dev))
            losses = []
            for it in range(5):
                if flat:
                    red.reset()
                else:
                    opt.zero_grad(set_to_none=True)
                out = model(samples)
                l0 = crit(out[0], to_dev(blob["targets"], dev), targets_cp=to_dev(tr["targets_cp0"], dev))
                l1 = crit_r(out[1], to_dev(tr["pseudo"], dev), targets_cp=to_dev(tr["targets_cp1"], dev))
                wd = tr["weight_dict"]
                total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
                total.backward()
                if flat:
                    red.finish()
                else:
                    torch.nn.utils.clip_grad_norm_(params, 0.1)
                opt.step()
                losses.append(float(total.detach()))
            if flat:
                red.remove()
            return losses, {n: p.detach().clone() for n, p in named}
        la, pa = run(False)
        lb, pb = run(True)
        print(f"[{prec}] torch stack {la}\n[{prec}] flat stack  {lb}")
        assert la[-1] < la[0]                                       # it trains
        tol = 2e-4 if prec == "bf16x3" else 2e-3
        for x, y in zip(la, lb):
            assert abs(x - y) <= tol * abs(x), (la, lb)
        # Adam turns rounding noise into O(lr) steps where the true gradient is zero (key-projection biases: softmax
        # shift invariance), so the worst parameter is only loosely bounded; the bulk must agree tightly
        errs = sorted(((rel(pb[n], pa[n]), n) for n in pa), reverse=True)
        print(errs[:4])
        if prec == "bf16x3":
            assert errs[0][0] < 2e-2, errs[:3]
        assert errs[len(errs) // 10][0] < (2e-5 if prec == "bf16x3" else 5e-3), errs[len(errs) // 10]   # bf16: rounding flips of the bf16 weight copies
    finally:
        K.LINEAR16_MIN_ROWS = old_rows
        K.set_precision("bf16")
