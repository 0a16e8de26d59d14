"""Pins the oracle (oracle/spe_oracle.py) against golden vectors captured from the REFERENCE itself
(tools/gen_golden.py, reference imported in the build container).  CPU only, fp32, rel <= 1e-5."""
import os

import pytest
import torch

from oracle import spe_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def close(a, b, tol=1e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert rel(a, b) < tol, rel(a, b)


@pytest.fixture(scope="module")
def ops():
    return torch.load(os.path.join(GOLD, "ops.pt"), weights_only=False)


def pref(sd, p):
    return {p + k: v for k, v in sd.items()}


@pytest.mark.parametrize("N", [12, 35])
def test_talking_heads_block(ops, N):
    g = ops[f"thattn_N{N}"]
    sd = pref(g["sd"], "b.")
    x = g["x"]
    close(O.talking_heads_attention(O.ln(x, sd, "b.norm1", 1e-6), sd, "b.attn", 4), g["attn_out"])
    close(O.layerscale_block(x, sd, "b", 4, 1e-6), g["block_out"])


def test_class_attention_block(ops):
    g = ops["ca_block"]
    sd = pref(g["sd"], "b.")
    cls, amap = O.class_attention_block(g["x"], g["cls"], sd, "b", 4, 1e-6)
    close(cls, g["out"])
    close(amap, g["map"])


@pytest.mark.parametrize("hw", [(14, 14), (50, 83), (5, 7)])
def test_pos_embed_bicubic(ops, hw):
    g = ops[f"posembed_{hw[0]}x{hw[1]}"]
    close(O.interpolate_pos_embed(g["pe"], (24, 24), hw), g["out"])


def test_position_embedding_sine(ops):
    g = ops["pos_sine"]
    close(O.position_embedding_sine(g["mask"], 16), g["out"])


@pytest.mark.parametrize("d", [32, 192])
def test_gen_sineembed(ops, d):
    g = ops[f"sineembed_d{d}"]
    close(O.gen_sineembed_for_position(g["pos"], d), g["out"])


def test_mha(ops):
    g = ops["mha"]
    out = O.mha_core(g["q"], g["k"], g["v"], 4, g["kpm"], g["sd"]["out_proj.weight"], g["sd"]["out_proj.bias"])
    close(out, g["out"])


def test_encoder_decoder_layers(ops):
    g = ops["enc_layer"]
    close(O.encoder_layer(g["src"], g["kpm"], g["pos"], pref(g["sd"], "e."), "e", 4), g["out"])
    g = ops["dec_layer"]
    sd = pref(g["sd"], "d.")
    for first, key in ((True, "out_first"), (False, "out_other")):
        out = O.decoder_layer(g["tgt"], g["memory"], g["kpm"], g["pos"], g["query_pos"], g["query_sine"], sd, "d", 4, first)
        close(out, g[key])


def test_giou(ops):
    g = ops["giou"]
    close(O.generalized_box_iou(g["a"], g["b"]), g["out"])
    close(O.box_iou(g["a"], g["b"])[0], g["iou"])


@pytest.mark.parametrize("tag", ["small", "many"])
def test_matcher(ops, tag):
    g = ops[f"matcher_{tag}"]
    idx = O.hungarian(g["outputs"], g["targets"])
    for (i, j), (ri, rj) in zip(idx, g["indices"]):
        assert torch.equal(i, ri) and torch.equal(j, rj)
    # the cost matrices the reference handed to SciPy (captured at its linear_sum_assignment call), incl. M > Q and M = 0
    for b, (t, c) in enumerate(zip(g["targets"], g["cost"])):
        mine = O.matcher_cost(g["outputs"]["pred_logits"][b], g["outputs"]["pred_boxes"][b], t["labels"], t["boxes"])
        assert mine.shape == c.shape
        if c.numel():
            close(mine, c)


@pytest.mark.parametrize("gam", [0.5, 2.0])
def test_focal(ops, gam):
    g = ops[f"focal_g{gam}"]
    close(O.weighted_sigmoid_focal_loss(g["x"], g["t"], 3.0, g["w"], 0.25, gam), g["out"])


# ------------------------------------------------------------------------------------------------
def cfg_from(blob):
    a = blob["args"]
    two = "Two_Branch" in a["backbone"]
    return O.make_cfg(embed_dim=32, depth=4 if two else 3, num_heads=4, num_cls_tokens=20, layer_to_det=a["layer_to_det"],
                      two_branch=two, pos_grid=(50, 84), nheads=a["nheads"], enc_layers=a["enc_layers"],
                      dec_layers=a["dec_layers"], dim_feedforward=a["dim_feedforward"], num_queries=a["num_queries"],
                      num_refines=1, num_det_classes=21, aux_loss=True)


def check_outputs(out, gold, tol=1e-5):
    for k in ("pred_logits", "pred_boxes", "x_logits", "x_cls_logits", "cams_cls"):
        close(out[k], gold[k], tol)
    close(out["x_patch"][0], gold["x_patch"][0], tol)
    assert torch.equal(out["x_patch"][1], gold["x_patch"][1])
    for a, b in zip(out["aux_outputs"], gold["aux_outputs"]):
        close(a["pred_logits"], b["pred_logits"], tol)
        close(a["pred_boxes"], b["pred_boxes"], tol)


@pytest.mark.parametrize("name", ["e2e_single", "e2e_two_branch"])
def test_end_to_end_eval(name):
    blob = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    cfg, sd, ev = cfg_from(blob), blob["state_dict"], blob["eval"]
    with torch.no_grad():
        out = O.model_forward(sd, cfg, blob["tensors"], blob["mask"])
        check_outputs(out[0], ev["out0"])
        check_outputs(out[1], ev["out1"])
        l0 = O.set_criterion(out[0], blob["targets"], refine=False)
        assert set(l0) == set(ev["loss0"])
        for k, v in ev["loss0"].items():
            assert abs(float(l0[k]) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), k
        pseudo = O.postprocess_refine(out[0], blob["targets"])
        for p, r in zip(pseudo, ev["pseudo"]):
            assert torch.equal(p["labels"], r["labels"])
            close(p["scores"], r["scores"])
            close(p["boxes"], r["boxes"])
        l1 = O.set_criterion(out[1], ev["pseudo"], refine=True)
        assert set(l1) == set(ev["loss1"])
        for k, v in ev["loss1"].items():
            assert abs(float(l1[k]) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), k
        post = O.postprocess(out[0], torch.stack([t["orig_size"] for t in blob["targets"]]), 10)
        for p, r in zip(post, ev["postprocess"]):
            assert torch.equal(p["labels"], r["labels"])
            close(p["scores"], r["scores"])
            close(p["boxes"], r["boxes"])


@pytest.mark.parametrize("name", ["e2e_single", "e2e_two_branch"])
def test_end_to_end_train_step_grads(name):
    """Train-mode criteria with the reference's captured jittered targets -> total loss -> grads of
    every parameter."""
    blob = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    cfg, tr = cfg_from(blob), blob["train"]
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in blob["state_dict"].items()}
    out = O.model_forward(sd, cfg, blob["tensors"], blob["mask"])
    ind0, ind1 = [], []
    l0 = O.set_criterion(out[0], blob["targets"], refine=False, targets_cp=tr["targets_cp0"], indices_out=ind0)
    l1 = O.set_criterion(out[1], tr["pseudo"], refine=True, targets_cp=tr["targets_cp1"], indices_out=ind1)
    for got, ref in ((ind0, tr["indices0"]), (ind1, tr["indices1"])):
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            for (i, j), (ri, rj) in zip(a, b):
                assert torch.equal(i, ri) and torch.equal(j, rj)
    for l, ref in ((l0, tr["loss0"]), (l1, tr["loss1"])):
        for k, v in ref.items():
            assert abs(float(l[k].detach()) - float(v)) <= 1e-5 * max(1.0, abs(float(v))), k
    wd = tr["weight_dict"]
    total = sum(l0[k] * wd[k] for k in l0 if k in wd) + sum(l1[k] * wd[k] for k in l1 if k in wd)
    assert abs(float(total.detach()) - float(tr["total"])) <= 1e-5 * abs(float(tr["total"]))
    total.backward()
    n_checked = 0
    for k, gref in tr["grads"].items():
        if gref is None:
            assert sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0, k
            continue
        assert sd[k].grad is not None, k
        if float(gref.abs().max()) < 1e-7:   # analytically zero (proj_l.bias: softmax shift invariance)
            assert float(sd[k].grad.abs().max()) < 1e-6, k
            continue
        assert rel(sd[k].grad, gref) < 2e-4, (k, rel(sd[k].grad, gref))
        n_checked += 1
    assert n_checked > 100


def test_jitter_targets_shape_and_order():
    """K17: [jittered copies..., original] per GT, labels/scores repeated (conditional_detr.py:409-431)."""
    t = [{"boxes": torch.tensor([[0.5, 0.5, 0.2, 0.3], [0.3, 0.6, 0.1, 0.1]]), "labels": torch.tensor([3, 7]),
          "scores": torch.tensor([0.9, 0.4])}, {"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, dtype=torch.int64)}]
    g = torch.Generator().manual_seed(0)
    o = O.jitter_targets(t, 5, 0.1, generator=g)
    assert o[0]["boxes"].shape == (10, 4) and o[0]["labels"].tolist() == [3] * 5 + [7] * 5
    assert torch.equal(o[0]["boxes"][4], t[0]["boxes"][0]) and torch.equal(o[0]["boxes"][9], t[0]["boxes"][1])
    iou, _ = O.box_iou(O.box_cxcywh_to_xyxy(o[0]["boxes"][:4]), O.box_cxcywh_to_xyxy(t[0]["boxes"][:1]))
    assert (iou > 0.7).all() and not torch.equal(o[0]["boxes"][0], t[0]["boxes"][0])
    assert o[0]["scores"].tolist() == pytest.approx([0.9] * 5 + [0.4] * 5)
    assert o[1]["labels"].numel() == 0


def test_oracle_flip_merge_vs_reference_golden():
    """oracle.decouple_output against the reference's own engine_loc.decouple_output (engine_loc.py:99-124) run on seeded tensors by
    tools/gen_infer_golden.py: every key, recursively through aux_outputs, bit for bit (the merge is copies, 1 - x and maximum)."""
    import copy
    from oracle import spe_oracle as O
    blob = torch.load(os.path.join(GOLD, "infer.pt"), weights_only=False)
    for case in blob["decouple_output"]:
        got = O.decouple_output(copy.deepcopy(case["input"]), case["bs"])
        ref = case["output"]
        assert set(got) == set(ref)
        for k, v in ref.items():
            if k == "aux_outputs":
                for a, b in zip(got[k], v):
                    assert set(a) == set(b) and all(torch.equal(a[kk], b[kk]) for kk in b)
            else:
                assert torch.equal(got[k], v), k
