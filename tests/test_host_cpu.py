"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of include/spe_hip.h,
the product's parameter names/shapes equal the reference's, host logic of the boundary, and the
product FAILS LOUDLY instead of computing on the CPU."""
import argparse
import ctypes
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_header_symbol():
    from spe_amd import lib
    protos = lib.parse_header()
    assert len(protos) >= 19
    l = lib.load()
    for name, sig in protos.items():
        fn = getattr(l, name)            # AttributeError == missing export
        assert fn.restype is ctypes.c_int and len(fn.argtypes) == len(sig)
    assert l.spe_abi_version() == 7
    # ONE attention backward composition (ABI 7): the retired entries are neither declared nor exported
    for gone in ("spe_talking_fused", "spe_talking_fused_bits", "spe_talking_fused_plan", "spe_attn_merge", "spe_talking_flash_rows",
                 "spe_talking_flash_dv", "spe_talking_bwdq_pass1"):
        assert gone not in protos and not hasattr(l, gone), gone


def test_comm_library_exports_every_header_symbol():
    """libspe_comm.so (RCCL collectives, include/spe_comm.h): loads without a GPU, every declared entry point resolves, and
    calls before spe_comm_init report "not initialised" instead of crashing."""
    from spe_amd import comm
    so = comm.load()
    assert set(comm.PROTOS) == {"spe_comm_unique_id", "spe_comm_init", "spe_comm_world", "spe_comm_allreduce", "spe_comm_broadcast",
                                "spe_comm_destroy"}
    for name, sig in comm.PROTOS.items():
        assert len(getattr(so, name).argtypes) == len(sig)
    assert so.spe_comm_world(None, None) == -1
    assert so.spe_comm_allreduce(None, 0, 0, None) == -1
    assert so.spe_comm_destroy() == 0
    text = open(comm.HEADER).read()
    for ref in ("util/misc.py:414-436", "main.py:172", "models/conditional_detr.py:438-440"):
        assert ref in text, ref


def test_header_cites_reference_sites():
    text = open(os.path.join(ROOT, "include", "spe_hip.h")).read()
    for ref in ("models/cait.py", "models/matcher.py", "models/conditional_detr.py", "models/attention.py",
                "models/transformer.py", "util/box_ops.py"):
        assert ref in text, ref


def _build(blob):
    from spe_amd.models import build_model
    args = argparse.Namespace(**blob["args"])
    args.device = "cpu"
    return build_model(args), args


@pytest.mark.parametrize("name", ["e2e_single", "e2e_two_branch"])
def test_state_dict_matches_reference(name):
    """Checkpoint compatibility (--resume does a strict load, reference main.py:229)."""
    blob = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    (model, crit, crit_r, pp, rpp), args = _build(blob)
    ref = blob["state_dict"]
    own = model.state_dict()
    assert list(own.keys()) != [] and set(own) == set(ref)
    for k in ref:
        assert own[k].shape == ref[k].shape, k
    model.load_state_dict(ref, strict=True)
    assert args.hidden_dim == 32                                   # Backbone overwrites --hidden_dim (cait_backbone.py:85)
    assert model.backbone[0].body.patch_size == 16                 # read by ConditionalDETR_Refine.forward
    names = [n for n, _ in model.named_parameters()]
    assert any("backbone" in n for n in names) and any("blocks_token_only" in n for n in names)   # LR groups, main.py:177-188
    assert set(crit.weight_dict) == set(blob["train"]["weight_dict"])
    assert crit.losses == ["labels", "boxes", "cardinality", "image_label"] and crit_r.losses == ["labels", "boxes", "cardinality"]
    assert set(pp) == {"bbox"} and set(rpp) == {"bbox"}
    crit.update_hung_match_ratio(3)
    assert crit.matcher.match_ratio == 3 and crit.hung_match_ratio == 3


def test_no_cpu_fallback():
    """The product path must not compute on the CPU (and must not reach for the oracle)."""
    from spe_amd import lib, ops
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    (model, *_), _ = _build(blob)
    from spe_amd.util.misc import NestedTensor
    with pytest.raises(lib.SpeLibraryError):
        model(NestedTensor(blob["tensors"], blob["mask"]))
    with pytest.raises(lib.SpeLibraryError):
        ops.linear(torch.randn(4, 8), torch.randn(3, 8), None)
    import spe_amd
    src = ""
    for dp, _, fs in os.walk(os.path.dirname(spe_amd.__file__)):
        for f in fs:
            if f.endswith(".py"):
                src += open(os.path.join(dp, f)).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_nested_tensor_and_mask_downsample():
    from spe_amd.util.misc import nested_tensor_from_tensor_list
    a, b = torch.ones(3, 32, 48), torch.ones(3, 16, 64)
    nt = nested_tensor_from_tensor_list([a, b])
    assert nt.tensors.shape == (2, 3, 32, 64) and nt.mask.shape == (2, 32, 64)
    assert not nt.mask[0, :, :48].any() and nt.mask[0, :, 48:].all() and nt.mask[1, 16:].all() and not nt.mask[1, :16].any()
    assert float(nt.tensors[0, :, :, 48:].abs().sum()) == 0


def test_postprocess_refine_matches_reference_golden():
    """PostProcessRefine / PostProcess are host-side tensor plumbing: check them on the golden outputs."""
    from spe_amd.models.conditional_detr import PostProcess, PostProcessRefine
    blob = torch.load(os.path.join(GOLD, "e2e_single.pt"), weights_only=False)
    ev = blob["eval"]
    orig = torch.stack([t["orig_size"] for t in blob["targets"]])
    res = PostProcessRefine()(ev["out0"], orig, blob["targets"])
    for p, r in zip(res, ev["pseudo"]):
        assert torch.equal(p["labels"], r["labels"])
        assert torch.allclose(p["scores"], r["scores"]) and torch.allclose(p["boxes"], r["boxes"])
    post = PostProcess()(ev["out0"], orig, 10)
    for p, r in zip(post, ev["postprocess"]):
        assert torch.equal(p["labels"], r["labels"]) and torch.allclose(p["scores"], r["scores"]) and torch.allclose(p["boxes"], r["boxes"])


def test_jitter_targets_semantics():
    """One-to-many targets (conditional_detr.py:409-431): ratio rows per GT, jittered copies first (IoU>0.7),
    original last, labels/scores repeated - same contract as the oracle's restatement."""
    from oracle import spe_oracle as O
    from spe_amd.models.conditional_detr import jitter_targets
    torch.manual_seed(0)
    t = [{"boxes": torch.tensor([[0.5, 0.5, 0.2, 0.3], [0.3, 0.6, 0.1, 0.1]]), "labels": torch.tensor([3, 7]),
          "scores": torch.tensor([0.9, 0.4])}, {"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, dtype=torch.int64)}]
    o = jitter_targets(t, 5, 0.1)
    assert o[0]["boxes"].shape == (10, 4) and o[0]["labels"].tolist() == [3] * 5 + [7] * 5
    assert torch.equal(o[0]["boxes"][4], t[0]["boxes"][0]) and torch.equal(o[0]["boxes"][9], t[0]["boxes"][1])
    for g in range(2):
        iou, _ = O.box_iou(O.box_cxcywh_to_xyxy(o[0]["boxes"][5 * g:5 * g + 4]), O.box_cxcywh_to_xyxy(t[0]["boxes"][g:g + 1]))
        assert (iou > 0.7).all()
        assert (o[0]["boxes"][5 * g:5 * g + 4] != t[0]["boxes"][g]).any()
    assert o[1]["labels"].numel() == 0 and o[1]["boxes"].shape == (0, 4)
    assert t[0]["boxes"].shape == (2, 4)           # input untouched (deepcopy)


def test_registry_has_baseline_backbones():
    from spe_amd.models.cait import _REGISTRY
    for n in ("TSCAM_cait_XXS24", "TSCAM_cait_XXS36", "TSCAM_cait_XXS36_Two_Branch", "TSCAM_cait_S24", "TSCAM_cait_S36"):
        assert n in _REGISTRY


def test_deit_checkpoint_interop(tmp_path):
    """a DeiT-style CaiT checkpoint ('model' dict, 'module.'-prefixed keys, classifier heads of
    another shape) loads the way cait.py:1639-1663 does, and init_blocks_det_weight (cait.py:724-726) copies the last
    backbone blocks into the detection branch."""
    import torch
    from spe_amd.models import cait
    torch.manual_seed(0)
    factory, kw = cait.TSCAM_cait_XXS36_Two_Branch, {"num_classes": 20, "layer_to_det": 33}
    src, _ = factory(pretrained=False, **kw)
    sd = {"module." + k: torch.randn_like(v) if v.dtype.is_floating_point else v.clone() for k, v in src.state_dict().items()
          if not k.startswith("blocks_det")}
    sd["module.head.weight"] = torch.randn(1000, 192)      # ImageNet classifier of the released file: must be skipped
    path = tmp_path / "deit_cait.pth"
    torch.save({"model": sd}, path)
    dst, width = factory(pretrained=True, checkpoint_path=str(path), **kw)
    assert width == 192
    own = dst.state_dict()
    for k, v in sd.items():
        k = k[len("module."):]
        if k in own and own[k].shape == v.shape:
            assert torch.equal(own[k], v), k
    nb = len(dst.blocks_det)
    assert nb > 0
    for i in range(1, nb + 1):
        for (ka, a), (kb, b) in zip(dst.blocks[-i].state_dict().items(), dst.blocks_det[-i].state_dict().items()):
            assert ka == kb and torch.equal(a, b)


def _blob_images():
    import numpy as np
    rng = np.random.RandomState(7)
    imgs = []
    z = np.zeros((12, 16), np.uint8); imgs.append(("empty", z.copy()))
    a = z.copy(); a[3:9, 4:13] = 200; imgs.append(("rect", a))                      # 9 wide x 6 high
    a = z.copy(); a[2:11, 2:14] = 9; a[5:8, 6:10] = 0; imgs.append(("ring", a))       # a hole
    a = z.copy(); a[6, 7] = 1; imgs.append(("pixel", a))
    a = z.copy(); a[4, 2:12] = 5; a[4:10, 11] = 5; imgs.append(("thin", a))
    a = z.copy(); a[0, :] = 3; a[:, 0] = 3; a[-1, -1] = 3; imgs.append(("edges", a))
    for k in range(6):
        n = rng.rand(40, 56)
        for _ in range(3):                                                          # smooth -> blobs with holes
            n = (n + np.roll(n, 1, 0) + np.roll(n, -1, 0) + np.roll(n, 1, 1) + np.roll(n, -1, 1)) / 5
        imgs.append((f"noise{k}", ((n > np.percentile(n, 55 + 5 * k)) * 255).astype(np.uint8)))
    return imgs


def test_cam_contour_boxes_vs_oracle_and_ndimage():
    """Native border following (csrc/cambox.hip, host code) == the NumPy restatement, plus checks that do not share
    its algorithm: outer borders <-> 8-connected components, hole borders <-> enclosed 4-connected background regions
    (scipy.ndimage), and known answers (filled w x h rectangle: area (w-1)(h-1), box [x, y, x+w, y+h])."""
    import numpy as np
    import torch
    from scipy import ndimage
    from oracle import cam_oracle as CO
    from spe_amd import kernels as K
    for name, img in _blob_images():
        t = torch.from_numpy(img.copy())
        for ratio in (0.5, 0.0):
            got = K.cam_contour_boxes(t, ratio, max_boxes=4096).tolist()
            assert got == CO.multi_bboxes_from_image(img, ratio), (name, ratio)
        allb = K.cam_contour_boxes(t, 0.0, max_boxes=4096).tolist()
        if name == "empty":
            assert allb == [[0, 0, 1, 1]]
            continue
        fg = img != 0
        lab, n = ndimage.label(fg, structure=np.ones((3, 3)))
        comp = sorted([[s[1].start, s[0].start, s[1].stop, s[0].stop] for s in ndimage.find_objects(lab)])
        bg = np.pad(~fg, 1, constant_values=True)
        labb, nb = ndimage.label(bg)                                                # 4-connected background
        outside = labb[0, 0]
        holes = []
        for k, s in enumerate(ndimage.find_objects(labb), start=1):
            if k != outside:
                holes.append([s[1].start - 1 - 1, s[0].start - 1 - 1, s[1].stop - 1 + 1, s[0].stop - 1 + 1])   # un-pad, grow by 1
        assert sorted(allb) == sorted(comp + holes), name
    rect = [b for n_, im in _blob_images() if n_ == "rect" for b in CO.find_borders(im)]
    assert rect == [(8.0 * 5.0, 4, 3, 12, 8)]


def test_accuracy_and_postprocess_refine_multi():
    """util.misc.accuracy (reference util/misc.py:439-455) and PostProcessRefineMulti (conditional_detr.py:680-715) against
    loop restatements of their semantics."""
    import torch
    from spe_amd.models.conditional_detr import PostProcessRefineMulti
    from spe_amd.util.misc import accuracy
    g = torch.Generator().manual_seed(3)
    out = torch.randn(17, 9, generator=g)
    tgt = torch.randint(0, 9, (17,), generator=g)
    a1, a3 = accuracy(out, tgt, topk=(1, 3))
    assert abs(float(a1) - 100.0 * float((out.argmax(1) == tgt).float().mean())) < 1e-4
    top3 = out.topk(3, 1)[1]
    assert abs(float(a3) - 100.0 * float((top3 == tgt[:, None]).any(1).float().mean())) < 1e-4
    assert float(accuracy(out[:0], tgt[:0])[0]) == 0.0
    B, Q, Kc = 2, 12, 7
    logits, boxes = torch.randn(B, Q, Kc, generator=g), torch.rand(B, Q, 4, generator=g)
    targets = [{"labels": torch.tensor([3, 1, 3])}, {"labels": torch.tensor([6])}]
    res = PostProcessRefineMulti()({"pred_logits": logits, "pred_boxes": boxes}, torch.tensor([[10, 10], [10, 10]]), targets)
    prob = logits.sigmoid()
    for b, t in enumerate(targets):
        labs, scs, bxs = [], [], []
        for c in range(Kc):
            if c in t["labels"]:
                keep = (prob[b, :, c] >= 0.5 * prob[b, :, c].max()).nonzero().reshape(-1)
                labs += [c] * len(keep); scs.append(prob[b, keep, c]); bxs.append(boxes[b, keep])
        assert res[b]["labels"].tolist() == labs
        assert torch.equal(res[b]["scores"], torch.cat(scs)) and torch.equal(res[b]["boxes"], torch.cat(bxs))
