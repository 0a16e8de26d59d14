#!/usr/bin/env python
"""Headline benchmark: images/sec of one SPE training iteration of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = forward (CaiT-S24 TSCAM backbone + 6-layer conditional-DETR decoder, both proposal stages)
-> SetCriterion (train mode: one-to-many jitter, Hungarian matching) -> PostProcessRefine pseudo labels
-> SetCriterionRefine -> backward -> gradient all-reduce -> clip + AdamW, on synthetic 3x800x1333
batches of 2 images per GPU (BASELINE.json configs[1]).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def model_args(backbone="TSCAM_cait_S24", enc_layers=0, dec_layers=6, num_queries=100, dataset="coco", layer_to_det=23,
               dropout=0.1, nheads=8, dim_feedforward=2048, drop_path=0.0, attn_drop=0.0, backbone_drop=0.0):
    """reference main.py:37-146 defaults + BASELINE.json cfg2."""
    return argparse.Namespace(
        dataset_file=dataset, device="cuda", backbone=backbone, backbone_drop_rate=backbone_drop, drop_path_rate=drop_path,
        drop_block_rate=0.0, drop_attn_rate=attn_drop, layer_to_det=layer_to_det, lr_backbone=1e-5, masks=False, dilation=False,
        position_embedding="sine", hidden_dim=256, dropout=dropout, nheads=nheads, num_queries=num_queries,
        dim_feedforward=dim_feedforward, enc_layers=enc_layers, dec_layers=dec_layers, pre_norm=False, aux_loss=True,
        num_refines=1, frozen_weights=None, set_cost_class=2, set_cost_bbox=5, set_cost_giou=2, hung_match_ratio=5,
        box_jitter=0.1, cls_loss_coef=2, bbox_loss_coef=2, giou_loss_coef=2, img_label_loss_coef=1,
        img_label_tokens_loss_coef=1, focal_alpha=0.25, focal_gamma=2)


def synth_batch(seed, device, batch=2, H=800, W=1333, K=90, n_tgt=7):
    """BASELINE.md section 3 inputs."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, 3, H, W, generator=g)
    targets = []
    for _ in range(batch):
        labels = torch.randint(1, K + 1, (n_tgt,), generator=g)
        c = torch.rand(n_tgt, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(n_tgt, 2, generator=g) * 0.35 + 0.05
        il = torch.zeros(K, dtype=torch.int64)
        il[labels - 1] = 1
        targets.append({"boxes": torch.cat([c, wh], 1), "labels": labels, "img_label": il,
                        "orig_size": torch.tensor([H, W]), "labels_unique": torch.unique(labels)})
    mask = torch.zeros(batch, H, W, dtype=torch.bool)
    to = lambda t: t.to(device)
    return to(imgs), to(mask), [{k: to(v) for k, v in t.items()} for t in targets]


def pseudo_labels(rpp, out0, targets):
    """engine.py:295-308: stage-0 detections become the stage-1 targets."""
    orig = torch.stack([t["orig_size"] for t in targets])
    res = rpp["bbox"](out0, orig, targets)
    ps = []
    for t, r in zip(targets, res):
        p = dict(t)
        p.update({"labels": r["labels"].detach(), "boxes": r["boxes"].detach(), "scores": r["scores"].detach()})
        ps.append(p)
    return ps


_WVEC = {}


def weighted_total(l0, l1, wd):
    """sum_k weight_k * loss_k over both criteria (reference engine.py:88-93) as one stack and one dot product: the
    ~40 scalar losses would otherwise cost a multiply and an add launch each, forward and backward."""
    terms = [l0[k] for k in l0 if k in wd] + [l1[k] for k in l1 if k in wd]
    key = (tuple(k for k in l0 if k in wd), tuple(k for k in l1 if k in wd), terms[0].device)
    w = _WVEC.get(key)
    if w is None:
        w = _WVEC[key] = torch.tensor([float(wd[k]) for k in key[0]] + [float(wd[k]) for k in key[1]], dtype=torch.float32).to(key[2])
    return (torch.stack(terms) * w).sum()           # elementwise + reduce: no vendor-library (rocBLAS dot) kernel in the timed region


def host_cores():
    """Usable host cores: os.cpu_count() capped by the cgroup CPU quota (the GPU box reports 256 logical
    CPUs under a 16-CPU quota; oversubscribing it throttles the process by >50x)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def host_ram_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def _cpu_case(depth, enc_layers):
    """(model args, oracle cfg) of the bench workload with `depth` of the 24 backbone blocks."""
    from oracle import spe_oracle as O
    from spe_amd.models.cait import TSCAM_cait, _make, register_model
    name = "TSCAM_cait_S24" if depth == 24 else f"TSCAM_cait_S24_depth{depth}"
    if depth != 24:
        def fac(pretrained=False, _d=depth, **kw):
            return _make(TSCAM_cait, 384, _d, 8, 1e-5, False, **kw)
        fac.__name__ = name
        register_model(fac)
    a = model_args(backbone=name, enc_layers=enc_layers, layer_to_det=depth - 1)
    a.device = "cpu"
    cfg = O.make_cfg(embed_dim=384, depth=depth, num_heads=8, num_cls_tokens=90, layer_to_det=depth - 1, two_branch=False,
                     pos_grid=(50, 84), nheads=8, enc_layers=enc_layers, dec_layers=6, dim_feedforward=2048,
                     num_queries=100, num_refines=1, num_det_classes=91, aux_loss=True)
    return a, cfg


def cpu_baseline(enc_layers, gpu_model=None, gpu_eval=None):
    """The oracle (CPU fp32 restatement, parity-pinned to the reference) timed on this box's host cores on a bounded
    sample of the same workload: ONE 3x800x1333 image, S24 dims, forward + both criteria + backward.
    With >= 100 GB of free host RAM (the eager graph keeps ~2.7 GB per backbone block at N = 4150) all 24 blocks run -
    a measured number; otherwise 2 and 6 blocks are timed and the per-block time is scaled to 24 ("extrapolated": true).
    gpu_model / gpu_eval: when given, the oracle runs on the GPU model's CURRENT weights and its total loss is compared
    with the product's on the same image (BASELINE.json's "loss delta vs ref" of the benchmarked precision mode)."""
    from oracle import spe_oracle as O
    from spe_amd.models import build_model
    cores = host_cores()
    torch.set_num_threads(cores)
    img, mask, tg = synth_batch(99, "cpu", batch=1)
    ram = host_ram_gb()
    parity = None
    if ram >= 100.0:
        a, cfg = _cpu_case(24, enc_layers)
        if gpu_model is not None:
            sd = {k: v.detach().to("cpu", copy=True).requires_grad_(v.is_floating_point()) for k, v in gpu_model.state_dict().items()}
        else:
            torch.manual_seed(0)
            model, *_ = build_model(a)
            sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        got, pseudo = gpu_eval(img, mask, tg) if gpu_eval is not None else (None, None)
        t0 = time.perf_counter()
        # the stage-1 targets (detached inputs of criterion_refine) are the product's: an argmax-over-queries flip between
        # two nearly tied queries of a randomly initialised decoder must not pass for a loss error
        tot, *_ = O.total_loss(sd, cfg, img, mask, tg, pseudo=pseudo)
        tot.backward()
        t_img = time.perf_counter() - t0
        if gpu_eval is not None:
            ref = float(tot.detach())
            parity = {"loss_gpu": got, "loss_oracle_cpu": ref, "loss_rel_delta": abs(got - ref) / abs(ref),
                      "sample": "total weighted loss (both criteria, eval mode, no dropout), 1 image 3x800x1333, the benchmarked "
                                "model's current weights, product in the benchmarked precision mode vs the fp32 CPU oracle"}
        return ({"value": 1.0 / t_img, "unit": "images/sec", "cores": cores, "kind": "port", "extrapolated": False,
                 "sample": f"oracle fwd+criteria+bwd, 1 image 3x800x1333, S24 dims, all 24 backbone blocks: {t_img:.1f}s "
                           f"(host RAM available {ram:.0f} GB)"}, parity)
    times = {}
    for depth in (2, 6):
        a, cfg = _cpu_case(depth, enc_layers)
        torch.manual_seed(0)
        model, *_ = build_model(a)
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        t0 = time.perf_counter()
        tot, *_ = O.total_loss(sd, cfg, img, mask, tg)
        tot.backward()
        times[depth] = time.perf_counter() - t0
        del model, sd, tot
    per_block = max((times[6] - times[2]) / 4.0, 1e-9)
    rest = max(times[2] - 2 * per_block, 0.0)
    t_img = rest + 24 * per_block
    return ({"value": 1.0 / t_img, "unit": "images/sec", "cores": cores, "kind": "port", "extrapolated": True,
             "sample": (f"oracle fwd+criteria+bwd, 1 image 3x800x1333, S24 dims; host RAM {ram:.0f} GB < 100 GB, so timed with 2 and 6 "
                        f"backbone blocks ({times[2]:.1f}s, {times[6]:.1f}s), per-block {per_block:.2f}s scaled to 24 blocks + rest {rest:.1f}s")},
            parity)


def _load_profile(name):
    """(content, provenance) of a JSON under profiles/: numbers the bench line COPIES from an earlier measurement pass carry the
    file, its content hash and the label the pass wrote into it, so a stale copy is visible in the line itself."""
    import hashlib
    path = os.path.join(ROOT, "profiles", name)
    try:
        raw = open(path, "rb").read()
        obj = json.loads(raw)
        return obj, {"file": "profiles/" + name, "sha256": hashlib.sha256(raw).hexdigest()[:16], "label": obj.get("label") or obj.get("source"),
                     "measured_in_this_run": False}
    except Exception:
        return {}, {"file": "profiles/" + name, "missing": True}


def roofline_inputs():
    """HBM traffic per launch (PMC runs) and the measured peaks: profiles/roofline_inputs.json, written by
    tools/pmc_to_json.py from the rocprofv3 --pmc runs of this same command."""
    return _load_profile("roofline_inputs.json")


def parity_record():
    """Measured errors of the benchmarked precision mode against the REFERENCE at cfg2's token count
    (tests/test_config_golden.py on the GPU box -> tools/parity_summary.py -> profiles/parity_r06.json; the record names the measurement pass -
    the gpurun call whose `pytest -m gpu` produced it - in its `label`)."""
    obj, prov = _load_profile("parity_r06.json")
    if not obj:
        obj, prov = _load_profile("parity_r05.json")
    return obj, prov


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--enc-layers", type=int, default=0, help="0 = north_star headline (backbone+decoder); 3 = script value")
    # bf16s (default): forward products on split bf16 operands / fp16 attention operands, backward products on single bf16
    # operands - the mode that meets north_star's 1e-3 on logits and losses (tests/test_config_golden.py); bf16: single bf16
    # operands everywhere (round 2's headline); bf16x3: 3-term split everywhere, fp32 materialised attention
    ap.add_argument("--precision", default="bf16s", choices=["bf16", "bf16s", "bf16x3"])
    ap.add_argument("--backbone", default="TSCAM_cait_S24")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--queries", type=int, default=100)
    # the reference's launch scripts (scripts/run_voc0712.py:15-41): --backbone TSCAM_cait_XXS36_Two_Branch --layer-to-det 24
    # --enc-layers 3 --queries 300 --drop-path 0.2 --attn-drop 0.05 --backbone-drop 0.07 --height 512 --width 512 --batch 1
    ap.add_argument("--layer-to-det", type=int, default=23)
    ap.add_argument("--drop-path", type=float, default=0.0)
    ap.add_argument("--attn-drop", type=float, default=0.0)
    ap.add_argument("--backbone-drop", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # gradient wire format of the all-reduce: fp32 (DDP's, the default) or bf16 (half the xGMI bytes, one conversion pass each way)
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16"])
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # developer knobs for exercising the N > 1 code path on a one-GPU box: all ranks on one device, exchange through gloo
    # (RCCL refuses two ranks on one device).  Never set by the driver.
    backend = os.environ.get("SPE_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("SPE_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # explicit CU budget of the collective: at most 32 RCCL channels (one persistent workgroup each); GradAllReducer
        # reserves as many workgroup slots in the single-round attention grids (kernels.set_cu_reserve)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # the HIP library is built in-tree by __graft_entry__.build(); if this checkout has none (or a stale one), local
    # rank 0 builds it once and the other ranks wait for the file - there is no CPU fallback to run instead
    from spe_amd import build as _build
    if local == 0 and _build.needs_build():
        _build.build(verbose=(rank == 0))
    if world > 1:
        dist.barrier()
    from spe_amd import kernels as K
    from spe_amd import lib
    from spe_amd.dp import GradAllReducer
    from spe_amd.optim import FlatAdamW
    from spe_amd.models import build_model
    from spe_amd.util.misc import NestedTensor
    lib.load()
    K.set_precision(a.precision)
    K.manual_seed(1234 + rank)

    args = model_args(backbone=a.backbone, enc_layers=a.enc_layers, num_queries=a.queries, layer_to_det=a.layer_to_det,
                      drop_path=a.drop_path, attn_drop=a.attn_drop, backbone_drop=a.backbone_drop)
    torch.manual_seed(0)                      # identical replicas
    model, crit, crit_r, pp, rpp = build_model(args)
    model.to(dev).train()
    crit.to(dev).train()
    crit_r.to(dev).train()
    wd = crit.weight_dict
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = GradAllReducer(params, flatten_params=True, wire_dtype=torch.bfloat16 if a.wire == "bf16" else None)
    # reference main.py:177-191: AdamW, backbone parameters at lr_backbone; engine.py:161-165: clip_grad_norm_(0.1)
    groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad], "lr": 1e-4},
              {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-5}]
    opt = FlatAdamW(groups, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
    torch.manual_seed(1234 + rank)
    img, mask, targets = synth_batch(1234 + rank, dev, batch=a.batch, H=a.height, W=a.width)
    samples = NestedTensor(img, mask)

    def step():
        reducer.reset()
        out = model(samples)
        l0 = crit(out[0], targets)
        with torch.no_grad():
            ps = pseudo_labels(rpp, out[0], targets)
        l1 = crit_r(out[1], ps)
        total = weighted_total(l0, l1, wd)
        total.backward()
        reducer.finish()
        opt.step()                      # global-norm clip (0.1) + AdamW, fused on the flat buckets
        return total

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    # kernels timed live with HIP events on the launch stream: the dominant kernel (the query-major backward kernel of the fused
    # talking-heads attention: dS, dWl, dbl, dQ), the HBM-bound dK contraction and north_star's decoder cross-attention GEMM (the
    # memory-side projections ca_kcontent / ca_v / ca_kpos of reference models/transformer.py:389-419:
    # [B*S, d] x [d, d] = [8300 x 384] x [384 x 384] at cfg2)
    HBMK = "spe_attn_contract"
    DOMQ = "spe_talking_bwdq_pass2"       # query-major backward kernel + its dQ merge
    FLF, STATS = "spe_talking_flash_fwd", "spe_talking_stats"
    BWDK = "spe_talking_bwdk_pass1"       # key-major backward kernel (D, dWw, dbw, dV) + its merges
    body = model.backbone[0].body
    S_rows, d_model = a.batch * (a.height // 16) * (a.width // 16), body.embed_dim
    n_dec = args.dec_layers
    CAG_N = 2 * n_dec * d_model          # ca_kcontent + ca_v of all decoder layers in one launch (ops.multi_linear)
    CAG = f"spe_gemm_bf16nt:{S_rows},{CAG_N},{d_model}"
    K.enable_timing([DOMQ, HBMK, CAG, FLF, STATS, BWDK])
    reducer.measure = True
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    sync()
    dt_local = time.perf_counter() - t0
    peak_mem = torch.cuda.max_memory_allocated(dev)
    K_res = K.timing_results()
    K.enable_timing([])
    reducer.measure = False
    # one more (untimed) step with the entry points counted: the kernel set this throughput number belongs to
    lib.count_launches(True)
    step()
    torch.cuda.synchronize()
    kernel_set = lib.count_launches(False)
    per_rank = [dt_local]
    if world > 1:
        tl = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        gath = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(gath, tl)
        per_rank = [float(g.item()) for g in gath]
    dt = max(per_rank)
    loss_val = float(last.detach())
    # what the collective actually spanned: backend, group size and every rank's device as the process group reports them (a SCALE
    # line proves its own N), the RCCL version and channel cap, and the exposed all-reduce time of every rank
    exposed = [reducer.exposed_ms_mean()]
    dist_info = {"initialized": world > 1, "backend": None, "world_size": 1, "devices": [torch.cuda.get_device_name(dev)],
                 "wire": a.wire, "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS")}
    if world > 1:
        info = [None] * world
        dist.all_gather_object(info, {"rank": rank, "device": torch.cuda.get_device_name(dev), "index": local,
                                      "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")), "exposed_ms": exposed[0]})
        exposed = [i["exposed_ms"] for i in info]
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        dist_info.update({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
                          "devices": [f"rank {i['rank']}: cuda:{i['index']} {i['device']} {i['uuid']}" for i in info]})

    if rank == 0:
        imgs = a.batch * world * a.steps
        N = (a.height // 16) * (a.width // 16)
        Hh, dh_ = body.num_heads, body.embed_dim // body.num_heads
        default_backbone = a.backbone == "TSCAM_cait_S24" and a.height == 800 and a.width == 1333
        default_cfg = default_backbone and a.queries == 100 and a.batch == 2
        rin, rin_prov = roofline_inputs()
        par_all, par_prov = parity_record()
        kin = rin.get("kernels", {})
        pk = rin.get("peaks_measured", {})
        launches, mean_ms = K_res.get(DOMQ, (0, 0.0))
        # MFMA work of one launch: S = QK^T (a recomputation: SURVEY 8(d) does not count it), dP' = dO V^T (algorithmic) and dQ = dS K
        # (algorithmic) for all heads, 2*N*N*dh FLOP each.  `achieved` = the algorithmic products, `executed` adds the recomputed S.
        mf = (2.0 * N * N * dh_) * Hh * a.batch
        n_alg = 2
        ach = (n_alg + 1) * mf / (mean_ms * 1e-3) / 1e12 if mean_ms > 0 else 0.0
        ach_alg = n_alg * mf / (mean_ms * 1e-3) / 1e12 if mean_ms > 0 else 0.0
        # the PATH's algorithmic FLOPs per image (forward 1.16 TF at enc_layers 0 + 0.044 TF per encoder layer, x3 for fwd + bwd) x images / s
        path_tf_per_img = 3.0 * (1.16 + 0.0444 * a.enc_layers) if default_backbone else None
        # vector-pipe work that is left in this pass: the dWl outer product (2*H FLOP per score and head; the three head
        # mixes run on the matrix pipe since round 2 - S' = Wl S in fp32 on v_mfma_f32_4x4x1, dP and dS in bf16), reported
        # against the 157.3 TFLOP/s vector peak; exp2 (quarter rate) and the bf16 packing are not counted as FLOP
        valu_flop = 1 * (2.0 * Hh * Hh) * N * N * a.batch
        valu = valu_flop / (mean_ms * 1e-3) / 1e12 if mean_ms > 0 else 0.0
        # second-largest kernel family, HBM-bound: the streaming contractions over the blocked bf16 score tensors
        # (algorithmic bytes per launch: B*H*N*N*2 B read once; operands/outputs are < 2 % of that)
        c_launch, c_ms = K_res.get(HBMK, (0, 0.0))
        c_bytes = 2.0 * a.batch * Hh * N * N
        c_bw = c_bytes / (c_ms * 1e-3) / 1e9 if c_ms > 0 else 0.0
        # decoder cross-attention memory-side GEMM: ALGORITHMIC work 2*M*N*K FLOP; in bf16s the forward product runs on
        # split operands, i.e. the matrix pipe executes 3 MFMAs per algorithmic one and reads hi + lo parts of both operands
        g_launch, g_ms = K_res.get(CAG, (0, 0.0))
        # round 4: the memory-side projections run on IEEE fp16 single-term operands and leave as fp16 (ops._MemorySideKV: the consumer
        # packs fp16 MFMA operands anyway); with SPE_MEMKV=0 / bf16x3 the round-3 path (split bf16 operands in bf16s, fp32 output)
        from spe_amd import ops as _ops
        kv16 = _ops.MEMKV and a.precision != "bf16x3"
        split_fwd = a.precision == "bf16s" and not kv16
        g_flop = 2.0 * S_rows * CAG_N * d_model
        g_exec = g_flop * (3.0 if split_fwd else 1.0)
        g_bytes = (2.0 * S_rows * d_model + 2.0 * CAG_N * d_model) * (2.0 if split_fwd else 1.0) + (2.0 if kv16 else 4.0) * S_rows * CAG_N
        g_tf = g_flop / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        g_bw = g_bytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        g_floor_us = max(g_exec / 2.5e15, g_bytes / 8e12) * 1e6
        res = {
            "metric": "images/sec (whole node) at 3x800x1333 bs=2/GPU", "value": imgs / dt, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16/fp16 operands, fp32 accumulate", "precision": a.precision, "data": "synthetic",
            "precision_note": {"bf16s": "MFMA operands: forward qkv / proj / decoder Linear GEMMs split bf16 (hi + lo, 3 products), backbone MLP, attention forward "
                                        "and the decoder's memory-side projections IEEE fp16 (single term), every backward product single bf16; fp32 accumulation / "
                                        "residual stream / statistics / losses",
                               "bf16": "single bf16 MFMA operands (attention forward fp16), fp32 accumulation",
                               "bf16x3": "3-term split bf16 operands everywhere, fp32 materialised attention"}[a.precision],
            "config": {"workload": f"{a.backbone} (C={body.embed_dim}, depth {body.depth}, {Hh} heads) + {a.enc_layers}-layer encoder + 6-layer "
                                   f"conditional-DETR decoder x2 stages, {a.queries} queries, COCO heads (91), "
                                   f"{a.batch}x3x{a.height}x{a.width} per GPU (N={N} tokens), fwd + SetCriterion + "
                                   f"SetCriterionRefine + bwd + grad all-reduce + clip + AdamW",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "final_loss": loss_val,
                       "drop_rates": {"decoder": 0.1, "drop_path": a.drop_path, "attn_drop": a.attn_drop, "backbone_drop": a.backbone_drop}},
            "per_rank_ms_per_step": [t_ / a.steps * 1e3 for t_ in per_rank],
            "dp": {"cu_reserve": reducer.cu_reserve, "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                   "stats_pass_workgroups": K.STATS_NWG},
            "dist": dist_info,
            "hbm_peak_allocated_gb": round(peak_mem / 1e9, 2),           # torch allocator high-water mark through the timed steps (of 288 GB)
            "allreduce_exposed_ms_per_step": max(exposed), "allreduce_exposed_ms_per_rank": exposed,
            "roofline": {"bound": "mfma",
                         "kernel": "talking_bwdq_kernel<8,2,tail> + its dQ merge (query-major attention backward: dS, dWl, dbl, dQ = dS K accumulated in registers)",
                         "launches": launches,
                         # algorithmic work only (dP' = dO V^T, and dQ = dS K when the flash-skeleton pass ran); the recomputed S = Q K^T is `executed`
                         "avg_ms": mean_ms, "achieved": ach_alg, "executed": ach, "peak": 2500.0,
                         "peak_measured": pk.get("mfma_f16_16x16x32_tflops", pk.get("mfma_bf16_tflops")), "unit": "TFLOP/s", "frac": ach_alg / 2500.0,
                         "frac_executed": ach / 2500.0,
                         # the whole PATH against the same peak (algorithmic FLOPs per image x images / s / peak)
                         "path_tflop_per_image": path_tf_per_img,
                         "path_achieved": (path_tf_per_img * imgs / dt / world) if path_tf_per_img else None,
                         "path_frac": (path_tf_per_img * imgs / dt / world / 2500.0) if path_tf_per_img else None,
                         "traffic": kin.get("talking_bwdq_pass2", {}).get("traffic_bytes"),   # bytes/launch, PMC (profiles/roofline_inputs.json)
                         "traffic_provenance": rin_prov,
                         "limiter": "instruction issue + exposed waits of ONE wave per SIMD",
                         "note": ("priced against the MFMA roofline (its arithmetic is matrix work), bound by something else: one wave per SIMD (512 registers: Q / dO "
                                  "fragments and the 96 dQ accumulators in AccVGPRs) issues in order; per 16x16 tile of all heads ~540 instructions of which 160 are matrix "
                                  "instructions: the wave issues during ~2600 and keeps the matrix pipe busy ~1800 of the ~5100 cycles a tile takes (SQ counters: "
                                  "profiles/r06_attn_pmc.txt; region stamps: profiles/r06_bwdq_stamps.txt; DESIGN.md 4.1)"),
                         "valu_achieved": valu, "valu_peak": 157.3,
                         "valu_frac": valu / 157.3,
                         "hbm_kernel": {"bound": "hbm", "kernel": "attn_contract_kernel<3,true,*> (dK = scale dS^T Q: the one streaming read of the blocked bf16 dS)",
                                        "launches": c_launch, "avg_ms": c_ms, "achieved": c_bw, "peak": 8000.0,
                                        "peak_measured": pk.get("hbm_read_gbs"), "unit": "GB/s",
                                        "frac": c_bw / 8000.0, "traffic": kin.get("attn_contract_T", {}).get("traffic_bytes")},
                         "decoder_ca_gemm": {"kernel": (f"gemm_nt2_kernel<160,128,64,2,single,fp16 operands,fp16 out> " if kv16 else
                                                        f"gemm_nt2_kernel<128,128,{32 if split_fwd else 64},2,{'split' if split_fwd else 'single'}> ") +
                                                       f"[{S_rows}x{d_model}]x[{d_model}x{CAG_N}]: ca_kcontent + ca_v projections of the memory for all "
                                                       f"{n_dec} decoder layers in one launch (the reference runs {2 * n_dec} [{S_rows}x{d_model}]x"
                                                       f"[{d_model}x{d_model}] GEMMs per decoder pass)",
                                             "launches": g_launch, "avg_us": g_ms * 1e3, "flop": g_flop, "executed_flop": g_exec, "bytes": g_bytes,
                                             "achieved_tflops": g_tf, "mfma_frac": g_tf / 2500.0, "executed_mfma_frac": g_exec / max(g_ms, 1e-9) / 1e9 / 2500.0,
                                             "achieved_gbs": g_bw, "hbm_frac": g_bw / 8000.0, "bound": "epilogue + launch phases beside a 12-step main loop", "floor_us": g_floor_us,
                                             "note": "north_star's 60 % target is not met (the vendor GEMM: 48-52 us = 23-24 % for this product, tools/debug/mm_peak.py; an A-resident "
                                                     "persistent kernel measured 55-60 us, profiles/r06_gemm_ares.txt): at K = 384 epilogue and launch phases weigh as much as the "
                                                     "matrix pipe.  Round 4: IEEE fp16 single-term operands (11 significand bits at bf16's bytes and MFMA rate: "
                                                     "the keys / values are packed to fp16 MFMA operands by their consumer anyway) and an LDS-staged fp16 output "
                                                     "(76 MB instead of 153 MB of fp32) - isolated 155 -> 58 us = 503 TFLOP/s = 20 % of the bf16 peak "
                                                     "(tools/debug/cagemm_check.py), inside the step 154 -> ~78 us (cold operands); main loop ~36 us + stores "
                                                     "that do not overlap it (profiles/r03_gemm_ablation.txt); parity at cfg2_full / cfg5_full unchanged "
                                                     "(outputs <= 1.6e-4, weighted loss keys <= 4.4e-4)"}},
            "precision_contract": par_all.get(a.precision), "precision_contract_provenance": par_prov,
            # the other attention kernels (no N x N tensor in HBM but the backward's dS), timed live
            "flash_attention": {"stats": {"launches": K_res.get(STATS, (0, 0.0))[0], "avg_ms": K_res.get(STATS, (0, 0.0))[1]},
                                "forward": {"launches": K_res.get(FLF, (0, 0.0))[0], "avg_ms": K_res.get(FLF, (0, 0.0))[1]},
                                # key-major backward kernel: D, dWw, dbw and dV = P'd^T dO in one walk
                                "bwd_pass1_dv": {"launches": K_res.get(BWDK, (0, 0.0))[0], "avg_ms": K_res.get(BWDK, (0, 0.0))[1]},
                                "note": "kernel + its partial-result merge(s) per launch; P'd is neither stored nor saved for the backward"},
            # what ran: entry points of libspe_hip.so launched in one step, and every SPE_* developer knob that was set
            "kernel_set": dict(sorted(kernel_set.items())),
            "env_knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("SPE_")},
        }
        if world == 1 and not a.no_cpu_baseline and default_cfg:
            def gpu_eval(img1, mask1, tg1):
                model.eval(); crit.eval(); crit_r.eval()
                try:
                    with torch.no_grad():
                        tgd = [{k: v.to(dev) for k, v in t.items()} for t in tg1]
                        o = model(NestedTensor(img1.to(dev), mask1.to(dev)))
                        e0 = crit(o[0], tgd)
                        ps = pseudo_labels(rpp, o[0], tgd)
                        e1 = crit_r(o[1], ps)
                        keep = ("labels", "boxes", "scores")
                        return float(weighted_total(e0, e1, wd)), [{k: p[k].detach().cpu() for k in keep} for p in ps]
                finally:
                    model.train(); crit.train(); crit_r.train()
            try:
                res["cpu_baseline"], par = cpu_baseline(a.enc_layers, gpu_model=model, gpu_eval=gpu_eval)
                if par is not None:
                    res["loss_delta_vs_ref"] = par
            except Exception as e:      # the CPU leg must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": host_cores(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
