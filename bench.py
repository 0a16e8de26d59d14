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
               dropout=0.1, nheads=8, dim_feedforward=2048):
    """reference main.py:37-146 defaults + BASELINE.json cfg2."""
    return argparse.Namespace(
        dataset_file=dataset, device="cuda", backbone=backbone, backbone_drop_rate=0.0, drop_path_rate=0.0,
        drop_block_rate=0.0, drop_attn_rate=0.0, layer_to_det=layer_to_det, lr_backbone=1e-5, masks=False, dilation=False,
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
    return torch.dot(torch.stack(terms), w)


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


def cpu_baseline(enc_layers):
    """The oracle (CPU fp32 restatement, parity-pinned to the reference) timed on this box's host cores on a
    bounded sample of the same workload: ONE 3x800x1333 image, S24 dims, forward + both criteria + backward,
    with 2 and 6 of the 24 backbone blocks; the per-block time (their difference / 4) is scaled to 24 blocks."""
    from oracle import spe_oracle as O
    from spe_amd.models import build_model
    from spe_amd.models.cait import TSCAM_cait, _make, register_model
    cores = host_cores()
    torch.set_num_threads(cores)
    times = {}
    for depth in (2, 6):
        name = f"TSCAM_cait_S24_depth{depth}"

        def fac(pretrained=False, _d=depth, **kw):
            return _make(TSCAM_cait, 384, _d, 8, 1e-5, False, **kw)
        fac.__name__ = name
        register_model(fac)
        a = model_args(backbone=name, enc_layers=enc_layers, layer_to_det=depth - 1)
        a.device = "cpu"
        torch.manual_seed(0)
        model, *_ = build_model(a)
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
        cfg = O.make_cfg(embed_dim=384, depth=depth, num_heads=8, num_cls_tokens=90, layer_to_det=depth - 1, two_branch=False,
                         pos_grid=(50, 84), nheads=8, enc_layers=enc_layers, dec_layers=6, dim_feedforward=2048,
                         num_queries=100, num_refines=1, num_det_classes=91, aux_loss=True)
        img, mask, tg = synth_batch(99, "cpu", batch=1)
        t0 = time.perf_counter()
        tot, *_ = O.total_loss(sd, cfg, img, mask, tg)
        tot.backward()
        times[depth] = time.perf_counter() - t0
        del model, sd, tot
    per_block = max((times[6] - times[2]) / 4.0, 1e-9)
    rest = max(times[2] - 2 * per_block, 0.0)
    t_img = rest + 24 * per_block
    return {"value": 1.0 / t_img, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": (f"oracle fwd+criteria+bwd, 1 image 3x800x1333, S24 dims; timed with 2 and 6 backbone blocks "
                       f"({times[2]:.1f}s, {times[6]:.1f}s), per-block {per_block:.2f}s scaled to 24 blocks + rest {rest:.1f}s")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--enc-layers", type=int, default=0, help="0 = north_star headline (backbone+decoder); 3 = script value")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--backbone", default="TSCAM_cait_S24")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

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

    args = model_args(backbone=a.backbone, enc_layers=a.enc_layers, num_queries=a.queries)
    torch.manual_seed(0)                      # identical replicas
    model, crit, crit_r, pp, rpp = build_model(args)
    model.to(dev).train()
    crit.to(dev).train()
    crit_r.to(dev).train()
    wd = crit.weight_dict
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = GradAllReducer(params, flatten_params=True)
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
    # dominant kernel (largest single-shape kernel of the step: backward pass 2 of the fused talking-heads
    # attention) timed live with HIP events on the launch stream
    DOM, HBMK = "spe_talking_fused", "spe_attn_contract"
    K.enable_timing([DOM, HBMK])
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    sync()
    dt = time.perf_counter() - t0
    K_res = K.timing_results()
    K.enable_timing([])
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(last.detach())

    if rank == 0:
        imgs = a.batch * world * a.steps
        N = (a.height // 16) * (a.width // 16)
        Hh = 8
        # K.timing_results() keys fused launches by mode: "spe_talking_fused:3" = backward pass 2
        launches, mean_ms = K_res.get(DOM + ":3", (0, 0.0))
        # algorithmic MFMA work of one launch: S = QK^T and dP' = dO V^T for all heads, 2*N*N*dh FLOP each
        alg_flop = 2 * (2.0 * N * N * 48) * Hh * a.batch
        ach = alg_flop / (mean_ms * 1e-3) / 1e12 if mean_ms > 0 else 0.0
        # the kernel's real limiter is the fp32 head-mix VALU work (3 mixes + the dWl outer product, 2*H FLOP each
        # per score and head): reported alongside against the 157.3 TFLOP/s vector peak
        valu_flop = 4 * (2.0 * Hh * Hh) * N * N * a.batch
        valu = valu_flop / (mean_ms * 1e-3) / 1e12 if mean_ms > 0 else 0.0
        # second-largest kernel family, HBM-bound: the streaming contractions over the blocked bf16 score tensors
        # (algorithmic bytes per launch: B*H*N*N*2 B read once; operands/outputs are < 2 % of that)
        c_launch, c_ms = K_res.get(HBMK, (0, 0.0))
        c_bytes = 2.0 * a.batch * Hh * N * N
        c_bw = c_bytes / (c_ms * 1e-3) / 1e9 if c_ms > 0 else 0.0
        res = {
            "metric": "images/sec (whole node) at 3x800x1333 bs=2/GPU", "value": imgs / dt, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.precision == "bf16" else "bf16x3", "data": "synthetic",
            "config": {"workload": f"{a.backbone} (C=384, depth 24, 8 heads) + {a.enc_layers}-layer encoder + 6-layer "
                                   f"conditional-DETR decoder x2 stages, {a.queries} queries, COCO heads (91), "
                                   f"{a.batch}x3x{a.height}x{a.width} per GPU (N={N} tokens), fwd + SetCriterion + "
                                   f"SetCriterionRefine + bwd + grad all-reduce + clip + AdamW",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "final_loss": loss_val},
            "roofline": {"bound": "mfma", "kernel": "talking_fused_kernel<8,2,3> (attention backward pass 2)", "launches": launches,
                         "avg_ms": mean_ms, "achieved": ach, "peak": 2500.0, "peak_measured": 1240.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                         "traffic": 9.95e8,   # bytes/launch, profiles/r01_pmc_fetch_write_v3.txt (2*FETCH_SIZE + WRITE_SIZE)
                         "note": "not MFMA-bound: fp32 head mixes (no MFMA form), fragment loads and MFMA phases serialise at 2 waves/SIMD", "valu_achieved": valu, "valu_peak": 157.3,
                         "valu_frac": valu / 157.3,
                         "hbm_kernel": {"bound": "hbm", "kernel": "attn_contract_kernel<3,*> (PV / dV / dQ / dK over blocked bf16 scores)",
                                        "launches": c_launch, "avg_ms": c_ms, "achieved": c_bw, "peak": 8000.0, "peak_measured": 6200.0, "unit": "GB/s",
                                        "frac": c_bw / 8000.0, "traffic": 6.14e8}},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(a.enc_layers)
            except Exception as e:      # the CPU leg must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": host_cores(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
