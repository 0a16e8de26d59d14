"""Whole-step determinism: the training step of bench.py (forward, both criteria, backward, clip, AdamW) run twice from the same
initial state must leave bitwise equal gradients after step 1 and bitwise equal parameters after step 2.

    python tools/determinism_step.py [--depth 4] [--height 800 --width 1333] [--enc-layers 0] [--runs 3]

Prints, per run beyond the first, the parameters whose gradient / value differs from run 0 (name, max abs difference); exit
status 1 if any does.  tests/test_determinism_gpu.py runs the same function at a small size."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def run_steps(dev, depth=4, H=800, W=1333, batch=2, enc_layers=0, steps=2, precision="bf16s", noise=None, dropout=0.1):
    """-> (names, gradients after the first backward, parameters after `steps` steps)"""
    import bench
    from spe_amd import kernels as K
    from spe_amd.dp import GradAllReducer
    from spe_amd.optim import FlatAdamW
    from spe_amd.models import build_model
    from spe_amd.models.cait import TSCAM_cait, _make, register_model
    from spe_amd.util.misc import NestedTensor
    name = "TSCAM_cait_S24" if depth == 24 else f"TSCAM_cait_S24_depth{depth}"
    if depth != 24:
        def fac(pretrained=False, _d=depth, **kw):
            return _make(TSCAM_cait, 384, _d, 8, 1e-5, False, **kw)
        fac.__name__ = name
        try:
            register_model(fac)
        except Exception:
            pass
    K.set_precision(precision)
    K.manual_seed(77)
    args = bench.model_args(backbone=name, enc_layers=enc_layers, layer_to_det=depth - 1, dropout=dropout)
    torch.manual_seed(0)
    model, crit, crit_r, pp, rpp = build_model(args)
    with torch.no_grad():                       # LayerScale at a size where the blocks matter (reference init 1e-5 hides them)
        for n, p in model.named_parameters():
            if n.endswith("gamma_1") or n.endswith("gamma_2"):
                p.fill_(0.2)
    model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
    wd = crit.weight_dict
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    params = [p for _, p in named]
    reducer = GradAllReducer(params, flatten_params=True)
    groups = [{"params": [p for n, p in named if "backbone" not in n], "lr": 1e-4},
              {"params": [p for n, p in named if "backbone" in n], "lr": 1e-5}]
    opt = FlatAdamW(groups, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
    img, mask, targets = bench.synth_batch(4321, dev, batch=batch, H=H, W=W)
    samples = NestedTensor(img, mask)
    grads = None
    for s in range(steps):
        reducer.reset()
        out = model(samples)
        l0 = crit(out[0], targets)
        with torch.no_grad():
            ps = bench.pseudo_labels(rpp, out[0], targets)
        l1 = crit_r(out[1], ps)
        total = bench.weighted_total(l0, l1, wd)
        if noise is not None:
            noise()                             # foreign kernels between forward and backward: different scheduling run to run
        total.backward()
        reducer.finish()
        if s == 0:
            grads = [p.grad.detach().clone() if p.grad is not None else None for p in params]
        opt.step()
    torch.cuda.synchronize()
    return [n for n, _ in named], grads, [p.detach().clone() for p in params], float(total.detach())


def compare(names, a, b, what, limit=400):
    bad = []
    for n, x, y in zip(names, a, b):
        if x is None or y is None:
            continue
        if not torch.equal(x, y):
            bad.append((n, float((x - y).abs().max()), float(x.abs().max())))
    for n, d, m in bad[:limit]:
        print(f"   {what} differs: {n:70s} max|diff| {d:.3e} (max|x| {m:.3e})")
    return len(bad)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--enc-layers", type=int, default=0)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--precision", default="bf16s")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    xs = torch.randn(2048, 1024, device=dev)
    ref = None
    nbad = 0
    for r in range(a.runs):
        noise = None if r == 0 else (lambda r=r: [(xs @ xs.t()[:, :512]).sum() for _ in range(r)])
        names, g, p, loss = run_steps(dev, a.depth, a.height, a.width, a.batch, a.enc_layers, 2, a.precision, noise)
        print(f"run {r}: loss {loss!r}")
        if ref is None:
            ref = (g, p)
            continue
        ng = compare(names, ref[0], g, "gradient")
        npar = compare(names, ref[1], p, "parameter", 5)
        print(f"run {r}: {ng} gradients, {npar} parameters differ of {len(names)}")
        nbad += ng + npar
    print("DETERMINISTIC" if nbad == 0 else "NOT deterministic")
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
