"""Per-entry-point / per-GEMM-shape time of one cfg2 training step (HIP events around every libspe_hip call).
Run on the GPU box:  python tools/step_profile.py [--top 40]"""
import argparse
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from spe_amd import kernels as K
from spe_amd import lib
from spe_amd.dp import GradAllReducer
from spe_amd.optim import FlatAdamW
from spe_amd.models import build_model
from spe_amd.util.misc import NestedTensor


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib.load()
    K.set_precision("bf16")
    K.manual_seed(1234)
    args = bench.model_args()
    torch.manual_seed(0)
    model, crit, crit_r, pp, rpp = build_model(args)
    model.to(dev).train(); crit.to(dev).train(); crit_r.to(dev).train()
    wd = crit.weight_dict
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = GradAllReducer(params, flatten_params=True)
    opt = FlatAdamW(params, reducer, lr=1e-4, weight_decay=1e-4, max_grad_norm=0.1)
    img, mask, targets = bench.synth_batch(1234, dev)
    samples = NestedTensor(img, mask)

    def step():
        reducer.reset()
        out = model(samples)
        l0 = crit(out[0], targets)
        with torch.no_grad():
            ps = bench.pseudo_labels(rpp, out[0], targets)
        l1 = crit_r(out[1], ps)
        total = bench.weighted_total(l0, l1, wd)
        total.backward()
        reducer.finish()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    events = defaultdict(list)
    orig = lib.call
    names = {n: [an for _, an in sig] for n, sig in lib.PROTOS.items()}

    def timed(name, *args):
        key = name
        if name in ("spe_gemm_f32", "spe_gemm_ex"):
            d = dict(zip(names[name], args))
            key = "gemm M=%d N=%d K=%d %s%s b=%dx%d sk=%d" % (d["M"], d["N"], d["K"], "T" if d["transA"] else "N",
                                                              "T" if d["transB"] else "N", d["batch0"], d["batch1"], d["splitk"])
        elif name == "spe_gemm_bf16nt":
            d = dict(zip(names[name], args))
            key = "gemm16 M=%d N=%d K=%d act=%s bias=%d" % (d["M"], d["N"], d["K"], d.get("act"), int(bool(d.get("bias"))))
        elif name == "spe_gemm_bf16nt_ex":
            d = dict(zip(names[name], args))
            key = "gemm16ex M=%d N=%d K=%d act=%s aux=%d C=%d C2=%d o16=%d o16T=%d cs=%d" % (
                d["M"], d["N"], d["K"], d.get("act"), int(bool(d["aux"])), int(bool(d["C"])), int(bool(d["C2"])), int(bool(d["out16"])),
                int(bool(d["out16T"])), int(bool(d["colsum"])))
        elif name == "spe_cvt_bf16":
            d = dict(zip(names[name], args))
            key = "cvt R=%d C=%d T=%d cs=%d aux=%d" % (d["R"], d["C"], int(bool(d["outT"])), int(bool(d["colsum"])), int(bool(d["aux"])))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(name, *args)
        e1.record()
        events[key].append((e0, e1))
        return r

    lib.call = timed
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    step()
    w1.record()
    torch.cuda.synchronize()
    lib.call = orig
    rows = [(sum(x.elapsed_time(y) for x, y in ev), len(ev), k) for k, ev in events.items()]
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("step wall (with event overhead) %.2f ms ; libspe_hip calls %d, %.2f ms" % (w0.elapsed_time(w1), sum(r[1] for r in rows), tot))
    for t, n, k in rows[:a.top]:
        print("%8.3f ms %5d x %8.1f us  %s" % (t, n, t / n * 1e3, k))


if __name__ == "__main__":
    main()
