"""cfg4 matcher stress (BASELINE.json configs[3], SURVEY.md section 8: reference models/matcher.py:41-87): device cost matrix +
device Hungarian per criterion call (6 decoder layers x 2 images = 12 problems) at Q = 300, Kc in {91, 80}, M in {35, 100, 300}
targets per image, against the reference's path - the cost matrix copied to the host and scipy.optimize.linear_sum_assignment run
problem by problem - on this box's host cores.  Assignments must agree pair for pair.  Writes gpurun_out/r04_cfg4.json
(copied to profiles/r04_cfg4.json).  Run on the GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy.optimize import linear_sum_assignment
import bench
from spe_amd import kernels as K

dev = torch.device("cuda:0")


def ev_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


rows = []
L, B, Q = 6, 2, 300
for Kc in (91, 80):
    for M in (35, 100, 300):
        g = torch.Generator().manual_seed(Kc * 1000 + M)
        logits = torch.randn(L, B, Q, Kc, generator=g).to(dev)
        cxy = torch.rand(L, B, Q, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(L, B, Q, 2, generator=g) * 0.35 + 0.05
        boxes = torch.cat([cxy, wh], -1).to(dev)
        total = B * M
        tid = torch.randint(1, Kc, (total,), generator=g).to(torch.int32).to(dev)
        tb = torch.cat([torch.rand(total, 2, generator=g) * 0.6 + 0.2, torch.rand(total, 2, generator=g) * 0.35 + 0.05], 1).to(dev)
        toff = torch.tensor([i * M for i in range(B + 1)], dtype=torch.int32, device=dev)
        cost, err = K.matcher_cost(logits, boxes, tid, tb, toff, total, 2.0, 5.0, 2.0)
        srow, gidx, lidx = K.hungarian(cost, toff, L, B, Q, total)
        torch.cuda.synchronize()
        t_cost = ev_time(lambda: K.matcher_cost(logits, boxes, tid, tb, toff, total, 2.0, 5.0, 2.0))
        t_hung = ev_time(lambda: K.hungarian(cost, toff, L, B, Q, total))
        t_both = ev_time(lambda: K.hungarian(K.matcher_cost(logits, boxes, tid, tb, toff, total, 2.0, 5.0, 2.0)[0], toff, L, B, Q, total))
        # the reference's path: .cpu() of the cost matrix (a device -> host sync) and SciPy problem by problem (matcher.py:83-86)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ch = cost.cpu()
            pairs = []
            for l in range(L):
                for b in range(B):
                    blk = ch[l, Q * b * M:Q * (b + 1) * M].view(Q, M).numpy()
                    pairs.append(linear_sum_assignment(blk))
            best = min(best, time.perf_counter() - t0)
        t_lsa = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for l in range(L):
                for b in range(B):
                    linear_sum_assignment(ch[l, Q * b * M:Q * (b + 1) * M].view(Q, M).numpy())
            t_lsa = min(t_lsa, time.perf_counter() - t0)
        # pair-for-pair agreement with SciPy
        srow_c, gidx_c = srow.cpu(), gidx.cpu()
        same = True
        k = 0
        for l in range(L):
            for b in range(B):
                qi, ti = pairs[k]; k += 1
                o = l * total + b * M
                n = min(Q, M)
                dq = (srow_c[o:o + n] - (l * B + b) * Q).tolist()
                dt = (gidx_c[o:o + n] - b * M).tolist()
                same = same and sorted(zip(dq, dt)) == sorted(zip(qi.tolist(), ti.tolist()))
        # algorithmic bytes of the cost kernel: reads B*Q*(Kc + 4) + total*5 floats per layer, writes Q*total floats per layer
        bytes_cost = L * 4.0 * (B * Q * (Kc + 4) + total * 5 + Q * total)
        rows.append({"Q": Q, "Kc": Kc, "targets_per_image": M, "problems": L * B,
                     "device_cost_ms": t_cost, "device_hungarian_ms": t_hung, "device_cost_plus_hungarian_ms": t_both,
                     "host_copy_plus_scipy_ms": best * 1e3, "host_scipy_only_ms": t_lsa * 1e3,
                     "device_le_host": t_both <= best * 1e3, "pairs_equal_scipy": bool(same),
                     "cost_kernel_bytes": bytes_cost, "cost_kernel_gbs": bytes_cost / (t_cost * 1e-3) / 1e9,
                     "cost_kernel_hbm_frac": bytes_cost / (t_cost * 1e-3) / 8e12})
        print(rows[-1], flush=True)
out = {"label": "round 4 cfg4 matcher stress: spe_matcher_cost + spe_hungarian vs host copy + SciPy, per criterion call",
       "host_cores": bench.host_cores(), "device": torch.cuda.get_device_name(0),
       "note": "latency-bound: the cost matrices are 0.25 - 2.2 MB per call; the comparison that matters is device path vs the host "
               "round trip of the reference (a device->host sync per criterion call, 12 per iteration)", "rows": rows}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r04_cfg4.json", "w"), indent=1)
