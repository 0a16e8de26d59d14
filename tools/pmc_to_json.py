"""HBM traffic per launch from rocprofv3 PMC runs -> profiles/roofline_inputs.json (what bench.py's `roofline.traffic`
fields are read from - no literals in bench.py).

    # on the GPU box, each counter in its OWN run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_to_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/roofline_inputs.json "<note>"

Counter values are KB per launch (rocprofv3's unit for FETCH_SIZE / WRITE_SIZE); gfx950 correction from the guide's HBM
section: FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE as reported.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {
    "talking_stats": "talking_stats_kernel<8, 2, true,",
    "attn_contract_T": "attn_contract_kernel<3, true",
    "flash_fwd": "talking_flash_fwd_kernel<8, 2, true, false>",
    "flash_merge": "flash_merge_kernel",
    "talking_bwdq_pass2": "talking_bwdq_kernel<8, 2, true, false>",
    "talking_bwdk_pass1": "talking_bwdk_kernel<8, 2, true, false>",
    "gemm_nt": "gemm_nt",
}


def per_launch(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                for key, sub in KERNELS.items():
                    if sub in row["Kernel_Name"]:
                        acc[key] += float(row["Counter_Value"]); cnt[key] += 1
    return {k: acc[k] / cnt[k] for k in acc}, dict(cnt)


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fe, nf = per_launch(fetch_dir, "FETCH_SIZE")
    wr, nw = per_launch(write_dir, "WRITE_SIZE")
    res = {"note": note, "label": note, "units": "bytes per launch; fetch = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE", "kernels": {}}
    prev = out if os.path.exists(out) else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "roofline_inputs.json")
    if os.path.exists(prev):
        old = json.load(open(prev))
        for k in ("peaks_measured",):
            if k in old:
                res[k] = old[k]
    for k in KERNELS:
        if k in fe or k in wr:
            f, w = 2.0 * fe.get(k, 0.0) * 1024, wr.get(k, 0.0) * 1024
            res["kernels"][k] = {"fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w, "launches_sampled": [nf.get(k, 0), nw.get(k, 0)]}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
