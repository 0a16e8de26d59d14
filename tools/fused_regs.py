"""profiles/r02_fused_vgpr_scratch.txt: registers / scratch of every instantiation of the fused score kernel, from `hipcc -S`."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "spe_amd", "csrc", "attn_fused.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "f.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.dirname(src), src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
rows = []
for m in re.finditer(r"\.name:\s+(_Z20talking_fused_kernelILi(\d+)ELi(\d+)ELb(\d)ELi(\d)ELb(\d)ELi(\d)E\S*)\n(.*?)\.wavefront_size", txt, re.S):
    body = m.group(8)
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))
    rows.append((int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), g("vgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
print("# hipcc --offload-arch=gfx950 -O3 -S spe_amd/csrc/attn_fused.hip (ROCm 7.2): registers and scratch of every instantiation of")
print("# talking_fused_kernel<H, DSTEPS, TAIL16, MODE, DROP, KT>  (mode 0 statistics, 1 write, 2 backward pass 1, 3 backward pass 2)")
print("# H DSTEPS TAIL16 MODE DROP | vgpr_count  scratch_bytes  vgpr_spills  sgpr_spills  occupancy(waves/SIMD)")
for r in sorted(rows, key=lambda r: (-r[0], -r[1], -r[2], r[3], r[4])):
    occ = min(8, 512 // max(r[5], 1))
    print("  %d    %d      %d     %d    %d  |   %3d      %3d      %3d      %3d      %d" % (r + (occ,)))
