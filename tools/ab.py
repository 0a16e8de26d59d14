"""Build kernel variants for A/B timing on the GPU box.

  python tools/ab.py name1 "-DFOO=1" name2 "-DFOO=2 -DBAR=0" ...

writes build_ab/<name>.so (git-ignored, travels with gpurun), built with -DSPE_ABLATE - the only builds in which the timing-experiment
switches (SPE_DBG_*, SPE_ABL_*) of csrc/ exist (csrc/common.h); run a tool against one with
  SPE_HIP_LIB=build_ab/<name>.so python tools/time_fused.py
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spe_amd.build import CSRC, sources  # noqa: E402


def one(name, flags):
    out = os.path.join(ROOT, "build_ab", name + ".so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSPE_ABLATE", "-o", out] + flags.split() + sources()
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    pairs = list(zip(sys.argv[1::2], sys.argv[2::2]))
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(lambda p: one(*p), pairs):
            print(o)
