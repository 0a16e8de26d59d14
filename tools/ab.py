"""Build kernel variants for A/B timing on the GPU box.

  python tools/ab.py name1 "-DFOO=1" name2 "-DFOO=2 -DBAR=0" ...

writes build_ab/<name>.so (git-ignored, travels with gpurun), built with -DSPE_ABLATE - the only builds in which the timing-experiment
switches (SPE_DBG_*, SPE_ABL_*) of csrc/ exist (csrc/common.h); run a tool against one with
  SPE_HIP_LIB=build_ab/<name>.so python tools/debug/attn_time.py
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spe_amd.build import CSRC, OBJ, sources, flags_for  # noqa: E402

# a variant only recompiles the sources its flags can reach (ONLY=file1.hip,file2.hip in the environment) and links the product build's
# objects for everything else
ONLY = [f for f in os.environ.get("ONLY", "").split(",") if f]


def one(name, flags):
    out = os.path.join(ROOT, "build_ab", name + ".so")
    objs = []
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        prod = os.path.join(OBJ, base + ".o")
        if ONLY and os.path.basename(src) not in ONLY and os.path.exists(prod):
            objs.append(prod)
            continue
        obj = os.path.join(ROOT, "build_ab", name + "." + base + ".o")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags_for(src) + ["-DSPE_ABLATE"] + flags.split() + ["-c", src, "-o", obj], check=True, cwd=CSRC)
        objs.append(obj)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    pairs = list(zip(sys.argv[1::2], sys.argv[2::2]))
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(lambda p: one(*p), pairs):
            print(o)
