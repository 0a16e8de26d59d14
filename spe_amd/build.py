"""Build libspe_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Each csrc/*.hip is compiled to an object file on its own (in parallel; an object is reused while the content hash of its
source and of the shared headers is unchanged), then everything is linked into spe_amd/libspe_hip.so."""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libspe_hip.so")
COMM_LIB = os.path.join(HERE, "libspe_comm.so")      # RCCL collectives behind include/spe_comm.h (csrc/comm/)
STAMP = os.path.join(HERE, "libspe_hip.srchash")     # content hash of the sources the library was built from
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# per-file flags.  attn_flash_bwd.hip runs one wave per SIMD with its big operands / accumulators in AccVGPRs through inline assembly; the
# builtin matrix instructions around them must keep their results in ordinary VGPRs (see the file header)
EXTRA_FLAGS = {"attn_flash_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def flags_for(src):
    return FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")))


def _hash(files, extra=b""):
    h = hashlib.sha256(extra)
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _comm_sources():
    return sorted(glob.glob(os.path.join(CSRC, "comm", "*.hip"))) + [os.path.join(os.path.dirname(HERE), "include", "spe_comm.h")]


def source_hash():
    return _hash(sources() + _headers() + _comm_sources())


def needs_build():
    """True when there is no library or it was built from other sources.  Compared by content, not by mtime: a copied
    tree (the GPU box snapshot) does not keep modification times.  The collective library is optional (built best-effort)."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def _compile(src, hipcc, verbose):
    name = os.path.splitext(os.path.basename(src))[0]
    obj, tag = os.path.join(OBJ, name + ".o"), os.path.join(OBJ, name + ".hash")
    want = _hash([src] + _headers(), " ".join(flags_for(src)).encode())
    if os.path.exists(obj) and os.path.exists(tag) and open(tag).read().strip() == want:
        return obj
    cmd = [hipcc] + flags_for(src) + ["-c", src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(obj + ".tmp", obj)
    with open(tag, "w") as fh:
        fh.write(want + "\n")
    return obj


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in glob.glob(os.path.join(OBJ, "*.hash")):
            os.remove(f)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, hipcc, verbose), srcs))
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr, flush=True)       # stderr: bench.py's stdout carries exactly one JSON line
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB)                      # never leave a half-written library behind
    # the collective layer is its own small, OPTIONAL library (bench.py and the DP path default to torch.distributed): it needs
    # librccl.so (the ROCm one at link time; at run time the copy already loaded in the process - torch's - is reused).  Built
    # best-effort: a box without the RCCL development library still gets the kernel library and a valid stamp.
    ctmp = COMM_LIB + ".tmp.%d" % os.getpid()
    ccmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", ctmp,
            os.path.join(CSRC, "comm", "comm.hip"), "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(ccmd), file=sys.stderr, flush=True)
    try:
        subprocess.run(ccmd, check=True, cwd=CSRC)
        os.replace(ctmp, COMM_LIB)
    except (subprocess.CalledProcessError, OSError) as e:
        print(f"spe_amd.build: libspe_comm.so not built ({e}); spe_amd.comm.RcclComm is unavailable, torch.distributed still works",
              file=sys.stderr, flush=True)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
