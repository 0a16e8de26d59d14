"""Build libspe_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspe_hip.so")
STAMP = os.path.join(HERE, "libspe_hip.srchash")     # content hash of the sources the library was built from


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build():
    """True when there is no library or it was built from other sources.  Compared by content, not by mtime: a copied
    tree (the GPU box snapshot) does not keep modification times."""
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", tmp] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB)                      # never leave a half-written library behind
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
