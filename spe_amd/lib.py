"""ctypes binding of libspe_hip.so.  include/spe_hip.h is the single source of truth: the
prototypes are parsed from it, so a symbol declared there but missing from the library (or the
other way round) fails at import, loudly."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "spe_hip.h")
# SPE_HIP_LIB: developer override used by tools/ab.py to A/B kernel variants built with different -D flags
LIBPATH = os.environ.get("SPE_HIP_LIB") or os.path.join(HERE, "libspe_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "uint64_t": ctypes.c_uint64, "size_t": ctypes.c_size_t, "spe_stream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int spe_*(...)` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(spe_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        sig = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    sig.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, an = a.rsplit(" ", 1)
                    sig.append((_CTYPES[ty.replace("const ", "").strip()], an))
        protos[name] = sig
    return protos


class SpeLibraryError(RuntimeError):
    pass


_lib = None
PROTOS = parse_header()


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise SpeLibraryError(
            f"{LIBPATH} is missing - build it with `python -m spe_amd.build` (there is no CPU fallback)")
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of this process BEFORE our library
    # resolves the same SONAME, otherwise two runtimes coexist and our launches see no device (hipErrorNoDevice).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIBPATH)
    for name, sig in PROTOS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SpeLibraryError(f"libspe_hip.so does not export {name} declared in spe_hip.h") from e
        fn.restype = ctypes.c_int
        fn.argtypes = [t for t, _ in sig]
    _lib = lib
    return lib


_fn = {}
COUNTS = None            # entry point -> launches since count_launches(True); None = not counting


def count_launches(on=True):
    """Start (fresh counters) / stop counting the launches per entry point; -> the counters collected so far.  bench.py records
    them for one step so that a throughput number and a parity claim can be tied to the same kernel set."""
    global COUNTS
    prev = COUNTS
    COUNTS = {} if on else None
    return prev


def call(name, *args):
    """Invoke an entry point; raise on a non-zero status."""
    if COUNTS is not None:
        COUNTS[name] = COUNTS.get(name, 0) + 1
    f = _fn.get(name)
    if f is None:
        f = _fn[name] = getattr(load(), name)
    rc = f(*args)
    if rc != 0:
        raise SpeLibraryError(f"{name} failed with status {rc}")
