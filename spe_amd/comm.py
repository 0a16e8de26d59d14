"""RCCL collectives through the C ABI of libspe_comm.so (include/spe_comm.h) - the native collective layer of the
data-parallel path (reference util/misc.py:414-436, main.py:172,
models/conditional_detr.py:438-440).

One communicator per process (one process per GPU).  Rendezvous: rank 0 draws the RCCL unique id and publishes it
through a torch TCPStore on MASTER_ADDR:MASTER_PORT (the same environment contract as the reference's `env://`); the
collectives themselves never touch torch.distributed.  `spe_amd.dp.GradAllReducer(..., comm=RcclComm(...))` issues its
bucket all-reduces through this layer on a side HIP stream; by default it uses torch.distributed's "nccl" backend (the
same RCCL), which is what bench.py runs.
"""
import ctypes
import os

import torch

from . import lib as _lib

HEADER = os.path.join(os.path.dirname(_lib.HERE), "include", "spe_comm.h")
LIBPATH = os.path.join(_lib.HERE, "libspe_comm.so")
PROTOS = _lib.parse_header(HEADER)
ID_BYTES = 128
_DT = {torch.float32: 0, torch.bfloat16: 1}
_so = None


def load():
    """ctypes handle of libspe_comm.so with every prototype of spe_comm.h bound (missing symbol -> SpeLibraryError)."""
    global _so
    if _so is not None:
        return _so
    if not os.path.exists(LIBPATH):
        raise _lib.SpeLibraryError(f"{LIBPATH} is missing - build it with `python -m spe_amd.build`")
    so = ctypes.CDLL(LIBPATH)            # torch is imported: its librccl.so.1 is the RCCL of this process
    for name, sig in PROTOS.items():
        try:
            fn = getattr(so, name)
        except AttributeError as e:
            raise _lib.SpeLibraryError(f"libspe_comm.so does not export {name} declared in spe_comm.h") from e
        fn.restype = ctypes.c_int
        fn.argtypes = [t for t, _ in sig]
    _so = so
    return so


def _call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise _lib.SpeLibraryError(f"{name} failed with status {rc}")


class RcclComm:
    """The process' communicator.  rank / world default to RANK / WORLD_SIZE; the unique id travels through `store`
    (a torch.distributed.Store; default: TCPStore on MASTER_ADDR:MASTER_PORT, rank 0 hosts it)."""

    _generation = 0        # communicators created by this process so far (every rank creates them in the same order)

    def __init__(self, rank=None, world=None, store=None, device=None, port=None, timeout_s=300):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        if device is not None:
            torch.cuda.set_device(device)
        buf = ctypes.create_string_buffer(ID_BYTES)
        if self.world > 1:
            if store is None:
                import datetime
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    # the process group's own store, under a key prefix: no second port to collide with another job's
                    store = dist.PrefixStore("spe_comm", dist.distributed_c10d._get_default_store())
                else:
                    # SPE_COMM_PORT / `port`: an explicit rendezvous port; default MASTER_PORT + 1 (documented in INTEGRATION.md)
                    port = int(port or os.environ.get("SPE_COMM_PORT") or int(os.environ.get("MASTER_PORT", "29533")) + 1)
                    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, self.world, self.rank == 0,
                                          timeout=datetime.timedelta(seconds=timeout_s))
            # every communicator of the job gets its own key: the n-th RcclComm a process creates meets the n-th of every other rank
            # (a re-created reducer, several groups) - a fixed key would hand a later communicator the stale id of an earlier one
            RcclComm._generation += 1
            key = "spe_comm_id/%d" % RcclComm._generation
            if self.rank == 0:
                _call("spe_comm_unique_id", buf)
                store.set(key, buf.raw)
            else:
                buf = ctypes.create_string_buffer(store.get(key), ID_BYTES)
        else:
            _call("spe_comm_unique_id", buf)
        _call("spe_comm_init", self.rank, self.world, buf)
        self.stream = torch.cuda.Stream()          # collectives run here, ordered against the compute stream by events

    def all_reduce_async(self, t):
        """In-place sum of a contiguous fp32 / bf16 tensor on the communicator's side stream, after everything enqueued so
        far on the current stream.  -> event to wait for (torch.cuda.current_stream().wait_event(ev))."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DT
        ready = torch.cuda.Event()
        ready.record()
        self.stream.wait_event(ready)
        _call("spe_comm_allreduce", ctypes.c_void_p(t.data_ptr()), t.numel(), _DT[t.dtype], ctypes.c_void_p(self.stream.cuda_stream))
        done = torch.cuda.Event()
        done.record(self.stream)
        t.record_stream(self.stream)
        return done

    def all_reduce(self, t):
        torch.cuda.current_stream().wait_event(self.all_reduce_async(t))
        return t

    def broadcast(self, t, root=0):
        """In place, on the communicator's side stream like the all-reduces (one stream per communicator: the collectives of a
        communicator are then ordered by the stream, not by RCCL's internals); the current stream waits for the result."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DT
        ready = torch.cuda.Event()
        ready.record()
        self.stream.wait_event(ready)
        _call("spe_comm_broadcast", ctypes.c_void_p(t.data_ptr()), t.numel(), _DT[t.dtype], root, ctypes.c_void_p(self.stream.cuda_stream))
        done = torch.cuda.Event()
        done.record(self.stream)
        t.record_stream(self.stream)
        torch.cuda.current_stream().wait_event(done)
        return t

    def destroy(self):
        torch.cuda.synchronize()
        _call("spe_comm_destroy")
