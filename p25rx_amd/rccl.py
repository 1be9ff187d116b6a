"""ctypes binding of libp25fe_rccl.so (include/p25fe_rccl.h): the time-sharded step over RCCL behind the C ABI.

This is what a Rust host binds for the N > 1 path; bench.py --gpus N times exactly these calls.  torch
supplies device memory and the stream only.  No fallback: without the library the import of the step fails.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libp25fe_rccl.so")
ID_BYTES = 128
GATHER = {"none": 0, "root": 1, "all": 2, "root_exact": 3}

SYMBOLS = ["p25fe_rccl_unique_id", "p25fe_shard_create", "p25fe_shard_destroy", "p25fe_shard_dibit_cap", "p25fe_shard_step",
           "p25fe_shard_offsets", "p25fe_shard_stream_dev", "p25fe_shard_comm_ms", "p25fe_shard_comm_timing", "p25fe_shard_gather_ran",
           "p25fe_shard_step_pipelined", "p25fe_shard_join", "p25fe_shard_info", "p25fe_shard_prepare"]


class ShardInfo(C.Structure):
    """p25fe_shard_info_t (include/p25fe_rccl.h)"""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("rccl_ranks", C.c_int32), ("rccl_rank", C.c_int32), ("device", C.c_int32),
                ("comms", C.c_int32), ("pipe_layout", C.c_int32), ("gather_ran", C.c_int32), ("staged", C.c_int32), ("head_wait", C.c_int32),
                ("broken", C.c_int32), ("reserved", C.c_int32), ("steps", C.c_uint64), ("pci_bus_id", C.c_char * 32)]

_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    _lib.load()                                                  # libp25fe.so first (and torch's HIP runtime before both)
    if not os.path.exists(LIB_PATH):
        raise ImportError("libp25fe_rccl.so not built: make -C p25rx_amd/csrc rccl")
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.p25fe_rccl_unique_id.argtypes = [vp]
    L.p25fe_shard_create.argtypes = [vp, C.c_int, C.c_int, vp, sz, C.POINTER(vp)]
    L.p25fe_shard_destroy.argtypes = [vp]
    L.p25fe_shard_destroy.restype = None
    L.p25fe_shard_dibit_cap.argtypes = [vp]
    L.p25fe_shard_dibit_cap.restype = sz
    L.p25fe_shard_step.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.p25fe_shard_step_pipelined.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp]
    L.p25fe_shard_join.argtypes = [vp, vp]
    L.p25fe_shard_offsets.argtypes = [vp, vp]
    L.p25fe_shard_stream_dev.argtypes = [vp]
    L.p25fe_shard_stream_dev.restype = vp
    L.p25fe_shard_comm_ms.argtypes = [vp, C.POINTER(C.c_double * 3), C.POINTER(C.c_uint64)]
    L.p25fe_shard_comm_timing.argtypes = [vp, C.c_int]
    L.p25fe_shard_gather_ran.argtypes = [vp]
    L.p25fe_shard_info.argtypes = [vp, C.POINTER(ShardInfo)]
    L.p25fe_shard_prepare.argtypes = [vp, vp]
    _LIB = L
    return L


def unique_id():
    """rank 0: a fresh communicator id (128 bytes) to hand to the other ranks by any means"""
    buf = C.create_string_buffer(ID_BYTES)
    rc = load().p25fe_rccl_unique_id(buf)
    if rc:
        raise _lib.P25feError(rc, "p25fe_rccl_unique_id failed")
    return buf.raw


class ShardStep:
    """One rank of a time-sharded capture: p25fe_shard_create / _step / _offsets / _comm_ms.

    fe: a ONE-channel FrontEnd on this rank's GPU; comm_id: the 128 bytes of unique_id(), or None for the TEST HOOK (all ranks
    on one GPU, exchanges through the POSIX shared-memory object named by $P25FE_SHARD_SHM -- never the product path)."""

    def __init__(self, fe, rank, world, n_per_rank, comm_id):
        self.L = load()
        self.fe, self.rank, self.world, self.n = fe, rank, world, n_per_rank
        self.h = C.c_void_p()
        idbuf = C.create_string_buffer(comm_id, ID_BYTES) if comm_id is not None else None
        rc = self.L.p25fe_shard_create(fe.h, rank, world, idbuf, n_per_rank, C.byref(self.h))
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_create: " + _lib.load().p25fe_strerror(rc).decode())
        self.dibit_cap = int(self.L.p25fe_shard_dibit_cap(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.p25fe_shard_destroy(self.h)
            self.h = None

    __del__ = close

    def step(self, buf, dibits, result, gather="root", fmt=_lib.FMT_CF32, pipelined=False):
        """buf: device tensor [halo + n_per_rank, 2] (the halo part is overwritten); dibits: uint8 device row of dibit_cap bytes;
        result: uint8 device tensor of one p25fe_result_t.  Enqueues on torch's current stream.  pipelined: p25fe_shard_step_pipelined
        (only K1 on the current stream, the rest on the handle's receive stream; join() before reading the outputs)."""
        import torch
        st = C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)
        fn = self.L.p25fe_shard_step_pipelined if pipelined else self.L.p25fe_shard_step
        rc = fn(self.h, C.c_void_p(buf.data_ptr()), fmt, C.c_void_p(dibits.data_ptr()), C.c_void_p(result.data_ptr()), GATHER[gather], st)
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_step%s: " % ("_pipelined" if pipelined else "") + _lib.load().p25fe_strerror(rc).decode())

    def join(self, device=None):
        """torch's current stream waits for everything pipelined steps have enqueued"""
        import torch
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        rc = self.L.p25fe_shard_join(self.h, st)
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_join")

    def offsets(self):
        """after a synchronise: the world + 1 dibit offsets of the capture"""
        off = np.zeros(self.world + 1, dtype=np.uint64)
        rc = self.L.p25fe_shard_offsets(self.h, off.ctypes.data_as(C.c_void_p))
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_offsets: " + _lib.load().p25fe_strerror(rc).decode())
        return off

    def stream(self, torch, device, n):
        """the ordered dibit stream on the device (rank 0; every rank with gather 'all'): a uint8 tensor copy of its first n bytes"""
        class _Raw:                                               # the library's device buffer, seen through __cuda_array_interface__
            pass
        raw = _Raw()
        raw.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "version": 2,
                                        "data": (int(self.L.p25fe_shard_stream_dev(self.h)), False)}
        return torch.as_tensor(raw, device=device).clone()

    def info(self):
        """what the library itself knows about the job (p25fe_shard_info): RCCL's rank count, this rank's GPU, the agreed layout"""
        i = ShardInfo()
        rc = self.L.p25fe_shard_info(self.h, C.byref(i))
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_info")
        d = {k: getattr(i, k) for k, _ in ShardInfo._fields_ if k != "reserved"}
        d["pci_bus_id"] = i.pci_bus_id.decode(errors="replace")
        d["gather_ran"] = {b: a for a, b in GATHER.items()}.get(int(i.gather_ran), "?")
        return d

    def prepare(self, device=None):
        """once per stream: the side stream must not share a hardware queue with torch's current stream or the receive stream"""
        import torch
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        rc = self.L.p25fe_shard_prepare(self.h, st)
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_prepare")

    def comm_timing(self, every):
        """HIP events around the exchanges on every `every`-th step (0: never; library default 16)"""
        rc = self.L.p25fe_shard_comm_timing(self.h, int(every))
        if rc:
            raise _lib.P25feError(rc, "p25fe_shard_comm_timing")

    def gather_ran(self):
        """the gather mode the last step executed, as a key of GATHER"""
        v = int(self.L.p25fe_shard_gather_ran(self.h))
        return {b: a for a, b in GATHER.items()}.get(v, "?")

    def comm_ms(self):
        ms = (C.c_double * 3)()
        n = C.c_uint64(0)
        self.L.p25fe_shard_comm_ms(self.h, C.byref(ms), C.byref(n))
        return {"halo_send_recv": ms[0], "summary_all_gather": ms[1], "dibit_gather": ms[2], "steps": int(n.value)}
