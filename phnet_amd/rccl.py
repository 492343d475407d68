"""Data-path collectives issued STRAIGHT into RCCL on our own HIP streams - no torch.distributed Work objects.

Why (the hipErrorCapturedEvent abort of round 2, gpurun_out/r02_final_gputests.log): ProcessGroupNCCL puts every EAGER
collective on its watchdog's list and the watchdog thread polls the Work's end event (`WorkNCCL::isCompleted ->
hipEventQuery`) every ~100 ms until it has retired it.  That event was recorded on the process group's internal stream.
On HIP, `hipEventQuery` fails with hipErrorCapturedEvent when the stream the event was LAST RECORDED ON is capturing at
the time of the query - whether or not the record itself was captured (tests/tools/probe_event_capture.py shows exactly
this on the box).  A torch collective issued under capture makes the process group's internal stream join the capture;
if the watchdog still holds an un-retired eager Work of that group (a warm-up collective, the initial broadcast) its
next poll throws inside the watchdog thread and `terminate` takes the process down.  A sleep only shrinks the window.

Removed by construction here: collectives that may run under capture never go through torch.  They are `ncclAllReduce`
calls (ctypes into the librccl.so torch already loaded) on OUR streams, into communicators that two dedicated process
groups bootstrap (rendezvous, unique-id exchange and `ncclCommInitRank` stay torch's job; `_comm_ptr` hands out the
handle).  Those groups run exactly one eager warm-up collective each - before any capture - and are never used through
torch again, so their internal streams never capture; the default group (broadcast of the initial weights, barriers,
the max-over-ranks of the timing) is never used under capture either.  No watchdog list ever contains an event whose
stream captures.

Two communicators, so that the two kinds of traffic do not queue behind each other (one communicator serialises its
collectives): `small` - the SyncBatchNorm statistic exchanges (<= 8 KB, on the compute stream: the next kernel needs
them) - and `bulk` - the gradient buckets (tens to hundreds of MB, on a side stream, overlapped with the trunk's
backward).  With a single communicator the first bucket (77 % of the gradient bytes) would sit in front of every
SyncBatchNorm exchange of the backward that is supposed to hide it.
"""
import ctypes
import os
from typing import List, Optional

import torch
import torch.distributed as dist

NCCL_SUM = 0
_DTYPE = {torch.float32: 7, torch.float64: 8, torch.int64: 4, torch.int32: 2}       # rccl.h ncclDataType_t


def _librccl():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path)                                   # the copy torch's process groups use: same handle, same state
    lib.ncclAllReduce.restype = ctypes.c_int
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    lib.ncclCommCount.restype = ctypes.c_int
    lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    return lib


class RcclStreams:
    """all_reduce_(t): in place, SUM, on the CURRENT torch stream (communicator `small`).
    all_reduce_async_(t) + join(): on the side stream (communicator `bulk`); fork / join by stream waits, so under capture
    the bucket becomes a parallel branch of the graph."""

    def __init__(self, device: Optional[torch.device] = None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclStreams needs an initialised torch.distributed default group (rendezvous)")
        if dist.get_backend() != "nccl":
            raise RuntimeError(f"RcclStreams needs the nccl (= RCCL) backend, not {dist.get_backend()}")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("RcclStreams must be built before any capture (communicator bootstrap is eager)")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.lib = _librccl()
        self.world = dist.get_world_size()
        self.groups, self.comms = [], []
        for _ in range(2):
            pg = dist.new_group(backend="nccl")               # collective over ALL ranks: every rank calls it, same order
            warm = torch.ones(8, device=self.device)
            dist.all_reduce(warm, group=pg)                   # creates the communicator (lazy in torch); the ONLY torch call on it
            torch.cuda.synchronize(self.device)
            if float(warm[0]) != float(self.world):
                raise RuntimeError(f"RCCL warm-up all-reduce returned {float(warm[0])}, expected {self.world}")
            backend = pg._get_backend(self.device)
            comm = int(backend._comm_ptr())
            if comm == 0:
                raise RuntimeError("ProcessGroupNCCL._comm_ptr() returned a null communicator")
            n = ctypes.c_int(-1)
            self._check(self.lib.ncclCommCount(ctypes.c_void_p(comm), ctypes.byref(n)), "ncclCommCount")
            if n.value != self.world:
                raise RuntimeError(f"communicator spans {n.value} ranks, process group {self.world}")
            self.groups.append(pg)
            self.comms.append(comm)
        self.small, self.bulk = self.comms
        self.side = torch.cuda.Stream(device=self.device)
        self._forked = False
        self.calls = 0

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.ncclGetErrorString(rc).decode()} ({rc})")

    def _issue(self, t: torch.Tensor, comm: int, stream: torch.cuda.Stream):
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPE:
            raise ValueError(f"all_reduce needs a contiguous CUDA tensor of {list(_DTYPE)}, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
        p = ctypes.c_void_p(t.data_ptr())
        self._check(self.lib.ncclAllReduce(p, p, t.numel(), _DTYPE[t.dtype], NCCL_SUM, ctypes.c_void_p(comm),
                                           ctypes.c_void_p(stream.cuda_stream)), "ncclAllReduce")
        self.calls += 1

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        self._issue(t, self.small, torch.cuda.current_stream(self.device))
        return t

    def all_reduce_async_(self, t: torch.Tensor):
        """The tensor must stay alive until join() (the gradient arena does)."""
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)                            # the bucket's gradients are final on the compute stream
        self._issue(t, self.bulk, self.side)
        self._forked = True

    def join(self):
        if self._forked:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self._forked = False


_ACTIVE: List[Optional[RcclStreams]] = [None]


def install(device: Optional[torch.device] = None) -> RcclStreams:
    """Builds the two communicators (collective call: every rank, once, before any capture) and routes phnet_amd.parallel's
    data-path collectives through them."""
    if _ACTIVE[0] is None:
        _ACTIVE[0] = RcclStreams(device)
    return _ACTIVE[0]


def installed() -> Optional[RcclStreams]:
    return _ACTIVE[0]


def uninstall():
    _ACTIVE[0] = None
