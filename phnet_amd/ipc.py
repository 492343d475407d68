"""One-shot all-reduce of the small messages of the data-parallel step over peer-mapped buffers (csrc/ipc_allreduce.hip).

The SyncBatchNorm statistic exchanges (trainOL.py:141 nn.SyncBatchNorm; 2C+1 doubles forward, 2C floats backward, <= 8 KB) are
72 dependent collectives per step on the critical path.  `OneShotAllReduce` sets up, once, an exchange buffer per rank that
every peer maps through a hipIpc handle (handles travel through torch.distributed's object collectives - any backend), and
then reduces a message with ONE kernel launch per rank: payload written straight into every peer's buffer over xGMI, local
poll, contributions added in rank order (bit-identical sums on all ranks).  Capturable in a hipGraph (the sequence number lives
on the device).

Selected at run time: `phnet_amd.parallel` routes all-reduces of at most `max_bytes` through it once `install()` has run
(bench.py --small-allreduce ipc, or PHNET_SMALL_ALLREDUCE=ipc); the default stays stock RCCL (phnet_amd/rccl.py) until the
two have been measured against each other on a multi-GPU node - no such node was available to this build."""
import ctypes
from typing import List, Optional

import torch
import torch.distributed as dist

from ._lib import check, lib

_DTYPES = {torch.float32: 0, torch.float64: 1}


class OneShotAllReduce:
    def __init__(self, max_bytes: int = 16384, group=None, device: Optional[torch.device] = None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("OneShotAllReduce needs an initialised torch.distributed group (the handles travel through it)")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("OneShotAllReduce must be built before any capture (allocation and handle exchange are eager)")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.max_bytes = int(max_bytes)
        self.cap = self.max_bytes // 4                                   # granules per rank row
        nbytes = int(lib().phnet_ipc_buffer_bytes(self.world, self.max_bytes))
        ptr = ctypes.c_void_p()
        check(lib().phnet_ipc_alloc(nbytes, ctypes.byref(ptr)), "phnet_ipc_alloc")
        self.local = ptr.value
        handle = (ctypes.c_ubyte * 64)()
        check(lib().phnet_ipc_get_handle(self.local, handle), "phnet_ipc_get_handle")
        mine = (bytes(handle), self.device.index if self.device.index is not None else 0)
        everyone: List = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)             # host channel: any backend
        self.mapped, table = [], []
        for r, (h, _dev) in enumerate(everyone):
            if r == self.rank:
                table.append(self.local)
                continue
            p = ctypes.c_void_p()
            buf = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            check(lib().phnet_ipc_open_handle(buf, ctypes.byref(p)), "phnet_ipc_open_handle")
            self.mapped.append(p.value)
            table.append(p.value)
        self.peers = torch.tensor(table, dtype=torch.int64).to(self.device)          # device array of the mapped pointers
        self.ctrl = torch.tensor([1, 0], dtype=torch.int32).to(self.device)         # {sequence, error}
        torch.cuda.synchronize(self.device)
        dist.barrier(group=group)                                        # every buffer is zero-filled and mapped before the first use
        self.calls = 0

    def applies(self, t: torch.Tensor) -> bool:
        return t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES and 0 < t.numel() * t.element_size() <= self.max_bytes

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        if not self.applies(t):
            raise ValueError(f"one-shot all-reduce takes contiguous CUDA float32 / float64 tensors of at most {self.max_bytes} bytes")
        check(lib().phnet_oneshot_allreduce(t.data_ptr(), t.numel(), _DTYPES[t.dtype], self.peers.data_ptr(), self.rank, self.world,
                                            self.cap, self.ctrl.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
              "phnet_oneshot_allreduce")
        self.calls += 1
        return t

    def error(self) -> bool:
        """True when a peer's payload did not arrive within the kernel's bounded poll (host read: not under capture)."""
        return bool(int(self.ctrl[1].item()))

    def close(self):
        for p in self.mapped:
            lib().phnet_ipc_close_handle(p)
        self.mapped = []
        if self.local:
            lib().phnet_ipc_free(self.local)
            self.local = None


_ACTIVE: List[Optional[OneShotAllReduce]] = [None]


def install(max_bytes: int = 16384) -> OneShotAllReduce:
    """Collective: every rank calls it once, before any capture."""
    if _ACTIVE[0] is None:
        _ACTIVE[0] = OneShotAllReduce(max_bytes)
    return _ACTIVE[0]


def installed() -> Optional[OneShotAllReduce]:
    return _ACTIVE[0]


def uninstall():
    if _ACTIVE[0] is not None:
        _ACTIVE[0].close()
    _ACTIVE[0] = None
