"""Data-parallel host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  The path shards by clip (reference: DistributedSampler + DDP + SyncBatchNorm,
trainOL.py:34,83,141-146); the only data-path collectives are the gradient all-reduce (DDP buckets) and the
SyncBatchNorm statistic exchange implemented here.

Everything in this file works on tensors of either device, so the N>1 logic is testable with gloo on CPU.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the default group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"), init_method="env://")
    return rank, world, local_rank


def shard_indices(n_units: int, rank: int, world: int, shuffle_seed: Optional[int] = None) -> List[int]:
    """Clip indices of this rank: rank, rank+world, ... over a list padded by wrap-around to a multiple of `world`
    (the partition torch's DistributedSampler produces, trainOL.py:83)."""
    order = list(range(n_units))
    if shuffle_seed is not None:
        g = torch.Generator()
        g.manual_seed(shuffle_seed)
        order = torch.randperm(n_units, generator=g).tolist()
    total = ((n_units + world - 1) // world) * world
    pad = total - n_units
    if pad:
        order += (order * ((pad + n_units - 1) // max(n_units, 1) + 1))[:pad]
    return order[rank:total:world]


def active(group=None) -> bool:
    """True when collectives have to be issued.  PHNET_FORCE_COLLECTIVES=1 also issues them in a one-rank group (they are
    identities there): lets a single GPU exercise the real RCCL call path, incl. its coexistence with hipGraph capture."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("PHNET_FORCE_COLLECTIVES", "0") == "1"


# Every collective of the data path goes through run_collective(): a single choke point (tests count the collectives of a
# step through it).
_RUNNER = None


def run_collective(fn):
    """fn(): issues torch.distributed call(s) IN PLACE on tensors that outlive the call."""
    if _RUNNER is not None:
        return _RUNNER(fn)
    return fn()


def merge_batch_statistics(mean: torch.Tensor, var_biased: torch.Tensor, count: int, group=None):
    """Combine per-rank (mean, biased variance, element count) of one BatchNorm layer into the statistics of the
    union batch.  One all-reduce of a [3,C]-ish fp64 buffer: (count, count*mean, count*(var+mean^2))."""
    c = mean.numel()
    buf = torch.empty(2 * c + 1, dtype=torch.float64, device=mean.device)
    buf[0] = float(count)
    buf[1:1 + c] = mean.double() * count
    buf[1 + c:] = (var_biased.double() + mean.double() ** 2) * count
    if active(group):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    total = buf[0]
    g_mean = buf[1:1 + c] / total
    g_var = (buf[1 + c:] / total - g_mean ** 2).clamp_min(0.0)
    return g_mean.float(), g_var.float(), int(round(float(total)))


def _direct(t: torch.Tensor, group=None):
    """The RcclStreams transport (phnet_amd/rccl.py) when it is installed and applies: collectives of the data path then go
    straight into RCCL on our streams and no torch Work object exists (required under hipGraph capture: rccl.py says why)."""
    if group is not None or not t.is_cuda:
        return None
    from . import rccl
    return rccl.installed()


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM over the ranks (blocking for the STREAM, not the host)."""
    if active(group):
        if group is None and t.is_cuda:
            from . import ipc
            one = ipc.installed()                    # small messages over peer-mapped buffers, when selected (phnet_amd/ipc.py)
            if one is not None and one.applies(t):
                run_collective(lambda: one.all_reduce_(t))
                return t
        tr = _direct(t, group)
        if tr is not None:
            run_collective(lambda: tr.all_reduce_(t))
        else:
            _no_torch_collective_under_capture(t)
            run_collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group))
    return t


def _no_torch_collective_under_capture(t: torch.Tensor):
    """A torch.distributed collective under capture pulls the process group's internal stream into the capture; the group's
    watchdog thread then dies on its next poll of any not-yet-retired eager Work (hipErrorCapturedEvent, rccl.py).  Refused
    loudly instead of racing: install phnet_amd.rccl first."""
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("torch.distributed collective under hipGraph capture: call phnet_amd.rccl.install() before capturing "
                           "a data-parallel step (phnet_amd/rccl.py explains the watchdog abort this prevents)")


def average_gradients_(params, bucket_bytes: int = 96 << 20, group=None) -> int:
    """Bucketed gradient averaging for callers that do not use DDP: gradients are packed into flat fp32 buckets of
    ~bucket_bytes (few large collectives: xGMI links are point-to-point, per-link bound), all-reduced asynchronously,
    then scattered back.  Returns the number of collectives issued."""
    if not active(group):
        return 0
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        cur.append(g)
        size += g.numel() * 4
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    work = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for w, flat, b in work:
        w.wait()
        flat.div_(world)
        off = 0
        for g in b:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(buckets)


def allreduce_flat_(flat: torch.Tensor, chunks: int = 4, group=None) -> int:
    """Average a flat gradient arena across ranks with `chunks` large asynchronous all-reduces (fewer, larger
    collectives: xGMI is point-to-point, ring steps are per-link bound).  Returns the number of collectives."""
    if not active(group):
        return 0
    world = dist.get_world_size(group)
    n = flat.numel()
    step = (n + chunks - 1) // chunks
    work = [dist.all_reduce(flat[i:min(n, i + step)], op=dist.ReduceOp.SUM, group=group, async_op=True)
            for i in range(0, n, step)]
    for w in work:
        w.wait()
    flat.div_(world)
    return len(work)


class BucketReducer:
    """Gradient averaging over a flat arena whose element order follows the BACKWARD schedule (phnet_amd/arena.py:
    backward_order): bucket i is the range [bounds[i], bounds[i+1]) and `issue(i)` starts its asynchronous all-reduce as soon
    as the backward has written its last element - bucket 0 (the lane head: 77 % of the parameters of ResNet-34) goes out when
    the trunk's backward starts and hides behind it; the remaining buckets follow the trunk stages.  Few, large collectives:
    xGMI links are point-to-point, a ring step is per-link bound, so 4 messages of tens of MB beat DDP's 25 MB buckets.
    SUM reduction: the caller pre-divides the loss by the world size (so that the reduced arena IS the mean gradient)."""

    def __init__(self, flat: torch.Tensor, bounds: List[int], group=None):
        assert bounds[0] == 0 and bounds[-1] == flat.numel() and all(a <= b for a, b in zip(bounds, bounds[1:])), bounds
        self.flat, self.bounds, self.group = flat, list(bounds), group
        self.work = []
        self.issued = []
        self.direct = None

    @property
    def n_buckets(self) -> int:
        return len(self.bounds) - 1

    def issue(self, i: int):
        lo, hi = self.bounds[i], self.bounds[i + 1]
        self.issued.append(i)
        if hi <= lo or not active(self.group):
            return

        tr = _direct(self.flat, self.group)
        if tr is not None:
            self.direct = tr
            run_collective(lambda: tr.all_reduce_async_(self.flat[lo:hi]))
            return

        def go():
            _no_torch_collective_under_capture(self.flat)
            self.work.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        run_collective(go)

    def finish(self):
        """Issues whatever has not been issued yet and makes the current stream wait for every bucket."""
        for i in range(self.n_buckets):
            if i not in self.issued:
                self.issue(i)
        self.issued = []
        if not active(self.group):
            return

        def wait():
            for w in self.work:
                w.wait()
            self.work = []
            if self.direct is not None:
                self.direct.join()
                self.direct = None
        run_collective(wait)
