"""Worker bodies of the multi-process data-parallel tests (importable, so that torch.multiprocessing.spawn can pickle them).
CPU jobs use gloo on CPU tensors; GPU jobs are 2 ranks sharing cuda:0 with gloo (functional rehearsal of the N > 1 path on
the one-GPU box; on a multi-GPU node the same code runs over RCCL) or a one-rank RCCL group."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, fn, ret, env):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    torch.set_num_threads(2)
    if backend != "gloo" or fn.__name__.startswith("gpu_"):
        torch.cuda.set_device(0)
    dist.init_process_group(backend=backend, init_method="env://")
    try:
        if os.environ.get("PHNET_SMALL_ALLREDUCE") == "ipc":
            from phnet_amd import ipc
            ipc.install()                                     # SyncBatchNorm exchanges over peer-mapped buffers (phnet_amd/ipc.py)
        ret[rank] = fn(rank, world)
        if os.environ.get("PHNET_SMALL_ALLREDUCE") == "ipc":
            from phnet_amd import ipc
            ret[rank] = dict(ret[rank], ipc_calls=ipc.installed().calls, ipc_error=ipc.installed().error())
            dist.barrier()
            ipc.uninstall()
    finally:
        dist.destroy_process_group()


def run(fn, world=2, backend="gloo", env=None):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, free_port(), backend, fn, ret, env or {}), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


# ---------------------------------------------------------------------------------------------------- CPU jobs (gloo)
def cpu_allreduce_flat(rank, world):
    from phnet_amd import parallel
    flat = torch.arange(1003, dtype=torch.float32) * (rank + 1)
    n = parallel.allreduce_flat_(flat, chunks=4)
    want = torch.arange(1003, dtype=torch.float32) * sum(r + 1 for r in range(world)) / world
    return n, float((flat - want).abs().max())


def cpu_bucket_reducer(rank, world):
    """Buckets issued in backward order, out of order and partially: the result is the SUM over ranks of every element,
    exactly once, whatever the issue pattern; a recording runner sees every collective."""
    from phnet_amd import parallel
    out = []
    for pattern in ([0, 1, 2, 3], [2, 0], [], [3, 3 - 3]):
        flat = (torch.arange(1000, dtype=torch.float32) + 1) * (rank + 1)
        red = parallel.BucketReducer(flat, [0, 700, 700, 900, 1000])           # bucket 1 is empty
        seen = []
        parallel._RUNNER = lambda fn: (seen.append(1), fn())[1]
        try:
            for i in pattern:
                red.issue(i)
            red.finish()
        finally:
            parallel._RUNNER = None
        want = (torch.arange(1000, dtype=torch.float32) + 1) * sum(r + 1 for r in range(world))
        out.append((float((flat - want).abs().max()), len(seen), len(red.work), list(red.issued)))
    return out


def cpu_sync_statistics(rank, world):
    """The arithmetic of the device-resident SyncBatchNorm (csrc/norm.hip bn_local_sums / bn_finalize_sums /
    bn_bwd_means) spelled in torch on CPU with the SAME collectives (one fp64 all-reduce of 2C+1, one f32 all-reduce of
    2C) against BatchNorm over the union batch, with UNEQUAL shard sizes."""
    from phnet_amd import parallel
    torch.manual_seed(0)
    full = torch.randn(7, 16, 5, 3, dtype=torch.float64) * 2 + 1
    full.requires_grad_(True)
    gam, bet = torch.rand(16, dtype=torch.float64) + 0.5, torch.randn(16, dtype=torch.float64)
    y = torch.nn.functional.batch_norm(full, None, None, gam, bet, True, 0.1, 1e-5)
    gy = torch.randn_like(y)
    y.backward(gy)
    lo, hi = (0, 3) if rank == 0 else (3, 7)
    x = full.detach()[lo:hi].permute(1, 0, 2, 3).reshape(16, -1)
    g = gy[lo:hi].permute(1, 0, 2, 3).reshape(16, -1)
    sums = torch.cat([x.sum(1), (x * x).sum(1), torch.tensor([float(x.shape[1])], dtype=torch.float64)])
    parallel.allreduce_sum_(sums)
    n = sums[32]
    mean = sums[:16] / n
    var = (sums[16:32] / n - mean ** 2).clamp_min(0)
    invstd = 1 / torch.sqrt(var + 1e-5)
    xhat = (x - mean[:, None]) * invstd[:, None]
    bsum = torch.stack([(g * xhat).sum(1), g.sum(1)]).float()
    parallel.allreduce_sum_(bsum)
    c2, c1 = bsum[0].double() / n, bsum[1].double() / n
    dx = (gam * invstd)[:, None] * (g - c1[:, None] - xhat * c2[:, None])
    ref_dx = full.grad[lo:hi].permute(1, 0, 2, 3).reshape(16, -1)
    ref = full.detach().permute(1, 0, 2, 3).reshape(16, -1)
    return (float((mean - ref.mean(1)).abs().max()), float((var - ref.var(1, unbiased=False)).abs().max()),
            float((dx - ref_dx).abs().max()), int(n))


# ---------------------------------------------------------------------------------------------------- GPU jobs
def _tiny_model(sync_bn: bool):
    from oracle import phnet_cpu as O
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    from tests import synth
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
    cfg = make_cfg(img_h=g.img_h, img_w=g.img_w, arch=g.arch)
    model = RouterOL(cfg, Criterion4OL(cfg))
    model.load_state_dict(synth.make_state(g), strict=True)
    for m in model.detNet.transformer_Dec.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    if sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    return g, model.cuda().train()


def gpu_two_rank_step_vs_pair_fixture(rank, world):
    """BASELINE configs[2] semantics on two real ranks: one clip per rank, SyncBatchNorm statistics over both, gradient
    buckets issued in backward order (eager: gloo collectives cannot be stream-captured)."""
    import json
    from phnet_amd import parallel
    from phnet_amd.graphed import data_parallel_step
    from phnet_amd.optim import FlatAdamW
    from tests import synth
    g, model = _tiny_model(sync_bn=True)
    T = 3
    frames = synth.make_clip(g, T, seed=3407 + rank).cuda()
    lanes = synth.make_targets(g, T).cuda()
    rec = {"matched": [], "loss": []}

    def record(o, gt, diff, m, l):
        rec["matched"].append([[i for i in x.cpu().tolist() if i >= 0] for x in m])
        rec["loss"].append(float(l.detach()))
    undo = synth.observe_criterion(model.criterion, record)
    opt, arena = FlatAdamW.for_model(model, lr=0.0, weight_decay=0.0)                 # lr 0: the step leaves the weights alone
    reducer = parallel.BucketReducer(arena.flat, arena.bucket_bounds)
    parts, n_coll = [], [0]
    parallel._RUNNER = lambda fn: (n_coll.__setitem__(0, n_coll[0] + 1), fn())[1]
    try:
        loss = data_parallel_step(model, arena, reducer, opt, frames, lanes, 1.0, stage_done=parts.append)
    finally:
        parallel._RUNNER = None
    torch.cuda.synchronize()
    undo()
    names = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grad_names_resnet18.json")))
    params = dict(model.named_parameters())
    return {"loss": float(loss), "frame_loss": rec["loss"], "matched": rec["matched"], "parts": parts, "collectives": n_coll[0],
            "grad_norm": [float(params[k].grad.double().norm()) for k in names],
            "bn1_mean": model.backbone.backbone.model.bn1.running_mean.cpu().numpy(),
            "bn1_var": model.backbone.backbone.model.bn1.running_var.cpu().numpy(),
            "bounds": list(arena.bucket_bounds)}


def gpu_two_ranks_two_clips_each_vs_quad_fixture(rank, world):
    """BASELINE configs[2] as ONE workload (8 clips over 4 ranks = 2 clips per GPU, trainOL.py:141-146,205-212), scaled to the
    one-GPU box: 2 ranks x 2 clips per rank.  Inside a rank the two clips run batched ([B,T,3,H,W]: lane head batched across
    the clips, local BatchNorm sums over B*T frames); across the ranks SyncBatchNorm merges the sums - statistics over all four
    clips, as in the fixture (the reference's trunk run once over the four clips' frames)."""
    import json
    from phnet_amd import parallel
    from phnet_amd.graphed import data_parallel_step
    from phnet_amd.optim import FlatAdamW
    from tests import synth
    g, model = _tiny_model(sync_bn=True)
    T, B = 2, 2
    seeds = (3407, 3408, 3409, 3410)[rank * B:(rank + 1) * B]
    frames = torch.stack([synth.make_clip(g, T, seed=s) for s in seeds]).cuda()
    lanes = torch.stack([synth.make_targets(g, T)] * B).cuda()
    rec = {"matched": [], "loss": []}

    def record(o, gt, diff, m, l):
        rec["matched"].append([[i for i in x.cpu().tolist() if i >= 0] for x in m])
        rec["loss"].append(float(l.detach()))
    undo = synth.observe_criterion(model.criterion, record)
    opt, arena = FlatAdamW.for_model(model, lr=0.0, weight_decay=0.0)
    reducer = parallel.BucketReducer(arena.flat, arena.bucket_bounds)
    n_coll = [0]
    parallel._RUNNER = lambda fn: (n_coll.__setitem__(0, n_coll[0] + 1), fn())[1]
    try:
        loss = data_parallel_step(model, arena, reducer, opt, frames, lanes, 1.0)
    finally:
        parallel._RUNNER = None
    torch.cuda.synchronize()
    undo()
    names = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grad_names_resnet18.json")))
    params = dict(model.named_parameters())
    return {"loss": float(loss), "frame_loss": rec["loss"], "matched": rec["matched"], "collectives": n_coll[0],
            "grad_norm": [float(params[k].grad.double().norm()) for k in names],
            "bn1_mean": model.backbone.backbone.model.bn1.running_mean.cpu().numpy(),
            "bn1_var": model.backbone.backbone.model.bn1.running_var.cpu().numpy()}


def gpu_whole_step_graph_with_rccl_inside(rank, world):
    """One-rank RCCL group, collectives forced on: the data-parallel step (staged trunk, SyncBatchNorm exchanges, bucket
    all-reduces, AdamW) eagerly, then captured as ONE hipGraph with the RCCL collectives inside and replayed twice."""
    from phnet_amd import parallel
    from phnet_amd.graphed import GraphedTrainStep, data_parallel_step
    from phnet_amd.optim import FlatAdamW
    from tests import synth
    g, model = _tiny_model(sync_bn=True)
    T = 3
    frames, lanes = synth.make_clip(g, T, seed=3407).cuda(), synth.make_targets(g, T).cuda()
    opt, arena = FlatAdamW.for_model(model, lr=0.0, weight_decay=0.0)
    reducer = parallel.BucketReducer(arena.flat, arena.bucket_bounds)
    n_coll = [0]
    parallel._RUNNER = lambda fn: (n_coll.__setitem__(0, n_coll[0] + 1), fn())[1]
    try:
        loss = data_parallel_step(model, arena, reducer, opt, frames, lanes, 1.0)
    finally:
        parallel._RUNNER = None
    torch.cuda.synchronize()
    eager = arena.flat.clone()
    rv = model.backbone.backbone.model.bn1.running_var.clone()
    step = GraphedTrainStep(model, opt, frames, lanes, loss_divisor=1.0, warmup=0, arena=arena, reducer=reducer)
    l1 = float(step(frames)); torch.cuda.synchronize()
    g1 = arena.flat.clone()
    rv1 = model.backbone.backbone.model.bn1.running_var.clone()
    l2 = float(step(frames)); torch.cuda.synchronize()
    return {"loss": float(loss), "collectives": n_coll[0], "replay_loss": [l1, l2], "step_count": int(opt.step_count),
            "replay_grad_err": float((g1 - eager).abs().max() / eager.abs().max()),
            "replay_repeat_err": float((arena.flat - g1).abs().max() / g1.abs().max()),
            "running_var_moves": bool((rv1 != rv).any())}


def gpu_ddp_syncbn_wrap(rank, world):
    """The reference's wrapping, unchanged (trainOL.py:141-146): convert_sync_batchnorm + DistributedDataParallel(
    find_unused_parameters=True) around the HIP-backed model; one rank per process group on the GPU box (RCCL), collectives
    forced on so that the SyncBatchNorm exchange really goes through the process group.  Compared with the bare model."""
    from tests import synth
    g, ref = _tiny_model(sync_bn=False)
    T = 3
    frames, lanes = synth.make_clip(g, T, seed=3).cuda(), synth.make_targets(g, T).cuda()
    lr = ref({"frame": frames, "lanes": lanes})
    lr.backward()
    g2, model = _tiny_model(sync_bn=True)
    assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    ld = ddp({"frame": frames, "lanes": lanes})
    ld.backward()
    torch.cuda.synchronize()
    worst, worst_gate, names = 0.0, 0.0, []
    for (k, a), (_, b) in zip(ref.named_parameters(), model.named_parameters()):
        assert b.grad is not None, k
        if ".DWNets." in k and k.endswith((".0.bias", ".3.bias")):
            continue        # depth-wise conv bias in front of a LayerNorm over the whole plane: true gradient 0, fp32 noise only
        e = float((a.grad - b.grad).abs().max() / (a.grad.abs().max() + 1e-6))
        names.append((round(e, 5), k))
        if k.startswith("detNet.router."):
            worst_gate = max(worst_gate, e)        # (one anchor at the gate's ReLU threshold may flip between the two runs)
        else:
            worst = max(worst, e)
    print("worst gradient mismatches:", sorted(names)[-6:], flush=True)
    rm = float((ref.backbone.backbone.model.bn1.running_var - model.backbone.backbone.model.bn1.running_var).abs().max())
    return {"loss_ref": float(lr), "loss_ddp": float(ld), "worst_grad_rel": worst, "worst_gate_grad_rel": worst_gate, "running_var_err": rm}


def gpu_rccl_inside_capture(rank, world):
    """phnet_amd.rccl.RcclStreams: raw ncclAllReduce calls on our streams (compute stream + side stream with fork / join) captured
    in a hipGraph and replayed, next to EAGER torch collectives on the default group before and after the capture (their Work
    objects sit on the watchdog's list while we capture - harmless, because the default group's stream never captures); and
    phnet_amd.parallel refuses a torch collective under capture instead of racing the watchdog."""
    from phnet_amd import parallel, rccl
    t = torch.ones(1024, device="cuda")
    dist.all_reduce(t)                                            # eager torch collective: an un-retired Work when the capture starts
    tr = rccl.install()
    small = torch.ones(33, dtype=torch.float64, device="cuda")
    big = torch.ones(1 << 20, device="cuda")
    refused = False
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            small.mul_(2.0)
            parallel.allreduce_sum_(small)                        # -> tr.all_reduce_ on the capturing stream
            big.mul_(3.0)
            red = parallel.BucketReducer(big, [0, 1 << 19, 1 << 20])
            red.issue(0)                                          # -> side stream: a parallel branch of the graph
            small.add_(1.0)
            red.finish()
            big.add_(1.0)
            rccl.uninstall()
            try:
                parallel.allreduce_sum_(small)                    # torch collective under capture: must raise, not race
            except RuntimeError as e:
                refused = "under hipGraph capture" in str(e)
            rccl._ACTIVE[0] = tr
        dist.all_reduce(t)                                        # eager again, right after the capture
        g.replay(); g.replay()
        torch.cuda.synchronize()
        return {"captured": True, "refused": refused, "small": float(small[0]), "big": float(big[0]), "big_last": float(big[-1]),
                "calls": tr.calls, "eager": float(t[0])}
    except Exception as e:                                       # noqa: BLE001
        return {"captured": False, "error": f"{type(e).__name__}: {str(e)[:300]}"}
    finally:
        rccl.uninstall()


def gpu_oneshot_allreduce_two_processes(rank, world):
    """Two processes on the one card of the box (gloo carries the 64-byte handles): each maps the other's exchange buffer and
    reduces float64 / float32 messages of the SyncBatchNorm sizes with one launch per rank - eagerly, repeatedly (sequence
    numbers, both slots), and from a captured hipGraph replayed three times; through phnet_amd.parallel.allreduce_sum_ as the
    SyncBatchNorm path calls it.  (Functional: on one card the peers' stores do not cross xGMI.)"""
    from phnet_amd import ipc, parallel
    one = ipc.install(max_bytes=16384)
    out = {"world": one.world}
    errs = []
    for n, dt in ((1025, torch.float64), (1024, torch.float32), (129, torch.float64), (1, torch.float32), (2048, torch.float64)):
        for rep in range(3):
            gen = torch.Generator().manual_seed(100 * n + rep)
            full = torch.randn(world, n, generator=gen, dtype=torch.float64)
            mine = full[rank].to(dt).cuda()
            parallel.allreduce_sum_(mine)
            want = full.to(dt)[0].clone()
            for r in range(1, world):
                want = want + full.to(dt)[r]                  # rank order
            errs.append(float((mine.cpu() - want).abs().max()))
    out["eager_max_err"] = max(errs)
    out["calls_eager"] = one.calls
    # captured: the sequence number advances on the device, so every replay is a fresh exchange
    x = torch.full((257,), float(rank + 1), dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        x.mul_(2.0)
        parallel.allreduce_sum_(x)
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append(float(x[0]))
    out["replays"] = vals
    big = torch.ones(1 << 16, device="cuda")                  # above max_bytes: not taken by the one-shot path
    out["big_applies"] = one.applies(big)
    out["error_flag"] = one.error()
    dist.barrier()
    ipc.uninstall()
    return out
