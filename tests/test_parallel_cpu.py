"""N>1 host logic on CPU with the gloo backend, world_size 2 (the GPU path uses the same functions over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phnet_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    parallel.init_from_env("gloo")
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _stats_job(rank, world):
    torch.manual_seed(0)
    full = torch.randn(7, 16, 5, 3) * 2 + 1                      # union batch [N,C,H,W]
    mine = full[rank::world]
    x = mine.permute(1, 0, 2, 3).reshape(16, -1)
    m, v, n = parallel.merge_batch_statistics(x.mean(1), x.var(1, unbiased=False), x.shape[1])
    ref = full.permute(1, 0, 2, 3).reshape(16, -1)
    return (float((m - ref.mean(1)).abs().max()), float((v - ref.var(1, unbiased=False)).abs().max()), n, ref.shape[1])


def test_merged_batch_statistics_equal_union_batch():
    for dm, dv, n, nref in _run(_stats_job):
        assert dm < 1e-6 and dv < 1e-5 and n == nref


def _grad_job(rank, world):
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.zeros(s)) for s in [(3, 5), (7,), (2, 2, 2), (1000,)]]
    allg = [[torch.randn(p.shape, generator=torch.Generator().manual_seed(10 * r + i)) for i, p in enumerate(ps)] for r in range(world)]
    for p, g in zip(ps, allg[rank]):
        p.grad = g.clone()
    n = parallel.average_gradients_(ps, bucket_bytes=64)
    err = max(float((p.grad - sum(allg[r][i] for r in range(world)) / world).abs().max()) for i, p in enumerate(ps))
    return n, err


def test_bucketed_gradient_average():
    for n, err in _run(_grad_job):
        assert n >= 2 and err < 1e-6


def test_shard_indices_partition_like_distributed_sampler():
    for n, world in [(10, 2), (7, 2), (5, 4), (16, 8), (3, 8)]:
        parts = [parallel.shard_indices(n, r, world) for r in range(world)]
        assert len({len(p) for p in parts}) == 1                              # equal work per rank
        flat = [i for p in parts for i in p]
        assert set(flat) == set(range(n)) or n < world
        assert len(flat) == ((n + world - 1) // world) * world
        assert all(p[0] == r % n for r, p in enumerate(parts))
    a = parallel.shard_indices(100, 1, 4, shuffle_seed=3)
    b = parallel.shard_indices(100, 1, 4, shuffle_seed=3)
    assert a == b and a != parallel.shard_indices(100, 1, 4, shuffle_seed=4)


def test_single_process_is_a_no_op():
    m, v, n = parallel.merge_batch_statistics(torch.ones(4), torch.full((4,), 2.0), 10)
    assert torch.allclose(m, torch.ones(4)) and torch.allclose(v, torch.full((4,), 2.0)) and n == 10
    assert parallel.average_gradients_([torch.nn.Parameter(torch.zeros(2))]) == 0


# ---- round 2: the collectives of the overlapped data-parallel step (phnet_amd/parallel.py, tests/dp_workers.py) ---------------
def test_allreduce_flat_averages_the_arena():
    from tests import dp_workers as W
    for n, err in W.run(W.cpu_allreduce_flat):
        assert n == 4 and err < 1e-4


def test_bucket_reducer_sums_every_element_once_whatever_the_issue_order():
    from tests import dp_workers as W
    for per_pattern in W.run(W.cpu_bucket_reducer):
        for err, seen, pending, issued in per_pattern:
            assert err == 0.0 and pending == 0 and issued == []
            assert seen == 3 + 1                       # 3 non-empty buckets + the wait, each through run_collective


def test_sync_batchnorm_collectives_reproduce_union_batch_statistics_and_gradients():
    from tests import dp_workers as W
    for dm, dv, ddx, n in W.run(W.cpu_sync_statistics):
        assert dm < 1e-12 and dv < 1e-12 and ddx < 1e-6 and n == 7 * 15


def test_backward_ordered_arena_and_bucket_bounds():
    """FlatAdamW.for_model lays the decayed parameters out in the order the backward finishes them and cuts 4 buckets:
    lane head | neck + layer4 | layer3 | everything else (layer2, layer1, stem, all 1-D parameters, padding)."""
    import torch
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    from phnet_amd.optim import FlatAdamW, model_part, no_decay
    from phnet_amd.trunk import PARTS
    cfg = make_cfg(img_h=64, img_w=160, arch="resnet18")
    model = RouterOL(cfg, Criterion4OL(cfg))
    opt, arena = FlatAdamW.for_model(model)
    try:
        names = {id(p): n for n, p in model.named_parameters()}
        parts = [model_part(names[id(p)]) for p in arena.params]
        dims = [not no_decay(names[id(p)], p) for p in arena.params]
        n_dec = sum(dims)
        assert all(dims[:n_dec]) and not any(dims[n_dec:])                                  # decayed first (FlatAdamW contract)
        assert parts[:n_dec] == sorted(parts[:n_dec]) and parts[n_dec:] == sorted(parts[n_dec:])
        b = arena.bucket_bounds
        assert b[0] == 0 and b[-1] == arena.flat.numel() and b == sorted(b) and len(b) == 5
        for p in arena.params[:n_dec]:
            off, n = arena.offsets[id(p)]
            bucket = max(i for i in range(4) if b[i] <= off)
            assert off + n <= b[bucket + 1]                                                 # no parameter straddles a bucket
            want = {0: 0, 1: 1, 2: 1, 3: 2}.get(model_part(names[id(p)]), 3)
            assert bucket == want, (names[id(p)], bucket, want)
        assert set(arena.bucket_of_part) <= set(PARTS) and sorted(arena.bucket_of_part.values()) == [0, 1, 2, 3]
        assert opt.n_decay == sum(p.numel() for p in arena.params[:n_dec])
        head_share = b[1] / arena.numel
        assert head_share > 0.5                                                            # the bucket that hides behind the trunk backward
    finally:
        arena.release()


def test_run_collective_default_and_hook():
    from phnet_amd import parallel
    calls = []
    assert parallel.run_collective(lambda: calls.append("a") or 7) == 7
    parallel._RUNNER = lambda fn: ("hooked", fn())
    try:
        assert parallel.run_collective(lambda: 5) == ("hooked", 5)
    finally:
        parallel._RUNNER = None
    assert calls == ["a"]


def _reference_groups(model):
    """libs/utils/optimizer.py:41-55 set_weight_decay, restated: named_parameters() order, no decay on 1-D or `*.bias`."""
    has, no = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no if (len(p.shape) == 1 or name.endswith(".bias")) else has).append(p)
    return [{"params": has}, {"params": no, "weight_decay": 0.0}]


def test_flat_adamw_checkpoints_in_the_reference_numbering_on_the_real_model():
    """trainOL.py:128,182: optimizer.state_dict() / load_state_dict() of the reference's optim.AdamW, built by build_optimizer
    with set_weight_decay's two groups in named_parameters() order.  FlatAdamW lays the same parameters out in BACKWARD order
    (lane head first) - the checkpoint numbering must not follow the arena: group sizes, the (64,36) LayerNorm biases in the
    undecayed group, moments landing on the right parameters in both directions.  (Host logic only: no kernel runs.)"""
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    from phnet_amd.optim import FlatAdamW
    cfg = make_cfg(img_h=64, img_w=160, arch="resnet18")
    torch.manual_seed(0)
    ref = RouterOL(cfg, Criterion4OL(cfg))
    ours = RouterOL(cfg, Criterion4OL(cfg))
    ours.load_state_dict(ref.state_dict())
    topt = torch.optim.AdamW(_reference_groups(ref), lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-4)
    for i, p in enumerate(ref.parameters()):                      # a different recognisable gradient for every parameter
        p.grad = torch.full_like(p, 1e-3 * (i + 1))
    topt.step()
    tsd = topt.state_dict()
    fopt, arena = FlatAdamW.for_model(ours, lr=1.0, weight_decay=0.5)
    try:
        assert [len(g["params"]) for g in fopt.param_groups] == [len(g["params"]) for g in topt.param_groups]
        two_dim_bias = [n for n, p in ours.named_parameters() if p.dim() == 2 and n.endswith(".bias")]
        assert len(two_dim_bias) == 27                               # detNet.router.*: LayerNorm([64,36]) biases
        names = {id(p): n for n, p in ours.named_parameters()}
        undecayed = {names[id(p)] for p in fopt.param_groups[1]["params"]}
        assert set(two_dim_bias) <= undecayed
        for p in fopt.param_groups[1]["params"]:
            assert arena.offsets[id(p)][0] >= fopt.n_decay            # what the kernel decays = the first group, nothing else
        fopt.load_state_dict(tsd)
        assert fopt.param_groups[0]["lr"] == 5e-4 and fopt.param_groups[0]["weight_decay"] == 5e-4 and int(fopt.step_count) == 1
        for i, p in enumerate(ours.parameters()):
            off, n = arena.offsets[id(p)]
            g = 1e-3 * (i + 1)
            assert torch.allclose(fopt.exp_avg[off:off + n], torch.full((n,), 0.1 * g), rtol=1e-6), names[id(p)]
            assert torch.allclose(fopt.exp_avg_sq[off:off + n], torch.full((n,), 1e-3 * g * g), rtol=1e-5), names[id(p)]
        # and back: our checkpoint into a fresh reference-style optim.AdamW
        fsd = fopt.state_dict()
        assert [g["params"] for g in fsd["param_groups"]] == [g["params"] for g in tsd["param_groups"]]
        topt2 = torch.optim.AdamW(_reference_groups(ref), lr=1.0)
        topt2.load_state_dict(fsd)
        for k, st in tsd["state"].items():
            assert torch.equal(topt2.state_dict()["state"][k]["exp_avg"], st["exp_avg"])
            assert torch.equal(topt2.state_dict()["state"][k]["exp_avg_sq"], st["exp_avg_sq"])
    finally:
        arena.release()


def test_bench_gpus_n_launches_n_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must run TWO ranks: the script turns into a launcher before torch is
    imported, starts the ranks with torch.distributed.run (the reference's launch, command.sh: torchrun --nproc_per_node=4
    trainOL.py) and relays rank 0's JSON line, whose n_gpus is what the process group counted.  --rendezvous-only stops after
    the rendezvous (no GPU here); a --gpus / WORLD_SIZE mismatch is refused."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--rendezvous-only"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                    # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_env"] == 2 and d["backend"] == "gloo", d
    assert "launcher: starting 2 ranks" in r.stderr
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--rendezvous-only"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)
