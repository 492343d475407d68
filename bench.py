#!/usr/bin/env python3
"""Headline benchmark: clips/s of the per-clip training step (forward + backward + AdamW) of the HIP-backed PHNet
model on synthetic 5-frame 3x320x800 clips, ResNet-34 + router + lane head (BASELINE.json configs[1]), one clip per
GPU per step (data parallel, weak scaling).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant GEMM kernel's achieved TFLOP/s (algorithmic FLOPs / HIP-event time of its launches in the
                 timed region) against the dense f32-input MFMA peak;
  cpu_baseline - the CPU oracle (oracle/phnet_cpu.py, a restatement of the reference's PyTorch-CPU path) timed on this
                 box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_ranks_if_asked(argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: this process becomes a LAUNCHER - it starts the N ranks
    as children (`python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`, the reference's own launch:
    command.sh `torchrun --nproc_per_node=4 trainOL.py`), relays their output (rank 0 prints the JSON line) and exits with their
    status.  Runs before torch is imported: the launcher never touches the GPU, so nothing is ever exec'ed or forked from a
    process that has initialised HIP."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args(argv)[0].gpus
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    spawn_ranks_if_asked(sys.argv[1:])

import torch
import torch.distributed as dist

F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0        # ibid., dense bf16
# what bounds a GEMM kernel per arithmetic: algorithmic (= f32-product) TFLOP/s it could reach with the matrix pipe always busy
GEMM_PEAK = {"f32": (F32_MFMA_PEAK_TFLOPS, "f32 (v_mfma_f32_32x32x2_f32)", "dense f32-input MFMA peak"),
             "bf16x3": (BF16_MFMA_PEAK_TFLOPS / 6, "bf16 (6 x v_mfma_f32_32x32x16_bf16 per 16-deep step of an f32-exact product)",
                        "dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per product of the exact three-term split"),
             "split3_bf16": (BF16_MFMA_PEAK_TFLOPS / 6, "bf16 (6 MFMAs per product, split in registers)", "dense bf16 MFMA peak / 6"),
             "split_bf16": (BF16_MFMA_PEAK_TFLOPS / 3, "bf16 (3 MFMAs per product, two-term split)", "dense bf16 MFMA peak / 3")}
DTYPE = {"f32": "f32 (f32-input MFMA)",
         "bf16x3": "f32 storage / accumulation, every product as an EXACT three-term bf16 split (6 bf16 MFMAs, f32-MFMA accuracy)",
         "split3_bf16": "f32 storage / accumulation, exact three-term bf16 split in registers",
         "split_bf16": "f32 storage / accumulation, two-term bf16 split (3 MFMAs per product; not parity-grade)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="resnet34")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--clips-per-gpu", type=int, default=1,
                    help="clips per GPU and step (default 1 = BASELINE.json configs[1], the reference's train_batch).  B > 1 runs "
                         "the lane head batched across the clips with joint BatchNorm statistics = the reference's DDP + "
                         "SyncBatchNorm over B virtual ranks on one GPU; a different workload, reported as such")
    ap.add_argument("--batched-extra", type=int, default=8,
                    help="N = 1 only: after the headline measurement, also time B clips per GPU and step in a child process and "
                         "report it under the extra key 'batched' (0 = skip)")
    ap.add_argument("--mma", default="bf16x3", choices=["f32", "split_bf16", "split3_bf16", "bf16x3"],
                    help="arithmetic of the conv / linear GEMM kernels: bf16x3 (default, the headline: f32 operands split EXACTLY "
                         "into three bf16 terms while a tile is staged into LDS, 6 bf16 MFMAs per product, f32 accumulation - the "
                         "accuracy of the f32-input MFMA), f32 (v_mfma_f32_32x32x2_f32, the round-1 headline), or the two-term "
                         "split_bf16 (3 MFMAs per product, ~4x the rounding noise: not parity-grade)")
    ap.add_argument("--split-extra", type=int, default=1,
                    help="after the headline run, also time the step on the f32-input MFMA (the round-1 arithmetic) and report it "
                         "under the extra key 'f32_mfma' (0 = skip)")
    ap.add_argument("--inference-extra", type=int, default=1,
                    help="N = 1 only: also run the inference workload of BASELINE.json configs[4] (32 clips x 5 frames, lane head + "
                         "fused decode / NMS, hipGraph) in a child process and report it under the extra key 'inference' (0 = skip)")
    ap.add_argument("--inference", action="store_true", help="run ONLY the inference workload and print its JSON (used by the child)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=5)
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step in a hipGraph (always eager for N>1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only for "
                                                      "functional rehearsals of the N>1 path on a single GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="plumbing check, no GPU needed: the ranks rendezvous over --backend, count themselves with one all-reduce, "
                         "rank 0 prints {n_gpus, ...} and everybody exits")
    ap.add_argument("--collectives", default="rccl-streams", choices=["rccl-streams", "torch"],
                    help="N > 1, nccl backend: rccl-streams (default) = raw ncclAllReduce calls on our own streams, capturable "
                         "(phnet_amd/rccl.py); torch = torch.distributed Work objects, eager step only")
    ap.add_argument("--small-allreduce", default=os.environ.get("PHNET_SMALL_ALLREDUCE", "rccl"), choices=["rccl", "ipc"],
                    help="N > 1: how the SyncBatchNorm statistic exchanges (<= 8 KB, 72 per step) travel: rccl (default) = stock RCCL "
                         "all-reduces on the compute stream; ipc = one launch per rank over hipIpc-mapped peer buffers (phnet_amd/ipc.py)")
    ap.add_argument("--tune-k-tile", type=int, default=None, help="tuning aid: phnet_tune_force_k_tile code (-5 = generic 3x3 forward / dgrad kernel)")
    ap.add_argument("--wgrad-flags", type=int, default=None, help="tuning aid: hip_ops.tune_wgrad flags (8 = generic 3x3 weight-gradient kernel)")
    ap.add_argument("--force-dp", action="store_true",
                    help="rehearsal on ONE GPU: run the N > 1 code path (SyncBatchNorm containers, bucket reducer, RCCL collectives captured "
                         "in the graph) in a one-rank process group with the collectives forced on - what the data-parallel machinery "
                         "costs per step before any link is involved")
    return ap.parse_args()


def cpu_baseline(args):
    """Times the CPU oracle's fwd+bwd on `cpu_clips` clips of the same workload (rank 0, N=1 only)."""
    from oracle import phnet_cpu as O
    from tests import synth
    g = O.Geometry(img_h=args.height, img_w=args.width, arch=args.arch)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)                       # the GPU box gives one GPU a 16-core share; more threads only thrash
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline: {args.cpu_clips} clip(s) on {cores} threads ...", file=sys.stderr, flush=True)
    sd = synth.make_state(g)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k.split(".")[-1] not in ("prior_feat_ys", "prior_ys", "priors", "priors_on_featmap"):
            v.requires_grad_(True)
    frames, lanes = synth.make_clip(g, args.frames), synth.make_targets(g, args.frames)
    O.clip_forward(sd, frames, lanes, g, training=True).backward()        # one untimed warm-up clip (thread pool, allocator, caches)
    for v in sd.values():
        if v.requires_grad:
            v.grad = None
    t0 = time.perf_counter()
    for i in range(args.cpu_clips):
        loss = O.clip_forward(sd, frames, lanes, g, training=True)
        loss.backward()
        print(f"[bench] cpu baseline clip {i + 1}/{args.cpu_clips} done at {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": args.cpu_clips / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_clips} clips of {args.frames}x3x{args.height}x{args.width} fwd+bwd (no optimizer; after one untimed warm-up clip), "
                      f"oracle/phnet_cpu.py on torch CPU fp32, {dt:.1f} s"}


def _child_bench(args, flags, what):
    """Runs this script once more in a child process (so that it cannot disturb the run above) and returns its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), *flags, "--no-cpu-baseline", "--no-kernel-timer", "--batched-extra", "0",
           "--split-extra", "0", "--inference-extra", "0", "--arch", args.arch, "--frames", str(args.frames), "--height", str(args.height),
           "--width", str(args.width)]
    try:
        print(f"[bench] extra: {what} (child process)", file=sys.stderr, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:                                             # noqa: BLE001
        print(f"[bench] extra measurement ({what}) failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return None


def batched_extra(args):
    """Not the headline: the same step with B clips per GPU (lane head batched across the clips, joint BatchNorm statistics =
    the reference's DDP + SyncBatchNorm over B virtual ranks)."""
    d = _child_bench(args, ["--clips-per-gpu", str(args.batched_extra), "--steps", "6", "--warmup", "2"],
                     f"{args.batched_extra} clips per GPU and step")
    if d is None:
        return None
    return {"clips_per_gpu": args.batched_extra, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
            "note": "same model and step with B clips per GPU and step: the per-frame chain is serial only inside a clip, so the "
                    "lane head runs B*240 rows per kernel; BatchNorm statistics over all B*T frames = the reference's DDP + "
                    "SyncBatchNorm over B ranks.  A different workload than the headline (1 clip/GPU/step)."}


def f32_extra(args):
    """Not the headline: the same workload with the GEMM kernels on the f32-input MFMA (round 1's headline arithmetic)."""
    d = _child_bench(args, ["--mma", "f32", "--steps", "10", "--warmup", "3"], "f32-input MFMA, 1 clip per GPU and step")
    if d is None:
        return None
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
            "note": "--mma f32: v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain), the headline arithmetic of round 1; same results to "
                    "rounding (both arithmetics hold the reference goldens at the same tolerances, tests/test_model_gpu.py)"}


def inference_extra(args):
    """Not the headline: BASELINE.json configs[4]."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--inference", "--mma", args.mma, "--arch", args.arch, "--frames", str(args.frames),
           "--height", str(args.height), "--width", str(args.width)]
    try:
        print("[bench] extra: inference, 32 clips x 5 frames (child process)", file=sys.stderr, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:                                             # noqa: BLE001
        print(f"[bench] extra measurement (inference) failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return None


def run_inference(args):
    """BASELINE.json configs[4]: 32 clips x T frames, eval: backbone + lane head + fused decode / NMS of every frame, one
    hipGraph replay per clip, and B clips per replay with the lane head batched across the clips.  The class heads are
    redrawn so that scores straddle conf_threshold (phnet_amd.synthetic.spread_scores_): every frame feeds ~120 candidates
    to the NMS and positives to the cross-frame memory (with random-init heads nothing is kept and the decode is empty)."""
    from phnet_amd import hip_ops
    from phnet_amd.config import make_cfg
    from phnet_amd.graphed import GraphedInference
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.synthetic import make_clip, spread_scores_
    hip_ops.set_mma_mode(args.mma)
    if args.wgrad_flags is not None:
        hip_ops.tune_wgrad(args.wgrad_flags)
    if args.tune_k_tile is not None:
        hip_ops.tune_k_tile(args.tune_k_tile)
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    clips, T, H, W = 32, args.frames, args.height, args.width
    model = RouterOL(make_cfg(img_h=H, img_w=W, arch=args.arch), None).cuda().eval()
    spread_scores_(model)
    batch = [make_clip(H, W, T, seed=i).cuda() for i in range(4)]
    g = GraphedInference(model, batch[0])
    for i in range(3):
        g(batch[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = []
    for i in range(clips):
        rows, nums, anchors = g(batch[i % 4])
        if i >= clips - 4:
            kept.append(nums.clone())
    host = model.lanes_from_device(rows, nums)           # one D2H + Lane objects for the last clip (per-clip host work)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": f"{clips} clips x {T} frames 3x{H}x{W}, {args.arch}, eval: backbone + lane head + fused decode/NMS per frame, "
                       "hipGraph; class heads redrawn so that ~half of the 240 anchors pass conf_threshold",
           "gemm_arithmetic": args.mma, "clips_per_s": round(clips / dt, 2), "frames_per_s": round(clips * T / dt, 1),
           "ms_per_clip": round(dt / clips * 1e3, 2), "lanes_last_clip": [len(x) for x in host["lane_lines"]],
           "kept_per_frame_last_4_clips": torch.stack(kept).cpu().tolist()}
    with torch.no_grad():
        lines_probe = model.infer_device(batch[0])
    del lines_probe
    for B in (8, 32):
        big = torch.stack([batch[i % 4] for i in range(B)])
        gb = GraphedInference(model, big)
        for _ in range(2):
            gb(big)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(max(1, clips // B) * 2):
            rb, nb, ab = gb(big)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - tb
        out[f"clips_per_s_batched_{B}"] = round(max(1, clips // B) * 2 * B / dtb, 2)
        out[f"kept_lanes_batched_{B}"] = int(nb.sum())
        del gb
    out["frames_per_s_batched_32"] = round(out["clips_per_s_batched_32"] * T, 1)
    print(json.dumps(out), flush=True)


def rendezvous_only(args, rank, world):
    """The launch plumbing without a GPU: every rank joins the process group, one all-reduce counts them."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, init_method="env://")
        one = torch.ones(1)
        dist.all_reduce(one)
        seen = int(one.item())
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": seen, "world_size_env": world, "backend": args.backend if world > 1 else None}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.inference:
        return run_inference(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus == 1:                           # launched under torchrun without --gpus: the environment decides
            print(f"[bench] --gpus not given, WORLD_SIZE={world}: running {world} ranks", file=sys.stderr, flush=True)
        else:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it launches "
                             "the ranks itself) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    if args.rendezvous_only:
        return rendezvous_only(args, rank, world)
    dp = world > 1 or args.force_dp                  # the data-parallel code path
    if dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ["PHNET_FORCE_COLLECTIVES"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(0 if (args.share_gpu or world == 1) else local_rank)
        dist.init_process_group(backend=args.backend, init_method="env://")
        if dist.get_world_size() != world:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, the environment said {world}")
        if args.backend == "nccl" and args.collectives == "rccl-streams":
            from phnet_amd import rccl
            rccl.install()                           # two dedicated communicators, before anything is captured (phnet_amd/rccl.py)
        if args.small_allreduce == "ipc" and world > 1:
            from phnet_amd import ipc
            ipc.install()                            # exchange buffers mapped by every peer, before anything is captured
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from phnet_amd import hip_ops
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    from phnet_amd.synthetic import make_clip, make_targets

    hip_ops.set_mma_mode(args.mma)
    if args.wgrad_flags is not None:
        hip_ops.tune_wgrad(args.wgrad_flags)
    if args.tune_k_tile is not None:
        hip_ops.tune_k_tile(args.tune_k_tile)
    torch.manual_seed(3407)
    cfg = make_cfg(img_h=args.height, img_w=args.width, arch=args.arch)
    model = RouterOL(cfg, Criterion4OL(cfg)).to(dev).train()
    # random-init cls/reg heads (std 1e-3) give ~0.5 scores everywhere, like the reference at initialisation
    net = model
    from phnet_amd import parallel
    from phnet_amd.arena import GradArena
    if dp:
        # the reference's data-parallel model (trainOL.py:141-146): SyncBatchNorm containers + same initial weights on every
        # rank (DDP's constructor broadcast); gradient averaging is the overlapped bucket reducer below instead of DDP's hooks
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        net = model
        for t_ in list(model.parameters()) + list(model.buffers()):
            d_ = t_.data
            if d_.dim() == 4 and not d_.is_contiguous():
                d_ = d_.permute(0, 2, 3, 1)          # channels_last parameter: broadcast its dense OHWI view
            dist.broadcast(d_, src=0)
    # flat fp32 arenas: the HIP backward kernels accumulate straight into the gradient arena, one launch of the flat AdamW
    # updates every parameter (torch.optim.AdamW semantics and the reference's grouping: no decay on 1-D parameters)
    from phnet_amd.optim import FlatAdamW
    use_graph = not args.eager
    opt, arena = FlatAdamW.for_model(model, lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-4)
    T = args.frames
    lanes = make_targets(args.height, args.width, T).to(dev)
    clips = [make_clip(args.height, args.width, T, seed=3407 + rank + 17 * i).to(dev) for i in range(4)]
    CB = args.clips_per_gpu
    if CB > 1:                                      # [B,T,3,H,W] inputs: RouterOL batches the head across the clips
        lanes = torch.stack([lanes] * CB)
        clips = [torch.stack([make_clip(args.height, args.width, T, seed=3407 + rank + 17 * i + 101 * b) for b in range(CB)]).to(dev)
                 for i in range(2)]

    # N > 1: 4 gradient buckets in backward order (lane head | neck + layer4 | layer3 | rest), each all-reduced (SUM of
    # gradients pre-divided by the world size) as soon as the backward has finished it - the head bucket, 77 % of the
    # bytes, hides behind the whole trunk backward
    reducer = parallel.BucketReducer(arena.flat, arena.bucket_bounds) if dp else None
    from phnet_amd.graphed import GraphedTrainStep, data_parallel_step

    def step(i):
        if dp:
            return data_parallel_step(model, arena, reducer, opt, clips[i % len(clips)], lanes, T * CB * world)
        arena.zero()
        loss = net({"frame": clips[i % len(clips)], "lanes": lanes}) / (T * CB)
        loss.backward()
        opt.step()
        return loss

    graphed = None
    if use_graph and dp and (args.backend != "nccl" or args.collectives != "rccl-streams"):
        use_graph = False                            # only raw RCCL calls on our streams can be captured (phnet_amd/rccl.py)
    if use_graph:
        try:
            graphed = GraphedTrainStep(model, opt, clips[0], lanes, loss_divisor=T * CB * world, warmup=2, arena=arena, reducer=reducer)
            print("[bench] training step captured in a hipGraph" + (" (RCCL collectives inside)" if dp else ""), file=sys.stderr, flush=True)
        except Exception as e:                                       # noqa: BLE001
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr, flush=True)
            graphed = None
    eager_step = step
    if graphed is not None:
        def step(i):                                                 # noqa: F811
            return graphed(clips[i % len(clips)])
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        print(f"[bench] rank {rank} warm-up step {i + 1}/{args.warmup} done", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    timer_on = not args.no_kernel_timer and graphed is None
    if timer_on:
        hip_ops.TIMER = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)          # the job's time is its slowest rank's
    dt = float(tmax.item())
    records, hip_ops.TIMER = hip_ops.TIMER, None
    timer_note = "HIP events around every launch of the kernel inside the timed region (eager)"
    isolated = None
    if graphed is not None and not args.no_kernel_timer:
        # Launches inside a graph replay cannot be bracketed by events (hipEventRecord nodes carry no timestamps on
        # ROCm 7.2).  Instead: run ONE instrumented eager step right after the timed region to collect every GEMM launch
        # of a step (same kernels, same shapes, same buffers), then re-issue those launches grouped by kernel symbol
        # inside small hipGraphs and time each graph with a pair of events: device-side duration per launch without
        # host gaps.
        hip_ops.TIMER = []
        eager_step(0)
        torch.cuda.synchronize()
        records, hip_ops.TIMER = hip_ops.TIMER, None
        groups = {}
        for rec in records:
            groups.setdefault(rec[0], []).append(rec)
        isolated = {}
        side = torch.cuda.Stream()
        for sym, recs in groups.items():
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for rec in recs:
                        rec[5]()
                g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                isolated[sym] = e0.elapsed_time(e1) / 3.0 * 1e-3            # seconds for all launches of the symbol
            except Exception as e:                                      # noqa: BLE001
                print(f"[bench] isolated timing of {sym} failed: {e}", file=sys.stderr)
        timer_note = ("device time of the step's launches of this kernel symbol, re-issued back to back in a hipGraph "
                      "(same shapes and buffers as the step) and bracketed by HIP events; split-K launches include their reduce kernel")
    steps_timed = args.steps if graphed is None else 1
    print(f"[bench] rank {rank}: {args.steps} timed steps in {dt:.3f} s", file=sys.stderr, flush=True)

    if rank == 0:
        roof = None
        if records:
            # GEMM launches are grouped by kernel symbol (what rocprofv3 --stats also groups by)
            agg = {}
            for sym, splits, flops, e0, e1, _launch in (r[:6] for r in records):
                a = agg.setdefault(sym, [0, 0.0, 0.0])
                a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
            if isolated:
                for sym, sec_ in isolated.items():
                    agg[sym][2] = sec_
            total_gemm_s = sum(a[2] for a in agg.values())
            sym, (n, fl, sec) = max(agg.items(), key=lambda kv: kv[1][2])
            peak, mfma_dtype, peak_note = GEMM_PEAK[args.mma]
            # counters from separate rocprofv3 --pmc passes of this bench (tests/tools/r03_profiles.sh): quoted only when they were
            # taken on the SAME kernel sources as the ones running now (sha256 over phnet_amd/csrc), null otherwise
            prof, prof_note = {}, None
            try:
                from tests.tools.kernel_sha import kernel_sources_sha
                prof = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
                if prof.get("kernel_sources_sha256") != kernel_sources_sha(ROOT):
                    prof_note = ("profiles/r03_pmc_summary.json was taken on other kernel sources (sha256 "
                                 f"{str(prof.get('kernel_sources_sha256'))[:12]}... vs {kernel_sources_sha(ROOT)[:12]}... now): counters not quoted")
                    prof = {}
            except Exception as e:                         # noqa: BLE001
                prof, prof_note = {}, f"no counter summary: {type(e).__name__}"
            traffic = (prof.get("traffic", {}).get(sym) or {}).get("hbm_bytes_per_launch_corrected")
            mfma_util = prof.get("backbone_conv_mfma_util")
            # algorithmic bytes of the symbol's launches, averaged over the SAME launches the counters average over: every operand
            # read once + the result written once (f32 activations; packed weights of conv3p: 6 bytes per element)
            alg = [r[7] for r in records if r[0] == sym and len(r) > 7 and r[7]]
            alg_bytes = (sum(alg) / len(alg)) if alg else None
            ach = fl / sec / 1e12
            roof = {"bound": "mfma", "kernel": sym, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(alg_bytes) if alg_bytes else None,
                    "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if (traffic and alg_bytes) else None,
                    "counters_note": prof_note,
                    "peak_note": f"algorithmic (f32-product) TFLOP/s this arithmetic reaches with the matrix pipe always busy: {peak_note}; "
                                 "`achieved` counts each f32 multiply-add once, whatever number of MFMAs it costs",
                    "frac_of_f32_mfma_peak": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                    "traffic_note": "HBM bytes per launch of this kernel symbol (average over its launches of a step) from separate "
                                    "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench (profiles/r03_pmc_summary.json, taken on "
                                    "these kernel sources; FETCH_SIZE doubled per the gfx950 correction for 16-B/lane streaming reads)" if traffic else None,
                    "launches": n, "avg_launch_us": round(sec / n * 1e6, 2), "gflop_per_launch": round(fl / n / 1e9, 3),
                    "mfma_dtype": mfma_dtype,
                    "all_gemm_kernels": {k: {"launches": v[0], "TFLOP/s": round(v[1] / v[2] / 1e12, 2), "ms_per_step": round(v[2] / steps_timed * 1e3, 3)}
                                         for k, v in sorted(agg.items())},
                    "gemm_ms_per_step": round(total_gemm_s / steps_timed * 1e3, 3), "timing": timer_note,
                    "backbone_conv_mfma_util_pct": mfma_util}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args)
        batched = None
        if world == 1 and CB == 1 and args.batched_extra > 1 and use_graph:
            batched = batched_extra(args)
        out = {"metric": f"clips/s ({T}x3x{args.height}x{args.width}) fwd+bwd", "value": round(world * CB * args.steps / dt, 4), "unit": "clips/s",
               "n_gpus": dist.get_world_size() if dp else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": DTYPE[args.mma], "data": "synthetic",
               "config": {"workload": f"{T}-frame clip 3x{args.height}x{args.width}, {args.arch} + router + lane head, fwd+bwd+AdamW, "
                                      f"{CB} clip{'s' if CB > 1 else ''}/GPU/step, random-init weights", "parallelism": f"dp{world}" + (" (data-parallel code path forced on one rank)" if dp and world == 1 else ""),
                          "timed_region": "grad-arena memset + forward + loss + backward (+ when N>1: SyncBatchNorm statistic exchanges and "
                                          "4 RCCL gradient-bucket all-reduces overlapped with the trunk backward) + AdamW step",
                          "launch": ("hipGraph replay of the whole step" if not dp else
                                     "hipGraph replay of the whole step, RCCL collectives captured inside the graph")
                                    if graphed is not None else "eager"},
               "loss": round(float(loss.item()) * world, 4), "roofline": roof, "cpu_baseline": cpu}
        if batched is not None:
            out["batched"] = batched
        if world == 1 and CB == 1 and args.split_extra and use_graph and args.mma != "f32":
            sp = f32_extra(args)
            if sp is not None:
                out["f32_mfma"] = sp
        if world == 1 and CB == 1 and args.inference_extra:
            inf = inference_extra(args)
            if inf is not None:
                out["inference"] = inf
        print(json.dumps(out), flush=True)
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
