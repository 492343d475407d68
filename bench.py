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

import torch
import torch.distributed as dist

F32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


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
    ap.add_argument("--mma", default="f32", choices=["f32", "split_bf16", "split3_bf16"],
                    help="arithmetic of the conv / linear GEMM kernels: f32-input MFMA (default, the headline) or split-bf16 "
                         "(operands split into two bf16 terms, 3 bf16 MFMAs per product, f32 accumulation)")
    ap.add_argument("--split-extra", type=int, default=1,
                    help="after the headline run, also time the step in split-bf16 arithmetic and report it under the extra key "
                         "'split_bf16' (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=6)
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step in a hipGraph (always eager for N>1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only for "
                                                      "functional rehearsals of the N>1 path on a single GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
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
    t0 = time.perf_counter()
    for i in range(args.cpu_clips):
        loss = O.clip_forward(sd, frames, lanes, g, training=True)
        loss.backward()
        print(f"[bench] cpu baseline clip {i + 1}/{args.cpu_clips} done at {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": args.cpu_clips / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_clips} clips of {args.frames}x3x{args.height}x{args.width} fwd+bwd (no optimizer, no warm-up), "
                      f"oracle/phnet_cpu.py on torch CPU fp32, {dt:.1f} s"}


def _child_bench(args, flags, what):
    """Runs this script once more in a child process (so that it cannot disturb the run above) and returns its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), *flags, "--no-cpu-baseline", "--no-kernel-timer", "--batched-extra", "0",
           "--split-extra", "0", "--arch", args.arch, "--frames", str(args.frames), "--height", str(args.height),
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


def split_extra(args):
    """Not the headline: the headline workload (and the batched one) with the GEMM kernels in split-bf16 arithmetic."""
    out = {"note": "opt-in arithmetic (--mma split_bf16): every conv / linear GEMM splits its f32 operands in registers into two "
                   "bf16 terms and runs 3 bf16 MFMAs per product with f32 accumulation (csrc/igemm.h) - rounding noise 4-5e-6 of "
                   "a GEMM's output scale, ~4x the f32-input MFMA's; the model's refinement cascade amplifies that beyond the "
                   "1e-3 end-to-end parity bound the headline arithmetic meets (tests/test_model_gpu.py), so it is reported "
                   "here and never as the headline."}
    d = _child_bench(args, ["--mma", "split_bf16", "--steps", "10", "--warmup", "3"], "split-bf16 arithmetic, 1 clip per GPU and step")
    if d is None:
        return None
    out.update({"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"]})
    if args.batched_extra > 1:
        b = _child_bench(args, ["--mma", "split_bf16", "--clips-per-gpu", str(args.batched_extra), "--steps", "6", "--warmup", "2"],
                         f"split-bf16 arithmetic, {args.batched_extra} clips per GPU and step")
        if b is not None:
            out["batched"] = {"clips_per_gpu": args.batched_extra, "value": b["value"], "ms_per_step": b["ms_per_step"]}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0 if args.share_gpu else local_rank)
        dist.init_process_group(backend=args.backend, init_method="env://")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from phnet_amd import hip_ops
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    from phnet_amd.synthetic import make_clip, make_targets

    if args.mma != "f32":
        hip_ops.set_mma_mode(args.mma)
        args.no_kernel_timer = True                  # the per-symbol attribution below knows the f32 kernels only
    torch.manual_seed(3407)
    cfg = make_cfg(img_h=args.height, img_w=args.width, arch=args.arch)
    model = RouterOL(cfg, Criterion4OL(cfg)).to(dev).train()
    # random-init cls/reg heads (std 1e-3) give ~0.5 scores everywhere, like the reference at initialisation
    net = model
    from phnet_amd import parallel
    from phnet_amd.arena import GradArena
    if world > 1:
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
    reducer = parallel.BucketReducer(arena.flat, arena.bucket_bounds) if world > 1 else None
    from phnet_amd.graphed import GraphedTrainStep, data_parallel_step

    def step(i):
        if world > 1:
            return data_parallel_step(model, arena, reducer, opt, clips[i % len(clips)], lanes, T * CB * world)
        arena.zero()
        loss = net({"frame": clips[i % len(clips)], "lanes": lanes}) / (T * CB)
        loss.backward()
        opt.step()
        return loss

    graphed = None
    if use_graph and world > 1 and args.backend != "nccl":
        use_graph = False                            # only RCCL collectives can be captured; a gloo rehearsal runs eagerly
    if use_graph:
        try:
            graphed = GraphedTrainStep(model, opt, clips[0], lanes, loss_divisor=T * CB * world, warmup=2, arena=arena, reducer=reducer)
            print("[bench] training step captured in a hipGraph" + ("" if world == 1 else " (RCCL collectives inside)"), file=sys.stderr, flush=True)
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
            for sym, splits, flops, e0, e1, _launch in records:
                a = agg.setdefault(sym, [0, 0.0, 0.0])
                a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
            if isolated:
                for sym, sec_ in isolated.items():
                    agg[sym][2] = sec_
            total_gemm_s = sum(a[2] for a in agg.values())
            sym, (n, fl, sec) = max(agg.items(), key=lambda kv: kv[1][2])
            traffic = None
            try:                                           # PMC passes are separate rocprofv3 runs (profiles/README.md)
                short = sym.replace(", false", ", false").strip()
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                if short in pmc:
                    traffic = pmc[short]["hbm_bytes_per_launch_corrected"]
            except Exception:                              # noqa: BLE001
                traffic = None
            mfma_util = None
            try:                                           # separate PMC pass on the backbone conv shapes (profiles/README.md)
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_mfma_trunk.json")))
                vals = [v["mfma_util_pct"] for k, v in pm.items() if ", 16, " in k or "wgrad" in k]
                mfma_util = {"min": min(vals), "max": max(vals), "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on "
                             "tests/tools/bench_conv.py --trunk (ResNet-34 stage convs of a 5x320x800 clip), profiles/r01_pmc_mfma_trunk.json"}
            except Exception:                              # noqa: BLE001
                mfma_util = None
            ach = fl / sec / 1e12
            roof = {"bound": "mfma", "kernel": sym, "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes per launch of this kernel symbol from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                    "passes of this bench (profiles/r01_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 "
                                    "correction for 16-B/lane streaming reads)" if traffic else None,
                    "launches": n, "avg_launch_us": round(sec / n * 1e6, 2), "gflop_per_launch": round(fl / n / 1e9, 3),
                    "mfma_dtype": "f32 (v_mfma_f32_32x32x2_f32)",
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
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.mma == "f32" else "f32 storage and accumulation, bf16x2-split MFMA inputs", "data": "synthetic",
               "config": {"workload": f"{T}-frame clip 3x{args.height}x{args.width}, {args.arch} + router + lane head, fwd+bwd+AdamW, "
                                      f"{CB} clip{'s' if CB > 1 else ''}/GPU/step, random-init weights", "parallelism": f"dp{world}",
                          "timed_region": "grad-arena memset + forward + loss + backward (+ when N>1: SyncBatchNorm statistic exchanges and "
                                          "4 RCCL gradient-bucket all-reduces overlapped with the trunk backward) + AdamW step",
                          "launch": ("hipGraph replay of the whole step" if world == 1 else
                                     "hipGraph replay of the whole step, RCCL collectives captured inside the graph")
                                    if graphed is not None else "eager"},
               "loss": round(float(loss.item()) * world, 4), "roofline": roof, "cpu_baseline": cpu}
        if batched is not None:
            out["batched"] = batched
        if world == 1 and CB == 1 and args.split_extra and use_graph and args.mma == "f32":
            sp = split_extra(args)
            if sp is not None:
                out["split_bf16"] = sp
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
