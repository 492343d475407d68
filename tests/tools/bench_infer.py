"""Inference throughput (BASELINE.json configs[4]: 32 clips x 5 frames 3x320x800, ResNet-34 + lane head + HIP NMS,
hipGraph-captured).  Clips are independent and the head is per-clip sequential, so the batch is a loop of graph replays."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd.config import make_cfg
from phnet_amd.graphed import GraphedInference
from phnet_amd.libs.models.Router4OL import RouterOL
from phnet_amd.synthetic import make_clip

def main(clips=32, T=5, H=320, W=800, arch="resnet34"):
    mma = sys.argv[sys.argv.index("--mma") + 1] if "--mma" in sys.argv else "f32"      # f32 | split_bf16 | split3_bf16 (opt-in arithmetics)
    if mma != "f32":
        from phnet_amd import hip_ops
        hip_ops.set_mma_mode(mma)
    torch.manual_seed(0)
    model = RouterOL(make_cfg(img_h=H, img_w=W, arch=arch), None).cuda().eval()
    batch = [make_clip(H, W, T, seed=i).cuda() for i in range(4)]
    g = GraphedInference(model, batch[0])
    for i in range(3):
        g(batch[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(clips):
        rows, nums, anchors = g(batch[i % 4])
    host = model.lanes_from_device(rows, nums)           # one D2H + Lane objects for the last clip (per-clip host work)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # B clips per replay, lane head batched across the clips
    batched = {}
    for B in (4, 8, 16, 32):
        big = torch.stack([batch[i % 4] for i in range(B)])
        gb = GraphedInference(model, big)
        for _ in range(2):
            gb(big)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(clips // B):
            rb, nb, ab = gb(big)
        torch.cuda.synchronize()
        dtb = time.perf_counter() - tb
        batched[f"clips_per_s_batched_{B}"] = round((clips // B) * B / dtb, 2)
        del gb
    with torch.no_grad():
        t1 = time.perf_counter()
        for i in range(4):
            model({"frame": batch[i % 4], "lanes": None})
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t1) / 4
    print(json.dumps({"workload": f"{clips} clips x {T} frames 3x{H}x{W}, {arch}, eval, hipGraph", "gemm_arithmetic": mma, "clips_per_s": round(clips / dt, 2),
                      "frames_per_s": round(clips * T / dt, 1), "ms_per_clip_graph": round(dt / clips * 1e3, 2),
                      "ms_per_clip_eager_sync_free": round(eager * 1e3, 2), "lanes_last_clip": [len(x) for x in host["lane_lines"]],
                      **batched, "frames_per_s_batched_32": round(batched["clips_per_s_batched_32"] * T, 1)}))

if __name__ == "__main__":
    main()
