"""Which ATen device ops are still on the train step, and which source lines issue them.

    python tests/tools/profile_aten.py [--top 25]

Runs eager config-2 train steps under torch.profiler (CPU-side op records with Python stacks) and prints, per ATen op that
launches a device kernel, the call count per step and the phnet_amd source lines responsible.  Used to pick fusion targets
(DESIGN.md section 7)."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    from phnet_amd.synthetic import make_clip, make_targets
    from phnet_amd.arena import GradArena

    dev = torch.device("cuda", 0)
    cfg = make_cfg(img_h=320, img_w=800, arch="resnet34")
    torch.manual_seed(1234)
    model = RouterOL(cfg, Criterion4OL(cfg)).to(dev).train()
    clip, lanes = make_clip(320, 800, 5, seed=3407).to(dev), make_targets(320, 800, 5).to(dev)
    arena = GradArena(model.parameters())

    def step():
        arena.zero()
        loss = model({"frame": clip, "lanes": lanes}) / 5
        loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    interesting = ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::mul", "aten::cat", "aten::sum",
                   "aten::gelu", "aten::gelu_backward", "aten::native_dropout", "aten::native_dropout_backward", "aten::sigmoid",
                   "aten::index", "aten::div", "aten::sub", "aten::neg", "aten::where", "aten::clone", "aten::stack", "aten::mean",
                   "aten::sigmoid_backward", "aten::rsub", "aten::mul_", "aten::index_select", "aten::gather")
    per_op = collections.Counter()
    per_site = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        if ev.name not in interesting:
            continue
        site = None
        for fr in ev.stack:
            if "phnet_amd" in fr and "profile_aten" not in fr:
                site = fr.split("phnet_amd/")[-1]
                break
        if site is None:                                  # backward: name the autograd node (or enclosing op) instead
            par, chain = ev.cpu_parent, []
            while par is not None:
                if not par.name.startswith("aten::"):
                    chain.append(par.name.replace("autograd::engine::evaluate_function: ", ""))
                par = par.cpu_parent
            site = " <- ".join(chain[:2]) if chain else "<top level>"
        per_op[ev.name] += 1
        per_site[ev.name][site] += 1
    # forward sites of the autograd nodes behind those backward ops (anomaly mode records the forward traceback per node)
    with torch.autograd.detect_anomaly(check_nan=False):
        arena.zero()
        loss = model({"frame": clip, "lanes": lanes}) / 5
    seen, todo, node_sites = set(), [loss.grad_fn], collections.defaultdict(collections.Counter)
    while todo:
        fn = todo.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        todo.extend(f for f, _ in fn.next_functions)
        tb = fn.metadata.get("traceback_", [])
        site = "?"
        for line in reversed(tb):
            if "phnet_amd/" in line:
                site = line.strip().split("phnet_amd/")[-1].split("\n")[0]
                break
        node_sites[type(fn).__name__][site] += 1
    print("== autograd nodes per step by forward site ==")
    for name, sites in sorted(node_sites.items(), key=lambda kv: -sum(kv[1].values())):
        if name in ("AccumulateGrad",):
            continue
        print(f"{name}: {sum(sites.values())}")
        for site, c in sites.most_common(12):
            print(f"    {c:5d}  {site}")
    print("== ATen ops per step ==")
    for name, n in per_op.most_common(args.top):
        print(f"{name}: {n} calls/step")
        for site, c in per_site[name].most_common(8):
            print(f"    {c:5d}  {site}")


if __name__ == "__main__":
    main()
