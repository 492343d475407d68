"""Per (kernel symbol, grid) duration distribution from a rocprofv3 --kernel-trace CSV: median / min / max / total.
usage: kernel_dist.py <kernel_trace.csv> <out.txt> [steps]   (steps: divide totals to get ms per step)"""
import collections, csv, re, sys

steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    agg[(name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as out:
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        out.write(f"{name:62s} grid={grid:>8s} n/step={len(v) / steps:7.1f} med={v[len(v) // 2] / 1e3:7.1f}us "
                  f"min={v[0] / 1e3:7.1f} max={v[-1] / 1e3:7.1f} ms/step={sum(v) / 1e6 / steps:7.3f}\n")
