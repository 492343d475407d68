"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass: MFMA-pipe utilisation per kernel symbol.
util = sum(MFMA busy cycles over all SIMDs) / (kernel cycles * 256 CUs * 4 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections, csv, glob, json, re, sys

def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]

rows = collections.defaultdict(dict)
for path in glob.glob(sys.argv[1]):
    for r in csv.DictReader(open(path)):
        rows[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (_, k, grid), c in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and (k.startswith("conv") or k.startswith("wgrad")):
        a = agg[f"{k} grid={grid}"]
        a[0] += 1; a[1] += c["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += c["GRBM_GUI_ACTIVE"]
out = {}
for k, (n, busy, gui) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cycles = gui / 8.0
    out[k] = {"dispatches": n, "mfma_busy_cycles_per_dispatch": round(busy / n), "kernel_cycles_per_dispatch": round(cycles / n),
              "mfma_util_pct": round(100.0 * busy / (cycles * 1024.0), 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:16]:
    print(k, v)
