"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel HBM bytes per launch.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide (16 B/lane) streaming reads at exactly half."""
import collections, csv, glob, json, re, sys

def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]

def load(pattern, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg

f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(f, key=lambda k: -f[k][1]):
    if not (k.startswith("conv") or k.startswith("gate_") or k.startswith("layernorm") or k.startswith("bn_") or k.startswith("roi_")
            or k.startswith("channel_") or k.startswith("linear_") or k.startswith("dyn_") or k.startswith("attn_") or k.startswith("lane_") or k.startswith("frame_loss")):
        continue
    n = f[k][0]
    fk = f[k][1] / n
    wk = (w[k][1] / w[k][0]) if k in w else 0.0
    out[k] = {"launches": n, "FETCH_SIZE_KB_per_launch": round(fk, 1), "WRITE_SIZE_KB_per_launch": round(wk, 1),
              "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:12]:
    print(k, v)
