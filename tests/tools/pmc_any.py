"""Generic summary of rocprofv3 --pmc passes: per kernel symbol (+grid), the mean of every collected counter.
usage: pmc_any.py '<glob of *counter_collection.csv>' out.json [name-prefix]"""
import collections, csv, glob, json, re, sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


prefix = sys.argv[3] if len(sys.argv) > 3 else "conv"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in glob.glob(sys.argv[1], recursive=True):
    per_dispatch = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k.startswith(prefix):
            per_dispatch[(r["Dispatch_Id"], f"{k} grid={r['Grid_Size']}")][r["Counter_Name"]] = float(r["Counter_Value"])
    for (_, key), c in per_dispatch.items():
        for name, v in c.items():
            a = acc[key][name]
            a[0] += 1; a[1] += v
out = {k: {n: round(s / c, 1) for n, (c, s) in v.items()} for k, v in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, v)
