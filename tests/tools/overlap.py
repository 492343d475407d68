"""How much kernel execution overlaps in a rocprofv3 --kernel-trace CSV: sum of kernel durations vs. the union of their
[start, end) intervals, over the last `tail` fraction of the trace (the graph-replayed steps).
usage: overlap.py <kernel_trace.csv> [tail=0.4]"""
import csv, sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = rows[int(len(rows) * (1 - tail)):]
total = sum(e - s for s, e, _ in rows)
union, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}: sum of durations {total / 1e6:.2f} ms, union {union / 1e6:.2f} ms, span {span / 1e6:.2f} ms, "
      f"overlapped {100 * (total - union) / total:.1f} % of kernel time, idle {100 * (span - union) / span:.1f} % of the span")
