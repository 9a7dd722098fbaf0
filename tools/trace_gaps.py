"""Timeline summary of a rocprofv3 --kernel-trace CSV: per kernel name the count, mean duration, and for the
step kernels the mean start-to-start period and the idle gap between consecutive launches."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0)) for r in rows))
t0 = ev[0][0]
by = collections.defaultdict(list)
for s, e, n, g in ev:
    key = n.split("(")[0][:60] + f" grid={g}"
    by[key].append((s - t0, e - t0))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 5:
        continue
    v = v[len(v) // 4:]          # skip warm-up
    dur = sum(e - s for s, e in v) / len(v)
    per = (v[-1][0] - v[0][0]) / max(1, len(v) - 1)
    print(f"{k:90s} n={len(v):5d} dur {dur/1e3:8.2f} us  period {per/1e3:8.2f} us")
if len(sys.argv) > 2:
    # argv[2]: substring of a kernel name; print argv[4] events starting at the argv[3]-th match
    first = [i for i, x in enumerate(ev) if sys.argv[2] in x[2]]
    lo = first[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
    hi = lo + (int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    for s, e, n, g in ev[lo:hi]:
        print(f"{(s-t0)/1e3:10.2f} .. {(e-t0)/1e3:10.2f} us  ({(e-s)/1e3:7.2f})  {n.split('(')[0][:50]} grid={g}")
