#!/usr/bin/env python3
"""Best-of-rounds table from tools/tune_step output (variant x plane pad, ms per step)."""
import collections
import re
import sys

best = collections.defaultdict(list)
for line in open(sys.argv[1]):
    if line.startswith(("copy4", "#")):
        print(line.strip())
        continue
    m = re.match(r"pad\s+(\d+)\s+(.*?)\s{2,}([\d.]+) ms", line)
    if m:
        best[(m.group(2).strip(), int(m.group(1)))].append(float(m.group(3)))
names = sorted({k[0] for k in best})
pads = sorted({k[1] for k in best})
print("%-28s" % "variant (min ms over rounds)" + "".join("%9d" % p for p in pads))
for n in names:
    print("%-28s" % n + "".join("%9.4f" % min(best[(n, p)]) if (n, p) in best else "        -" for p in pads))
