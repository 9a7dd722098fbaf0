"""Mean of one PMC counter per kernel from a rocprofv3 --pmc counter_collection CSV.
python tools/pmc_mean.py <counter_collection.csv> <COUNTER> [min launches, default 3]"""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if len(v) >= (int(sys.argv[3]) if len(sys.argv) > 3 else 3):
        print(f"{k},{sys.argv[2]},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f}")
