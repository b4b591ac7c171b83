"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (sum over dispatches)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"]
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); calls[k] += 1
for k, d in acc.items():
    print(k, "dispatches", calls[k])
    for c, v in sorted(d.items()):
        print("   %-28s %.4g   per-dispatch %.4g" % (c, v, v / max(1, calls[k])))
