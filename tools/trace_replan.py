"""Analyse a rocprofv3 --kernel-trace CSV: duration histogram of the relax launches and the kernel
timeline of a few replans (gaps included)."""
import csv
import glob
import sys
import collections

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]
hist = collections.defaultdict(lambda: [0] * 12)
for s, e, n in rows:
    if "k_relax" in n:
        hist[short(n)][min((e - s) // 4000, 11)] += 1
for k, v in sorted(hist.items()):
    print("%-28s 4us bins: %s" % (k, v))
# find the replans: a replan starts with k_patch_apply
starts = [i for i, r in enumerate(rows) if "k_patch_apply" in r[2] or "k_patch_small" in r[2]]
for si in starts[150:153]:
    j = si
    t0 = rows[si][0]
    line = []
    while j < len(rows) and (j == si or ("k_patch_apply" not in rows[j][2] and "k_patch_small" not in rows[j][2])):
        s, e, n = rows[j]
        line.append("%s@%.0f+%.1f" % (short(n).replace("k_relax", "R").replace("k_", ""), (s - t0) / 1000, (e - s) / 1000))
        j += 1
    print("replan: " + " ".join(line))
