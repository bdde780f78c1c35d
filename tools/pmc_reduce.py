"""Reduce a rocprofv3 --pmc <COUNTER> --kernel-trace run (output-format csv) to the per-kernel mean of the counter.
    python tools/pmc_reduce.py <dir with *counter_collection.csv> <COUNTER> > per_kernel.csv"""
import csv, glob, os, sys
from collections import defaultdict
d, counter = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: [0.0, 0])
for f in files:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("Kernel_Name,Dispatches,Mean_%s" % counter)
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print('"%s",%d,%.6g' % (k.replace('"', "'"), n, s / n))
