"""Condense rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch."""
import csv, glob, sys, re, collections
d = sys.argv[1]
agg = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"])
        if "rocprim" in n:
            n = "rocprim::*"
        k = (n, r["Counter_Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
print("%-40s %-14s %8s %16s %16s" % ("kernel", "counter", "calls", "sum", "mean/dispatch"))
for (n, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-40s %-14s %8d %16.1f %16.1f" % (n[:40], c, a[0], a[1], a[1] / a[0]))
