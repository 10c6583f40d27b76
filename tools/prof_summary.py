"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short per-kernel table."""
import csv, glob, sys, re, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = re.sub(r"\(.*", "", n)
    if "rocprim" in n:
        m = re.search(r"(segmented_radix_sort|radix_sort_onesweep|scan_impl|histogram|lookback|transform|partition)", n)
        n = "rocprim::" + (m.group(1) if m else "other")
    dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(n, [0, 0.0, 0.0, 0])
    a[0] += 1; a[1] += dt; a[2] = max(a[2], dt); a[3] = int(r["VGPR_Count"])
tot = sum(a[1] for a in agg.values())
print("%-46s %6s %10s %10s %10s %6s %5s" % ("kernel", "calls", "total_ms", "avg_ms", "max_ms", "pct", "vgpr"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-46s %6d %10.3f %10.3f %10.3f %6.2f %5d" % (n[:46], a[0], a[1], a[1] / a[0], a[2], 100 * a[1] / tot, a[3]))
