"""Timeline of the LAST telr_map call inside a rocprofv3 --kernel-trace CSV directory: every kernel (rocPRIM merged by
name) with start / end relative to the call's first kernel, plus the idle gap before it on the device."""
import csv, glob, sys, os, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("void k_sketch<", "void k_sketch32<", "void k_sketch_hpc<"))][-1]
# a map call may run several sketch tiles back to back: rewind to the first of the run
while idx > 0 and rows[idx - 1]["Kernel_Name"].startswith(("void k_sketch<", "void k_sketch32<", "void k_sketch_hpc<", "k_sketch_compact")):
    idx -= 1
t0 = int(rows[idx]["Start_Timestamp"]); busy_end = t0
for r in rows[idx:]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    if "rocprim" in n:
        m = re.search(r"(segmented_radix_sort|radix_sort_onesweep|scan_impl|lookback|partition|histogram|transform)", n); n = "rocprim::" + (m.group(1) if m else "other")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - busy_end) / 1e6
    if e - s > 20000 or gap > 0.05:
        print("%-34s q%-3s grid %9s  %8.3f -> %8.3f  (%7.3f ms)%s" % (n[:34], r.get("Queue_Id", "?")[-3:], r["Grid_Size_X"], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6,
                                                                      "   idle before: %.3f ms" % gap if gap > 0.05 else ""))
    busy_end = max(busy_end, e)
