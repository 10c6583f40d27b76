import csv, glob, sys, os, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
# last k_segments<0> marks the start of the last step's DP part
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_segments<0>")][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    if "rocprim" in n:
        m = re.search(r"(segmented_radix_sort|radix_sort_onesweep|scan_impl|lookback|partition)", n); n = "rocprim::" + (m.group(1) if m else "other")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s > 0.05 or n.startswith(("k_", "void k_")):
        print("%-34s stream %-3s grid %9s  %8.3f -> %8.3f  (%.3f ms)" % (n[:34], r["Stream_Id"], r["Grid_Size_X"], s, e, e - s))
