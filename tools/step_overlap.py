#!/usr/bin/env python3
"""Where is a map step serial?  Reads a rocprofv3 kernel_trace.csv, takes the steady-state window between the first kernels of
two consecutive telr_map calls (k_sketch32 bursts separated by the longest gaps), and prints: wall, time with 0 / 1 / >= 2 kernels
running, and per kernel its EXCLUSIVE time (nothing else running) next to its total -- the exclusive stretches are what a
faster kernel or more overlap would shorten.
usage: step_overlap.py kernel_trace.csv [first_step last_step]"""
import csv, sys, collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
# steps: a k_sketch kernel that starts more than 15 ms after the previous sketch kernel's START begins a new range; steps are
# found from the caller's point of view by the option --ranges-per-step (default 4)
rps = 4
sk = [s for s, e, n in rows if n.startswith("k_sketch")]
starts = [sk[0]] + [b for a, b in zip(sk, sk[1:]) if b - a > 15e6]
steps = starts[::rps]
lo_i, hi_i = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, len(steps) - 1)
t0, t1 = steps[lo_i], steps[hi_i]
ev = []
for s, e, n in rows:
    s2, e2 = max(s, t0), min(e, t1)
    if e2 > s2:
        ev.append((s2, 1, n)); ev.append((e2, -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
run = collections.Counter(); excl = collections.Counter(); tot = collections.Counter(); hist = collections.Counter()
last = t0
for t, d, n in ev:
    k = sum(run.values())
    dt = t - last
    if dt > 0:
        hist[min(k, 3)] += dt
        if k == 1:
            excl[next(x for x, c in run.items() if c > 0)] += dt
        for x, c in run.items():
            if c > 0:
                tot[x] += dt
    run[n] += d
    last = t
wall = (t1 - t0) / 1e6; nst = hi_i - lo_i
print("window: steps %d..%d, %.1f ms per step" % (lo_i, hi_i, wall / nst))
print("per step: idle %.1f ms, one kernel %.1f ms, two %.1f ms, three or more %.1f ms" % tuple(hist[k] / 1e6 / nst for k in range(4)))
print("%-40s %10s %10s" % ("kernel", "alone ms", "running ms"))
for n, v in sorted(excl.items(), key=lambda kv: -kv[1])[:25]:
    print("%-40s %10.2f %10.2f" % (n[:40], v / 1e6 / nst, tot[n] / 1e6 / nst))
