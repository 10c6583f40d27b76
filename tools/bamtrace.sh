mkdir -p gpurun_out/prof gpurun_out/kt
export TMPDIR=/tmp
# data cache first (the bench forks workers to make the data set: not under the profiler)
python3 bench.py --no-cpu-baseline --loci 0 --no-stream-leg --steps 1 --warmup 0 --bam-leg none > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --steps 2 --warmup 1 > gpurun_out/prof/bam_under_rocprof.json 2>/dev/null
f=$(ls gpurun_out/kt/*/*kernel_stats.csv 2>/dev/null | head -1); echo $f
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("bam","bgzf","widen","iota","blk_first")): print(n[:40], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
