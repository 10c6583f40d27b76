# kernel trace of the stage-1-to-BAM leg (every step under its own timeout: a profiler run that hangs costs box minutes)
mkdir -p gpurun_out/prof gpurun_out/kt
export TMPDIR=/tmp
timeout 200 python3 bench.py --no-cpu-baseline --loci 0 --no-stream-leg --steps 1 --warmup 0 --bam-leg none > /dev/null 2>&1
rm -rf gpurun_out/kt/*
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --steps 2 --warmup 1 > gpurun_out/prof/r03_bench_bam_under_rocprof.json 2>/dev/null
echo "rocprof exit $?"
f=$(ls gpurun_out/kt/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/prof_summary.py gpurun_out/kt | grep -E "^kernel|bam|bgzf|widen|iota|blk_first" | tee gpurun_out/prof/r03_bam_kernel_trace_summary.txt
