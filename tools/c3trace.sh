mkdir -p gpurun_out/kt4; export TMPDIR=/tmp
timeout 200 python3 bench.py --config c3 --no-cpu-baseline --loci 0 --no-stream-leg --steps 1 --warmup 0 --bam-leg none --no-shard-leg > /dev/null 2>&1
rm -rf gpurun_out/kt4/*
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt4 -- python3 bench.py --config c3 --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --steps 2 --warmup 1 --bam-leg none --no-shard-leg > /dev/null 2>&1
echo "rocprof exit $?"
python3 tools/prof_summary.py gpurun_out/kt4 | head -24
