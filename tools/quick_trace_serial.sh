#!/bin/bash
# Kernel trace of map steps with every class on the main stream and no range pipelining (TELR_SERIAL=1 TELR_PIPELINE=1): the
# stand-alone duration of every kernel, without the co-running kernels of another range or class.  Iteration aid.
# usage (through gpurun): bash tools/quick_trace_serial.sh <tag> [config] [extra bench args]
set -u
tag=${1:-qts}; cfg=${2:-c2}; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
export TELR_SERIAL=1 TELR_PIPELINE=1
timeout 900 python3 bench.py --config $cfg --data-cache $cache --bam-leg none --loci 0 --no-stream-leg --no-default-aligner-leg --no-shard-leg --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/${tag}_bench_plain.json 2>$out/${tag}_bench_plain.err || { echo "plain run failed"; tail -5 $out/${tag}_bench_plain.err; exit 1; }
B="python3 bench.py --config $cfg --data-cache $cache --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --no-default-aligner-leg --no-shard-leg --bam-leg none"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- $B --steps 2 --warmup 1 "$@" > $out/${tag}_bench_under_rocprof.json 2>/dev/null
python3 tools/prof_summary.py gpurun_out/kt > $out/${tag}_kernel_trace_summary.txt
rm -rf gpurun_out/kt $cache
python3 -c "
import json
d=json.loads(open('$out/${tag}_bench_plain.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'])
print({k:round(v,1) for k,v in d['stage_ms_per_step'].items()})
"
head -45 $out/${tag}_kernel_trace_summary.txt
