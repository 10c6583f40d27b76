#!/bin/bash
# the trace-back tap (gpurun_variants/libtelrhip_tbprof.so, built with -DTB_PROF): per k_traceback_pk launch, the waves' own durations
# against the launch's span.   usage (through gpurun): bash tools/tbprof.sh [config] [bench args]
set -u
cfg=${1:-c2}; shift
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/c.XXXX)
TELR_SERIAL=1 TELR_PIPELINE=1 TELR_LIB=$PWD/gpurun_variants/libtelrhip_tbprof.so timeout 800 python3 bench.py --config $cfg --data-cache $cache --bam-leg none --loci 0 --no-stream-leg --no-default-aligner-leg --no-shard-leg --no-cpu-baseline --no-upstream-check --steps 1 --warmup 1 "$@" 2>&1 >/dev/null | grep "tb prof" | tail -12
rm -rf $cache
