#!/bin/bash
# host-side marks of every batch (TELR_TRACE=host) for the tree's library and for a variant: where does a range wait for the host?
# usage (through gpurun): bash tools/host_trace_ab.sh <config> <variant> [bench args]
set -u
cfg=$1; var=$2; shift; shift
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --no-upstream-check --loci 0 --no-stream-leg --no-shard-leg --bam-leg none --no-default-aligner-leg --steps 2 --warmup 2 $*"
$B > /dev/null 2>&1
TELR_TRACE=host $B 2> gpurun_out/host_default.txt > /dev/null
TELR_TRACE=host TELR_LIB=$PWD/gpurun_variants/libtelrhip_$var.so $B 2> gpurun_out/host_$var.txt > /dev/null
rm -rf $cache
for f in gpurun_out/host_default.txt gpurun_out/host_$var.txt; do echo "== $f"; grep "\[host" $f | tail -120 | awk '{print}' | head -150; done
