#!/bin/bash
# A/B of prebuilt library variants (gpurun_variants/libtelrhip_<name>.so, built in the container with hipcc -D...) against the tree's library on one box:
# usage (through gpurun): bash tools/ab_variants.sh <config> <steps> [bench args] -- name1 name2 ...
set -u
cfg=$1; steps=$2; shift; shift
X=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do X="$X $1"; shift; done; shift
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --loci 0 --no-stream-leg --no-shard-leg --bam-leg none --no-default-aligner-leg --steps $steps --warmup 2 $X"
$B > /dev/null 2>&1
for rep in 1 2; do
  for name in default "$@"; do
    if [ "$name" = default ]; then out=$($B 2>/dev/null); else out=$(TELR_LIB=$PWD/gpurun_variants/libtelrhip_$name.so $B 2>/dev/null); fi
    echo "$name $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), {k:round(v,1) for k,v in d['stage_ms_per_step'].items() if k in ('seed','dp','chain')})")"
  done
done
rm -rf $cache
