#!/bin/bash
# A/B of an environment switch on one box: bash tools/ab.sh "<VAR=VALUE>" [config] [steps]  -> ms per step, alternating runs
set -u
sw=$1; cfg=${2:-c2}; steps=${3:-8}
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --loci 0 --no-stream-leg --no-shard-leg --bam-leg none --steps $steps --warmup 2"
$B > /dev/null 2>&1
for rep in 1 2 3; do
  for mode in default "$sw"; do
    if [ "$mode" = default ]; then out=$($B 2>/dev/null); else out=$(env $sw $B 2>/dev/null); fi
    echo "$mode $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2))")"
  done
done
rm -rf $cache
