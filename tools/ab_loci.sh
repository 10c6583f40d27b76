#!/bin/bash
# A/B of environment assignments on the TE-loci leg (loci/s, seconds of each pass), same box, alternating:
# usage (through gpurun): bash tools/ab_loci.sh <config> -- name1=ENV=VALUE[,ENV2=VALUE2] ...
set -u
cfg=$1; shift; shift
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --no-upstream-check --no-stream-leg --no-shard-leg --bam-leg none --no-default-aligner-leg --no-polish-leg --steps 2 --warmup 1"
$B > /dev/null 2>&1
for rep in 1 2 3; do
  for name in default "$@"; do
    case "$name" in
      default) out=$($B 2>/dev/null);;
      *) envs=${name#*=}; out=$(env ${envs//,/ } $B 2>/dev/null);;
    esac
    echo "${name%%=*} $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['te_loci']; print(round(d['te_loci_per_s']), [round(x,4) for x in t['seconds_of_each_pass']], t['merged_table_sha256'][:10], round(d['ms_per_step'],1))")"
  done
done
rm -rf $cache
