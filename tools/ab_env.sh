#!/bin/bash
# A/B of environment switches (TELR_AB tokens) and prebuilt library variants against the tree's library on one box, alternating runs:
# usage (through gpurun): bash tools/ab_env.sh <config> <steps> [bench args] -- name1[=ENV=VALUE] name2 ...
#   a name with '=': the run gets that environment assignment (e.g. skip=TELR_AB=dbg_tb_skip); a plain name: gpurun_variants/libtelrhip_<name>.so
set -u
cfg=$1; steps=$2; shift; shift
X=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do X="$X $1"; shift; done; shift
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --no-upstream-check --loci 0 --no-stream-leg --no-shard-leg --bam-leg none --no-default-aligner-leg --steps $steps --warmup 2 $X"
$B > /dev/null 2>&1
for rep in 1 2 3; do
  for name in default "$@"; do
    case "$name" in
      default) out=$($B 2>/dev/null);;
      *=*) out=$(env "${name#*=}" $B 2>/dev/null);;
      *) out=$(TELR_LIB=$PWD/gpurun_variants/libtelrhip_$name.so $B 2>/dev/null);;
    esac
    echo "${name%%=*} $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), {k:round(v,1) for k,v in d['stage_ms_per_step'].items() if k in ('seed','sort','chain','backtrack','select_host','dp','k_traceback','k_dp_pk')})")"
  done
done
rm -rf $cache
