#!/bin/bash
# A/B of environment assignments on the polishing passes (pile-up and window consensus, seconds of each pass), same box, alternating:
# usage (through gpurun): bash tools/ab_polish.sh <config> -- name1=ENV=VALUE ...
set -u
cfg=$1; shift; shift
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --no-upstream-check --no-stream-leg --no-shard-leg --bam-leg none --no-default-aligner-leg --steps 1 --warmup 0"
$B > /dev/null 2>&1
for rep in 1 2 3; do
  for name in default "$@"; do
    case "$name" in
      default) out=$($B 2>/dev/null);;
      *) envs=${name#*=}; out=$(env ${envs//,/ } $B 2>/dev/null);;
    esac
    echo "${name%%=*} $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['te_loci']['polish_pileup']; print('pileup', round(p['loci_per_s']), [round(x,4) for x in p['seconds_of_each_pass']], 'poa', round(p['poa']['loci_per_s']), [round(x,4) for x in p['poa']['seconds_of_each_pass']], 'call sets', {k:(v.get('recovered_exact_chrom_family_strand_pos20', v.get('error')) if isinstance(v,dict) else None) for k,v in p.get('call_set_ab',{}).items() if k!='what'})")"
  done
done
rm -rf $cache
