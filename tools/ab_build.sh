#!/bin/bash
# A/B of a compile-time constant on one box: bash tools/ab_build.sh "<-DNAME=VALUE>" [config] [steps]
# (the variant is built into a copy of the library under /tmp and selected with TELR_LIB)
set -u
flag=$1; cfg=${2:-c3}; steps=${3:-4}
cd "$GRAFT_REPO_ROOT"
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
B="python3 bench.py --config $cfg --data-cache $cache --no-cpu-baseline --loci 0 --no-stream-leg --no-shard-leg --bam-leg none --steps $steps --warmup 1"
$B > /dev/null 2>&1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flag -o /tmp/libtelrhip_variant.so telr_amd/csrc/telr_engine.hip -lz || exit 1
for rep in 1 2; do
  for mode in default variant; do
    if [ "$mode" = default ]; then out=$($B 2>/dev/null); else out=$(TELR_LIB=/tmp/libtelrhip_variant.so $B 2>/dev/null); fi
    echo "$mode $flag $(echo "$out" | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],2), {k:round(v,1) for k,v in d['stage_ms_per_step'].items() if k in ('seed','dp','sort')})")"
  done
done
rm -rf $cache
