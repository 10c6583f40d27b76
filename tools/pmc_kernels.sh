#!/bin/bash
# PMC counters of chosen kernels, one counter group per pass (iteration aid).
# usage (through gpurun): [LOCI=-1] [BENCH_ARGS="--preset ngmlr-ont"] bash tools/pmc_kernels.sh <tag> <config> <kernel regex> <group> [<group> ...]   (a group = counters separated by '+')
set -u
tag=$1; cfg=$2; pat=$3; shift; shift; shift
X=${BENCH_ARGS:-}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
timeout 900 python3 bench.py --config $cfg --data-cache $cache --bam-leg none --loci ${LOCI:-0} --no-stream-leg --no-default-aligner-leg --no-shard-leg --no-cpu-baseline --steps 1 --warmup 0 $X > /dev/null 2>$out/${tag}_gen.err || { echo "generation failed"; tail -5 $out/${tag}_gen.err; exit 1; }
B="python3 bench.py --config $cfg --data-cache $cache --require-cache --no-cpu-baseline --loci ${LOCI:-0} --no-stream-leg --no-default-aligner-leg --no-shard-leg --bam-leg none $X"
: > $out/${tag}_pmc.txt
for g in "$@"; do
  timeout 900 rocprofv3 --pmc ${g//+/ } --output-format csv -d gpurun_out/pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc | grep -E "^kernel|$pat" >> $out/${tag}_pmc.txt; rm -rf gpurun_out/pmc
done
rm -rf $cache
cat $out/${tag}_pmc.txt
