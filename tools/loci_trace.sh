#!/bin/bash
# Kernel trace of the TE-loci leg with its polish passes (pile-up and window POA): per-kernel totals.  Iteration aid.
# usage (through gpurun): bash tools/loci_trace.sh <tag> [config]
set -u
tag=${1:-loci}; cfg=${2:-c2}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
A="--config $cfg --data-cache $cache --bam-leg none --no-stream-leg --no-default-aligner-leg --no-shard-leg --no-cpu-baseline --no-upstream-check --steps 1 --warmup 0"
TELR_TRACE=host timeout 900 python3 bench.py $A > $out/${tag}_bench_plain.json 2>$out/${tag}_bench_plain.err || { echo "plain run failed"; tail -5 $out/${tag}_bench_plain.err; exit 1; }
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py $A --require-cache > $out/${tag}_bench_under_rocprof.json 2>/dev/null
python3 tools/prof_summary.py gpurun_out/kt > $out/${tag}_kernel_trace_summary.txt
rm -rf gpurun_out/kt $cache
python3 -c "
import json
d=json.loads(open('$out/${tag}_bench_plain.json').read().strip().splitlines()[-1])
p=d['te_loci']['polish_pileup']
print('poa phases', p['poa'].get('phases_s'))
print('loci/s',round(d['te_loci_per_s']),'pileup pass s',round(p['seconds'],3),'poa pass s',round(p['poa']['seconds'],3),'poa loci/s',round(p['poa']['loci_per_s']))
"
grep "host poa" $out/${tag}_bench_plain.err | tail -6
grep -n "k_poa\|k_pileup\|k_pile" $out/${tag}_kernel_trace_summary.txt | cut -c1-140
