#!/bin/bash
# rocprofv3 evidence for the reference's DEFAULT aligner path (the ngmlr-* presets): kernel trace (two ranges in flight, and every kernel alone on the
# device: TELR_SERIAL=1 TELR_PIPELINE=1) + FETCH_SIZE / WRITE_SIZE / SQ counters of one step.
# usage (from the repo root, through gpurun):  bash tools/collect_ngmlr_profiles.sh <tag> <config> [--preset P]     e.g.  r05_ngmlr_ont_c2 c2 --preset ngmlr-ont ; r05_ngmlr_pacbio_c3 c3
set -u
tag=$1; cfg=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
X="--no-default-aligner-leg --no-upstream-check"
timeout 1200 python3 bench.py --config $cfg --data-cache $cache --bam-leg none --loci 0 --no-stream-leg --no-shard-leg --cpu-sample-reads 12000 $X "$@" > $out/${tag}_bench.json 2>$out/${tag}_bench.err || { echo "plain bench run failed"; tail -5 $out/${tag}_bench.err; exit 1; }
B="python3 bench.py --config $cfg --data-cache $cache --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --no-shard-leg --bam-leg none $X"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- $B --steps 3 --warmup 1 "$@" > $out/${tag}_bench_under_rocprof.json 2>/dev/null
python3 tools/prof_summary.py gpurun_out/kt > $out/${tag}_kernel_trace_summary.txt
rm -rf gpurun_out/kt
TELR_SERIAL=1 TELR_PIPELINE=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- $B --steps 2 --warmup 1 "$@" > /dev/null 2>&1
python3 tools/prof_summary.py gpurun_out/kt > $out/${tag}_kernel_trace_serial_summary.txt
rm -rf gpurun_out/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc -- $B --steps 1 --warmup 0 "$@" > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc > $out/${tag}_pmc_$c.txt; rm -rf gpurun_out/pmc
done
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc -- $B --steps 1 --warmup 0 "$@" > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc > $out/${tag}_pmc_SQ.txt; rm -rf gpurun_out/pmc
rm -rf $cache
python3 -c "
import json
d=json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'], 'parity', d['cpu_baseline'].get('parity',{}).get('identical'))"
head -16 $out/${tag}_kernel_trace_serial_summary.txt
