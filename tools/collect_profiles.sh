#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/prof/ (copy what should be judged into profiles/).
# usage (from the repo root, through gpurun):  bash tools/collect_profiles.sh r03 [bench config, default c2]
# The data set is generated once (plain run, forked generator) and cached in a private directory: the profiled runs load it
# (--require-cache: rocprofv3 and a forking child do not mix, so a missing cache is an error, not a reason to generate).
set -u
tag=${1:-r03}; cfg=${2:-c2}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out
cache=$(mktemp -d /tmp/telr_cache.XXXXXX)
hipcc -O3 --offload-arch=gfx950 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip || { echo "ubench build failed"; exit 1; }
timeout 1200 python3 bench.py --config $cfg --data-cache $cache --bam-leg device --files-leg > $out/${tag}_bench_default.json 2>$out/${tag}_bench_default.err || { echo "plain bench run failed"; tail -5 $out/${tag}_bench_default.err; exit 1; }
ls $cache/*.npz > /dev/null || { echo "no data cache"; exit 1; }
B="python3 bench.py --config $cfg --data-cache $cache --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --no-default-aligner-leg --no-shard-leg --bam-leg none"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- $B --steps 3 --warmup 1 > $out/${tag}_bench_under_rocprof.json 2>/dev/null
python3 tools/prof_summary.py gpurun_out/kt > $out/${tag}_kernel_trace_summary.txt
python3 tools/step_timeline.py gpurun_out/kt > $out/${tag}_step_timeline.txt
cp "$(find gpurun_out/kt -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_raw.csv
rm -rf gpurun_out/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc -- $B --steps 1 --warmup 0 > $out/${tag}_bench_under_pmc_$c.json 2>/dev/null
  python3 tools/pmc_summary.py gpurun_out/pmc > $out/${tag}_pmc_$c.txt; rm -rf gpurun_out/pmc
done
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc > $out/${tag}_pmc_SQ.txt; rm -rf gpurun_out/pmc
./tools/ubench/valu_rate > $out/${tag}_valu_rate.txt 2>&1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench/segsort_bench tools/ubench/segsort_bench.hip && ./tools/ubench/segsort_bench > $out/${tag}_segsort_ubench.txt 2>&1
rm -rf $cache
head -14 $out/${tag}_kernel_trace_summary.txt
grep "^k_dp_pk \|^k_traceback_pk\|^void k_seed" $out/${tag}_pmc_*.txt
