#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/prof/ (copy what should be judged into profiles/).
# usage (from the repo root, through gpurun):  bash tools/collect_profiles.sh r02 [bench config, default c2]
# The data set is generated once (plain run, forked generator) and cached under /tmp: the profiled runs load it
# (rocprofv3 and a forking child do not mix).
set -u
tag=${1:-r02}; cfg=${2:-c2}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; mkdir -p $out
timeout 900 python3 bench.py --config $cfg --data-cache /tmp/tb > $out/${tag}_bench_default.json 2>/dev/null
B="python3 bench.py --config $cfg --data-cache /tmp/tb --no-cpu-baseline --loci 0"
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
head -14 $out/${tag}_kernel_trace_summary.txt
grep "^k_dp_pk \|^k_traceback_pk" $out/${tag}_pmc_*.txt
