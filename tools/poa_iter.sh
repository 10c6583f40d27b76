#!/bin/bash
# POA iteration aid: the consensus parity tests, the loci leg's kernel trace, and (with gpurun_variants/libtelrhip_poaprof.so, built with
# -DPOA_PROF) the share of a window's wave time per phase.   usage (through gpurun): bash tools/poa_iter.sh <tag> [config]
set -u
tag=${1:-poa}; cfg=${2:-c2}
timeout 900 python3 -m pytest tests/test_gpu_consensus.py -x -q -m gpu 2>&1 | tail -3
bash tools/loci_trace.sh $tag $cfg
if [ -f gpurun_variants/libtelrhip_poaprof.so ]; then
  cache=$(mktemp -d /tmp/c.XXXX)
  TELR_LIB=$PWD/gpurun_variants/libtelrhip_poaprof.so timeout 800 python3 bench.py --config $cfg --data-cache $cache --bam-leg none --no-stream-leg --no-default-aligner-leg --no-shard-leg --no-cpu-baseline --no-upstream-check --steps 1 --warmup 0 2>&1 >/dev/null | grep "poa prof" | tail -2
  rm -rf $cache
fi
