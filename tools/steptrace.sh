mkdir -p gpurun_out/prof gpurun_out/kt2
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline --loci 0 --no-stream-leg --steps 1 --warmup 0 --bam-leg none > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt2 -- python3 bench.py --require-cache --no-cpu-baseline --loci 0 --no-stream-leg --steps 6 --warmup 1 --bam-leg none > gpurun_out/prof/step_under_rocprof.json 2>/dev/null
f=$(ls gpurun_out/kt2/*/*kernel_trace.csv | head -1)
python3 tools/step_overlap.py $f | tee gpurun_out/prof/r03_step_overlap.txt
python3 -c "import json;d=json.load(open('gpurun_out/prof/step_under_rocprof.json'));print(d['value'], d['ms_per_step'])"
